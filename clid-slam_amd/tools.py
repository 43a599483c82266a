"""Host-side helpers with the reference's names and signatures (utils/tools.py).

Only the functions the SDF-training hot path and its immediate callers use are present:
`setup_optimizer` (:205-255), `get_gradient` (:298-311), `freeze_model`/`unfreeze_model`
(:314-325), `get_time` (:385-392), `voxel_down_sample_torch` (:639-682; needed by
`NeuralPoints.update`).  They are torch-level plumbing; the compute of the hot path is in the HIP
library.
"""
from __future__ import annotations

import os
import time

import torch
import torch.nn as nn
from torch import optim
from torch.autograd import grad


def setup_optimizer(config, neural_point_feat, mlp_geo_param=None, mlp_sem_param=None, mlp_color_param=None,
                    poses=None, lr_ratio=1.0):
    """Same parameter groups / hyper-parameters as utils/tools.py:205-255.  `Mapper.mapping` does
    NOT use this object (its Adam is the fused HIP kernel with identical arithmetic); it is kept for
    callers that drive their own optimisation loop."""
    lr_cur = config.lr * lr_ratio
    groups = []
    if mlp_geo_param is not None:
        groups.append({"params": mlp_geo_param, "lr": lr_cur, "weight_decay": 0.0})
    if getattr(config, "semantic_on", False) and mlp_sem_param is not None:
        groups.append({"params": mlp_sem_param, "lr": lr_cur, "weight_decay": 0.0})
    if getattr(config, "color_on", False) and mlp_color_param is not None:
        groups.append({"params": mlp_color_param, "lr": lr_cur, "weight_decay": 0.0})
    if poses is not None:
        groups.append({"params": poses, "lr": config.lr_pose, "weight_decay": config.weight_decay})
    groups.append({"params": neural_point_feat, "lr": lr_cur, "weight_decay": config.weight_decay})
    if getattr(config, "opt_adam", True):
        return optim.Adam(groups, betas=(0.9, 0.99), eps=config.adam_eps)
    return optim.SGD(groups, momentum=0.9)


def get_gradient(inputs, outputs):
    """d outputs / d inputs through autograd (utils/tools.py:298-311).  With the HIP-backed
    `NeuralPoints.query_feature` / `Decoder.sdf` this runs their hand-written backward kernels."""
    d_points = torch.ones_like(outputs, requires_grad=False, device=outputs.device)
    return grad(outputs=outputs, inputs=inputs, grad_outputs=d_points, create_graph=True, retain_graph=True,
                only_inputs=True)[0]


def freeze_model(model: nn.Module):
    for child in model.children():
        for param in child.parameters():
            param.requires_grad = False


def unfreeze_model(model: nn.Module):
    for child in model.children():
        for param in child.parameters():
            param.requires_grad = True


def get_time():
    """utils/tools.py:385-392.  (The fused mapping loop does not call this: the reference's six
    per-iteration synchronisations are replaced by stream ordering.)"""
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    return time.time()


_VOX_WS = {}


def voxel_down_sample_launch(points: torch.Tensor, voxel_size: float, value: torch.Tensor = None, n_dev: torch.Tensor = None):
    """First half of the voxel down-sampling on device tensors (`clid_voxel_down_sample_launch`): everything is enqueued,
    nothing is waited for.  Returns a handle for `voxel_down_sample_finish`; no other voxel down-sampling may run on the
    device in between (they share one workspace)."""
    from . import _lib

    lib = _lib.load()
    pts = points.detach()
    if pts.dtype != torch.float32 or not pts.is_contiguous():
        pts = pts.to(torch.float32).contiguous()
    n = pts.shape[0]
    need = int(lib.clid_voxel_workspace_bytes(n))
    ws = _VOX_WS.get(pts.device)
    if ws is None or ws.numel() < need:
        ws = _VOX_WS[pts.device] = torch.empty(int(need * 1.25) + 256, device=pts.device, dtype=torch.uint8)
    out = torch.empty(n, device=pts.device, dtype=torch.int64)
    val = None if value is None else value.detach().to(device=pts.device, dtype=torch.float32).contiguous()
    # n_dev (device int64[1]): only the first *n_dev rows of `points` count (a count the caller has not read back yet)
    _lib.check(lib.clid_voxel_down_sample_launch(pts.data_ptr(), n, float(voxel_size), _lib.ptr(val), _lib.ptr(n_dev), ws.data_ptr(),
                                                 out.data_ptr(), _lib.stream()), "clid_voxel_down_sample_launch")
    return (pts, val, ws, out, n)


def voxel_down_sample_async(points: torch.Tensor, voxel_size: float, count_out: torch.Tensor, n_dev: torch.Tensor = None):
    """The voxel down-sampling without a host round trip (`clid_voxel_down_sample_async`): returns the index tensor [n] whose
    first count_out[0] entries (device int64; count_out[1] != 0: ordering failed, see include/clid_native.h) are the result.
    The consumers take list and count on the device (LocalPointCloudMap.update_map / NeuralPoints.update).  None: not
    applicable (more than 2^21 points) -- the caller takes the two-phase path."""
    from . import _lib

    n = points.shape[0]
    if not (0 < n <= (1 << 21)) or points.dtype != torch.float32 or not points.is_contiguous():
        return None
    lib = _lib.load()
    need = int(lib.clid_voxel_workspace_bytes(n))
    ws = _VOX_WS.get(points.device)
    if ws is None or ws.numel() < need:
        ws = _VOX_WS[points.device] = torch.empty(int(need * 1.25) + 256, device=points.device, dtype=torch.uint8)
    out = torch.empty(n, device=points.device, dtype=torch.int64)
    _lib.check(lib.clid_voxel_down_sample_async(points.data_ptr(), n, float(voxel_size), _lib.ptr(n_dev), ws.data_ptr(), out.data_ptr(),
                                                count_out.data_ptr(), _lib.stream()), "clid_voxel_down_sample_async")
    return out


def voxel_down_sample_finish(handle):
    """Second half: the one host round trip; returns the indices (ascending linear voxel id)."""
    from . import _lib

    _, _, ws, out, n = handle
    m = _lib.load().clid_voxel_down_sample_finish(n, ws.data_ptr(), out.data_ptr(), _lib.stream())
    if m < 0:
        _lib.check(m, "clid_voxel_down_sample_finish")
    return out[:m]


def _voxel_down_sample_hip(points: torch.Tensor, voxel_size: float, value: torch.Tensor = None):
    """`clid_voxel_down_sample` (csrc/mapops.hip): bounding box -> hash insert with a 64-bit atomicMin per voxel ->
    splitters / partition / bucket sort of the occupied voxels; one host round trip instead of ~40 torch ops."""
    return voxel_down_sample_finish(voxel_down_sample_launch(points, voxel_size, value))


def voxel_down_sample_torch(points: torch.Tensor, voxel_size: float):
    """Indices of one point per voxel: the one closest to the voxel centre, distance quantised to
    1000 levels, ties to the lowest index (same selection rule as utils/tools.py:639-682).  Device tensors go
    through the HIP kernels of csrc/mapops.hip; host tensors (tests, tools) through the same rule written with
    torch sorts (deterministic, unlike the reference's scatter_reduce)."""
    if points.is_cuda and points.shape[0] > 0 and points.dim() == 2 and points.shape[1] == 3:
        return _voxel_down_sample_hip(points, voxel_size)
    quant = 1000
    grid = torch.floor(points / voxel_size)
    center = (grid + 0.5) * voxel_size
    dist = ((points - center) ** 2).sum(dim=1) ** 0.5
    qd = (dist / dist.max() * (quant - 1)).long()
    cell = grid.long() - torch.floor(points.min(dim=0)[0] / voxel_size).long()
    # The reference linearises the voxel coordinates with stride v = max(coordinate) (NOT max + 1,
    # utils/tools.py:659-661), so voxels whose coordinate equals v alias another voxel and are merged
    # with it.  Reproduced on purpose: it decides which neural points exist.
    v = cell.max()
    flat = cell[:, 0] + cell[:, 1] * v + cell[:, 2] * v * v
    n = points.shape[0]
    key = qd * n + torch.arange(n, device=points.device)
    # sort by (voxel, quantised distance, index); first of each voxel wins
    order = torch.argsort(key)
    order = order[torch.argsort(flat[order], stable=True)]
    fs = flat[order]
    first = torch.ones_like(fs, dtype=torch.bool)
    first[1:] = fs[1:] != fs[:-1]
    return order[first]


def voxel_down_sample_min_value_torch(points: torch.Tensor, voxel_size: float, value: torch.Tensor):
    """Indices of one point per voxel: the one with the smallest `value` (quantised to 1000 levels of value.max()),
    lowest index among equals; voxels in ascending order of the reference's linear voxel id (stride = max cell
    coordinate, as above).  Selection rule of utils/tools.py:685-724; device tensors through the HIP kernels, host
    tensors written with sorts (deterministic)."""
    if (points.is_cuda and points.shape[0] > 0 and points.dim() == 2 and points.shape[1] == 3 and value.shape[0] == points.shape[0]
            and os.environ.get("CLID_FUSED_MAINTENANCE", "1") != "0"):
        return _voxel_down_sample_hip(points, voxel_size, value)
    cell = torch.floor(points / voxel_size).long() - torch.floor(points.min(dim=0)[0] / voxel_size).long()
    v = cell.max()
    flat = cell[:, 0] + cell[:, 1] * v + cell[:, 2] * v * v
    q = (value / value.max() * 999).long()
    n = points.shape[0]
    key = q * n + torch.arange(n, device=points.device)
    order = torch.argsort(key)
    order = order[torch.argsort(flat[order], stable=True)]
    fs = flat[order]
    first = torch.ones_like(fs, dtype=torch.bool)
    first[1:] = fs[1:] != fs[:-1]
    return order[first]


def transform_torch(points: torch.Tensor, transformation: torch.Tensor):
    """Rigid transform of [N,3] points by a 4x4 matrix, evaluated in the points' dtype
    (utils/tools.py:590-609)."""
    if points.is_cuda and points.dtype == torch.float32 and points.dim() == 2 and points.shape[1] == 3 and points.shape[0] > 0:
        import ctypes as C

        from . import _lib

        lib = _lib.load()
        x = points.detach().contiguous()
        out = torch.empty_like(x)
        T = transformation.detach().to("cpu", torch.float32)  # 16 numbers; `.to(points)` in the reference
        pose = (C.c_float * 12)(*T[:3, :].reshape(-1).tolist())
        _lib.check(lib.clid_transform_points(x.data_ptr(), x.shape[0], pose, out.data_ptr(), _lib.stream()),
                   "clid_transform_points")
        return out
    T = transformation.to(points)
    return points @ T[:3, :3].T + T[:3, 3]


def transform_batch_torch(points: torch.Tensor, transformation: torch.Tensor):
    """Per-point rigid transform: points [N,3], transformation [N,4,4] (utils/tools.py:612-636)."""
    T = transformation.to(points)
    return torch.bmm(T[:, :3, :3], points.unsqueeze(-1)).squeeze(-1) + T[:, :3, 3]


def feature_pca_torch(data, principal_components=None, principal_dim: int = 3, down_rate: int = 1,
                      project_data: bool = True, normalize: bool = True):
    """utils/tools.py:858-916: PCA of an [N, D] tensor -> (projected [N, principal_dim] or None, components [D, k]).
    The covariance is symmetric, so `eigh` replaces the reference's general `eig` (same subspace, eigenvector signs are
    arbitrary in both); projections are min-max normalised to [0, 1] per component like the reference."""
    n = data.shape[0]
    centered = data - data.mean(dim=0)
    if principal_components is None:
        used = centered[::down_rate]
        if used.shape[0] <= principal_dim:
            raise ValueError("not enough data for the PCA (down_rate too large or too few points)")
        cov = used.T @ used / max(n - 1, 1)
        evals, evecs = torch.linalg.eigh(cov.double())
        order = torch.argsort(evals, descending=True)[:principal_dim]
        principal_components = evecs[:, order].to(data.dtype)
    projected = None
    if project_data:
        projected = centered @ principal_components
        if normalize:
            lo, hi = projected.min(dim=0).values, projected.max(dim=0).values
            projected = (projected - lo) / (hi - lo).clamp_min(1e-12)
    return projected, principal_components


class PointCloudArrays:
    """What the GUI packets need of an Open3D point cloud (gui/gui_utils.py:52-130 reads `.points` / `.colors` and converts
    them to numpy): returned by the `*_o3d` helpers when open3d is not installed."""

    def __init__(self, points, colors=None):
        import numpy as np

        self.points = np.asarray(points, dtype=np.float64)
        self.colors = None if colors is None else np.asarray(colors, dtype=np.float64)

    def __len__(self):
        return self.points.shape[0]


def point_cloud_o3d(points_np, colors_np=None):
    """An open3d.geometry.PointCloud when open3d is importable, else a PointCloudArrays with the same two fields."""
    try:
        import open3d as o3d

        pc = o3d.geometry.PointCloud()
        pc.points = o3d.utility.Vector3dVector(points_np)
        if colors_np is not None:
            pc.colors = o3d.utility.Vector3dVector(colors_np)
        return pc
    except Exception:  # open3d absent (this image) or stubbed
        return PointCloudArrays(points_np, colors_np)


def save_implicit_map(run_path, neural_points, mlp_dict, with_footprint: bool = True):
    """utils/tools.py:347-367: `<run_path>/model/pin_map.pth` = {"neural_points": the whole NeuralPoints module, one
    state_dict (or None) per decoder key} -- the object contract vis_pin_map.py:118-127 loads (it then re-hashes the map
    with recreate_hash and calls compute_feature_principle_components) -- plus memory_footprint.npy."""
    import os

    import numpy as np

    map_model = {"neural_points": neural_points}
    for key in list(mlp_dict.keys()):
        map_model[key] = None if mlp_dict[key] is None else mlp_dict[key].state_dict()
    os.makedirs(os.path.join(run_path, "model"), exist_ok=True)
    model_save_path = os.path.join(run_path, "model", "pin_map.pth")
    torch.save(map_model, model_save_path)
    if with_footprint:
        np.save(os.path.join(run_path, "memory_footprint.npy"), np.array(neural_points.memory_footprint))
    return model_save_path


def load_decoders(loaded_model, mlp_dict, freeze_decoders: bool = True):
    """utils/tools.py:370-382: restore the decoders saved by save_implicit_map."""
    for key in list(loaded_model.keys()):
        if key != "neural_points" and loaded_model[key] is not None and mlp_dict.get(key) is not None:
            mlp_dict[key].load_state_dict(loaded_model[key])
            if freeze_decoders:
                freeze_model(mlp_dict[key])
