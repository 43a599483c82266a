"""Dense SDF inference for meshing ("next" row N3): the query part of
`Mesher.query_points` (utils/mesher.py:38-163) as one fused HIP launch per chunk.  Marching cubes and
Open3D mesh handling stay with the reference (skimage / open3d, out of scope)."""
from __future__ import annotations

import ctypes as C
import math

import torch

from . import _lib


def query_points(neural_points, sdf_mlp, config, coord, bs=None, query_sdf=True, query_sem=False, query_color=False,
                 query_mask=True, query_locally=False, mask_min_nn_count: int = 4, out_torch: bool = True):
    """Same 4-tuple as the reference method: (sdf_pred [N], None, None, mc_mask [N]); `out_torch=False`
    returns numpy arrays.  Semantic / colour heads are outside the hot-path scope."""
    if query_sem or query_color:
        raise NotImplementedError("semantic / colour queries are outside the hot-path scope")
    lib = _lib.load()
    bs = int(bs or config.infer_bs)
    n = coord.shape[0]
    dev = neural_points.neural_points.device
    sdf = torch.zeros(n, device=dev, dtype=torch.float32)
    nn_cnt = torch.zeros(n, device=dev, dtype=torch.int32)
    view, keep = neural_points._map_view(bool(query_locally))
    W1, b1, W2, b2 = sdf_mlp.flat_params()
    for i in range(math.ceil(n / bs)):
        x = coord[i * bs:(i + 1) * bs].to(dev, torch.float32).contiguous()
        m = x.shape[0]
        _lib.check(
            lib.clid_sdf_query(C.byref(view), _lib.ptr(W1), _lib.ptr(b1), _lib.ptr(W2), _lib.ptr(b2),
                               float(sdf_mlp.sdf_scale), _lib.ptr(x), m, sdf[i * bs:].data_ptr(),
                               nn_cnt[i * bs:].data_ptr(), _lib.stream()),
            "clid_sdf_query",
        )
    sdf_pred = sdf if query_sdf else None
    mc_mask = (nn_cnt >= mask_min_nn_count).to(torch.float32) if query_mask else None
    if not out_torch:
        sdf_pred = None if sdf_pred is None else sdf_pred.cpu().numpy().astype("float64")
        mc_mask = None if mc_mask is None else mc_mask.cpu().numpy().astype("float64")
    return sdf_pred, None, None, mc_mask


class Mesher:
    """`Mesher` with the reference's constructor and `query_points` signature (utils/mesher.py:20-47) for the part that is
    on the hot path: the dense SDF / mask query.  Marching cubes and mesh export need skimage / open3d and stay with the
    reference (out of scope): those methods raise."""

    def __init__(self, config, neural_points, decoders: dict):
        self.config = config
        self.silence = getattr(config, "silence", True)
        self.neural_points = neural_points
        self.sdf_mlp = decoders["sdf"]
        self.sem_mlp = decoders.get("semantic")
        self.color_mlp = decoders.get("color")
        self.device = config.device
        self.cur_device = self.device
        self.dtype = config.dtype

    def query_points(self, coord, bs, query_sdf=True, query_sem=False, query_color=False, query_mask=True,
                     query_locally=False, mask_min_nn_count: int = 4, out_torch: bool = False):
        """utils/mesher.py:38-163: (sdf_pred, sem_pred, color_pred, mc_mask)."""
        return query_points(self.neural_points, self.sdf_mlp, self.config, coord, bs, query_sdf, query_sem, query_color,
                            query_mask, query_locally, mask_min_nn_count, out_torch)

    def recon_aabb_mesh(self, *args, **kwargs):
        raise NotImplementedError("marching cubes / mesh export (utils/mesher.py:371-667) need skimage + open3d: out of scope")

    mc_mesh = recon_aabb_mesh
