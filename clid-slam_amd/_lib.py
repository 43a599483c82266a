"""ctypes binding of libclid_native.so (include/clid_native.h).

The product path has NO fallback: if the library is missing or a call fails, a RuntimeError is
raised.  Device pointers come from torch tensors (`.data_ptr()`), the stream from
`torch.cuda.current_stream()`; no torch types cross the ABI.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# (CLID_NATIVE_LIB: a variant build of the same sources, `python clid-slam_amd/build.py --variant NAME -D...` -- A/B of compile-time switches)
LIB_PATH = os.environ.get("CLID_NATIVE_LIB") or os.path.join(_HERE, "lib", "libclid_native.so")

F, D, H, K = 8, 11, 64, 6
MLP_PARAMS = H * D + H + H + 1  # 833
GRAD_FEAT_OFFSET = 836     # compact layout: [836 | (M+1) x 8]
GRAD_ROW16 = 16            # 16-float accumulation rows: 8 gradients | certainty increment | 7 unused
GRAD_FEAT_OFFSET16 = 848   # [848 | (M+1) x 16], rows 64-byte aligned (include/clid_native.h)


def grad_offset(stride: int) -> int:
    return GRAD_FEAT_OFFSET16 if stride == GRAD_ROW16 else GRAD_FEAT_OFFSET

_vp = C.c_void_p
_i32 = C.c_int32
_i64 = C.c_int64
_f32 = C.c_float


class MapView(C.Structure):
    _fields_ = [
        ("tab", _vp), ("tab_pos", _vp), ("pos4", _vp), ("feat", _vp), ("cert", _vp), ("ts_update", _vp), ("delta", _vp),
        ("filter", _vp),
        ("log2cap", _i32), ("M", _i32), ("P", _i32), ("buffer_size", _i32),
        ("resolution", _f32), ("max_valid_dist2", _f32), ("layer_norm", _i32), ("log2filter", _i32),
        ("weighted_first", _i32), ("stencil_nc", _i32),
        ("cdir_hdr", _vp), ("cdir_words", _vp), ("cdir_pos", _vp), ("stencil_rows", _vp),
    ]


class TrainArgs(C.Structure):
    _fields_ = [
        ("pool_coord", _vp), ("pool_label", _vp), ("pool_ts", _vp), ("pool_weight", _vp), ("index", _vp),
        ("bs", _i32), ("decimation", _i32), ("batch_offset", _i64),
        ("fd_eps", _f32), ("inv_n_main", _f32), ("inv_n_eik", _f32), ("sigma", _f32), ("weight_e", _f32),
        ("loss_weight_on", _i32), ("eikonal_mode", _i32), ("train_decoder", _i32),
        ("W1", _vp), ("b1", _vp), ("W2", _vp), ("b2", _vp),
        ("sdf_scale", _f32), ("defer_reduce", _i32),
        ("grad", _vp), ("ws", _vp), ("loss_out", _vp),
        ("debug_flags", _i32), ("grad_stride", _i32),
        # per-call switches and aids (ABI 3: no process-global state in the library)
        ("decode_variant", _i32), ("pipeline", _i32), ("sdf_dbg", _vp), ("prof", _vp),
        # touched-row bookkeeping of the hoisted-search loop (NULL = dense exchange / dense Adam sweep)
        ("touch_ws", _vp), ("touch_stride", _i64), ("touch_iter", _i32), ("touch_all", _i32), ("cbuf", _vp), ("p2p", _vp), ("decode_each_neighbour", _i32),
        # overlapped search schedule (ABI 6): clid_sched_create() object, iterations per side launch, block cap of a side launch
        ("sched", _vp), ("side_group", _i32), ("side_blocks", _i32),
        # config.ekional_add_to: 0 all / 1 surface / 2 freespace, |label| threshold, 1 / subset size per iteration (device floats)
        ("eik_mask", _i32), ("eik_mask_range", _f32), ("eik_inv_n", _vp),
        # ABI 7: config.main_loss_type (0 bce / 1 sdf_l1 / 2 sdf_l2 / 3 zhong); ba_done_flag: per-frame poses [n_pose][12] fp32
        ("main_loss_type", _i32), ("n_pose", _i32), ("pool_pose", _vp),
        # config.proj_correction_on: labels scaled by |cos(g, x - origin of the sample's frame)|; frame poses [n][12] fp32
        ("proj_correction", _i32), ("n_frame_pose", _i32), ("frame_pose", _vp),
        # config.consistency_loss_on: gradient probe output, dL/dg input, partial-row placement (see the header)
        ("g_out", _vp), ("c_extra", _vp), ("partial_row0", _i32), ("partial_rows_extra", _i32),
        # ABI 8: sharded dense exchange -- [n_dec_copies][848] decoder-gradient copies inside the all-reduced buffer (see the header)
        ("dec_copies", _vp), ("n_dec_copies", _i32), ("dec_ranks", _i32),
    ]


class TrackCall(C.Structure):
    """clid_track_call: what stays fixed over the iterations of the measurement model on one scan."""
    _fields_ = [
        ("mv", MapView),
        ("W1", _vp), ("b1", _vp), ("W2", _vp), ("b2", _vp),
        ("sdf_scale", _f32), ("min_nn", _i32), ("min_grad_norm", _f32), ("max_grad_norm", _f32), ("max_sdf_std", _f32), ("N", _i32),
        ("pc_imu", _vp), ("sdf_out", _vp), ("grad_out", _vp), ("pmap_out", _vp), ("valid_out", _vp),
    ]


class AdamArgs(C.Structure):
    _fields_ = [
        ("feat", _vp), ("grad", _vp), ("m", _vp), ("v", _vp),
        ("W1", _vp), ("b1", _vp), ("W2", _vp), ("b2", _vp), ("m_mlp", _vp), ("v_mlp", _vp),
        ("n_feat", _i64),
        ("lr", _f32), ("beta1", _f32), ("beta2", _f32), ("eps", _f32), ("weight_decay", _f32),
        ("step", _i32), ("train_decoder", _i32), ("grad_stride", _i32),
        ("cert", _vp), ("n_cert", _i32), ("pad0", _i32),
    ]


class CloudView(C.Structure):
    _fields_ = [
        ("buffer_pt_index", _vp), ("points", _vp), ("neighbor_idx", _vp), ("buffer_size", _i64),
        ("n_points", _i32), ("P", _i32),
        ("resolution", _f32), ("max_valid_range", _f32), ("eta_threshold", _f32), ("dist_threshold", _f32),
    ]


class SamplerParams(C.Structure):
    _fields_ = [
        ("surface_sample_range_m", _f32), ("free_sample_begin_ratio", _f32), ("free_sample_end_dist_m", _f32),
        ("dist_weight_scale", _f32), ("max_range", _f32),
        ("surface_sample_n", _i32), ("free_front_n", _i32), ("free_behind_n", _i32), ("dist_weight_on", _i32),
        ("behind_dropoff_on", _i32),
        ("pose", _f32 * 12), ("reserved", _i32 * 2),
    ]


_SIGS = {
    "clid_abi_version": (C.c_int, []),
    "clid_last_error": (C.c_char_p, []),
    "clid_table_build": (C.c_int, [_vp, _i32, _vp, _vp, _i64, _f32, _vp, _vp, _i32, _i32, _f32, _vp, _vp, _i32, _vp, _vp, _i32,
                                   _vp]),
    "clid_cdir_build": (C.c_int, [_vp, _i32, _vp, _vp, _i32, _vp, _i32, _i64, _f32, _vp, _vp, _i64, _vp, _i64, _vp, _vp]),
    "clid_radius_search": (C.c_int, [C.POINTER(MapView), _vp, _i32, _vp, _vp, _vp]),
    "clid_query_certainty": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _i32, _f32, _f32, _vp, _i32, _vp, _vp]),
    "clid_query_fwd": (C.c_int, [C.POINTER(MapView), _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "clid_query_bwd": (C.c_int, [C.POINTER(MapView), _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "clid_mlp_sdf_fwd": (C.c_int, [_vp, _vp, _vp, _vp, _f32, _vp, _i32, _vp, _vp]),
    "clid_mlp_sdf_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _f32, _vp, _vp, _i32, _vp, _vp, _vp]),
    "clid_sdf_grad_x": (C.c_int, [C.POINTER(MapView), _vp, _vp, _vp, _vp, _f32, _vp, _i32, _vp, _vp, _vp, _vp, _vp]),
    "clid_sdf_query": (C.c_int, [C.POINTER(MapView), _vp, _vp, _vp, _vp, _f32, _vp, _i32, _vp, _vp, _vp]),
    "clid_track_model": (C.c_int, [C.POINTER(MapView), _vp, _vp, _vp, _vp, _f32, C.POINTER(_f32), C.POINTER(_f32), _i32,
                         _f32, _f32, _f32, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "clid_track_model_dev": (C.c_int, [C.POINTER(MapView), _vp, _vp, _vp, _vp, _f32, _vp, _vp, _i32,
                         _f32, _f32, _f32, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "clid_loss_fwd_bwd": (C.c_int, [_vp, _vp, _vp, _i32, _f32, _i32, _vp, _i32, _f32, _vp, _vp, _vp, _vp]),
    "clid_adam_step": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _f32, _i32, _i32, _vp]),
    "clid_train_workspace_floats": (_i64, [_i32, _i32, _i32]),
    "clid_train_fwd_bwd": (C.c_int, [C.POINTER(MapView), C.POINTER(TrainArgs), _vp]),
    "clid_train_adam": (C.c_int, [C.POINTER(AdamArgs), C.POINTER(TrainArgs), _vp]),
    "clid_mapping_run": (C.c_int, [C.POINTER(MapView), C.POINTER(TrainArgs), C.POINTER(AdamArgs), _i32, _vp, _i64, _vp, _vp]),
    "clid_debug_task_cover": (C.c_int, [_i32, _i64, _i32, _i32, _vp, _vp, _vp]),
    "clid_region_sdf": (C.c_int, [C.POINTER(CloudView), _vp, _i32, _vp, _vp, _vp]),
    "clid_sample_frame": (C.c_int, [C.POINTER(CloudView), C.POINTER(SamplerParams), _vp, _i32, _vp, _vp, _vp, _vp, _vp,
                                    _vp, _vp, _vp]),
    "clid_transform_points": (C.c_int, [_vp, _i32, C.POINTER(_f32), _vp, _vp]),
    "clid_voxel_workspace_bytes": (_i64, [_i32]),
    "clid_voxel_down_sample": (C.c_int, [_vp, _i32, _f32, _vp, _vp, _vp]),
    "clid_voxel_down_sample_launch": (C.c_int, [_vp, _i32, _f32, _vp, _vp, _vp, _vp, _vp]),
    "clid_voxel_down_sample_finish": (C.c_int, [_i32, _vp, _vp, _vp]),
    "clid_voxel_down_sample_async": (C.c_int, [_vp, _i32, _f32, _vp, _vp, _vp, _vp, _vp]),
    "clid_voxel_down_sample_min_value": (C.c_int, [_vp, _i32, _f32, _vp, _vp, _vp, _vp]),
    "clid_map_rehash": (C.c_int, [_vp, _vp, _i32, _f32, _vp, _i64, _vp]),
    "clid_map_gather": (C.c_int, [_vp, _i32, _i64] + [_vp] * 13),
    "clid_map_prune_workspace_bytes": (_i64, [_i64]),
    "clid_map_prune_select": (C.c_int, [_vp, _vp, _i64, _vp, _i32, _f32, _f32, _i32, _vp, _vp, _vp, _vp]),
    "clid_pool_workspace_bytes": (_i64, [_i64]),
    "clid_pool_filter": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i64, C.POINTER(C.c_double), C.c_double,
                                   _i64, C.c_uint64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "clid_pool_filter_after": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i64, C.POINTER(C.c_double), C.c_double,
                                         _i64, C.c_uint64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "clid_cloud_workspace_bytes": (_i64, [_i64]),
    "clid_cloud_update": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _vp, _i64, _f32, C.POINTER(C.c_double), C.c_double, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "clid_map_insert_workspace_bytes": (_i64, [_i32]),
    "clid_map_insert": (C.c_int, [_vp, _i32, _vp, _i64, _f32, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _i32, _i32, _i32, _f32, _f32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "clid_local_window_workspace_bytes": (_i64, [_i64]),
    "clid_local_window": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _f32, _i32, _i32, _i32, C.POINTER(C.c_double), C.c_double, _i32,
                                    _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _vp]),
    "clid_local_to_global": (C.c_int, [_vp, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "clid_sample_compact_workspace_bytes": (_i64, [_i64]),
    "clid_sample_compact": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _vp, C.c_float, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "clid_new_sample_workspace_bytes": (_i64, [_i64]),
    "clid_new_sample_select": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _i32, C.c_float, C.c_float, _vp, _vp, _i64, C.c_float,
                                         C.c_float, _i64, _vp, _vp, _vp, _vp, _vp]),
    "clid_read_back": (C.c_int, [_vp, _i32, _vp, _vp]),
    "clid_mapping_prep_workspace_bytes": (_i64, [_i32, _i32]),
    "clid_mapping_prep": (C.c_int, [_vp, _i64, _vp, _i32, _i32, _i32, _i64, _vp, _i64, C.c_uint64, C.c_uint64, _vp, C.c_float,
                                    _vp, _i32, _i32, _i32, _vp]),
    "clid_debug_prep_draw": (_i64, [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64]),
    "clid_train_partial_rows": (_i32, [_vp]),
    "clid_consistency_couple": (C.c_int, [_vp, _vp, _vp, _i32, _i32, C.c_float, _vp, _vp, _vp, _vp]),
    "clid_debug_scan_scratch_bytes": (_i64, [_i64]),
    "clid_debug_scan": (C.c_int, [_vp, _vp, _i64, _i32, _vp, _vp]),
    "clid_comm_unique_id": (C.c_int, [_vp]),
    "clid_comm_init": (C.c_int, [_vp, _i32, _i32, C.POINTER(_vp)]),
    "clid_comm_size": (C.c_int, [_vp]),
    "clid_p2p_blob_bytes": (_i64, []),
    "clid_p2p_create": (C.c_int, [_i32, _i32, _i64, _vp, _vp]),
    "clid_p2p_connect": (C.c_int, [_vp, _vp]),
    "clid_p2p_selftest": (C.c_int, [_vp, _vp]),
    "clid_p2p_world": (_i32, [_vp]),
    "clid_p2p_capacity": (_i64, [_vp]),
    "clid_p2p_buffer": (_vp, [_vp]),
    "clid_p2p_allreduce": (C.c_int, [_vp, _i64, _vp]),
    "clid_p2p_allreduce_or": (C.c_int, [_vp, _vp, _i64, _vp]),
    "clid_p2p_status": (C.c_int, [_vp, _vp]),
    "clid_p2p_set_timeout": (C.c_int, [_vp, C.c_double]),
    "clid_p2p_agree": (C.c_int, [_vp, _vp, _vp]),
    "clid_debug_p2p_fail": (C.c_int, [_vp, _vp]),
    "clid_p2p_destroy": (C.c_int, [_vp]),
    "clid_debug_copy": (C.c_int, [_vp, _vp, _i64, _vp]),
    "clid_comm_available": (C.c_int, []),
    "clid_comm_allreduce": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _vp]),
    "clid_comm_destroy": (C.c_int, [_vp]),
    "clid_mapping_run_dist": (C.c_int, [C.POINTER(MapView), C.POINTER(TrainArgs), C.POINTER(AdamArgs), _i32, _vp, _i64, _vp,
                                        _vp, _i64, C.POINTER(_i64), _vp]),
    "clid_touch_stride": (_i64, [_i32]),
    "clid_touch_workspace_bytes": (_i64, [_i32, _i32]),
    "clid_train_touch_scan": (C.c_int, [C.POINTER(TrainArgs), _i32, _i32, _i32, C.POINTER(_i32), _vp]),
    "clid_train_chunk_iters": (_i32, [C.POINTER(TrainArgs)]),
    "clid_train_decode_kernel": (C.c_int, [C.POINTER(MapView), C.POINTER(TrainArgs)]),
    "clid_train_search_floats": (_i64, [_i32, _i64, _i32, _i32, _i32]),
    "clid_train_search_tasks": (_i32, [_i32, _i64, _i32, _i32]),
    "clid_train_search": (C.c_int, [C.POINTER(MapView), C.POINTER(TrainArgs), _i32, _vp, _i64, _vp, _vp]),
    "clid_train_decode": (C.c_int, [C.POINTER(MapView), C.POINTER(TrainArgs), _vp, _vp]),
    "clid_track_model_call": (C.c_int, [C.POINTER(TrackCall), _vp, _vp, _i32, _vp, _vp, _vp, C.c_double, _vp]),
    "clid_track_valid_count": (C.c_int, [_vp, _i32, _vp, _vp, C.c_double, _vp]),
    "clid_track_rows": (C.c_int, [C.POINTER(TrackCall), _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "clid_pinned_alloc": (C.c_int, [_i64, C.POINTER(_vp), C.POINTER(_vp)]),
    "clid_pinned_free": (None, [_vp]),
    "clid_sched_create": (C.c_int, [_vp, _i32, _i32, C.POINTER(_vp)]),
    "clid_sched_destroy": (None, [_vp]),
    "clid_debug_cu_census": (C.c_int, [_vp, _vp, _i32, _i32, _vp]),
    "clid_profile_create": (_vp, []),
    "clid_profile_read": (C.c_int, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_int), _vp]),
    "clid_profile_destroy": (None, [_vp]),
}

EXPORTS = tuple(_SIGS)
_lib = None


def _env_int(name: str, default: int, lo: int, hi: int) -> int:
    try:
        v = int(os.environ.get(name, default))
    except ValueError:
        return default
    return v if lo <= v <= hi else default


# Defaults of the loop's per-call switches (clid_train_args.decode_variant / .pipeline); a `Mapper` may override them with
# its attributes of the same names.  They were process-global setters inside the library in ABI 2.
DECODE_VARIANT = _env_int("CLID_DECODE", 1, 0, 2)   # 0 VALU kernel, 1 tile kernel fp32 MFMA, 2 tile kernel bf16 MFMA
PIPELINE = _env_int("CLID_PIPELINE", 1, 0, 1)       # 1 hoisted searches, 0 one fused search+decode launch per iteration
# overlapped search schedule (clid_train_args.sched): defaults of CLID_SIDE / CLID_SIDE_CUS / CLID_SIDE_GROUP / CLID_SIDE_BLOCKS
SIDE_DEFAULT, SIDE_CUS_DEFAULT, SIDE_GROUP_DEFAULT, SIDE_BLOCKS_DEFAULT = "0", "", 1, 0


def load():
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python clid-slam_amd/build.py` "
            "(there is no CPU/PyTorch fallback for the HIP path)"
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.clid_abi_version() != 8:
        raise RuntimeError("libclid_native.so ABI version mismatch; rebuild")
    _lib = lib
    return lib


def replica_generator(owner, config, device, salt: int):
    """Random draws that must be IDENTICAL on every rank of a data-parallel group (new neural-point features, the
    sampler's noise, the pool-capacity filter): under torch.distributed with world > 1 they come from a generator owned
    by `owner` and seeded from config.seed (+ salt), independent of the ranks' global RNG state.  Single process: None,
    i.e. torch's global generator exactly like the reference."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return None
    gens = owner.__dict__.setdefault("_replica_gens", {})
    key = (str(device), salt)
    if key not in gens:
        g = torch.Generator(device=device)
        g.manual_seed(int(getattr(config, "seed", 42)) * 1000003 + salt)
        gens[key] = g
    return gens[key]


_comm = None  # RCCL communicator of this process: None = not tried, False = unavailable, else a C pointer


def rccl_comm(dist):
    """The process-wide RCCL communicator behind the C ABI (clid_comm_*), created collectively on first use from the
    torch.distributed group: rank 0's unique id is broadcast, every rank runs ncclCommInitRank on its current device.
    Returns None (callers keep the torch.distributed path) when the backend is not RCCL ("nccl") or RCCL cannot be
    initialised on EVERY rank."""
    global _comm
    if _comm is not None:
        return _comm or None
    _comm = False
    if dist.get_backend() != "nccl" or os.environ.get("CLID_RCCL", "1") == "0":
        return None
    lib = load()
    dev = torch.device("cuda", torch.cuda.current_device())
    rank, world = dist.get_rank(), dist.get_world_size()
    # every rank must be able to resolve librccl BEFORE anyone enters the collective ncclCommInitRank: a rank whose dlopen
    # fails would return at once and leave the others waiting inside it
    ok = torch.tensor([1 if lib.clid_comm_available() == 1 else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if int(ok.item()) != 1:
        return None
    ident = torch.zeros(129, dtype=torch.uint8, device=dev)  # [128] id | ok flag
    if rank == 0:
        buf = (C.c_uint8 * 128)()
        if lib.clid_comm_unique_id(buf) == 0:
            ident[:128] = torch.tensor(list(buf), dtype=torch.uint8)
            ident[128] = 1
    dist.broadcast(ident, 0)
    host = ident.cpu()
    ok = torch.zeros(1, dtype=torch.int32, device=dev)
    ptr_out = _vp()
    if int(host[128]) == 1:
        idb = (C.c_uint8 * 128)(*host[:128].tolist())
        if lib.clid_comm_init(idb, rank, world, C.byref(ptr_out)) == 0 and lib.clid_comm_size(ptr_out) == world:
            ok[0] = 1
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if int(ok.item()) == 1:
        _comm = ptr_out
        return _comm
    if ptr_out:
        lib.clid_comm_destroy(ptr_out)
    return None


_p2p = None  # peer-mapped exchange object of this process: None = not tried, False = unavailable, else (pointer, capacity)
E_P2P_TIMEOUT = -4  # include/clid_native.h CLID_E_P2P_TIMEOUT


class P2pTimeout(RuntimeError):
    """A flag wait of the peer-mapped exchange gave up on some rank (agreed across the ranks: every rank raises).  The
    sums of the call are invalid; `Mapper.mapping` restores its saved state and repeats the call over RCCL."""


def p2p_default() -> bool:
    """The per-iteration payload of a sharded run goes over RCCL (what north_star names) unless the peer-mapped transport
    is asked for: CLID_P2P=1, or `Mapper.exchange_transport = "p2p"`."""
    return os.environ.get("CLID_P2P", "0") == "1"


def p2p_disable() -> None:
    """Rule the peer-mapped exchange out for the rest of this process (after a timeout: its epoch counters and error word
    are no longer trustworthy).  The object stays allocated -- peers may still have it mapped."""
    global _p2p
    _p2p = False



def p2p_exchange(dist, need_bytes: int):
    """The process-wide peer-mapped exchange object behind the C ABI (clid_p2p_*, csrc/p2p.hip), created collectively on
    first use (and re-created when `need_bytes` outgrows it -- every rank sees the same map, so every rank decides alike):
    each rank allocates its exchange buffers, the HIP IPC blobs are gathered over torch.distributed, every rank maps its
    peers and runs the collective self-test; the object is used only if EVERY rank reports success (MIN), else None and the
    callers keep RCCL / torch.distributed for the payload.  CLID_P2P=0 switches it off."""
    global _p2p
    if _p2p is False:
        return None
    if _p2p is not None and _p2p[1] >= need_bytes:
        return _p2p[0]
    lib = load()
    old = _p2p
    _p2p = False
    rank, world = dist.get_rank(), dist.get_world_size()
    if world > 8:
        return None
    dev = torch.device("cuda", torch.cuda.current_device())
    wire = dev if dist.get_backend() == "nccl" else torch.device("cpu")

    def agree(flag: bool) -> bool:
        t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=wire)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return int(t.item()) == 1

    torch.cuda.synchronize(dev)
    if old:
        lib.clid_p2p_destroy(old[0])
    cap = max(2 * int(need_bytes), 16 << 20)
    nb = int(lib.clid_p2p_blob_bytes())
    blob = (C.c_uint8 * nb)()
    obj = _vp()
    made = lib.clid_p2p_create(rank, world, cap, C.byref(obj), blob) == 0
    mine = torch.tensor(list(blob), dtype=torch.uint8, device=wire)
    parts = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    ok = agree(made)
    if ok:
        flat = torch.cat(parts).cpu().tolist()
        ok = agree(lib.clid_p2p_connect(obj, (C.c_uint8 * (nb * world))(*flat)) == 0)
    if ok:
        try:  # how long a flag wait polls before it gives up (default 600 s: a benign skew between ranks must not abort a run)
            lib.clid_p2p_set_timeout(obj, float(os.environ.get("CLID_P2P_TIMEOUT_S", "600")))
        except ValueError:
            pass
        ok = agree(lib.clid_p2p_selftest(obj, stream()) == 0)
    if not ok:
        if made:
            torch.cuda.synchronize(dev)
            lib.clid_p2p_destroy(obj)
        return None
    _p2p = (obj, cap)
    return obj


def p2p_likely(dist, wanted=None) -> bool:
    """True while the peer-mapped exchange is asked for (`wanted`, default CLID_P2P=1) and has not been ruled out for this
    process (not yet tried, or set up): the sharded loop then prefers the compact exchange at every map size, because that is
    the payload the object carries."""
    wanted = p2p_default() if wanted is None else bool(wanted)
    return wanted and _p2p is not False and dist.get_world_size() <= 8


_sched = {}  # device index -> (clid_sched pointer or None, side_group, side_blocks)


def cu_mask_words(spec: str, n_cus: int = 256, n_xcd: int = 8):
    """uint32 words of a CU mask for hipExtStreamCreateWithCUMask.  Bit i of the mask = compute unit i in the driver's
    numbering, which walks the XCDs first (bit i -> XCD i % 8, measured: tools/cu_mask_census.py).  `spec`:
      "percu:N"  N compute units of every XCD (bits 0 .. 8 N - 1);   "xcd:K"  every compute unit of XCDs 0 .. K - 1;
      "hex:..."  the mask itself, most significant word first;       "" / "none"  no mask (None)."""
    spec = (spec or "").strip().lower()
    if spec in ("", "none", "0"):
        return None
    bits = [0] * n_cus
    kind, _, val = spec.partition(":")
    if kind == "percu":
        for i in range(min(int(val) * n_xcd, n_cus)):
            bits[i] = 1
    elif kind == "xcd":
        for i in range(n_cus):
            bits[i] = 1 if (i % n_xcd) < int(val) else 0
    elif kind == "hex":
        v = int(val, 16)
        for i in range(n_cus):
            bits[i] = (v >> i) & 1
    else:
        raise ValueError(f"CU mask spec {spec!r}")
    if not any(bits):
        return None
    words = [sum(bits[32 * w + b] << b for b in range(32)) for w in range(n_cus // 32)]
    return words


def sched(device=None):
    """The process-wide schedule object of the overlapped search schedule for `device` (clid_train_args.sched): a side
    stream -- confined to the compute units of CLID_SIDE_CUS when set -- and its events, created on first use.  Returns
    (pointer or None, side_group, side_blocks).  CLID_SIDE = 0 / 1 switches the schedule off / on (default: see below),
    CLID_SIDE_GROUP = iterations per side launch (0 = growing groups), CLID_SIDE_BLOCKS = block cap of a side launch,
    CLID_SIDE_PRIO = -1 / 0 / 1."""
    dev = None if device is None else torch.device(device).index
    if dev is None:
        dev = torch.cuda.current_device()
    hit = _sched.get(dev)
    if hit is not None:
        return hit
    out = (None, 0, 0)
    if os.environ.get("CLID_SIDE", SIDE_DEFAULT) == "1":
        lib = load()
        words = cu_mask_words(os.environ.get("CLID_SIDE_CUS", SIDE_CUS_DEFAULT))
        obj = _vp()
        with torch.cuda.device(dev):
            arr = (C.c_uint32 * len(words))(*words) if words else None
            check(lib.clid_sched_create(arr, len(words) if words else 0, _env_int("CLID_SIDE_PRIO", 0, -1, 1), C.byref(obj)),
                  "clid_sched_create")
        out = (obj, _env_int("CLID_SIDE_GROUP", SIDE_GROUP_DEFAULT, 0, 32), _env_int("CLID_SIDE_BLOCKS", SIDE_BLOCKS_DEFAULT, 0, 1 << 16))
    _sched[dev] = out
    return out


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().clid_last_error()
        text = f"{what} failed ({rc}): {msg.decode() if msg else ''}"
        if rc == E_P2P_TIMEOUT:
            raise P2pTimeout(text)
        raise RuntimeError(text)


def ptr(t) -> int:
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream() -> int:
    """hipStream_t of torch's current stream on the current device (the raw getter is ~10x cheaper than building a
    torch.cuda.Stream object per call; this sits in front of every launch)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def read_counts(t: torch.Tensor, n: int):
    """The first `n` int64 counts of device tensor `t` as Python ints: ONE synchronising read-back through the library's
    pinned landing buffer (`.item()` / `.tolist()` copy into pageable memory, several times slower)."""
    if not (t.is_cuda and t.dtype == torch.int64 and t.is_contiguous() and t.numel() >= n and n <= 32):
        return [int(v) for v in t[:n].tolist()]
    host = (C.c_int64 * n)()
    check(load().clid_read_back(t.data_ptr(), 8 * n, host, stream()), "clid_read_back")
    return list(host)


def small_to_host(t: torch.Tensor) -> torch.Tensor:
    """`t.detach().cpu()` for a tensor of at most 256 bytes (a pose) through the pinned landing buffer."""
    t = t.detach()
    if not (t.is_cuda and t.is_contiguous() and 0 < t.numel() * t.element_size() <= 256):
        return t.cpu()
    out = torch.empty(t.shape, dtype=t.dtype)
    check(load().clid_read_back(t.data_ptr(), t.numel() * t.element_size(), out.data_ptr(), stream()), "clid_read_back")
    return out


def require_cuda(t: torch.Tensor, name: str, dtype=None) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on the GPU (got {t.device}); the HIP path has no CPU fallback")
    if dtype is not None and t.dtype != dtype:
        raise RuntimeError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")
    return t


def low_priority_stream(device):
    """A side stream of the LOWEST priority the device offers: the frame's large pool compaction (200 us of bandwidth-bound
    launches) runs on it next to the map growth's chain of small dependent launches on the caller's stream -- at equal priority
    those starve behind the big launch's waves (k_vox_splitters: 12 -> 138 us, profiles/r04_frame_trace.txt).  CLID_POOL_PRIO=0
    keeps a stream of default priority (A/B)."""
    import os
    import torch

    if os.environ.get("CLID_POOL_PRIO", "1") == "0":
        return torch.cuda.Stream(device=device)
    hip = load()  # the HIP runtime the native library is linked against (dlsym walks its dependencies): the one its launches use
    least, greatest = C.c_int(0), C.c_int(0)
    with torch.cuda.device(device):
        try:
            get_range, create = hip.hipDeviceGetStreamPriorityRange, hip.hipStreamCreateWithPriority
        except AttributeError:
            return torch.cuda.Stream(device=device)
        if get_range(C.byref(least), C.byref(greatest)) != 0 or least.value == greatest.value:
            return torch.cuda.Stream(device=device)
        handle = C.c_void_p()
        if create(C.byref(handle), C.c_uint(1), C.c_int(least.value)) != 0:  # 1 = hipStreamNonBlocking
            return torch.cuda.Stream(device=device)
    return torch.cuda.ExternalStream(handle.value, device=device)
