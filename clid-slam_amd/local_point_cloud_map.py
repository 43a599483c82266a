"""Raw-point voxel map used for the region-specific SDF labels ("next" row N2).

Drop-in for `LocalPointCloudMap` (model/local_point_cloud_map.py:11-153): same constructor, attributes
(`buffer_pt_index`, `local_point_cloud_map`, `neighbor_idx`, `max_valid_range`, `map_size`, ...) and methods.
Map maintenance (`insert_points`, `update_map`) is a handful of torch ops on the device, once per frame;
`region_specific_sdf_estimation` -- the 7-probe / top-4 / plane-fit estimate evaluated for every near-surface
sample -- is the HIP kernel `k_region_sdf` (csrc/sampler.hip) behind `clid_region_sdf`.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from .tools import voxel_down_sample_async, voxel_down_sample_torch


class LocalPointCloudMap:
    def __init__(self, config) -> None:
        self.config = config
        self.idx_dtype = torch.int64
        self.dtype = config.dtype
        self.device = config.device
        self.resolution = config.local_voxel_size_m
        self.buffer_size = int(config.local_buffer_size)
        self.buffer_pt_index = torch.full((self.buffer_size,), -1, dtype=self.idx_dtype, device=self.device)
        self.local_point_cloud_map = torch.empty((0, 3), dtype=torch.float32, device=self.device)
        # NOT the neural-point primes: the middle one differs (model/local_point_cloud_map.py:27-29)
        self.primes = torch.tensor([73856093, 19349663, 83492791], dtype=self.idx_dtype, device=self.device)
        self.neighbor_idx = None
        self.max_valid_range = None
        self._neighbor_i32 = None
        self.set_search_neighborhood()
        self.map_size = config.local_map_size

    # ------------------------------------------------------------------ table
    def voxel_hash(self, points: torch.Tensor) -> torch.Tensor:
        """model/local_point_cloud_map.py:38-41 (fmod keeps the sign; negative values index from the end)."""
        cells = torch.floor(points / self.resolution).to(self.primes)
        return torch.fmod((cells * self.primes).sum(-1), self.buffer_size)

    def _write_slots(self, table: torch.Tensor, slots: torch.Tensor, first_index: int) -> None:
        """table[slots] = arange + first_index where several entries may name one slot: the reference's plain
        indexed assignment keeps the LAST one when run sequentially; amax is the same choice, deterministic
        on the GPU."""
        slots = torch.where(slots < 0, slots + self.buffer_size, slots)
        vals = torch.arange(slots.shape[0], dtype=self.idx_dtype, device=table.device) + first_index
        table.scatter_reduce_(0, slots, vals, reduce="amax", include_self=True)

    def insert_points(self, points: torch.Tensor) -> None:
        """:43-61: one point per voxel of this scan, appended where the voxel's slot is still empty."""
        sample_points = points[voxel_down_sample_torch(points, self.resolution)]
        slots = self.voxel_hash(sample_points)
        pos = torch.where(slots < 0, slots + self.buffer_size, slots)
        empty = torch.nonzero(self.buffer_pt_index[pos] == -1).flatten()
        fresh = sample_points.index_select(0, empty)
        self._write_slots(self.buffer_pt_index, slots.index_select(0, empty), self.local_point_cloud_map.shape[0])
        self.local_point_cloud_map = torch.cat((self.local_point_cloud_map, fresh), 0)

    def update_map(self, sensor_position: torch.Tensor, points: torch.Tensor) -> None:
        """:63-72: insert, keep what is within `map_size` of the sensor, rebuild the slot table."""
        if self._update_map_fused(sensor_position, points):
            return
        self.insert_points(points)
        near = torch.norm(self.local_point_cloud_map - sensor_position, dim=-1) < self.map_size
        self.local_point_cloud_map = self.local_point_cloud_map[near].contiguous()
        table = torch.full((self.buffer_size,), -1, dtype=self.idx_dtype, device=self.device)
        self._write_slots(table, self.voxel_hash(self.local_point_cloud_map), 0)
        self.buffer_pt_index = table

    def _update_map_fused(self, sensor_position: torch.Tensor, points: torch.Tensor) -> bool:
        """update_map in one enqueue (csrc/mapops.hip clid_cloud_update) + ONE count read-back; the point array and the slot
        table ping-pong between two buffers each.  False = not applicable (CPU tensors), the torch path runs."""
        if not (points.is_cuda and self.buffer_pt_index.is_cuda and self.local_point_cloud_map.dtype == torch.float32
                and points.dtype == torch.float32 and self.buffer_size < (1 << 30)):
            return False
        lib = _lib.load()
        dev = points.device
        if getattr(self, "_count_pending", False):
            self._finish_count()
        defer = getattr(self, "_defer_counts", None)
        self._defer_counts = None
        vox = self.__dict__.pop("_defer_vox", None)  # device int64[2] left by Mapper.process_frame: [voxel count | ordering failed]
        s_idx = n_s_dev = None
        if defer is not None and vox is not None:
            # nothing on the host needs the scan's voxel count: the down-sampling stays in flight, the update takes its index
            # list and count on the device (sized for every scan point), the caller reads the block with the frame's last read-back
            scan = points.contiguous()
            s_idx = voxel_down_sample_async(scan, self.resolution, vox)
        if s_idx is not None:
            samples, n_s_dev = scan, vox[0:1]
        else:
            samples = points[voxel_down_sample_torch(points, self.resolution)].contiguous()
        old = self.local_point_cloud_map.contiguous()
        n_a, n_s = int(old.shape[0]), int(samples.shape[0])
        n = n_a + n_s
        hint = getattr(self, "_sensor_pos_host", None)  # (tensor, host tuple) left by Mapper.process_frame: no read-back
        if hint is not None and hint[0] is sensor_position:
            sp = hint[1]
        else:
            sp = [float(v) for v in sensor_position.detach().reshape(-1)[:3].tolist()]
        state = getattr(self, "_pp", None)
        if state is None or state["dev"] != dev:
            state = self._pp = {"dev": dev, "side": 0, "pts": [None, None], "tab": [None, None]}
        side = 1 - state["side"]
        if state["pts"][side] is None or state["pts"][side].shape[0] < n:
            state["pts"][side] = torch.empty((max(int(n * 1.5), 1 << 16), 3), device=dev, dtype=torch.float32)
        if state["tab"][side] is None or state["tab"][side].data_ptr() == self.buffer_pt_index.data_ptr():
            state["tab"][side] = torch.empty((self.buffer_size,), device=dev, dtype=self.idx_dtype)
        out_pts, out_tab = state["pts"][side], state["tab"][side]
        need = int(lib.clid_cloud_workspace_bytes(n))
        if getattr(self, "_cloud_ws", None) is None or self._cloud_ws.numel() < need or self._cloud_ws.device != dev:
            self._cloud_ws = torch.empty(int(need * 1.5) + 1024, device=dev, dtype=torch.uint8)
            self._cloud_counts = torch.zeros(2, device=dev, dtype=torch.int64)
        counts = defer if defer is not None else self._cloud_counts
        _lib.check(lib.clid_cloud_update(
            old.data_ptr(), n_a, samples.data_ptr(), n_s, self.buffer_pt_index.data_ptr(), out_tab.data_ptr(), self.buffer_size,
            float(self.resolution), (C.c_double * 3)(*sp), float(self.map_size), int(sensor_position.dtype == torch.float64),
            out_pts.data_ptr(), counts.data_ptr(), self._cloud_ws.data_ptr(), _lib.ptr(s_idx), _lib.ptr(n_s_dev), _lib.stream()),
            "clid_cloud_update")
        state["side"] = side
        self.buffer_pt_index = out_tab
        if defer is not None:
            # the caller reads the count later (Mapper.process_frame: with the frame's last read-back); until then the map is
            # its first n rows -- an upper bound: rows beyond the true count are referenced by no slot of the table
            self.local_point_cloud_map = out_pts[:n]
            self._count_pending, self._pending = True, (out_pts, defer, vox if s_idx is not None else None,
                                                        (sensor_position, points) if s_idx is not None else None)
            return True
        kept = _lib.read_counts(self._cloud_counts, 1)[0]  # the one host round trip (sizes the map)
        self.local_point_cloud_map = out_pts[:kept]
        return True

    def _finish_count(self, kept=None, vox_failed=None):
        """Settle a deferred update_map: `kept` (and the failure flag of the scan's voxel pass, when that stayed in flight) as
        read by the caller, or read back here."""
        out_pts, counts, vox, redo = self._pending
        if kept is None:
            kept = _lib.read_counts(counts, 1)[0]
            vox_failed = None if vox is None else _lib.read_counts(vox, 2)[1]
        self.local_point_cloud_map = out_pts[:int(kept)]
        self._count_pending, self._pending = False, None
        if vox is not None and vox_failed:
            # the scan's voxel ids were too wide for the device-side ordering (a bounding box beyond 2^17 voxels per axis: an
            # outlier point).  That pass published zero voxels, so the update above inserted nothing (crop + table rebuild only);
            # the same scan goes in again through the pass with the library sort (one extra round trip, this frame only)
            self.vox_fallbacks = getattr(self, "vox_fallbacks", 0) + 1
            self._defer_counts = None
            self.__dict__.pop("_defer_vox", None)
            self.update_map(redo[0], redo[1])

    def set_search_neighborhood(self, num_nei_cells: int = 1, search_alpha: float = 0.2) -> None:
        """:74-96: the cells within (num_nei_cells + search_alpha) of the centre cell (7 by default)."""
        r = torch.arange(-num_nei_cells, num_nei_cells + 1, device=self.primes.device, dtype=self.primes.dtype)
        dx = torch.stack(torch.meshgrid(r, r, r, indexing="ij"), dim=-1).reshape(-1, 3)
        self.neighbor_idx = dx[torch.sum(dx**2, dim=-1) < (num_nei_cells + search_alpha) ** 2]
        self.max_valid_range = 1.732 * (num_nei_cells + 1) * self.resolution
        self._neighbor_i32 = self.neighbor_idx.to(torch.int32).contiguous()

    # ------------------------------------------------------------------ kernel-side view
    def _cloud_view(self):
        """(CloudView, tensors to keep alive) for the HIP entry points."""
        _lib.require_cuda(self.buffer_pt_index, "buffer_pt_index", torch.int64)
        pts = _lib.require_cuda(self.local_point_cloud_map.contiguous(), "local_point_cloud_map", torch.float32)
        nb = self._neighbor_i32
        if nb.device != pts.device:
            nb = self._neighbor_i32 = nb.to(pts.device)
        v = _lib.CloudView()
        v.buffer_pt_index, v.points, v.neighbor_idx = self.buffer_pt_index.data_ptr(), pts.data_ptr(), nb.data_ptr()
        v.buffer_size, v.n_points, v.P = self.buffer_size, pts.shape[0], nb.shape[0]
        v.resolution, v.max_valid_range = float(self.resolution), float(self.max_valid_range)
        v.eta_threshold, v.dist_threshold = 0.2, 0.1  # estimate_plane's defaults (:156-158)
        return v, (pts, nb, self.buffer_pt_index)

    def region_specific_sdf_estimation(self, points: torch.Tensor):
        """:98-153: (|SDF| estimate [N], surface mask [N] bool) for world-frame sample points."""
        lib = _lib.load()
        x = _lib.require_cuda(points.detach().to(torch.float32).contiguous(), "points", torch.float32)
        n = x.shape[0]
        sdf_abs = torch.empty(n, device=x.device, dtype=torch.float32)
        mask = torch.empty(n, device=x.device, dtype=torch.uint8)
        view, keep = self._cloud_view()
        _lib.check(lib.clid_region_sdf(C.byref(view), x.data_ptr(), n, sdf_abs.data_ptr(), mask.data_ptr(),
                                       _lib.stream()), "clid_region_sdf")
        surface_mask = mask.bool()
        if not self.config.silence:
            print(surface_mask.sum().item() / max(surface_mask.numel(), 1))
        return sdf_abs, surface_mask
