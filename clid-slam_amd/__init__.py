"""clid-slam_amd: MI355X-native (gfx950 HIP) implementation of CLID-SLAM's per-scan SDF training
inner loop behind the reference's own Python call signatures.

    from clid_slam_amd import NeuralPoints, Decoder, Mapper        # drop-ins for model/*.py, utils/mapper.py
    from clid_slam_amd import LocalPointCloudMap, DataSampler     # ... model/local_point_cloud_map.py, utils/data_sampler.py
    from clid_slam_amd.loss import sdf_bce_loss
    from clid_slam_amd.tools import get_gradient, setup_optimizer

The compute lives in libclid_native.so (include/clid_native.h); see DESIGN.md / INTEGRATION.md.
"""
# (The package does not touch the process environment.  The loop is a chain of short dependent launches whose ~600 bytes of by-value
# argument structs should sit in device memory -- HIP_FORCE_DEV_KERNARG=1, this ROCm's default, read by the HIP runtime when it
# starts: decode 11.8 -> 13.8 us, Adam 4.3 -> 6.4 us per launch without it (profiles/r05_kernarg_ab.jsonl).  The entry scripts
# (bench.py, bench_sequence.py) pin it before importing torch; a host application does the same if its environment turns it off:
# INTEGRATION.md, "Environment".)
from .config import HotPathConfig  # noqa: F401
from .data_sampler import DataSampler  # noqa: F401
from .decoder import Decoder  # noqa: F401
from .local_point_cloud_map import LocalPointCloudMap  # noqa: F401
from .mapper import Mapper  # noqa: F401
from .neural_points import NeuralPoints  # noqa: F401

__all__ = ["NeuralPoints", "Decoder", "Mapper", "LocalPointCloudMap", "DataSampler", "HotPathConfig"]
