"""In-kernel phase timing of k_train_fused (developer tool, GPU box only).

Builds a -DCLID_TIMING variant of the library into /tmp, runs a few iterations and prints the median
cycle count between the s_memtime stamps of wave 0 of the first 256 blocks (each stamp drains
vmcnt/lgkmcnt first, so the numbers are serialised phase latencies, not pipelined costs)."""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
csrc = os.path.join(ROOT, "clid-slam_amd", "csrc")
out = "/tmp/libclid_timing.so"
srcs = [os.path.join(csrc, f) for f in ("api.hip", "comm.hip", "p2p.hip", "table.hip", "celldir.hip", "query.hip", "query_tile.hip", "track_tile.hip", "train.hip", "train_analytic.hip", "train_wf0.hip", "train_tile.hip", "mlp.hip", "sampler.hip", "mapops.hip")]
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
                       "-ffp-contract=on", "-DCLID_TIMING", "-Wno-unused-value", "-shared", *srcs, "-ldl", "-o", out])
import clid_slam_amd
from clid_slam_amd import _lib
_lib.LIB_PATH = out
import torch, bench
from clid_slam_amd import HotPathConfig
cfg = HotPathConfig(); cfg.device = "cuda:0"
nm, dec, mp, scene = bench.build_scene(cfg, "cuda:0")
mp.mapping(8); torch.cuda.synchronize()  # (>= 6 iterations: batches in Morton order)
lib = C.CDLL(out)
buf = (C.c_longlong * (256 * 32))()
assert lib.clid_debug_read_stamps(buf) == 0
a = np.array(buf, dtype=np.int64).reshape(256, 32)
names8 = {0: "task start", 1: "coords", 2: "probed", 3: "selected", 4: "fence", 5: "fwd r0", 6: "fwd r1", 7: "fence2", 8: "bwd r0", 9: "bwd r1", 10: "fence3", 24: "loop end", 25: "flushed"}
names = {0: "A start", 1: "A buckets", 2: "A pos4", 3: "A inserted", 4: "A selected", 5: "A blended", 6: "A mlp",
         8: "B start", 9: "B buckets", 10: "B pos4", 11: "B inserted", 12: "B selected", 13: "B blended", 14: "B mlp",
         16: "stashed", 17: "bwdA start", 18: "bwdA end", 20: "bwdB start", 21: "bwdB end", 24: "loop end", 25: "flushed"}
import os
names = names8
if os.environ.get("CLID_DECODE", "1") != "0":  # the tile (matrix-core) decode kernel has its own stamps
    assert lib.clid_debug_read_stamps_tile(buf) == 0
    a = np.array(buf, dtype=np.int64).reshape(256, 32)
    names = {0: "start", 1: "weights", 2: "record", 8: "numbered", 3: "gathered", 4: "blended", 5: "mlp fwd", 6: "loss", 7: "dh+df",
             9: "atomics", 10: "stamps", 11: "dW1", 12: "loop end", 13: "flushed"}
keys = sorted(names)
if 13 in names:
    keys = [0, 1, 2, 8, 3, 4, 5, 6, 7, 9, 10, 11, 12, 13]  # program order of the tile kernel's stamps
print("phase deltas (median / p90 cycles at 100 MHz s_memtime? raw units), relative to previous stamp:")
prev = None
for k in keys:
    if prev is not None:
        d = a[:, k] - a[:, prev]
        print(f"{names[prev]:>12s} -> {names[k]:<12s} median {np.median(d):9.0f}  p90 {np.percentile(d, 90):9.0f}")
    prev = k
print("total", np.median(a[:, max(keys)] - a[:, 0]))
