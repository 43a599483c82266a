"""Developer tool (GPU box): in-kernel phase timing of k_batch_sort_bucket (-DCLID_TIMING build of mapops.hip; serialised stamps)."""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
csrc, objdir = os.path.join(ROOT, "clid-slam_amd", "csrc"), os.path.join(ROOT, "clid-slam_amd", "lib", "obj")
out = "/tmp/libclid_timing.so"
v = "/tmp/variant_mapops.o"
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-ffp-contract=on", "-w",
                       "-DCLID_TIMING", *sys.argv[1:], "-c", os.path.join(csrc, "mapops.hip"), "-o", v])
objs = [v if o == "mapops.o" else os.path.join(objdir, o) for o in sorted(os.listdir(objdir))]
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-ldl", "-o", out])
import clid_slam_amd
from clid_slam_amd import _lib, HotPathConfig
_lib.LIB_PATH = out
import torch, bench
cfg = HotPathConfig(); cfg.device = "cuda:0"
nm, dec, mp, scene = bench.build_scene(cfg, "cuda:0")
for _ in range(5): mp.mapping(20)
torch.cuda.synchronize()
lib = C.CDLL(out)
buf = (C.c_longlong * (256 * 32))()
assert lib.clid_debug_read_stamps_mapops(buf) == 0
a = np.array(buf, dtype=np.int64).reshape(256, 32)[:256]
names = {0: "start", 1: "samples loaded", 2: "ranked", 3: "scanned + counted", 4: "kept in LDS", 5: "ordered (bitonic)", 6: "written"}
keys = [0, 1, 2, 3, 4, 5, 6]
prev = None
for k in keys:
    if prev is not None:
        d = a[:, k] - a[:, prev]
        print(f"{names[prev]:>18s} -> {names[k]:<18s} median {np.median(d):8.0f}  p90 {np.percentile(d, 90):8.0f}")
    prev = k
print("total", np.median(a[:, 6] - a[:, 0]))
