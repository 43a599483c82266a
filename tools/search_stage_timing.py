"""Developer tool (GPU box): stage latencies of ONE unit (tile = two 8-query tasks) of the cell-directory search launch
(k_search_tiles<., 1>) from in-kernel s_memtime stamps (-DCLID_TIMING build; every stamp drains vmcnt / lgkmcnt first, so the
figures are serialised stage latencies of wave 0 of the first 256 blocks, last unit of the wave), for a launch of ONE iteration
(one unit per wave: the shape a search riding inside another launch would have) and of 20 iterations (the hoisted launch).
usage: python tools/search_stage_timing.py   -> JSON lines"""
import ctypes as C, json, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
csrc, objdir = os.path.join(ROOT, "clid-slam_amd", "csrc"), os.path.join(ROOT, "clid-slam_amd", "lib", "obj")
out = "/tmp/libclid_timing_search.so"
v = "/tmp/variant_train.o"
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-ffp-contract=on", "-w",
                       "-mllvm", "-amdgpu-kernarg-preload-count=16", "-DCLID_TIMING", "-c", os.path.join(csrc, "train.hip"), "-o", v])
objs = [v if o == "train.o" else os.path.join(objdir, o) for o in sorted(os.listdir(objdir)) if o.endswith(".o")]
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-ldl", "-o", out])
from clid_slam_amd import _lib, HotPathConfig
_lib.LIB_PATH = out
import torch, bench
cfg = HotPathConfig(); cfg.device = "cuda:0"
nm, dec, mp, scene = bench.build_scene(cfg, "cuda:0")
lib = C.CDLL(out)
CLOCK_MHZ = 100.0  # s_memtime ticks at the 100 MHz reference clock on this part (calibrated below against a timed launch)
names = [(12, "task start"), (13, "index -> pool coordinates"), (14, "cells + stencil rows, directory words requested"),
         (16, "directory words -> hit list in LDS"), (15, "hit positions -> distances -> K winners"),
         (18, "winners' pos4 rows, IDW weights, blended offset"), (19, "records of both tasks stored"), (20, "tile numbered (LDS hash)")]
for iters in (1, 20):
    for _ in range(3):
        mp.mapping(iters)
    torch.cuda.synchronize()
    buf = (C.c_longlong * (256 * 32))()
    assert lib.clid_debug_read_stamps(buf) == 0
    a = np.array(buf, dtype=np.int64).reshape(256, 32)
    rec = {"launch": f"{iters} iteration(s)", "unit": "s_memtime ticks (median over 256 blocks' wave 0, last unit)", "stages": {}}
    prev = None
    for k, nme in names:
        if prev is not None:
            d = a[:, k] - a[:, prev]
            rec["stages"][f"{dict(names)[prev]} -> {nme}"] = [float(np.median(d)), float(np.percentile(d, 90))]
        prev = k
    rec["second task 12 -> 18"] = float(np.median(a[:, 18] - a[:, 12]))
    rec["records + numbering 18 -> 20"] = float(np.median(a[:, 20] - a[:, 18]))
    print(json.dumps(rec))
