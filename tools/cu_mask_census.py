#!/usr/bin/env python3
"""Developer tool (GPU box): which compute units does a CU-masked stream reach?  For every mask spec a census launch
(clid_debug_cu_census) of short one-wave blocks that each hold their CU for a while; prints blocks per XCC and the number of
distinct (XCC, SE, SH, CU) ids seen.  Answers how hipExtStreamCreateWithCUMask numbers its bits on this part."""
import collections
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import clid_slam_amd  # noqa: E402,F401
from clid_slam_amd import _lib  # noqa: E402

lib = _lib.load()
torch.cuda.init()
torch.zeros(1, device="cuda")
specs = sys.argv[1:] or ["none", "percu:1", "percu:4", "percu:8", "xcd:1", "xcd:2", "hex:ff", "hex:ff00", "hex:" + "f" * 8]
n = 4096
for spec in specs:
    words = _lib.cu_mask_words(spec)
    obj = C.c_void_p()
    arr = (C.c_uint32 * len(words))(*words) if words else None
    _lib.check(lib.clid_sched_create(arr, len(words) if words else 0, 0, C.byref(obj)), "clid_sched_create")
    out = (C.c_int32 * n)()
    _lib.check(lib.clid_debug_cu_census(obj, out, n, 40000, None), "census")
    lib.clid_sched_destroy(obj)
    per_xcc = collections.Counter()
    cus = set()
    for v in out:
        xcc, hw = (v >> 16) & 0xF, v & 0xFFFF
        cu, sh, se = (hw >> 8) & 0xF, (hw >> 12) & 1, (hw >> 13) & 7
        per_xcc[xcc] += 1
        cus.add((xcc, se, sh, cu))
    per_xcc_cus = collections.Counter(c[0] for c in cus)
    print(json.dumps({"spec": spec, "bits_set": sum(bin(w).count("1") for w in (words or [])), "distinct_cus": len(cus),
                      "blocks_per_xcc": dict(sorted(per_xcc.items())), "cus_per_xcc": dict(sorted(per_xcc_cus.items()))}))
