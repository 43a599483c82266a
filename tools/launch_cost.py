"""Host cost of one kernel launch through the C ABI (enqueue-only time of many tiny clid_adam_step launches).
usage: python tools/launch_cost.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clid_slam_amd import _lib

lib = _lib.load()
n = 256
p, g, m, v = (torch.zeros(n, device="cuda") for _ in range(4))
for stream_kind in ("default", "side"):
    st = torch.cuda.Stream() if stream_kind == "side" else torch.cuda.current_stream()
    with torch.cuda.stream(st):
        s = _lib.stream()
        for _ in range(200):
            lib.clid_adam_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), n, 0.01, 0.9, 0.99, 1e-15, 0.0, 1, 1, s)
        torch.cuda.synchronize()
        N = 4000
        t0 = time.perf_counter()
        for _ in range(N):
            lib.clid_adam_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), n, 0.01, 0.9, 0.99, 1e-15, 0.0, 1, 1, s)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"{stream_kind:8s} stream: enqueue {1e6 * (t1 - t0) / N:.2f} us/launch, total {1e6 * (t2 - t0) / N:.2f} us/launch")
