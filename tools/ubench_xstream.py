#!/usr/bin/env python3
"""Developer tool (GPU box): what does a cross-stream dependency cost?  N tiny kernels back to back on ONE stream against the
same N kernels alternating between TWO streams with an event record + stream wait per hop (the shape of the overlapped search
schedule, clid_train_args.sched), and against two streams without any dependency.  Also with a CU-masked second stream."""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import clid_slam_amd  # noqa: E402,F401
from clid_slam_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
x = torch.zeros(4096, device=dev)
y = torch.zeros(4096, device=dev)
N = 400


def timed(fn, reps=5):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e6 / N


def one_stream():
    for _ in range(N):
        x.add_(1.0)


def make_two(s2):
    s1 = torch.cuda.current_stream()

    def pingpong():
        for i in range(N // 2):
            x.add_(1.0)
            e = torch.cuda.Event()
            e.record(s1)
            s2.wait_event(e)
            with torch.cuda.stream(s2):
                x.add_(1.0)
                e2 = torch.cuda.Event()
                e2.record(s2)
            s1.wait_event(e2)

    def independent():
        for i in range(N // 2):
            x.add_(1.0)
            with torch.cuda.stream(s2):
                y.add_(1.0)

    def join_only():  # side stream runs ahead; the main stream waits for an event that completed long ago
        evs = []
        with torch.cuda.stream(s2):
            for i in range(N // 2):
                y.add_(1.0)
                e = torch.cuda.Event()
                e.record(s2)
                evs.append(e)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(N // 2):
            s1.wait_event(evs[i])
            x.add_(1.0)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e6 / (N // 2)

    return pingpong, independent, join_only


out = {"one_stream_us_per_kernel": timed(one_stream)}
s2 = torch.cuda.Stream()
pp, ind, jo = make_two(s2)
out["two_streams_pingpong_us_per_kernel"] = timed(pp)
out["two_streams_independent_us_per_kernel"] = timed(ind)
out["wait_on_completed_event_then_kernel_us"] = min(jo() for _ in range(3))
lib = _lib.load()
words = _lib.cu_mask_words("percu:8")
obj = C.c_void_p()
arr = (C.c_uint32 * len(words))(*words)
hip = lib
handle = C.c_void_p()
if hip.hipExtStreamCreateWithCUMask(C.byref(handle), C.c_uint32(len(words)), arr) == 0:
    sm = torch.cuda.ExternalStream(handle.value, device=dev)
    pp, ind, jo = make_two(sm)
    out["masked_pingpong_us_per_kernel"] = timed(pp)
    out["masked_independent_us_per_kernel"] = timed(ind)

    def masked_only():
        with torch.cuda.stream(sm):
            for _ in range(N):
                y.add_(1.0)
    out["masked_one_stream_us_per_kernel"] = timed(masked_only)
print(json.dumps(out))
