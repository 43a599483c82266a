"""Developer tool: timeline of ONE Mapper.mapping(10) call from a rocprofv3 kernel trace (start offset, duration, gap).
   on the GPU box:  rocprofv3 --kernel-trace -d gpurun_out/tl -o tl --output-format csv -- python tools/frame_timeline.py run
                    python tools/frame_timeline.py show gpurun_out/tl/.../tl_kernel_trace.csv"""
import csv, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if sys.argv[1] == "run":
    import time, torch, bench
    from clid_slam_amd import HotPathConfig
    cfg = HotPathConfig(); cfg.device = "cuda:0"
    nm, dec, mp, scene = bench.build_scene(cfg, "cuda:0")
    mp.reserve(10)
    for _ in range(40):
        mp.mapping(10); torch.cuda.synchronize()
else:
    rows = list(csv.DictReader(open(sys.argv[2])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # the last call = everything after the last gap > 20 us that precedes >= 20 kernels
    starts = [int(r["Start_Timestamp"]) for r in rows]; ends = [int(r["End_Timestamp"]) for r in rows]
    cut = [i for i in range(1, len(rows)) if starts[i] - ends[i - 1] > 15000]
    calls = [(a, b) for a, b in zip(cut, cut[1:] + [len(rows)]) if b - a >= 20]
    a, b = calls[-2]
    t0 = starts[a]
    print("kernels", b - a, "span_us", (ends[b - 1] - t0) / 1e3, "idle before the call us", (starts[a] - ends[a - 1]) / 1e3)
    for i in range(a, b):
        n = rows[i]["Kernel_Name"].split("(")[0][-46:]
        print("%8.1f  dur %7.2f  gap %6.2f  %s" % ((starts[i] - t0) / 1e3, (ends[i] - starts[i]) / 1e3, (starts[i] - ends[i - 1]) / 1e3, n))
