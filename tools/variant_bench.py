"""Developer tool: build a variant of the library with extra -D flags into /tmp on the GPU box and run bench.py
against it (the committed library is untouched).   usage: python tools/variant_bench.py "-DCLID_DECODE_WAVES=5" [bench args... | --sequence [frames]]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
flags = sys.argv[1].split()
csrc = os.path.join(ROOT, "clid-slam_amd", "csrc")
out = "/tmp/libclid_variant.so"
srcs = [os.path.join(csrc, f) for f in ("api.hip", "comm.hip", "p2p.hip", "table.hip", "celldir.hip", "query.hip", "query_tile.hip", "track_tile.hip", "train.hip", "train_analytic.hip", "train_wf0.hip", "train_tile.hip", "mlp.hip", "sampler.hip", "mapops.hip")]
only = os.environ.get("VARIANT_SRCS", "").split()  # recompile just these sources, link the rest from the committed build's objects
CC = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-ffp-contract=on", "-Wno-unused-value", "-Wno-unused-result", "-w"]
if only:
    objdir = os.path.join(ROOT, "clid-slam_amd", "lib", "obj")
    objs = []
    for f in srcs:
        b = os.path.basename(f)
        o = os.path.join(objdir, b.replace(".hip", ".o"))
        if b in only:
            o = "/tmp/variant_" + b.replace(".hip", ".o")
            subprocess.check_call(CC + flags + ["-c", f, "-o", o])
        objs.append(o)
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-ldl", "-o", out])
else:
    subprocess.check_call(CC + flags + ["-shared", *srcs, "-ldl", "-o", out])
import clid_slam_amd  # noqa
from clid_slam_amd import _lib
_lib.LIB_PATH = out
if len(sys.argv) > 2 and sys.argv[2] == "--sequence":  # the large-map point of the sequence workload instead of bench.py
    import bench, bench_sequence as BS
    frames = int(sys.argv[3]) if len(sys.argv) > 3 else 50
    cfg, rows, checks, (nm, dec, mp) = BS.run(frames, "cuda:0", quiet=True)
    kernels, _ = bench.kernel_report(_lib.load(), mp, 10, cfg.bs, cfg.gradient_decimation, nm.local_count(), 1)
    tail = rows[frames // 2:]
    print(flags, "M_local", nm.local_count(), "median mapping ms", round(sorted(r["t_mapping_ms"] for r in tail)[len(tail) // 2], 3),
          [(k["kernel"][:16], k["avg_us"]) for k in kernels])
    sys.exit(0)
sys.argv = ["bench.py", "--no-cpu-baseline"] + sys.argv[2:]
import io, contextlib
buf = io.StringIO()
import bench
with contextlib.redirect_stdout(buf):
    bench.main()
d = json.loads(buf.getvalue().strip().splitlines()[-1])
print(sys.argv[1:], flags, os.environ.get("CLID_SEARCH_BLOCKS"), round(d["ms_per_step"] * 1e3, 2),
      "frame-regime", round((d.get("per_frame_regime") or {}).get("ms_per_step", 0) * 1e3, 2), [(k["kernel"][:16], k["avg_us"]) for k in d["roofline"]["kernels"]])
