"""Developer tool (GPU box): time Mesher.query_points (row N3) against a variant build of one source.
usage: python tools/time_sdf_query.py "<-D flags>" [points]   (VARIANT_SRCS names the sources to recompile, default query_tile.hip)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
flags = sys.argv[1].split()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4194304
csrc = os.path.join(ROOT, "clid-slam_amd", "csrc")
objdir = os.path.join(ROOT, "clid-slam_amd", "lib", "obj")
only = os.environ.get("VARIANT_SRCS", "query_tile.hip").split()
out = "/tmp/libclid_variant.so"
objs = []
for o in sorted(os.listdir(objdir)):
    src = o.replace(".o", ".hip")
    if src in only:
        v = "/tmp/variant_" + o
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-ffp-contract=on", "-w",
                               *flags, "-c", os.path.join(csrc, src), "-o", v])
        objs.append(v)
    else:
        objs.append(os.path.join(objdir, o))
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-ldl", "-o", out])
import torch
import clid_slam_amd  # noqa
from clid_slam_amd import HotPathConfig, _lib, mesher
_lib.LIB_PATH = out
import bench
_lib.load()
cfg = HotPathConfig(); cfg.device = "cuda:0"
nm, dec, mp, scene = bench.build_scene(cfg, "cuda:0")
mp.mapping(50)
gen = torch.Generator().manual_seed(7)
pool = scene["coord"]
x = (pool[torch.randint(0, pool.shape[0], (n,), generator=gen)] + 0.05 * torch.randn((n, 3), generator=gen)).cuda().contiguous()
res = {}
for mode in ("1", "0"):
    os.environ["CLID_SDF_TILE"] = mode
    for _ in range(3): mesher.query_points(nm, dec, cfg, x, query_locally=False)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(10): mesher.query_points(nm, dec, cfg, x, query_locally=False)
    b.record(); torch.cuda.synchronize()
    res["tile" if mode == "1" else "lane16"] = round(a.elapsed_time(b) / 10, 4)
print(flags, n, "ms per call", res, "G points/s", {k: round(n / v / 1e6, 2) for k, v in res.items()})
