"""Build recipe for libclid_native.so (hand-written gfx950 HIP kernels + the C ABI of
include/clid_native.h).  hipcc cross-compiles without a GPU; the .so is written IN-TREE
(clid-slam_amd/lib/) so it travels to the GPU box with the repository snapshot.

    python clid-slam_amd/build.py            # or: __graft_entry__.build()
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libclid_native.so")
SOURCES = ["api.hip", "comm.hip", "p2p.hip", "table.hip", "celldir.hip", "query.hip", "query_tile.hip", "track_tile.hip", "train.hip", "train_analytic.hip", "train_wf0.hip", "train_tile.hip", "mlp.hip", "sampler.hip", "mapops.hip"]
HEADERS = [os.path.join(CSRC, "common.hpp"), os.path.join(CSRC, "train_common.hpp"), os.path.join(CSRC, "search8.hpp"), os.path.join(HERE, "..", "include", "clid_native.h")]
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
    "-munsafe-fp-atomics",      # global_atomic_add_f32 instead of a CAS loop
    "-ffp-contract=on",         # FMA only inside one expression; exact-order code uses __f*_rn
    "-Wno-unused-result", "-Wno-unused-value",
    # the leading scalar / pointer arguments of a kernel arrive in SGPRs with the wave instead of through an s_load of the
    # kernel-argument segment (decode 11.75 -> 11.5 us: the record request goes out earlier; profiles/r05_kernarg_ab.jsonl)
    "-mllvm", "-amdgpu-kernarg-preload-count=16",
]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stamp() -> str:
    h = hashlib.sha256()
    for f in [os.path.join(CSRC, s) for s in SOURCES] + HEADERS + [os.path.abspath(__file__)]:
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def build_variant(name: str, extra_flags) -> str:
    """Developer aid (A/B of compile-time switches): the library built with `extra_flags` as lib/libclid_native_<name>.so; run
    with CLID_NATIVE_LIB=<that path> (clid-slam_amd/_lib.py)."""
    objdir = os.path.join(LIBDIR, "obj_" + name)
    os.makedirs(objdir, exist_ok=True)
    out = os.path.join(LIBDIR, f"libclid_native_{name}.so")
    hipcc = _hipcc()

    def one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        subprocess.run([hipcc, *FLAGS, *extra_flags, "-c", os.path.join(CSRC, src), "-o", obj], check=True)
        return obj

    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(one, SOURCES))
    subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-ldl", "-o", out], check=True)
    return out


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    stamp_file = LIB + ".stamp"
    stamp = _stamp()
    if not force and os.path.exists(LIB) and os.path.exists(stamp_file):
        if open(stamp_file).read().strip() == stamp:
            return LIB
    hipcc = _hipcc()
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)

    def compile_one(src: str) -> str:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [hipcc, *FLAGS, "-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-ldl", "-o", LIB]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    with open(stamp_file, "w") as fh:
        fh.write(stamp)
    if verbose:
        print(f"built {LIB}")
    return LIB


if __name__ == "__main__":
    if "--variant" in sys.argv:  # python build.py --variant NAME -DFLAG=0 ...
        i = sys.argv.index("--variant")
        print(build_variant(sys.argv[i + 1], sys.argv[i + 2:]))
    else:
        build(force="--force" in sys.argv)
