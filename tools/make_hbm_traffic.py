"""Developer tool: profiles/rNN_hbm_traffic.json (what bench.py reports as roofline.traffic / the issue roof) from the per-launch
counter means of a profile run (tools/profile_rNN.sh).
usage: python tools/make_hbm_traffic.py [profiles/r05_pmc_summary.txt [profiles/r05_resource_usage.txt]]"""
import ast, json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r05_pmc_summary.txt")
tag = re.search(r"(r\d\d)_", os.path.basename(src)).group(1)
res_src = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", tag + "_resource_usage.txt")
cfg, rows = None, {}
for line in open(src):
    m = re.match(r"===== (cfg\d) ", line)
    if m:
        cfg = m.group(1); rows[cfg] = {}; continue
    if line.startswith("====="):
        cfg = None; continue
    m = re.match(r"(fetch|write|sq|mfma) (\S.*?) (\{.*\}) launches (\d+)", line)
    if cfg and m:
        rows[cfg].setdefault(m.group(2), {}).update(ast.literal_eval(m.group(3)))
        rows[cfg][m.group(2)]["launches_" + m.group(1)] = int(m.group(4))
def pick(d, pat):
    return next(v for k, v in d.items() if re.search(pat, k))
def resources(pat):
    """VGPRs / AGPRs / LDS / occupancy of the instantiation matching `pat` (tools/resusage.sh output)."""
    try:
        for line in open(res_src):
            if re.search(pat, line):
                kv = dict(x.split("=") for x in line.split("|")[1].split())
                return {"vgpr": int(kv["VGPRs"]), "agpr": int(kv["AGPRs"]), "scratch_bytes_per_lane": int(kv["ScratchSize"]),
                        "lds_bytes_per_block": int(kv["LDS"]), "occupancy_waves_per_simd": int(kv["Occupancy"])}
    except OSError:
        pass
    return None
def issue(r):
    """VALU-port model of a launch (MI355X_MICROARCH.md: a wave64 VALU instruction occupies its SIMD-32 for 2 cycles, the matrix
    pipe is per SIMD and SQ_VALU_MFMA_BUSY_CYCLES counts its cycles): cycles the VALU / matrix port of ALL SIMDs is busy =
    2 x (VALU instructions - MFMA instructions) + MFMA busy cycles; the MFMA count is busy cycles / 32 for the fp32 16x16x4 form
    (32 cycles per SIMD), / 16 for bf16 16x16x32 (~17).  SALU / LDS / VMEM issue from other waves on their own ports."""
    if "SQ_INSTS_VALU" not in r:
        return None
    busy = r.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) or 0.0
    per = 16.0 if (r.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0) or 0.0) > (r.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0) or 0.0) else 32.0
    n_mfma = busy / per
    return {"valu_insts": r["SQ_INSTS_VALU"], "salu_insts": r.get("SQ_INSTS_SALU"), "lds_insts": r.get("SQ_INSTS_LDS"),
            "vmem_insts": (r.get("SQ_INSTS_VMEM_RD", 0.0) or 0.0) + (r.get("SQ_INSTS_VMEM_WR", 0.0) or 0.0), "waves": r.get("SQ_WAVES"),
            "mfma_busy_cycles": busy, "mfma_insts_est": n_mfma,
            "valu_port_cycles": 2.0 * max(r["SQ_INSTS_VALU"] - n_mfma, 0.0) + busy,
            "wave_cycles": r.get("SQ_WAVE_CYCLES"), "wait_any": r.get("SQ_WAIT_ANY")}
def commit():
    try:
        return subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
    except Exception:
        return "unknown"
def traffic(r):
    return int((2 * r["FETCH_SIZE"] + r.get("WRITE_SIZE", 0.0)) * 1024)
c2 = rows["cfg2"]
dec, adam, srch = pick(c2, r"k_decode_tile<0"), pick(c2, r"k_adam_all"), pick(c2, r"k_search_tiles<false, 1>|k_search_tiles<0, 1>|k_search_tiles")
iters, n_l = dec["launches_fetch"], srch["launches_fetch"]
Q, bs, M = 26218, 16384, 23496
out = {
    "source": {"pmc_summary": os.path.basename(src), "commit_when_generated": commit(),
               "note": "offline PMC passes: stale once the kernels change -- bench.py prints this stamp next to the figures"},
    "how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/profile_" + tag + ".sh) on `python bench.py "
           "--no-cpu-baseline --config <cfg> --steps 64 --warmup 5 --frame-calls 0`, MI355X, per-launch means "
           "(profiles/" + tag + "_pmc_summary.txt); counters are KiB. traffic_bytes = 2 x FETCH_SIZE + WRITE_SIZE: the x2 is the guide's "
           "gfx950 correction for FETCH_SIZE, calibrated in round 1 on this repo's own access patterns (tools/calib_fetch.hip, "
           "profiles/r01_pmc_calibration.txt); atomic rows are booked as writes only. Infinity-Cache hits are included in these "
           "memory-side counters.",
    "workload": {"bs_per_gpu": bs, "decimation": 10, "neural_points_local": M},
    "k_decode_tile<fp32 MFMA>": {
        "fetch_kb": dec["FETCH_SIZE"], "write_kb": dec["WRITE_SIZE"], "traffic_bytes": traffic(dec), "algorithmic_bytes": 752 * Q,
        "valu_insts_per_wave": round(dec["SQ_INSTS_VALU"] / dec["SQ_WAVES"], 1), "salu_insts_per_wave": round(dec["SQ_INSTS_SALU"] / dec["SQ_WAVES"], 1),
        "lds_insts_per_wave": round(dec["SQ_INSTS_LDS"] / dec["SQ_WAVES"], 1), "vmem_rd_per_wave": round(dec["SQ_INSTS_VMEM_RD"] / dec["SQ_WAVES"], 2),
        "vmem_wr_per_wave": round(dec["SQ_INSTS_VMEM_WR"] / dec["SQ_WAVES"], 2), "waves": int(dec["SQ_WAVES"]),
        "mfma_busy_cycles_per_launch": dec.get("SQ_VALU_MFMA_BUSY_CYCLES"),
        "mfma_busy_cycles_per_simd": round(dec.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024, 1),
        "issue": issue(dec), "resources": resources(r"k_decode_tile<0, false, 4, true, 0, false>|k_decode_tile<0, false, 4, true>"),
        "note": "row numbering read from the search launch's number blocks; batches in Morton order"},
    "k_search_tiles (cell-directory search + tile numbering, per iteration of a <=32-iteration launch)": {
        "fetch_kb_per_launch": srch["FETCH_SIZE"], "write_kb_per_launch": srch["WRITE_SIZE"], "launches": n_l, "iterations": iters,
        "traffic_bytes": int(traffic(srch) * n_l / iters), "algorithmic_bytes": 688 * Q + 24 * bs,
        "valu_insts_per_tile": round(srch["SQ_INSTS_VALU"] * n_l / (iters * 1640), 1),
        "issue_per_iteration": {k: (v * n_l / iters if isinstance(v, (int, float)) else v) for k, v in (issue(srch) or {}).items()},
        "resources": resources(r"k_search_tiles<false, 1>")},
    "k_adam_all": {"fetch_kb": adam["FETCH_SIZE"], "write_kb": adam["WRITE_SIZE"], "traffic_bytes": traffic(adam),
                   "algorithmic_bytes": 256 * (M + 1) + 28 * 833, "issue": issue(adam), "resources": resources(r"k_adam_all")},
}
c3 = rows.get("cfg3", {})
if c3:
    d3 = pick(c3, r"k_decode_tile<1")
    out["cfg3"] = {"workload": {"bs_per_gpu": 65536}, "k_decode_tile<bf16 MFMA>": {
        "fetch_kb": d3["FETCH_SIZE"], "write_kb": d3.get("WRITE_SIZE"), "traffic_bytes": traffic(d3),
        "mfma_busy_cycles_per_launch": d3.get("SQ_VALU_MFMA_BUSY_CYCLES"), "mfma_mops_bf16": d3.get("SQ_INSTS_VALU_MFMA_MOPS_BF16"),
        "mfma_mops_f32": d3.get("SQ_INSTS_VALU_MFMA_MOPS_F32"), "issue": issue(d3),
        "resources": resources(r"k_decode_tile<1, false, 2, false, 0, false>|k_decode_tile<1, false, 2, false>")}}
json.dump(out, open(os.path.join(ROOT, "profiles", tag + "_hbm_traffic.json"), "w"), indent=1)
print("decode", out["k_decode_tile<fp32 MFMA>"]["traffic_bytes"], "search/iter", traffic(srch) * n_l // iters, "adam", traffic(adam))
