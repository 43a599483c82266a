"""Developer tool: profiles/r04_hbm_traffic.json (what bench.py reports as roofline.traffic) from the per-launch counter means of
a profile run (tools/profile_r04.sh).   usage: python tools/make_hbm_traffic.py [profiles/r04_pmc_summary.txt]"""
import ast, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r04_pmc_summary.txt")
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
def traffic(r):
    return int((2 * r["FETCH_SIZE"] + r.get("WRITE_SIZE", 0.0)) * 1024)
c2 = rows["cfg2"]
dec, adam, srch = pick(c2, r"k_decode_tile<0"), pick(c2, r"k_adam_all"), pick(c2, r"k_search_tiles<false, 1>|k_search_tiles<0, 1>|k_search_tiles")
iters, n_l = dec["launches_fetch"], srch["launches_fetch"]
Q, bs, M = 26218, 16384, 23496
out = {
    "how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/profile_r04.sh) on `python bench.py "
           "--no-cpu-baseline --config <cfg> --steps 64 --warmup 5 --frame-calls 0`, MI355X, per-launch means "
           "(profiles/r04_pmc_summary.txt); counters are KiB. traffic_bytes = 2 x FETCH_SIZE + WRITE_SIZE: the x2 is the guide's "
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
        "note": "row numbering read from the search launch's number blocks; batches in Morton order"},
    "k_search_tiles (cell-directory search + tile numbering, per iteration of a <=32-iteration launch)": {
        "fetch_kb_per_launch": srch["FETCH_SIZE"], "write_kb_per_launch": srch["WRITE_SIZE"], "launches": n_l, "iterations": iters,
        "traffic_bytes": int(traffic(srch) * n_l / iters), "algorithmic_bytes": 688 * Q + 24 * bs,
        "valu_insts_per_tile": round(srch["SQ_INSTS_VALU"] * n_l / (iters * 1640), 1)},
    "k_adam_all": {"fetch_kb": adam["FETCH_SIZE"], "write_kb": adam["WRITE_SIZE"], "traffic_bytes": traffic(adam),
                   "algorithmic_bytes": 256 * (M + 1) + 28 * 833},
}
c3 = rows.get("cfg3", {})
if c3:
    d3 = pick(c3, r"k_decode_tile<1")
    out["cfg3"] = {"workload": {"bs_per_gpu": 65536}, "k_decode_tile<bf16 MFMA>": {
        "fetch_kb": d3["FETCH_SIZE"], "write_kb": d3.get("WRITE_SIZE"), "traffic_bytes": traffic(d3),
        "mfma_busy_cycles_per_launch": d3.get("SQ_VALU_MFMA_BUSY_CYCLES"), "mfma_mops_bf16": d3.get("SQ_INSTS_VALU_MFMA_MOPS_BF16"),
        "mfma_mops_f32": d3.get("SQ_INSTS_VALU_MFMA_MOPS_F32")}}
json.dump(out, open(os.path.join(ROOT, "profiles", "r04_hbm_traffic.json"), "w"), indent=1)
print("decode", out["k_decode_tile<fp32 MFMA>"]["traffic_bytes"], "search/iter", traffic(srch) * n_l // iters, "adam", traffic(adam))
