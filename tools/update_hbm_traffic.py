"""Developer tool: rebuild profiles/r02_hbm_traffic.json (what bench.py reports as roofline.traffic) from the per-launch
counter means of a profile run.   usage: python tools/update_hbm_traffic.py gpurun_out/r02/pmc_summary.txt"""
import ast, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r02_pmc_summary.txt")
cfg, rows = None, {}
for line in open(src):
    m = re.match(r"===== (cfg\d) ", line)
    if m:
        cfg = m.group(1); rows[cfg] = {}; continue
    if line.startswith("====="):
        cfg = None; continue
    m = re.match(r"(fetch|write|sq|mfma) (clid::\S.*?) (\{.*\}) launches (\d+)", line)
    if cfg and m:
        rows[cfg].setdefault(m.group(2), {}).update(ast.literal_eval(m.group(3)))
        rows[cfg][m.group(2)]["launches_" + m.group(1)] = int(m.group(4))
path = os.path.join(ROOT, "profiles", "r02_hbm_traffic.json")
out = json.load(open(path))
def pick(d, pat):
    return next(v for k, v in d.items() if re.search(pat, k))
c2 = rows["cfg2"]
dec, adam, srch = pick(c2, r"k_decode_tile<0"), pick(c2, r"k_adam_all"), pick(c2, r"k_train_fused8<1>")
iters = dec["launches_fetch"]
e = out["k_decode_tile<fp32 MFMA>"]
e.update(fetch_kb=dec["FETCH_SIZE"], write_kb=dec["WRITE_SIZE"], traffic_bytes=int((2 * dec["FETCH_SIZE"] + dec["WRITE_SIZE"]) * 1024),
         valu_insts_per_wave=round(dec["SQ_INSTS_VALU"] / dec["SQ_WAVES"], 1), salu_insts_per_wave=round(dec["SQ_INSTS_SALU"] / dec["SQ_WAVES"], 1),
         lds_insts_per_wave=round(dec["SQ_INSTS_LDS"] / dec["SQ_WAVES"], 1), vmem_rd_per_wave=round(dec["SQ_INSTS_VMEM_RD"] / dec["SQ_WAVES"], 2),
         vmem_wr_per_wave=round(dec["SQ_INSTS_VMEM_WR"] / dec["SQ_WAVES"], 2), waves=int(dec["SQ_WAVES"]),
         mfma_busy_cycles_per_launch=dec["SQ_VALU_MFMA_BUSY_CYCLES"], mfma_busy_cycles_per_simd=round(dec["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024, 1),
         note="batches in Morton order of the samples' voxels (calls of >= 6 iterations): neighbouring queries share neural points, "
              "so the per-tile merge leaves fewer rows and the gathers hit in L2; traffic is about half the algorithmic bytes")
s = out[next(k for k in out if k.startswith("k_train_fused8<1>"))]
n_l = srch["launches_fetch"]
s.update(fetch_kb_per_launch=srch["FETCH_SIZE"], write_kb_per_launch=srch["WRITE_SIZE"], launches=n_l, iterations=iters,
         traffic_bytes=int((2 * srch["FETCH_SIZE"] + srch["WRITE_SIZE"]) * 1024 * n_l / iters),
         valu_insts_per_wave_task=round(srch["SQ_INSTS_VALU"] * n_l / (iters * 3278), 1))
a = out["k_adam_all"]
a.update(fetch_kb=adam["FETCH_SIZE"], write_kb=adam["WRITE_SIZE"], traffic_bytes=int((2 * adam["FETCH_SIZE"] + adam["WRITE_SIZE"]) * 1024))
c3 = rows.get("cfg3", {})
if c3:
    d3 = pick(c3, r"k_decode_tile<1")
    out["cfg3"]["k_decode_tile<bf16 MFMA>"].update(fetch_kb=d3["FETCH_SIZE"], write_kb=d3.get("WRITE_SIZE"),
        traffic_bytes=int((2 * d3["FETCH_SIZE"] + d3.get("WRITE_SIZE", 0)) * 1024),
        mfma_busy_cycles_per_launch=d3["SQ_VALU_MFMA_BUSY_CYCLES"], mfma_mops_bf16=d3["SQ_INSTS_VALU_MFMA_MOPS_BF16"],
        mfma_mops_f32=d3["SQ_INSTS_VALU_MFMA_MOPS_F32"])
json.dump(out, open(path, "w"), indent=1)
print("decode", e["traffic_bytes"], "search/iter", s["traffic_bytes"], "adam", a["traffic_bytes"])
