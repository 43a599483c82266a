#!/bin/bash
# developer tool: bench.py under each decode kernel (CLID_DECODE 0 = 16-lane VALU kernel, 1 = tile fp32 MFMA, 2 = tile bf16 MFMA)
out=${1:-gpurun_out/variants}
mkdir -p "$out"
for v in ${VARIANTS:-0 1 2}; do
  for st in ${STEPS:-200 20}; do
    python bench.py --decode $v --steps "$st" --warmup 5 --no-cpu-baseline ${BENCH_ARGS:-} > "$out/v${v}_s${st}.json" 2> "$out/err.log" || tail -5 "$out/err.log"
  done
done
python - "$out" <<'EOF'
import json, sys, glob, os
for f in sorted(glob.glob(os.path.join(sys.argv[1], "v*_s*.json"))):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, "unreadable", e); continue
    r = d["roofline"]
    pf = d.get("per_frame_regime") or {}
    print(os.path.basename(f), round(d["ms_per_step"] * 1e3, 2), "us/step |", "mapping(10):", round(pf.get("ms_per_step", 0) * 1e3, 2), "us/step |",
          {k["kernel"].split(" ")[0]: k["avg_us"] for k in r["kernels"]}, "frac", r["frac"], "loss", round(d["final_loss"]["total"], 5))
EOF
