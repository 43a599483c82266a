#!/bin/bash
# developer tool (GPU box): the overlapped search schedule again, on the 199 + 16-register decode kernel (two decode waves now leave
# 82 registers per SIMD lane: a 64-register search wave fits beside them, which it did not when r05_side_schedule_ab was taken)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; out=gpurun_out/side_ab2.jsonl; mkdir -p gpurun_out; : > $out
run() {
  for st in "20 5" "200 20"; do set -- "$@"; s=${st% *}; w=${st#* }
  line=$(env "$@" python bench.py --steps $s --warmup $w --no-cpu-baseline --frame-calls 0 2>/dev/null | tail -1)
  python - "$line" "$*" <<'PY' | tee -a $out
import json, sys
try:
    d = json.loads(sys.argv[1])
    ks = {k["kernel"].split(" ")[0].split("<")[0]: k["avg_us"] for k in d["roofline"]["kernels"]}
    print(json.dumps({"env": sys.argv[2], "steps": d["steps"], "ms_per_step": round(d["ms_per_step"], 5), "kernels_us": ks, "loss": round(d["final_loss"]["total"], 5)}))
except Exception as e:
    print(json.dumps({"env": sys.argv[2], "error": str(e), "raw": sys.argv[1][:200]}))
PY
  done
}
run CLID_SIDE=0
for grp in 0 2 4 8; do
  run CLID_SIDE=1 CLID_SIDE_GROUP=$grp
  run CLID_SIDE=1 CLID_SIDE_GROUP=$grp CLID_SIDE_PRIO=-1
  run CLID_SIDE=1 CLID_SIDE_GROUP=$grp CLID_SIDE_BLOCKS=256 CLID_SIDE_PRIO=-1
done
run CLID_SIDE=0
