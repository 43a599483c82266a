#!/bin/bash
# developer tool (GPU box): A/B of the overlapped search schedule (clid_train_args.sched) on the driver's command shape.
# usage: tools/side_ab.sh OUTFILE [steps] ; one JSON line per configuration: env, ms_per_step, per-kernel us
out=${1:-gpurun_out/side_ab.jsonl}; steps=${2:-20}; warm=5
: > "$out"
run() {  # env assignments as arguments
  line=$(env "$@" python bench.py --steps $steps --warmup $warm --no-cpu-baseline --frame-calls 0 2>/dev/null | tail -1)
  python - "$line" "$*" <<'PY' >> "$out"
import json, sys
try:
    d = json.loads(sys.argv[1])
    ks = {k["kernel"].split(" ")[0].split("<")[0]: k["avg_us"] for k in d["roofline"]["kernels"]}
    print(json.dumps({"env": sys.argv[2], "steps": d["steps"], "ms_per_step": round(d["ms_per_step"], 5), "gpu_ms": round(d["timed_region_split"]["gpu_ms"], 4),
                      "wall_ms": round(d["timed_region_split"]["wall_ms"], 4), "kernels_us": ks, "loss": d["final_loss"]["total"]}))
except Exception as e:
    print(json.dumps({"env": sys.argv[2], "error": str(e), "raw": sys.argv[1][:200]}))
PY
}
run CLID_SIDE=0
run CLID_SIDE=0
for grp in 1 2 0; do
  run CLID_SIDE=1 CLID_SIDE_GROUP=$grp
  for blocks in 64 128 256; do run CLID_SIDE=1 CLID_SIDE_GROUP=$grp CLID_SIDE_BLOCKS=$blocks; done
  for cus in percu:4 percu:8 percu:12 percu:16 xcd:1 xcd:2; do run CLID_SIDE=1 CLID_SIDE_GROUP=$grp CLID_SIDE_CUS=$cus; done
done
run CLID_SIDE=1 CLID_SIDE_GROUP=1 CLID_SIDE_PRIO=-1
run CLID_SIDE=1 CLID_SIDE_GROUP=0 CLID_SIDE_PRIO=-1
run CLID_SIDE=1 CLID_SIDE_GROUP=1 CLID_SIDE_PRIO=1
run CLID_SIDE=0
cat "$out"
