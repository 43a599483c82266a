#!/bin/bash
# developer tool (GPU box): trees (git worktrees under the git-ignored _old/, each built with its own clid-slam_amd/build.py) alternating on ONE box, the driver's command shape (and --steps 200):  tools/r06_ab.sh <label> [reps] [trees...]
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; out=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p "$out"
label=$1; reps=${2:-3}; shift; shift; trees=${@:-"_old/r05 ."}
: > $out/ab_$label.jsonl
for rep in $(seq 1 $reps); do for tree in $trees; do for steps in 20 200; do
  (cd $tree && python bench.py --no-cpu-baseline --frame-calls 0 --steps $steps --warmup 5 $CLID_AB_ARGS 2>> $out/log.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k={x['kernel']:round(x['avg_us'],2) for x in (d.get('roofline') or {}).get('kernels',[])}
print(json.dumps({'tree':'$tree','rep':$rep,'steps':$steps,'ms_per_step':round(d['ms_per_step'],5),'split':{a:round(b,4) for a,b in (d.get('timed_region_split') or {}).items()},'kernels_us':k}))") >> $out/ab_$label.jsonl
done; done; done
cat $out/ab_$label.jsonl | cut -c1-600
