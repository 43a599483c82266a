#!/bin/bash
# developer tool (GPU box): variant builds of the library (clid-slam_amd/build.py --variant) alternating on ONE box: tools/r06_lib_ab.sh <label> <reps> <lib names ...>   ("default" = the tree's library)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; out=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p "$out"
label=$1; reps=$2; shift; shift
: > $out/lib_ab_$label.jsonl
for rep in $(seq 1 $reps); do for v in "$@"; do for steps in 20 200; do
  lib=""; [ "$v" != default ] && lib="$GRAFT_REPO_ROOT/clid-slam_amd/lib/libclid_native_$v.so"
  CLID_NATIVE_LIB=$lib python bench.py --no-cpu-baseline --frame-calls 0 --steps $steps --warmup 5 $CLID_AB_ARGS 2>> $out/log.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k={x['kernel'][:24]:round(x['avg_us'],2) for x in (d.get('roofline') or {}).get('kernels',[])}
print(json.dumps({'lib':'$v','rep':$rep,'steps':$steps,'ms_per_step':round(d['ms_per_step'],5),'gpu_ms':round((d.get('timed_region_split') or {}).get('gpu_ms',0),4),'kernels_us':k}))" >> $out/lib_ab_$label.jsonl
done; done; done
cat $out/lib_ab_$label.jsonl
