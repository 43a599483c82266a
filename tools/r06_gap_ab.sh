#!/bin/bash
# developer tool (GPU box): round-4 tree (git worktree add _old/r04 b1dbd25 && (cd _old/r04 && python clid-slam_amd/build.py); _old/ is git-ignored) and HEAD alternating on ONE box -- the driver's command shape,
# then a kernel trace of each, cut per mapping() call by tools/trace_calls.py  -> gpurun_out/r06/gap_ab.jsonl
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; out=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p "$out"
A="--no-cpu-baseline --frame-calls 0 --steps 20 --warmup 5"
: > $out/gap_ab.jsonl
for rep in 1 2 3; do for tree in _old/r04 .; do
  (cd $tree && python bench.py $A 2>> $out/log.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'tree':'$tree','rep':$rep,'ms_per_step':d['ms_per_step'],'kernels':d.get('kernels'),'split':d.get('timed_region_split')}))") >> $out/gap_ab.jsonl
done; done
for tree in _old/r04 .; do
  n=$(echo $tree | tr -d './_'); n=${n:-head}
  (cd $tree && timeout 600 rocprofv3 --kernel-trace -d $out/tr_$n -o t --output-format csv -- python bench.py $A > /dev/null 2>> $out/log.txt)
  f=$(find $out/tr_$n -name 't_kernel_trace.csv' | head -1)
  python tools/trace_calls.py "$f" 20 > $out/calls_$n.jsonl
  rm -rf $out/tr_$n
done
tail -3 $out/log.txt; cat $out/gap_ab.jsonl | cut -c1-400
