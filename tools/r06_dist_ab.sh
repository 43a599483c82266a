#!/bin/bash
# developer tool (GPU box): the sharded loop, 2 ranks on ONE GPU over gloo (dry run of the launch sequence), r05 tree vs HEAD alternating
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; out=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p "$out"
: > $out/dist_ab.jsonl
for rep in 1 2; do for tree in _old/r05 .; do
  (cd $tree && timeout 600 python bench.py --gpus 2 --backend gloo --steps 20 --warmup 5 --no-cpu-baseline --frame-calls 0 2>> $out/log.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k={x['kernel'][:28]:[round(x['avg_us'],2), x.get('launches')] for x in (d.get('roofline') or {}).get('kernels',[])}
print(json.dumps({'tree':'$tree','rep':$rep,'ms_per_step':round(d['ms_per_step'],5),'exchange':(d['config'].get('gradient_exchange') or {}).get('rccl_dense',{}).get('mode'),'kernels':k}))") >> $out/dist_ab.jsonl
done; done
cat $out/dist_ab.jsonl | cut -c1-500
