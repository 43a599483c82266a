#!/bin/bash
# developer tool (GPU box): round-5 batch b -- sampler ambiguity diagnostic, full GPU suite, large-launch decode A/B (3 vs 2 waves per SIMD)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; out=gpurun_out/r05b; mkdir -p $out
python tools/debug_sampler_amb.py > $out/sampler_amb.jsonl 2> $out/sampler_amb.err
python tools/ubench_xstream.py > $out/xstream.json 2> $out/xstream.err
B="python bench.py --no-cpu-baseline --frame-calls 0"
: > $out/wps_ab.jsonl
for rep in 1 2; do for flag in 0 16; do
  for args in "--config cfg3 --steps 100 --warmup 10" "--config cfg4 --steps 50 --warmup 5" "--bs 65536 --decode 1 --steps 100 --warmup 10" "--config cfg3 --bs 262144 --steps 50 --warmup 5"; do
    CLID_DEBUG_FLAGS=$flag $B $args 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(json.dumps({'flags': $flag, 'args': '$args', 'ms_per_step': round(d['ms_per_step'],5), 'kernels_us': {k['kernel'].split(' ')[0]: k['avg_us'] for k in d['roofline']['kernels']}}))" >> $out/wps_ab.jsonl
  done; done; done
timeout 1500 python -m pytest tests -m gpu -q > $out/pytest_gpu.txt 2>&1
tail -30 $out/pytest_gpu.txt
cat $out/sampler_amb.jsonl | cut -c1-1500; cat $out/xstream.json; cat $out/wps_ab.jsonl
