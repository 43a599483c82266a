#!/bin/bash
# developer tool (GPU box): round-5 batch g -- early record request (variant build) A/B + parity on the variant
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; out=gpurun_out/r05g; mkdir -p $out
CLID_NATIVE_LIB=clid-slam_amd/lib/libclid_native_early.so timeout 900 python -m pytest tests/test_tile_decode.py tests/test_hip_parity.py -m gpu -q -x -k "g6 or tile or mapping" > $out/pytest_early.txt 2>&1; tail -3 $out/pytest_early.txt
: > $out/early_ab.jsonl
for rep in 1 2 3; do for lib in "" clid-slam_amd/lib/libclid_native_early.so; do
  for args in "--steps 20 --warmup 5" "--steps 200 --warmup 20" "--steps 100 --warmup 10 --layer-norm --freeze-decoder"; do
    CLID_NATIVE_LIB=$lib python bench.py --no-cpu-baseline --frame-calls 0 $args 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(json.dumps({'lib': '$lib' or 'default', 'args': '$args', 'ms_per_step': round(d['ms_per_step'],5), 'kernels_us': {k['kernel'].split(' ')[0]: k['avg_us'] for k in d['roofline']['kernels']}}))" >> $out/early_ab.jsonl
  done; done; done
cat $out/early_ab.jsonl
