#!/bin/bash
# developer tool (GPU box): default build against a variant build (lib/libclid_native_$1.so), alternating bench runs -> gpurun_out/lib_ab_$1.jsonl
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; v=$1; : > gpurun_out/lib_ab_$v.jsonl
CLID_NATIVE_LIB=clid-slam_amd/lib/libclid_native_$v.so timeout 600 python -m pytest tests/test_tile_decode.py tests/test_hip_parity.py -m gpu -q -x -k "g6 or tile" 2>&1 | tail -2
for rep in 1 2 3; do for lib in "" clid-slam_amd/lib/libclid_native_$v.so; do
  for args in "--steps 20 --warmup 5" "--steps 200 --warmup 20" "--config cfg3 --steps 100 --warmup 10"; do
    CLID_NATIVE_LIB=$lib python bench.py --no-cpu-baseline --frame-calls 0 $args 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(json.dumps({'lib': '$lib' or 'default', 'args': '$args', 'ms_per_step': round(d['ms_per_step'],5), 'kernels_us': {k['kernel'].split(' ')[0]: k['avg_us'] for k in d['roofline']['kernels']}}))" | tee -a gpurun_out/lib_ab_$v.jsonl
  done; done; done
