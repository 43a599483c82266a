#!/bin/bash
# developer tool (GPU box): record requests of the analytic-eikonal iteration one round ahead (default) against -DCLID_ANALYTIC_PF=0
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; : > gpurun_out/analytic_pf_ab.jsonl
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_dist_gpu.py -m gpu -q -x -k "analytic or g6 or g4" 2>&1 | tail -2
for rep in 1 2 3; do for lib in "" clid-slam_amd/lib/libclid_native_anopf.so; do
  for args in "--analytic --steps 100 --warmup 10" "--analytic --bs 65536 --steps 50 --warmup 5"; do
    CLID_NATIVE_LIB=$lib python bench.py --no-cpu-baseline --frame-calls 0 $args 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(json.dumps({'lib': 'no prefetch' if '$lib' else 'prefetch', 'args': '$args', 'ms_per_step': round(d['ms_per_step'],5), 'kernels_us': {k['kernel'].split(' ')[0]: k['avg_us'] for k in d['roofline']['kernels']}}))" | tee -a gpurun_out/analytic_pf_ab.jsonl
  done; done; done
