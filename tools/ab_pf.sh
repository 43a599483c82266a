#!/bin/bash
# developer tool (GPU box): LDS-DMA record prefetch of the multi-tile decode launches (default) against -DCLID_TILE_PF=0
# (lib/libclid_native_nopf.so): parity at 65 536 / 262 144 samples, then alternating bench runs -> gpurun_out/pf_ab.jsonl
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; : > gpurun_out/pf_ab.jsonl
timeout 900 python -m pytest tests/test_tile_decode.py tests/test_hip_parity.py -m gpu -q -x -k "tile or 65536 or 262144 or large or sharded or full_size or sweep" 2>&1 | tail -3
for rep in 1 2 3; do for lib in "" clid-slam_amd/lib/libclid_native_nopf.so; do
  for args in "--config cfg3 --steps 100 --warmup 10" "--config cfg4 --steps 50 --warmup 5" "--bs 65536 --decode 1 --steps 100 --warmup 10" "--steps 200 --warmup 20"; do
    CLID_NATIVE_LIB=$lib python bench.py --no-cpu-baseline --frame-calls 0 $args 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(json.dumps({'lib': 'no prefetch' if '$lib' else 'prefetch', 'args': '$args', 'ms_per_step': round(d['ms_per_step'],5), 'kernels_us': {k['kernel'].split(' ')[0]: k['avg_us'] for k in d['roofline']['kernels']}}))" | tee -a gpurun_out/pf_ab.jsonl
  done; done; done
