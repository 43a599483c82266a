#!/bin/bash
# developer tool (GPU box): class-preserving batch order (default) against the plain order (CLID_ORDER_CLASSES=0): parity tests,
# then alternating bench runs -> gpurun_out/order_ab.jsonl
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; : > gpurun_out/order_ab.jsonl
timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_sequence.py tests/test_dist_gpu.py tests/test_touched_rows.py -m gpu -q -x 2>&1 | tail -4
for rep in 1 2 3; do for lib in 1 0; do
  for args in "--steps 20 --warmup 5" "--steps 200 --warmup 20" "--config cfg3 --steps 100 --warmup 10"; do
    CLID_ORDER_CLASSES=$lib python bench.py --no-cpu-baseline --frame-calls 0 $args 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(json.dumps({'CLID_ORDER_CLASSES': $lib, 'args': '$args', 'ms_per_step': round(d['ms_per_step'],5), 'kernels_us': {k['kernel'].split(' ')[0]: k['avg_us'] for k in d['roofline']['kernels']}}))" | tee -a gpurun_out/order_ab.jsonl
  done; done; done
for m in 1 0 1 0; do CLID_ORDER_CLASSES=$m timeout 300 python bench_sequence.py --frames 120 --quiet 2>/dev/null | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); s=j['steady_state']
print(json.dumps({'sequence_CLID_ORDER_CLASSES': $m, 'scans_per_s': round(s['scans_per_s'],1), 'process_frame_ms': round(s['median_process_frame_ms'],4), 'mapping_ms': round(s['median_mapping_ms'],4)}))" | tee -a gpurun_out/order_ab.jsonl; done
