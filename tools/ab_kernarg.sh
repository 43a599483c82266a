#!/bin/bash
# developer tool (GPU box): HIP runtime knobs that move the cost of a launch boundary: HIP_FORCE_DEV_KERNARG (kernel arguments in device
# memory instead of host-coherent memory: the ~600 bytes of by-value structs of every launch are fetched across PCIe otherwise)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; : > gpurun_out/kernarg_ab.jsonl
for k in 0 1; do echo "HIP_FORCE_DEV_KERNARG=$k"; HIP_FORCE_DEV_KERNARG=$k timeout 100 tools/ubench_chain.bin 200 | head -1; done
for rep in 1 2 3; do for k in 0 1; do
  for args in "--steps 20 --warmup 5" "--steps 200 --warmup 20" "--config cfg3 --steps 100 --warmup 10"; do
    HIP_FORCE_DEV_KERNARG=$k python bench.py --no-cpu-baseline --frame-calls 0 $args 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(json.dumps({'HIP_FORCE_DEV_KERNARG': $k, 'args': '$args', 'ms_per_step': round(d['ms_per_step'],5), 'kernels_us': {k['kernel'].split(' ')[0]: k['avg_us'] for k in d['roofline']['kernels']}}))" | tee -a gpurun_out/kernarg_ab.jsonl
  done; done; done
for k in 0 1 0 1; do HIP_FORCE_DEV_KERNARG=$k timeout 300 python bench_sequence.py --frames 120 --quiet 2>/dev/null | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); s=j['steady_state']
print(json.dumps({'sequence_HIP_FORCE_DEV_KERNARG': $k, 'scans_per_s': round(s['scans_per_s'],1), 'process_frame_ms': round(s['median_process_frame_ms'],4), 'mapping_ms': round(s['median_mapping_ms'],4)}))" | tee -a gpurun_out/kernarg_ab.jsonl; done
