#!/bin/bash
# developer tool (GPU box): process_frame with the pool compaction held back until the voxel pass is through (CLID_POOL_GATE=1)
# against the default, alternating runs of 120 frames on one box -> gpurun_out/pool_gate_ab.jsonl
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; : > gpurun_out/pool_gate_ab.jsonl
timeout 600 python -m pytest tests/test_sampler_gpu.py tests/test_mapops_gpu.py -m gpu -x -q 2>&1 | tail -2
for r in 1 2 3; do for m in 0 1; do CLID_POOL_GATE=$m timeout 300 python bench_sequence.py --frames 120 --quiet 2>/dev/null | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); s=j['steady_state']
print(json.dumps({'CLID_POOL_GATE': $m, 'scans_per_s': round(s['scans_per_s'],1), 'process_frame_ms': round(s['median_process_frame_ms'],4), 'mapping_ms': round(s['median_mapping_ms'],4)}))" | tee -a gpurun_out/pool_gate_ab.jsonl; done; done
