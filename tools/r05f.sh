#!/bin/bash
# developer tool (GPU box): round-5 batch f -- block-level dW1 flush (parity + A/B against the per-wave build), fused h_model rows
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; out=gpurun_out/r05f; mkdir -p $out
timeout 1200 python -m pytest tests/test_tile_decode.py tests/test_hip_parity.py tests/test_sequence.py tests/test_touched_rows.py -m gpu -q -x > $out/pytest.txt 2>&1; tail -5 $out/pytest.txt
: > $out/blk_ab.jsonl
for rep in 1 2 3; do for lib in "" clid-slam_amd/lib/libclid_native_blk0.so; do
  for args in "--steps 20 --warmup 5" "--steps 200 --warmup 20"; do
    CLID_NATIVE_LIB=$lib python bench.py --no-cpu-baseline --frame-calls 0 $args 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(json.dumps({'lib': '$lib' or 'default (BLK)', 'args': '$args', 'ms_per_step': round(d['ms_per_step'],5), 'kernels_us': {k['kernel'].split(' ')[0]: k['avg_us'] for k in d['roofline']['kernels']}}))" >> $out/blk_ab.jsonl
  done; done; done
cat $out/blk_ab.jsonl
timeout 600 python bench_next.py --no-cpu-baseline 2> $out/next.err | head -1 | cut -c1-900
