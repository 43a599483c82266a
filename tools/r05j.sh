#!/bin/bash
# developer tool (GPU box): round-5 batch j -- decode: operands-first block flush + paired merge chains (default build) against
# the build before them (lib/libclid_native_early.so); parity on the default build; host phases
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; out=gpurun_out/r05j; mkdir -p $out
timeout 1200 python -m pytest tests/test_tile_decode.py tests/test_hip_parity.py tests/test_sequence.py tests/test_touched_rows.py -m gpu -q -x > $out/pytest.txt 2>&1; tail -4 $out/pytest.txt
: > $out/ab.jsonl
for rep in 1 2 3; do for lib in "" clid-slam_amd/lib/libclid_native_early.so; do
  for args in "--steps 20 --warmup 5" "--steps 200 --warmup 20" "--steps 100 --warmup 10 --layer-norm --freeze-decoder"; do
    CLID_NATIVE_LIB=$lib python bench.py --no-cpu-baseline --frame-calls 0 $args 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(json.dumps({'lib': '$lib' or 'default', 'args': '$args', 'ms_per_step': round(d['ms_per_step'],5), 'kernels_us': {k['kernel'].split(' ')[0]: k['avg_us'] for k in d['roofline']['kernels']}}))" >> $out/ab.jsonl
  done; done; done
cat $out/ab.jsonl
python tools/mapping_host_phases.py 2>&1 | tail -7
