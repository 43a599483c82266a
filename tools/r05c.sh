#!/bin/bash
# developer tool (GPU box): round-5 batch c -- LDS-operand decode (3 waves/SIMD, no scratch) parity + A/B, ekional_add_to tests, N1 bound call
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; out=gpurun_out/r05c; mkdir -p $out
CLID_DEBUG_FLAGS=16 timeout 900 python -m pytest tests/test_tile_decode.py -m gpu -q -x > $out/pytest_tile_flag16.txt 2>&1; tail -4 $out/pytest_tile_flag16.txt
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_tile_decode.py -m gpu -q -k "ekional or g8 or tracking or pickle or 262144 or mapping_loop_g6" > $out/pytest_new.txt 2>&1; tail -6 $out/pytest_new.txt
B="python bench.py --no-cpu-baseline --frame-calls 0"
: > $out/wps_ab.jsonl
for rep in 1 2; do for flag in 0 16; do
  for args in "--config cfg3 --steps 100 --warmup 10" "--config cfg4 --steps 50 --warmup 5" "--bs 65536 --decode 1 --steps 100 --warmup 10" "--config cfg3 --bs 262144 --steps 50 --warmup 5"; do
    CLID_DEBUG_FLAGS=$flag $B $args 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(json.dumps({'flags': $flag, 'args': '$args', 'ms_per_step': round(d['ms_per_step'],5), 'kernels_us': {k['kernel'].split(' ')[0]: k['avg_us'] for k in d['roofline']['kernels']}}))" >> $out/wps_ab.jsonl
  done; done; done
cat $out/wps_ab.jsonl
timeout 600 python bench_next.py --no-cpu-baseline > $out/next_rows.jsonl 2> $out/next_rows.err; head -c 1500 $out/next_rows.jsonl; tail -3 $out/next_rows.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --frame-calls 0 | tail -1 | cut -c1-300
