#!/bin/bash
# N back-to-back runs of the driver's command shape: the spread of ms_per_step and of the timed region's split
# usage: tools/bench_repeat.sh N [extra bench.py flags]
n=${1:-10}; shift
for i in $(seq 1 $n); do
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --frame-calls 0 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['ms_per_step'],5), {k:round(v,3) for k,v in d['timed_region_split'].items()})"
done
