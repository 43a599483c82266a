#!/bin/bash
# one bench.py run (driver's command shape), printing ms/step, the timed region's split and the per-kernel dispatch durations
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --frame-calls ${FRAME_CALLS:-0} "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['ms_per_step'],5), {k:round(v,3) for k,v in d['timed_region_split'].items()}, d['per_frame_regime'] and round(d['per_frame_regime']['median_ms_per_call'],4))
for k in d['roofline']['kernels']: print('   ', k['kernel'][:60], k['avg_us'])"
