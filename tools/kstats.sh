#!/bin/bash
# rocprofv3 kernel stats of one bench.py command: tools/kstats.sh <tag> [env assignments and bench flags...]
# e.g. tools/kstats.sh s2 CLID_SEARCH=2 --config cfg2     -> gpurun_out/prof_<tag>/..._kernel_stats.csv, top kernels printed
tag=$1; shift
envs=(); while [[ "$1" == *=* ]]; do envs+=("$1"); shift; done
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
env "${envs[@]}" timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o stats --output-format csv -- \
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --frame-calls 20 "$@" > gpurun_out/prof_$tag.json 2> gpurun_out/prof_$tag.err
python - "$tag" <<'PY'
import csv, glob, sys
f = glob.glob(f"gpurun_out/prof_{sys.argv[1]}/**/*kernel_stats.csv", recursive=True)
for r in list(csv.DictReader(open(f[0])))[:14]:
    print(f'{r["Name"][:90]:90s} calls {r["Calls"]:>6s} total_us {float(r["TotalDurationNs"])/1e3:10.1f} avg_us {float(r["AverageNs"])/1e3:8.2f}')
PY
