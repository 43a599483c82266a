#!/bin/bash
# PMC pass of one bench.py command for the clid kernels: tools/kpmc.sh <tag> "<counters>" [ENV=.. ...] [bench flags]
tag=$1; shift; ctr=$1; shift
envs=(); while [[ "$1" == *=* ]]; do envs+=("$1"); shift; done
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
env "${envs[@]}" timeout 600 rocprofv3 --kernel-trace --pmc $ctr -d gpurun_out/pmc_$tag -o pmc --output-format csv -- \
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --frame-calls 0 "$@" > /dev/null 2> gpurun_out/pmc_$tag.err
python - "$tag" <<'PY'
import csv, collections, glob, re, sys
fn = glob.glob(f"gpurun_out/pmc_{sys.argv[1]}/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(fn)):
    k = r["Kernel_Name"]
    if "clid::" in k:
        agg[re.sub(r"\(.*", "", k).replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    # the LAST launch of each kernel (the profiled 20-iteration pass) and the mean
    print(k[:50], {c: (round(x[-1]), round(sum(x) / len(x))) for c, x in v.items()}, "launches", len(next(iter(v.values()))))
PY
