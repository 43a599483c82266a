#!/bin/bash
# developer tool (GPU box): rocprofv3 kernel stats of the driver's command shape, rows matching $1 (regex)  [tree dir as $2]
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT/${2:-.}"; out=$GRAFT_REPO_ROOT/gpurun_out/r06/ks_$$; mkdir -p $out
timeout 600 rocprofv3 --kernel-trace --stats -d $out -o s --output-format csv -- python bench.py --no-cpu-baseline --frame-calls 0 --steps 20 --warmup 5 > /dev/null 2>> $GRAFT_REPO_ROOT/gpurun_out/r06/log.txt
f=$(find $out -name 's_kernel_stats.csv' | head -1)
python - "$f" "$1" <<'PY'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], r["Name"]):
        print("%-60s calls %4s avg %9.2f us min %9.2f max %9.2f" % (re.sub(r"\(.*", "", r["Name"])[:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
rm -rf $out
