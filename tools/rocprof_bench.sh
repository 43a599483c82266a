#!/bin/bash
# developer tool: rocprofv3 kernel trace + stats of bench.py (args after the output tag are passed to bench.py)
tag=${1:-r02}; shift
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; out=gpurun_out/prof_$tag; mkdir -p "$out"
timeout 600 rocprofv3 --kernel-trace --stats -d "$out" -o stats --output-format csv -- python bench.py --no-cpu-baseline --frame-calls 0 "$@" > "$out/bench.json" 2> "$out/log.txt"
f=$(find "$out" -name 'stats_kernel_stats.csv' | head -1); t=$(find "$out" -name 'stats_kernel_trace.csv' | head -1)
head -12 "$f"
python tools/trace_gaps.py "$t" 900 | grep -v "n=   [0-9] "
python - "$out/bench.json" <<'EOF'
import json, sys
d = json.load(open(sys.argv[1])); print(round(d["ms_per_step"]*1e3, 2), "us/step", [(k["kernel"].split(" ")[0], k["avg_us"]) for k in d["roofline"]["kernels"]])
EOF
