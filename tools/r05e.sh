#!/bin/bash
# developer tool (GPU box): round-5 batch e -- full GPU suite on the current tree + smoke + headline
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; out=gpurun_out/r05e; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -q > $out/pytest_gpu.txt 2>&1; tail -15 $out/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py --steps 20 --warmup 5 > $out/bench_s20.json 2> $out/bench.err; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05e/bench_s20.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("ms_per_step", "value")}, d["roofline"]["binds"], d["roofline"]["issue"] and d["roofline"]["issue"]["issue_frac"], d["roofline"]["counters_from"])
PY
