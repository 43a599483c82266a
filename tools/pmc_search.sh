#!/bin/bash
# developer tool (GPU box): instruction mix of the search launch, directory walk (flags 0) vs probing (flags 8)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; out=gpurun_out/r04/pmc; mkdir -p $out
CFG=${1:-cfg4}
for f in 0 8; do
  A="python bench.py --config $CFG --steps 32 --warmup 3 --frame-calls 0 --no-cpu-baseline"
  CLID_DEBUG_FLAGS=$f timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_BUSY_CYCLES -d $out/sq_$f -o sq --output-format csv -- $A > /dev/null 2>> $out/log.txt
  CLID_DEBUG_FLAGS=$f timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY -d $out/sq2_$f -o sq --output-format csv -- $A > /dev/null 2>> $out/log.txt
done
python - <<'PY'
import csv, collections, glob, re
for f in (0, 8):
    for tag in ("sq", "sq2"):
        for fn in sorted(glob.glob(f"gpurun_out/r04/pmc/{tag}_{f}/**/*_counter_collection.csv", recursive=True)):
            agg = collections.defaultdict(lambda: collections.defaultdict(list))
            for r in csv.DictReader(open(fn)):
                k = r["Kernel_Name"]
                if "k_search" in k:
                    agg[re.sub(r"\(.*", "", k).replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
            for k, v in agg.items():
                print("flags", f, k[:40], {c: round(sum(x) / len(x)) for c, x in v.items()}, "launches", len(next(iter(v.values()))))
PY
tail -3 $out/log.txt
