cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/pmc4
for f in 0; do
CLID_DEBUG_FLAGS=$f timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY -d gpurun_out/pmc4 -o f$f --output-format csv -- python bench.py --no-cpu-baseline --steps 6 --warmup 2 > gpurun_out/pmc4/log$f.txt 2>&1
CLID_DEBUG_FLAGS=$f timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS_ATOMIC SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d gpurun_out/pmc4 -o g$f --output-format csv -- python bench.py --no-cpu-baseline --steps 6 --warmup 2 > gpurun_out/pmc4/logg$f.txt 2>&1
done
python - <<'PY'
import csv, collections, glob
for fn in sorted(glob.glob("gpurun_out/pmc4/*_counter_collection.csv")):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(fn)):
        if "fused" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(fn.split("/")[-1].split("_")[0], {c: round(sum(x)/len(x)/3280) for c,x in agg.items()})
PY
