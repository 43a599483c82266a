cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/pmc5
rocprofv3 -L 2>/dev/null | grep -oE "TCP_[A-Z_0-9a-z]+|TCC_[A-Z_0-9a-z]+|TA_[A-Z_0-9a-z]+" | sort -u | tr "\n" " " > gpurun_out/pmc5/counters.txt
for f in 0; do
CLID_DEBUG_FLAGS=$f timeout 300 rocprofv3 --kernel-trace --pmc TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_ATOMIC_sum -d gpurun_out/pmc5 -o f$f --output-format csv -- python bench.py --no-cpu-baseline --steps 6 --warmup 2 > gpurun_out/pmc5/log$f.txt 2>&1
done
python - <<'PY'
import csv, collections, glob
for fn in sorted(glob.glob("gpurun_out/pmc5/f*_counter_collection.csv")):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(fn)):
        if "fused" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(fn.split("/")[-1].split("_")[0], {c: round(sum(x)/len(x)) for c,x in agg.items()})
PY
tail -3 gpurun_out/pmc5/log0.txt | cut -c1-300
