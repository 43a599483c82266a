# developer tool: HBM traffic of the training kernels (separate --pmc passes, MI355X_MICROARCH.md section HBM)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/pmc6
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc6 -o fetch --output-format csv -- python bench.py --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/pmc6/log_f.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc6 -o write --output-format csv -- python bench.py --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/pmc6/log_w.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum -d gpurun_out/pmc6 -o ea --output-format csv -- python bench.py --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/pmc6/log_e.txt 2>&1
python - <<'PY'
import csv, collections, glob
for fn in sorted(glob.glob("gpurun_out/pmc6/*_counter_collection.csv")):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fn)):
        k=r["Kernel_Name"]
        if "clid::" in k: agg[k[:28]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items(): print(fn.split("/")[-1].split("_")[0], k, {c: round(sum(x)/len(x),1) for c,x in v.items()})
PY
