# developer tool: kernel stats + HBM traffic of the training kernels (separate --pmc passes, MI355X_MICROARCH.md section HBM)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/pmc8
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/pmc8 -o stats --output-format csv -- python bench.py --no-cpu-baseline > gpurun_out/pmc8/bench_stats.json 2> gpurun_out/pmc8/log_s.txt
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc8 -o fetch --output-format csv -- python bench.py --no-cpu-baseline --steps 64 --warmup 5 > gpurun_out/pmc8/log_f.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc8 -o write --output-format csv -- python bench.py --no-cpu-baseline --steps 64 --warmup 5 > gpurun_out/pmc8/log_w.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_BUSY_CYCLES -d gpurun_out/pmc8 -o sq --output-format csv -- python bench.py --no-cpu-baseline --steps 64 --warmup 5 > gpurun_out/pmc8/log_q.txt 2>&1
python - <<'PY'
import csv, collections, glob, re
for fn in sorted(glob.glob("gpurun_out/pmc8/**/*_counter_collection.csv", recursive=True)):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fn)):
        k=r["Kernel_Name"]
        if "clid::" in k: agg[re.sub(r"\(.*", "", k).replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items(): print(fn.split("/")[-1].split("_")[0], k, {c: round(sum(x)/len(x),1) for c,x in v.items()}, "launches", len(next(iter(v.values()))))
for fn in sorted(glob.glob("gpurun_out/pmc8/**/stats_kernel_stats.csv", recursive=True)):
    print(open(fn).read()[:3000])
PY
