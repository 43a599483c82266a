#!/bin/bash
# developer tool (GPU box): hand-written exclusive scans (default build) against hipcub::DeviceScan (lib/libclid_native_scanlib.so,
# -DCLID_SCAN_LIB=1) on the sequence workload, alternating -> gpurun_out/scan_ab.jsonl
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; : > gpurun_out/scan_ab.jsonl
timeout 600 python -m pytest tests/test_mapops_gpu.py tests/test_sampler_gpu.py -m gpu -q -x 2>&1 | tail -2
for r in 1 2 3 4 5; do for lib in "" clid-slam_amd/lib/libclid_native_scanlib.so; do CLID_NATIVE_LIB=$lib timeout 300 python bench_sequence.py --frames 200 --quiet 2>/dev/null | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); s=j['steady_state']
print(json.dumps({'scans': 'hipcub' if '$lib' else 'hand-written', 'scans_per_s': round(s['scans_per_s'],1), 'process_frame_ms': round(s['median_process_frame_ms'],4), 'mapping_ms': round(s['median_mapping_ms'],4)}))" | tee -a gpurun_out/scan_ab.jsonl; done; done
cd /tmp; for lib in "" clid-slam_amd/lib/libclid_native_scanlib.so; do rm -rf /tmp/ft; (cd $GRAFT_REPO_ROOT; CLID_NATIVE_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ft -o ft --output-format csv -- python bench_sequence.py --frames 60 --quiet > /dev/null 2>&1); python - <<PY
import csv,glob
f=glob.glob("/tmp/ft/**/ft_kernel_stats.csv", recursive=True)[0]
rows=[r for r in csv.DictReader(open(f))]
tot=0
print("== ${lib:-hand-written}")
for r in rows:
    n=r["Name"]
    if "scan" in n.lower() or "lookback" in n.lower() or "rocprim" in n.lower():
        print("  ", n[:90], r["Calls"], r["AverageNs"])
PY
done
