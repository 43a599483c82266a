"""Launch-by-launch account of the mapping() calls in a rocprofv3 --kernel-trace CSV: the trace is cut at every
k_mapping_prep launch; for each call with at least `min_iters` decode launches the span (first start -> last end) is split
into kernel time per kernel name and idle time per (previous kernel -> next kernel) pair, and everything that is not the
decode -> Adam chain is listed in launch order.
usage: python tools/trace_calls.py <kernel_trace.csv> [min_iters]"""
import collections
import csv
import json
import re
import sys


def short(n):
    n = re.sub(r"^void ", "", n.split("(")[0])
    n = n.replace("clid::", "")
    return n[:48]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    min_iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ev = [(short(r["Kernel_Name"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
    cuts = [i for i, e in enumerate(ev) if e[0].startswith("k_mapping_prep")] + [len(ev)]
    out = []
    for a, b in zip(cuts, cuts[1:]):
        call = ev[a:b]
        n_dec = sum(1 for e in call if e[0].startswith("k_decode_tile") or e[0].startswith("k_train_analytic"))
        if n_dec < min_iters:
            continue
        # the call ends with the last k_local_to_global (assign_local_to_global) or the last Adam launch
        last = max(i for i, e in enumerate(call) if e[0].startswith("k_adam_all") or e[0].startswith("k_local_to_global"))
        call = call[: last + 1]
        t0 = call[0][1]
        span = call[-1][2] - t0
        dur = collections.defaultdict(lambda: [0, 0])
        gap = collections.defaultdict(lambda: [0, 0])
        seq = []
        for i, (n, s, e) in enumerate(call):
            dur[n][0] += 1
            dur[n][1] += e - s
            g = s - call[i - 1][2] if i else 0
            if i:
                k = (call[i - 1][0], n)
                gap[k][0] += 1
                gap[k][1] += g
            chain = n.startswith("k_decode_tile") or n.startswith("k_adam_all")
            if not chain or i < 8:
                seq.append({"k": n, "at_us": round((s - t0) / 1e3, 2), "dur_us": round((e - s) / 1e3, 2), "gap_before_us": round(g / 1e3, 2)})
        rec = {
            "iters": n_dec, "launches": len(call), "span_us": round(span / 1e3, 2),
            "kernel_us": {k: [v[0], round(v[1] / 1e3, 2)] for k, v in sorted(dur.items(), key=lambda kv: -kv[1][1])},
            "gap_us": {f"{k[0]} -> {k[1]}": [v[0], round(v[1] / 1e3, 2)] for k, v in sorted(gap.items(), key=lambda kv: -kv[1][1])},
            "sum_kernels_us": round(sum(v[1] for v in dur.values()) / 1e3, 2),
            "sum_gaps_us": round(sum(v[1] for v in gap.values()) / 1e3, 2),
            "non_chain_launches": seq,
        }
        out.append(rec)
    for rec in out:
        print(json.dumps(rec))


if __name__ == "__main__":
    main()
