"""CPU: the committed bench line (profiles/r01_bench_v6_default_run.json, produced by `python bench.py` on an
MI355X) carries every key of the driver's contract plus the roofline / cpu_baseline objects, with consistent
arithmetic; bench.py's flags parse and default to a single GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_recorded_bench_line_matches_the_contract():
    d = json.load(open(os.path.join(ROOT, "profiles", "r01_bench_v6_default_run.json")))
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"].split(",")[0] == base["metric"].split(",")[0]
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    # value = samples per step / step time
    assert abs(d["value"] - d["config"]["global_batch"] / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) <= 1e-9
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e9) <= 1e-6 * r["achieved"]
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0


def test_bench_flags_parse():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in out.stdout


def test_newest_committed_headline_was_measured_on_these_sources():
    """The newest profiles/rNN_bench_cfg2_steps20.json (the driver's command shape, kept per round) carries the sha256 of the kernel
    sources, the C header and the host side of mapping() it was measured on (bench.source_stamp): it must be THIS tree's.  A change to
    csrc/ or mapper.py after the round's measurement set fails here until tools/profile_rNN.sh + collect_rNN.sh are run again
    (the round-5 set predated the last kernel and host changes)."""
    import glob
    import re

    sys.path.insert(0, ROOT)
    import bench

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_cfg2_steps20.json")),
                   key=lambda f: int(re.search(r"r(\d+)_", os.path.basename(f)).group(1)))
    assert files, "no committed headline run"
    newest = json.load(open(files[-1]))
    assert "source_stamp" in newest, f"{os.path.basename(files[-1])} carries no source stamp (measured before round 6?)"
    assert newest["source_stamp"] == bench.source_stamp(), (
        f"{os.path.basename(files[-1])} was measured on other sources ({newest['source_stamp']}) than this tree "
        f"({bench.source_stamp()}): re-run the measurement set")
    assert newest["steps"] == 20 and newest["warmup"] == 5 and newest["n_gpus"] == 1
