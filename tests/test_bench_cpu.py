"""CPU checks of bench.py's bookkeeping: a committed PMC traffic record rides along only when it was measured on the kernel
sources the library in use was built from (the `csrc_hash` stamp)."""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_committed_traffic_records_carry_a_source_stamp():
    """Every round-3 PMC record names the kernel sources it was measured on; bench.py passes it on only when that stamp is
    the one of the library in use (while kernels are being edited the records are stale and the line says so)."""
    import bench
    here = bench.csrc_hash()
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r03_*_hbm_traffic.json")))
    assert len(files) >= 5, files
    for f in files:
        stamp = json.load(open(f)).get("csrc_hash")
        assert isinstance(stamp, str) and len(stamp) == 12, f
        rec, at = bench.stamped_traffic(os.path.basename(f))
        assert (rec is not None) == (stamp == here) and at is not None


def test_stale_traffic_record_is_dropped(monkeypatch):
    import bench
    monkeypatch.setattr(bench, "csrc_hash", lambda: "0" * 12)
    rec, at = bench.stamped_traffic("r03_wino4_gemm_hbm_traffic.json")
    assert rec is None and "dropped" in at
    rec, at = bench.stamped_traffic("no_such_file.json")
    assert rec is None and at is None


def test_default_bench_line_of_the_round_carries_both_halves_of_the_metric():
    """profiles/r03_bench_default.json is the line `python bench.py` printed on the MI355X for the sources in the tree."""
    d = json.load(open(os.path.join(ROOT, "profiles", "r03_bench_default.json")))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "train"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["scaling"] == "weak" and d["data"] == "synthetic" and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert d["cpu_baseline"]["kind"] in ("port", "reference") and d["cpu_baseline"]["cores"] >= 1
    t = d["train"]
    assert t["unit"] == "samples/s" and t["value"] > 0 and len(t["loss_terms"]) == 6
    assert abs(t["roofline"]["frac"] - t["roofline"]["achieved"] / t["roofline"]["peak"]) < 1e-3
