"""The line ``bench.py`` prints is the record the driver parses out of a 16-KB tail of stdout: it must stay ONE compact JSON object
(< 4 KB) carrying the contract's keys, ``roofline`` and ``cpu_baseline`` -- round 4's 24-KB line left the driver's record unparsed.
Input here: the full records of earlier runs (``profiles/archive/r04_bench.json``: the 24-KB N = 1 record itself) pushed through
``bench.compact_line`` / ``bench.emit``; no GPU, no timing."""
import json
import os
import sys
import types

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench   # noqa: E402

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline")


def _full_n1():
    return json.load(open(os.path.join(REPO, "profiles", "archive", "r04_bench.json")))


def test_compact_line_of_the_round4_record(tmp_path, monkeypatch):
    full = _full_n1()
    assert len(json.dumps(full)) > 16384            # (the record that did not parse)
    monkeypatch.setattr(bench, "REPO", str(tmp_path))
    line = bench.emit(full, types.SimpleNamespace())
    assert "\n" not in line and len(line) < bench.LINE_LIMIT == 4096, len(line)
    r = json.loads(line)
    for k in CONTRACT:
        assert k in r, k
    assert r["value"] == pytest.approx(full["value"], rel=1e-4) and r["ms_per_step"] == pytest.approx(full["ms_per_step"], rel=1e-4)
    assert set(("workload", "mode", "candidates_per_step", "horizon_steps")) <= set(r["config"])
    rf = r["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms", "bytes_per_launch"):
        assert k in rf, k
    assert rf["bound"] == "hbm" and rf["frac"] == pytest.approx(rf["achieved"] / rf["peak"], rel=1e-3)
    cb = r["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["all_cores"]["cores"] >= 1 and cb["numpy_loop"]["value"] > 0
    assert set(("p50", "p90")) <= set(r["plan_latency_ms"])
    assert "cfg4" in r["side_configs"] and len(r["side_configs"]["cfg4"]) == 4
    # everything else is in the detail file the line names
    detail = json.load(open(os.path.join(str(tmp_path), r["detail"])))
    assert detail["configs"]["cfg4"]["draw"]["roofline"]["kernel"] == full["configs"]["cfg4"]["draw"]["roofline"]["kernel"]


def test_line_sheds_riders_not_contract_keys(tmp_path, monkeypatch):
    """should side records ever outgrow the limit, riders go, the contract's keys stay"""
    full = _full_n1()
    full["configs"] = {f"cfg{i}x": full["configs"]["cfg4"] for i in range(400)}
    monkeypatch.setattr(bench, "REPO", str(tmp_path))
    line = bench.emit(full, types.SimpleNamespace())
    r = json.loads(line)
    assert len(line) < 4096 and "side_configs" not in r
    for k in CONTRACT + ("cpu_baseline",):
        assert k in r, k


def test_compact_line_multi_gpu_keys():
    full = _full_n1()
    full.update(n_gpus=8, exchange="MailboxExchange", exchange_ms_per_step=0.0071, ranks_seen_by_rccl=8, wait_mode="yield",
                other_transport={"value": 3.1e9, "ms_per_step": 0.16, "exchange": "CollectiveExchange", "exchange_ms_per_step": 0.05},
                one_gpu_same_grid={"value": 4.5e8, "ms_per_step": 0.138}, strong_scaling={"value": 2.0e9, "ms_per_step": 0.25},
                rehearsal=False, strong={"sharded": full["configs"]["cfg4"]}, weak={"sharded": full["configs"]["cfg4"]})
    r = bench.compact_line(full, "gpurun_out/bench_detail_n8.json")
    line = json.dumps(r, separators=(",", ":"))
    assert len(line) < 4096
    for k in ("exchange", "exchange_ms_per_step", "other_transport", "ranks_seen_by_rccl", "one_gpu_same_grid", "strong_scaling", "detail"):
        assert k in r, k
    assert "strong" not in r and "weak" not in r


def test_numpy_loop_calibration_file():
    """tests/golden/cpu_calibration.json (written by tests/golden/time_reference.py in the build container): data only"""
    cal = json.load(open(os.path.join(REPO, "tests", "golden", "cpu_calibration.json")))
    assert set(cal["cases"]) == {"cfg1_ref_l3", "cfg2_ref"}
    for c in cal["cases"].values():
        assert c["reference_candidates_per_s"] > 0 and c["numpy_loop_candidates_per_s"] > 0
    assert 0.5 < cal["ratio_numpy_loop_to_reference"] < 2.0   # the restatement runs the reference's kind of program


def test_committed_round_records_belong_to_the_committed_library():
    """the judged records under profiles/ name the library they were measured on (its source hash): they must be the library the committed
    sources build -- a kernel change without a new measurement pass shows up here -- and the committed bench line must be what the driver
    can parse (one object, < 4 KB, roofline consistent with its own numbers)."""
    from commonroad_rp_amd import _capi
    here = _capi.source_hash()
    prof = os.path.join(REPO, "profiles")
    for name in ("r05_pmc_traffic.json", "r05_fp64_flops.json"):
        d = json.load(open(os.path.join(prof, name)))
        hashes = {v.get("source_hash") for v in d.values() if isinstance(v, dict)}
        assert hashes == {here}, (name, hashes, here)
    for name in ("r05_sq_cfg3.json", "r05_sq_cfg5_fused.json"):
        assert json.load(open(os.path.join(prof, name)))["source_hash"] == here, name
    for name in ("r05_fuzz_parity.txt", "r05_full_scale_parity.txt"):
        assert here in open(os.path.join(prof, name)).readline(), name
    text = open(os.path.join(prof, "r05_bench.json")).read().strip()
    assert "\n" not in text and len(text) < 4096
    r = json.loads(text)
    for k in CONTRACT + ("cpu_baseline",):
        assert k in r, k
    rf = r["roofline"]
    assert rf["frac"] == pytest.approx(rf["achieved"] / rf["peak"], rel=1e-3)
    assert rf["achieved"] == pytest.approx(rf["bytes_per_launch"] / (rf["kernel_ms"] * 1e-3) / 1e9, rel=1e-3)
    assert rf["traffic"] is not None and rf["traffic"] >= rf["bytes_per_launch"]          # (counters of THIS library: else bench.py reports null)
    assert r["value"] == pytest.approx(r["config"]["candidates_per_step"] / (r["ms_per_step"] * 1e-3), rel=1e-3)
    assert rf["kernel_ms"] < r["ms_per_step"]
