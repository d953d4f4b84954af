"""The line bench.py prints is what the driver parses out of the last 8 KB of stdout (the r04 line had grown to
27.6 KB and was recorded as `parsed: null`).  CPU tests of the compaction on canned full records."""
import copy
import importlib.util
import json
import os

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config")


def _bench_module():
    spec = importlib.util.spec_from_file_location("ego_bench", os.path.join(REPO, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _canned():
    return json.load(open(os.path.join(REPO, "profiles", "r04", "s6_bench_line.json")))   # a real full record (27.6 KB as one line)


def _check(line: dict, full: dict):
    text = json.dumps(line, separators=(",", ":"))
    assert len(text.encode()) < 6000, len(text)
    back = json.loads(text, parse_constant=lambda c: (_ for _ in ()).throw(ValueError(c)))   # strict: no NaN / Infinity tokens
    for k in CONTRACT:
        assert k in back, k
    assert back["config"]["workload"].startswith("OmniBlender barbershop") and "configs[1]" in back["config"]["workload"]
    rf = back["roofline"]
    for k in ("bound", "kernel", "unit", "achieved", "peak", "frac", "traffic", "ms"):
        assert k in rf, k
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-5
    # every `frac` in the line is a physical fraction - an algorithmic byte rate that the caches inflate above the
    # HBM peak goes to `algorithmic_frac`, never to `frac`
    for name, r in [("headline", rf)] + [(n, ln.get("roofline") or {}) for n, ln in (back.get("secondary") or {}).items() if "error" not in ln]:
        assert r.get("frac") is None or 0 < r["frac"] <= 1.0, (name, r)
    cb = back["cpu_baseline"]
    assert set(cb) == {"value", "unit", "cores", "kind", "sample"} and cb["kind"] in ("port", "reference")
    assert back["parity"]["max_abs_rgb_err"] <= 1e-4
    assert abs(back["value"] - full["value"]) <= 1e-5 * full["value"]
    assert abs(back["ms_per_step"] - full["ms_per_step"]) <= 1e-5 * full["ms_per_step"]
    return back


def test_compact_line_of_a_real_record_fits_the_drivers_window():
    b = _bench_module()
    full = _canned()
    assert len(json.dumps(full)) > 20000
    back = _check(b.compact_line(full, [os.path.join(REPO, "bench_full.json")]), full)
    assert set(back["secondary"]) == set(full["secondary"])
    for name, ln in back["secondary"].items():
        assert ln["value"] > 0 and ln["ms_per_step"] > 0 and ln["roofline"]["bound"] in ("hbm", "mfma")
    assert back["full_record"] == ["bench_full.json"]
    # what was measured in the bench process and what was read from the tracked counter passes is said in the line itself
    assert back["roofline"]["measured_here"] == ["ms", "achieved", "frac"]
    assert back["roofline"]["from_counter_pass"]["file"].startswith("profiles/")


def test_an_algorithmic_byte_rate_above_the_hbm_peak_is_never_printed_as_frac():
    b = _bench_module()
    full = copy.deepcopy(_canned())
    full["secondary"]["render_big_grid"]["roofline"].update(bound="hbm", frac=1.98, achieved=15840.0, peak=8000.0, hbm_counter_frac=0.36, traffic=1.79e9)
    full["secondary"]["render_fresh_rays"]["roofline"].update(bound="hbm", frac=2.27, achieved=18160.0, peak=8000.0, hbm_counter_frac=None, traffic=None)
    back = _check(b.compact_line(full, []), full)
    big, fresh = back["secondary"]["render_big_grid"]["roofline"], back["secondary"]["render_fresh_rays"]["roofline"]
    assert big["frac"] == 0.36 and big["algorithmic_frac"] == 1.98
    assert fresh["frac"] is None and fresh["algorithmic_frac"] == 2.27
    for k in ("issue_frac", "l1_frac"):   # what binds the headline kernel is in the driver's record itself
        assert 0 < back["roofline"][k] <= 1.2, k


def test_compact_line_sheds_optional_blocks_rather_than_outgrow_the_window():
    b = _bench_module()
    full = _canned()
    fat = copy.deepcopy(full)
    for k in range(40):   # forty more secondaries: the optional block must go, the contract keys must stay
        fat["secondary"][f"extra_{k}"] = copy.deepcopy(full["secondary"]["erp"])
    fat["config"]["workload"] = fat["config"]["workload"] + " x" * 4000
    fat["cpu_baseline"]["sample"] = "s" * 5000
    fat["dtype"] = "d" * 3000
    back = _check(b.compact_line(fat, []), fat)
    assert "secondary" not in back


def test_compact_line_handles_missing_blocks_and_non_finite_numbers():
    b = _bench_module()
    full = _canned()
    lean = {k: v for k, v in full.items() if k not in ("secondary", "parity", "cpu_baseline")}
    lean["cpu_baseline"] = None
    lean["roofline"] = dict(full["roofline"], traffic=None, hbm_counter_frac=float("nan"))
    line = b.compact_line(lean, [])
    text = json.dumps(line, separators=(",", ":"))
    assert "NaN" not in text and len(text) < 6000
    back = json.loads(text)
    assert back["cpu_baseline"] is None and back["parity"] is None and back["roofline"]["traffic"] is None
    assert back["roofline"]["from_counter_pass"] is None
