"""The bench line's contract, checked on the committed lines of the current round (profiles/r04_*_bench.json, written by
bench.py on the GPU box): the keys the driver reads, the metric of BASELINE.json, and the arithmetic that ties
`roofline.achieved` to the algorithmic bytes and the dominant kernel's measured launch time."""
from __future__ import annotations

import importlib.util
import json
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
LINES = sorted((ROOT / "profiles").glob("r04_*_bench.json")) or sorted((ROOT / "profiles").glob("r03_*_bench.json"))


def _bench_module():
    spec = importlib.util.spec_from_file_location("bench_contract", ROOT / "bench.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_lines_exist():
    assert LINES, "no committed bench line of the current round under profiles/"


@pytest.mark.parametrize("path", LINES, ids=lambda p: p.name)
def test_bench_line_contract(path):
    d = json.loads(path.read_text())
    base = json.loads((ROOT / "BASELINE.json").read_text())
    assert d["metric"] == base["metric"] and d["unit"] == "grid-cells/s"
    for k in ("value", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in d, k
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32"
    assert "synthetic" in d["data"] and "workload" in d["config"] and "model" not in d["config"]
    cells = d["config"]["cells_per_gpu"] * d["n_gpus"]
    assert d["value"] == pytest.approx(cells / (d["ms_per_step"] * 1e-3), rel=1e-3)
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-4)
    assert r["achieved"] == pytest.approx(r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9, rel=1e-4)
    assert r["traffic"] is None or r["traffic"] >= 0.9 * r["algorithmic_bytes_per_launch"]
    if path.name.startswith("r04"):
        # provenance of the counter figure: the committed PMC file it was read from, or null + the reason
        src = r["traffic_source"]
        if r["traffic"] is None:
            assert src.startswith("none:")
        else:
            f = ROOT / src.split(" ")[0]
            assert f.is_file() and f.parent.name == "profiles", src
            k = json.loads(f.read_text())["kernels"][r["kernel"].split("(")[0]]
            assert k["traffic_bytes"] == r["traffic"]
    # the algorithmic bytes are SURVEY §8d's per-cell figure times the cells of one launch
    b = _bench_module()
    method = d["config"]["workload"].split(":")[0].split("_")[0]
    tb, ab = b.algorithmic_bytes_per_cell(method, d["config"]["days"], d["config"]["groups"], 20, trend_array=False)
    per_cell = tb if r["kernel"].startswith("k_train") else ab
    assert r["algorithmic_bytes_per_launch"] == per_cell * d["config"]["cells_per_gpu"]
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["unit"] == "grid-cells/s" and c["cores"] >= 1 and c["value"] > 0 and c["sample"]


def test_algorithmic_bytes_match_the_survey():
    b = _bench_module()
    # cfg3-like EQM/DQM doy tables, T = 10950, G = 365, Q = 20
    assert b.algorithmic_bytes_per_cell("eqm", 10950, 365, 20) == (146000, 146000)
    assert b.algorithmic_bytes_per_cell("dqm", 10950, 365, 20) == (147460, 191260)
    assert b.algorithmic_bytes_per_cell("dqm", 10950, 365, 20, trend_array=False) == (147460, 147460)  # the timed step: no trend array
    assert b.algorithmic_bytes_per_cell("qdm", 10950, 365, 20) == (146000, 116800)
    assert b.algorithmic_bytes_per_cell("eqm", 10950, 12, 20) == (89520, 89520)


def test_default_size_is_one_per_gpu_problem_at_every_n():
    """129 600 cells per GPU (north_star's 1440x720 cube over 8) whatever N: BENCH's N = 1 is SCALE's N = 1"""
    b = _bench_module()
    assert b.DEFAULT_CELLS_PER_GPU == 1440 * 720 // 8 == 129600
    src = (ROOT / "bench.py").read_text()
    main = src[src.index("def main():"):src.index("def gather_leg(")]
    assert "args.cells = DEFAULT_CELLS_PER_GPU" in main and "world == 8" not in main


def test_traffic_provenance():
    b = _bench_module()
    t, src = b.pmc_traffic("eqm_doy31", 64800, 10950, "k_train_cell3")
    assert t is not None and src.startswith("profiles/") and (ROOT / src.split(" ")[0]).is_file()
    t, src = b.pmc_traffic("eqm_doy31", 12345, 10950, "k_train_cell3")
    assert t is None and src.startswith("none:") and "12345" in src


def test_cpu_sample_downloads_only_its_rows(monkeypatch):
    """cpu_baseline / api_level fetch a VIEW of the first n series, never the cube"""
    import numpy as np

    from xsdba_amd import _lib

    seen = []

    def fake_download(self):
        seen.append(self.nbytes)
        return np.zeros(self.size, self.dtype)

    monkeypatch.setattr(_lib.DeviceArray, "download", fake_download)
    b = _bench_module()
    cube = _lib.DeviceArray.from_ptr(None, 4096, (129600, 10950), np.float32)
    rows = b.first_rows(None, cube, 8, 10950)
    assert rows.shape == (8, 10950) and seen == [8 * 10950 * 4]
    src = (ROOT / "bench.py").read_text()
    body = src[src.index("def cpu_baseline("):src.index('if __name__ == "__main__"')]
    assert "dev.download()" not in body and "first_rows(ctx, dev, n, T)" in body
