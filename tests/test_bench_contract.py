"""bench.py's output contract: one JSON line with the fields the driver reads.  The CPU part
checks the helpers; the GPU part runs the script as the driver does (a small workload)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_valid_pairs_counts_labels_up_to_the_column_index():
    # 4K x 256: the figure the benchmark's unit is defined on
    assert bench.valid_pairs(2160, 3840, 256) == 2052864000
    assert bench.valid_pairs(1, 3, 40) == 1 + 2 + 3
    assert bench.valid_pairs(2, 5, 2) == 2 * (1 + 2 + 2 + 2 + 2)


def test_committed_traffic_record_matches_the_dominant_kernels():
    t = bench.pmc_traffic("4k_256", "ds::fused2_iteration_kernel<float, 64, 12, 0>")
    assert t is not None and t > 3.0e10
    assert bench.pmc_traffic("4k_256", "ds::no_such_kernel") is None
    for name, w in bench.WORKLOADS.items():
        assert w["dtype"] in ("f32", "f64") and w["D"] <= 256, name


@pytest.mark.gpu
def test_bench_line_has_the_fields_the_driver_reads():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "1080p_128", "--steps", "5",
                          "--warmup", "1"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines          # exactly one line on stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["dtype"] == "f32" and d["scaling"] in ("strong", "weak")
    assert d["unit"] == "G pixel*label*iters/s" and "workload" in d["config"]
    assert abs(d["value"] - bench.valid_pairs(1080, 1920, 128) * 5 / (d["ms_per_step"] * 5e-3) / 1e9) <= 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.05 < r["frac"] < 1.0
    assert "traffic" in r and "kernel" in r
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    assert d["value"] > 100 * c["value"]
    # the companions of a one-GPU run: the same handle over several hundred steps
    assert d["sustained"]["steps"] >= 400 and d["sustained"]["ms_per_step"] > 0
    assert r["placement"]["mode"] == "default"   # the headline is timed as a default deployment runs
