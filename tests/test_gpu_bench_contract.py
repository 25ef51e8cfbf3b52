"""bench.py's output contract: ONE line of JSON on stdout (whatever the
libraries underneath print), with the fields the driver reads."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def run_bench(*extra, env=None):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] +
                       list(extra), cwd=ROOT, capture_output=True, text=True,
                       timeout=600, env=dict(os.environ, **(env or {})))
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [x for x in p.stdout.split("\n") if x.strip()]
    assert len(lines) == 1, p.stdout[:2000]
    return json.loads(lines[0])


def test_headline_line():
    d = run_bench("--slices", "256", "--steps", "2", "--warmup", "1",
                  "--verify", "2", "--cpu-sample", "4", "--no-extras")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup",
              "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"] == "bitstream_parsed_mbit_per_s" and d["unit"] == "Mbit/s"
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["dtype"] == "u8"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    assert r["kernel"] == "h264_amd_k_decode_slices" and r["kernel_ms"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0
    assert "sample" in c and c["unit"] == "Mbit/s"
    assert d["verified_slices_vs_oracle"] == 2
    assert d["value"] > 1000 * c["value"] / 1000  # both positive, same unit


def test_one_line_under_rccl_and_the_other_workloads():
    # one rank over RCCL: the library prints a banner when its communicator
    # comes up; it must not reach stdout
    d = run_bench("--slices", "256", "--steps", "1", "--warmup", "1",
                  "--verify", "0", "--no-extras",
                  env={"H264_AMD_FORCE_DIST": "1", "MASTER_PORT": "29611"})
    assert d["n_gpus"] == 1
    d = run_bench("--workload", "c5", "--slices", "256", "--steps", "2",
                  "--warmup", "1")
    assert d["metric"] == "bitstream_rewritten_mbit_per_s"
    assert d["roofline"]["kernel"] == "h264_amd_k_write_slices"
    assert d["byte_identical_slices_checked_per_rank"] >= 1
    d = run_bench("--workload", "c4", "--mib", "8", "--steps", "1",
                  "--warmup", "1")
    assert d["scaling"] == "strong" and d["n_gpus"] == 1
    assert d["verified_slices_vs_oracle"] >= 1
    d = run_bench("--workload", "c3", "--slices", "64", "--steps", "2",
                  "--warmup", "1")
    assert d["roofline"]["kernel"] == "h264_amd_k_decode_cabac"
    assert d["bins_per_s"] > 0 and d["ms_per_step"] >= d["roofline"]["kernel_ms"]
    assert d["cpu_baseline"]["cores"] == 1 and d["cpu_baseline"]["kind"] == "port"
    assert d["verified_slices_vs_oracle"] >= 1 and "spec-only" in d["parity"]
