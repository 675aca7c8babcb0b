"""GPU: bench.py end to end at a small size — the single-GPU contract line, and the multi-rank flow
rehearsed with two processes on ONE card (gloo, host-staged halos; RCCL needs one GPU per rank and
is exercised by the driver's multi-GPU bench)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
SMALL = ["--width", "2048", "--height", "1536", "--steps", "2", "--warmup", "1", "--iters-per-step", "16",
         "--no-cpu-baseline", "--no-configs", "--converge-cap", "256"]


def last_json(text):
    lines = [ln for ln in text.strip().splitlines() if ln.startswith("{")]
    assert lines, text[-2000:]
    return json.loads(lines[-1])


def test_bench_contract_line_single_gpu():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *SMALL], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = last_json(out.stdout)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["dtype"] == "f64" and d["vs_baseline"] is None
    assert d["value"] > 1e10 and d["roofline"]["bound"] == "hbm" and d["roofline"]["achieved"] > 0
    assert "workload" in d["config"]
    assert 0 < d["roofline"]["frac"] <= 1.0 and "traffic" in d["roofline"]       # a fraction of the 8 TB/s peak
    assert d["roofline"]["launches_timed"] * d["roofline"]["iterations_per_launch"] == 2 * 16
    pc = d["parity_check"]
    assert pc["iterations"] == 48 and pc["abs_sum_equal"] and pc["residual_sums_equal"] and pc["bands_equal"]
    assert pc["oracle_bands"]["bit_identical"] is True and pc["oracle_bands"]["iterations"] == 16       # ... and against the oracle
    assert "iters_to_1e-5" in d and d["rel_residual_after_timed"][0] == 48
    # the untimed reference-order run on the same system
    ro = d["reference_order"]
    assert ro["iterations"] == 128 and ro["pixel_updates_per_s"] > 1e8 and 0 < ro["rel_residual_after"] < 1.0


def test_bench_two_ranks_on_one_card_matches_single():
    single = last_json(subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *SMALL], capture_output=True,
                                      text=True, timeout=600).stdout)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--same-device",
           "--ghost", "16", *SMALL]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    d = last_json(out.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"
    # same iterates on the partitioned grid: the residual trace agrees to reduction rounding
    a, b = d["rel_residual_after_timed"], single["rel_residual_after_timed"]
    assert a[0] == b[0] and abs(a[1] - b[1]) <= 1e-12 * b[1] + 1e-18


def test_bare_command_starts_its_own_ranks():
    """`python bench.py --gpus 2 ...` with NO launcher (no WORLD_SIZE in the environment): bench.py starts the two rank
    processes itself before anything touches the GPU, picks the ghost depth from timed candidates and prints one line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--same-device", *SMALL]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 1e9
    mg = d["multi_gpu"]
    assert mg["ghost"] in (32, 64, 128) and mg["iters_per_exchange"] == mg["ghost"] // 2
    assert set(mg["ghost_candidates_ms_per_iteration"]) == {"32", "64", "128"}
    assert mg["exchange_ms"] > 0 and mg["interval_ms"] > 0 and mg["halo_path"].startswith("torch.distributed")
    assert "ghost " + str(mg["ghost"]) in d["config"]["partition"]
    single = last_json(subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *SMALL], capture_output=True,
                                      text=True, timeout=600).stdout)
    a, b = d["rel_residual_after_timed"], single["rel_residual_after_timed"]
    assert a[0] == b[0] and abs(a[1] - b[1]) <= 1e-12 * b[1] + 1e-18
