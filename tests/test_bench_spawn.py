"""CPU: `python bench.py --gpus N` without a launcher starts N rank processes itself — before the parent imports
torch or touches a GPU — with the rendezvous environment the contract names, and returns the worst exit code."""
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_gpus_n_spawns_ranks_with_the_rendezvous_environment():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300, env=env)
    # no GPU here: every rank must stop with the no-device message (a loud failure, never a CPU fallback) ...
    assert out.returncode != 0
    assert out.stderr.count("bench.py needs an MI355X") == 3, out.stderr[-2000:]
    assert "torch.distributed.run" not in out.stderr              # ... and nobody asks for an external launcher any more


def test_spawn_ranks_environment(monkeypatch, tmp_path):
    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module("bench")
    seen = []

    class P:
        def __init__(self, cmd, env=None):
            seen.append((cmd, env))

        def poll(self):
            return 0

        def wait(self):
            return 0

    import subprocess as sp
    monkeypatch.setattr(sp, "Popen", P)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    assert bench.spawn_ranks(4) == 0
    assert len(seen) == 4
    ports = {e["MASTER_PORT"] for _, e in seen}
    assert len(ports) == 1
    for r, (cmd, e) in enumerate(seen):
        assert cmd[0] == sys.executable and cmd[1].endswith("bench.py") and cmd[2:] == ["--gpus", "4", "--steps", "3"]
        assert e["RANK"] == e["LOCAL_RANK"] == str(r) and e["WORLD_SIZE"] == "4" and e["MASTER_ADDR"] == "127.0.0.1"
        assert e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_launcher_world_size_must_match():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], capture_output=True, text=True,
                         timeout=300, env=env)
    assert out.returncode != 0 and "--gpus 4 but the launcher started 2" in out.stderr


def test_cpu_baseline_child_reports_its_phases(tmp_path):
    """bench.py's CPU baseline runs in a child process (no torch, no GPU): one JSON line per sample and a wall-clock line per
    phase in the log, so that a run that is cut short still says where the time went."""
    import json
    log = tmp_path / "phases.log"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-baseline-child", "96,160", "--cpu-iters", "3",
                          "--cpu-threads", "2", "--cpu-log", str(log)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [json.loads(ln) for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert [d["sample"] for d in lines] == [96, 160]
    for d in lines:
        assert d["iters"] == 3 and d["value"] > 0 and d["index_type"] == "int" and d["kind"] in ("reference", "port")
        assert set(d["phases_s"]) >= {"warm", "generator", "rhs"}
    text = log.read_text()
    for what in ("warm (first touch", "generator (closed form", "b = A x_true", "gaussSeidel, 3 sweeps"):
        assert text.count(what) == 2, what


def test_the_headline_system_needs_the_wide_index_type():
    sys.path.insert(0, ROOT)
    import bench
    assert not bench.cpu_needs_wide_index(4096) and not bench.cpu_needs_wide_index(14336)
    assert bench.cpu_needs_wide_index(16384)                   # 5 x 16384^2 > 2^30: getNearestIndex's (end + idx) / 2 overflows int
    assert 50.0 < bench.cpu_system_gb(16384) < 70.0
