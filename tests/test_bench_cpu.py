"""bench.py host logic that needs no GPU: `--gpus N` started plainly must become an N-rank torchrun job."""
import importlib.util
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_launcher_command_line():
    b = _bench()
    cmd = b.launcher_command(8, ["--gpus", "8", "--steps", "20", "--warmup", "5"], 29777)
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29777"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]   # the invocation's own arguments, unchanged


def test_self_launch_decision():
    b = _bench()
    assert not b.needs_self_launch(1, {})                                    # one GPU: this process is the bench
    assert b.needs_self_launch(8, {})                                        # plain start with N > 1: launch N ranks
    assert b.needs_self_launch(2, {"WORLD_SIZE": "1"})
    assert not b.needs_self_launch(8, {"RANK": "3", "WORLD_SIZE": "8"})      # already a rank of torchrun's job
    assert not b.needs_self_launch(8, {"WORLD_SIZE": "8"})


def test_plain_multi_gpu_start_spawns_ranks_and_reports_failure_loudly():
    """No GPU here: the two ranks refuse to run ("the library has no CPU path"); the launcher must relay that as a JSON line with
    value null and a non-zero exit code - never a silent one-GPU number."""
    env = dict(os.environ, OMP_NUM_THREADS="1")
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0
    last = [ln for ln in r.stdout.splitlines() if ln.strip()][-1]
    out = json.loads(last)
    assert out["value"] is None and out["n_gpus"] == 2
    assert "--nproc-per-node=2" in out["launcher"]
    assert "bench.py needs a GPU" in r.stderr                                # the children really ran bench.py as ranks
