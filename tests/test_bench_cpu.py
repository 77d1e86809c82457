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


def test_multi_rank_line_carries_baseline_parity_and_kuf():
    """for world > 1 rank 0 used to drop `cpu_baseline`, `parity` and `kuf_roofline`.  The helpers that produce
    them are exercised here for a 2-rank job on a GPU-less box with a stand-in for the device handles (the stand-in answers
    svgp_elbo_partial / svgp_prior_kl from the oracle: test infrastructure, this file only): the three keys are present, the parity
    figure is the local-partial one and the baseline says whose shard it timed."""
    import types

    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import svgp_oracle as o

    b = _bench()
    b.CONFIGS["T"] = (3000, 48, 2, b.SE, b.GAUSS, "f64", 9)
    n, M, d, family, lik, dtype, cid = b.CONFIGS["T"]
    p = b.synth(cid, n, M, d, family, lik, dtype, 0)
    sva = o.SVA(o.Kernel(family, p["variance"], p["inv_l"]), p["z"], p["m"], p["Lq"], jitter=p["jitter"])

    class Model:
        def elbo_partial(self, data, off, length):
            e = o.elbo(sva, p["x"][:, off:off + length], p["y"][off:off + length], lik=lik, sigma2=p["sigma2"], num_data=float(length))
            return np.array([e + o.prior_kl(sva), float(length), 0.0, 0.0])

        def prior_kl(self):
            return o.prior_kl(sva), 0.0

        def elbo(self, *a):
            raise AssertionError("svgp_elbo is collective under a communicator: rank 0 must not call it alone")

        def kuf(self, data, off, length, fetch=True):
            return None

    ctx = types.SimpleNamespace(timing=lambda: types.SimpleNamespace(ms_kuf=0.5))
    args = types.SimpleNamespace(cpu_sample=2500)
    got = b.baseline_and_parity(args, "T", p, Model(), None, {"elbo": 0.0}, 2, 40.0)
    assert {"cpu_baseline", "parity", "gpu_over_cpu"} <= set(got), got
    assert got["parity"]["ok"] and got["parity"]["rel_err"] < 1e-10 and "svgp_elbo_partial" in got["parity"]["via"]
    assert got["cpu_baseline"]["value"] > 0 and "rank 0's shard" in got["cpu_baseline"]["scope"]
    assert got["gpu_over_cpu"] == (40.0 / 2) / got["cpu_baseline"]["value"]
    kr = b.measure_kuf("T", ctx, Model(), None, None, None)
    assert kr["bound"] == "hbm" and kr["achieved"] > 0 and kr["bytes_per_launch"] == 8 * (M * n + n * d + M * d)
    src = open(os.path.join(ROOT, "bench.py")).read()
    main_src = src[src.index("def main():"):]
    # main() attaches all three on rank 0 whatever the world size
    assert 'if rank == 0 and not args.no_kuf:' in main_src and 'if rank == 0 and not args.no_cpu_baseline:' in main_src
    assert "world == 1 and not args.no_kuf" not in main_src and "world == 1 and not args.no_cpu_baseline" not in main_src
