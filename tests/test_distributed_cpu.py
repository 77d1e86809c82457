"""world_size-2 gloo test of the data-parallel ELBO host logic (approxgp/distributed.py): shard ranges,
ONE all-reduce of {ΣE, n, n_neg, chol}, KL subtracted once.  The per-shard partial sums come from the
CPU oracle here (test infrastructure); on the GPU ranks they come from svgp_elbo_partial."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from approxgp.distributed import combine, shard_range


def test_shard_ranges_cover_everything():
    for n in (1, 7, 100, 262144 * 8 + 3):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def test_combine():
    assert combine([10.0, 5, 0, 0], 1.5, 50) == pytest.approx(10.0 * 10 - 1.5)
    with pytest.raises(ArithmeticError):
        combine([10.0, 5, 0, 1], 1.5, 50)


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (os.path.join(root, "approximategps.jl_amd"), os.path.join(root, "oracle")):
        sys.path.insert(0, p)
    import torch.distributed as dist

    import svgp_oracle as o
    from approxgp.distributed import allreduce_partials, combine, shard_range

    dist.init_process_group("gloo", rank=rank, world_size=world)
    x, y, sva, s2 = o.synth_problem(21, 301, 12, 3)
    lo, hi = shard_range(301, rank, world)
    t = o.elbo_terms(sva, x[:, lo:hi], y[lo:hi], sigma2=s2)
    total = allreduce_partials([t.expectation, hi - lo, 0, 0])
    val = combine(total, t.kl, 1e4)
    if rank == 0:
        full = o.elbo(sva, x, y, sigma2=s2, num_data=1e4)
        np.save(out, np.array([val, full]))
    dist.destroy_process_group()


def _grad_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (os.path.join(root, "approximategps.jl_amd"), os.path.join(root, "oracle")):
        sys.path.insert(0, p)
    import torch.distributed as dist

    import svgp_oracle as o
    from approxgp.distributed import allreduce_value_and_gradient, shard_range

    dist.init_process_group("gloo", rank=rank, world_size=world)
    N = 203
    x, y, nc, s2 = o.synth_problem(22, N, 9, 2, family=o.KERNEL_MATERN52)
    lo, hi = shard_range(N, rank, world)
    errs = []
    for sva in (nc, o.SVA(nc.kernel, nc.z, nc.m + 0.3, 0.7 * nc.Lq, jitter=1e-4, mean_const=0.15, centered=True)):
        # what svgp_elbo_grad_shard returns on this rank: scale = num_data / n_global on the shard's sum, KL / world
        val, g = o.elbo_grad(sva, x[:, lo:hi], y[lo:hi], sigma2=s2, num_data=5e3 * (hi - lo) / N, kl_weight=1.0 / world)
        tot, gt = allreduce_value_and_gradient(val, g)
        ref, gr = o.elbo_grad(sva, x, y, sigma2=s2, num_data=5e3)
        errs += [abs(tot - ref) / abs(ref)] + [float(np.abs(np.asarray(gt[k]) - np.asarray(gr[k])).max() / max(np.abs(np.asarray(gr[k])).max(), 1e-12))
                                                for k in ("variance", "lik_sigma2", "mean_const", "inv_lengthscale", "z", "m", "Lq")]
    if rank == 0:
        np.save(out, np.array(errs))
    dist.destroy_process_group()


def test_two_rank_gloo_gradient_equals_single_process(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "gres.npy")
    mp.spawn(_grad_worker, args=(2, port, out), nprocs=2, join=True)
    assert np.load(out).max() < 1e-10


def test_two_rank_gloo_elbo_equals_single_process(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "res.npy")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    val, full = np.load(out)
    assert val == pytest.approx(full, rel=1e-12)


def _failing_worker(rank, world, port, out):
    """Rank 1's evaluation raises (a DomainError-like condition that depends on its shard): both ranks must come out of
    the step with an exception instead of rank 0 waiting in the all-reduce forever (distributed.py)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "approximategps.jl_amd"))
    import datetime

    import torch.distributed as dist

    from approxgp.distributed import ShardedELBO

    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))

    class Model:
        def elbo_partial(self, data, off, length):
            if rank == 1:
                raise ArithmeticError("negative variance on this shard")
            return np.array([-12.5, float(length), 0.0, 0.0])

        def prior_kl(self):
            return 0.75, 0.0

        def elbo_grad(self, data, off, length, shard=None):
            #  the gradient step: rank 1 cannot even size its gradient blocks (the "NULL model" case of the library)
            if rank == 1:
                raise ValueError("no model on this rank")
            g = dict(variance=0.1, lik_sigma2=0.2, mean_const=0.3, inv_lengthscale=np.ones(2), z=np.ones((2, 3)), m=np.ones(3), Lq=np.eye(3))
            return -3.0, None, g

    sh = ShardedELBO(Model(), None, 1000.0)
    try:
        sh.step(0, 10)
        res = "no exception"
    except ArithmeticError as e:
        res = "own:" + str(e)
    except RuntimeError as e:
        res = "peer:" + str(e)
    try:
        sh.step_grad(0, 10, 20, world)
        res += "|grad:no exception"
    except ValueError as e:
        res += "|grad own:" + str(e)
    except RuntimeError as e:
        res += "|grad peer:" + str(e)
    # and the group is still usable afterwards: a healthy step on both ranks
    class Healthy(Model):
        def elbo_grad(self, data, off, length, shard=None):
            g = dict(variance=0.1, lik_sigma2=0.2, mean_const=0.3, inv_lengthscale=np.ones(2), z=np.ones((2, 3)), m=np.ones(3), Lq=np.eye(3))
            return -3.0, None, g
    v, g = ShardedELBO(Healthy(), None, 1000.0).step_grad(0, 10, 20, world)
    res += f"|after:{v:.1f}:{float(g['variance']):.1f}"
    with open(f"{out}.{rank}", "w") as f:
        f.write(res)
    dist.destroy_process_group()


def test_failing_rank_does_not_hang_the_others(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "fail")
    mp.spawn(_failing_worker, args=(2, port, out), nprocs=2, join=True)
    r0, r1 = open(out + ".0").read().split("|"), open(out + ".1").read().split("|")
    assert r0[0].startswith("peer:a rank failed") and r1[0].startswith("own:negative variance")
    # the gradient step (its all-reduce is sized by the model): the failing rank's own error, RuntimeError on its peer, no hang
    assert r0[1].startswith("grad peer:a rank failed before the gradient all-reduce") and r1[1].startswith("grad own:no model on this rank")
    assert r0[2] == r1[2] == "after:-6.0:0.2"   # the process group survived: the next step sums the two ranks


def test_synthetic_shards_share_the_model():
    """bench.py --gpus N: every rank draws its own data shard from the SURVEY 8d recipe, but the model (z, m, Lq, kernel)
    must be rank 0's on every rank (approxgp/synthetic.py), and shard 0 is exactly what the parity tests use."""
    from approxgp.synthetic import synth_arrays

    a0 = synth_arrays(6, 500, 40, 8)
    a3 = synth_arrays(6, 500, 40, 8, shard=3)
    for k in ("z", "m", "Lq", "inv_lengthscale"):
        np.testing.assert_array_equal(a0[k], a3[k])
    assert a0["variance"] == a3["variance"] and a0["jitter"] == a3["jitter"]
    assert not np.array_equal(a0["x"], a3["x"]) and not np.array_equal(a0["y"], a3["y"])
    np.testing.assert_allclose(a0["z"], a0["x"][:, :40], atol=1e-2)      # z = the first M points of x + 1e-3 noise
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import svgp_oracle as o

    x, y, sva, s2 = o.synth_problem(6, 500, 40, 8)
    np.testing.assert_array_equal(x, a0["x"])
    np.testing.assert_array_equal(sva.z, a0["z"])
    f32 = synth_arrays(6, 100, 10, 4, dtype=np.float32)
    assert f32["jitter"] == 1e-3 and np.array_equal(f32["x"], f32["x"].astype(np.float32).astype(np.float64))
