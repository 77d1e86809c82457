"""Two ranks sharing the one GPU of the test box (gloo for the collective, the HIP library for every partial):
the data-parallel ELBO and value-and-gradient of approxgp/distributed.py against a single-process evaluation.
The 8-GPU RCCL launch itself is the driver's; this covers everything but the transport."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rank(rank, world, port, out, centered):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    for p in (os.path.join(ROOT, "approximategps.jl_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    import torch.distributed as dist

    import svgp_oracle as o
    from approxgp import _ffi
    from approxgp.distributed import ShardedELBO, shard_range
    from helpers import device_model

    dist.init_process_group("gloo", rank=rank, world_size=world)
    N, M, d, batch = 4001, 96, 3, 700
    x, y, nc, s2 = o.synth_problem(71, N, M, d, family=o.KERNEL_MATERN52)
    sva = o.SVA(nc.kernel, nc.z, nc.m + 0.2, 0.8 * nc.Lq, jitter=1e-4, mean_const=0.1, centered=True) if centered else nc
    lo, hi = shard_range(N, rank, world)
    ctx = _ffi.Context(0)
    model = device_model(ctx, sva, sigma2=s2)
    data = _ffi.DeviceData(ctx, x[:, lo:hi], y[lo:hi], np.float64)        # this rank's shard only
    sh = ShardedELBO(model, data, num_data=float(N))
    full = sh.step(0, hi - lo)                                             # whole data set, sharded
    val, g = sh.step_grad(100, batch, n_global=world * batch, world=world)  # a minibatch window per rank
    if rank == 0:
        ref_full = o.elbo(sva, x, y, sigma2=s2)
        idx = np.concatenate([np.arange(shard_range(N, r, world)[0] + 100, shard_range(N, r, world)[0] + 100 + batch)
                              for r in range(world)])
        ref_val, ref_g = o.elbo_grad(sva, x[:, idx], y[idx], sigma2=s2, num_data=float(N))
        errs = [abs(full - ref_full) / abs(ref_full), abs(val - ref_val) / abs(ref_val)]
        errs += [float(np.abs(np.asarray(g[k], dtype=np.float64).reshape(np.shape(ref_g[k]), order="F") - np.asarray(ref_g[k])).max()
                       / max(np.abs(np.asarray(ref_g[k])).max(), 1e-12)) for k in ("variance", "inv_lengthscale", "z", "m", "Lq", "mean_const")]
        np.save(out, np.array(errs))
    model.free()
    data.free()
    dist.destroy_process_group()


@pytest.mark.parametrize("centered", [False, True])
def test_two_ranks_on_one_gpu_match_single_process(tmp_path, centered):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "res.npy")
    mp.spawn(_rank, args=(2, port, out, centered), nprocs=2, join=True)
    errs = np.load(out)
    assert errs[0] < 1e-8 and errs[1] < 1e-8, errs      # fp64 ELBO contract
    assert errs[2:].max() < 1e-6, errs                  # gradient blocks, relative to their max-norm
