"""Data-parallel ELBO: the expectation term Σ_i E_q[log p(y_i|f_i)] is a plain sum over points
(SVA:355-359), so each rank evaluates its own shard / minibatch with the HIP library and ONE
all-reduce combines {ΣE, n, n_neg, status flags}; the M-sized work (Kuu, Cholesky, KL) is replicated
and the KL is subtracted once after the reduce.

The collective lives INSIDE the library (csrc/comm.hip): `attach_comm_via_torch` hands every rank's context one
rank of an RCCL communicator, after which `DeviceModel.elbo` / `elbo_grad` are collective — ncclAllReduce on the
device-resident result vector, on the context's stream, no host hop.  torch.distributed is used once, to carry the
128-byte communicator id from rank 0 to the others.  The host-side path below (`allreduce_partials` + `combine`, any
torch.distributed backend) remains for hosts that run their own collective and for the gloo CPU tests of the
combination rule.  Pure host logic: nothing here computes a partial sum itself."""
from __future__ import annotations

import numpy as np


def shard_range(n: int, rank: int, world: int):
    """Contiguous shard [lo, hi) of n points for `rank` (first n % world ranks get one more point)."""
    if not (0 <= rank < world):
        raise ValueError("rank outside world")
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def attach_comm_via_torch(ctx, group=None):
    """Give `ctx` its rank of a library-owned RCCL communicator spanning the torch.distributed world (one process per GPU).
    Rank 0 draws the id with svgp_comm_unique_id, torch.distributed broadcasts the 128 bytes, every rank attaches."""
    import torch.distributed as dist

    from . import _ffi

    world, rank = dist.get_world_size(group), dist.get_rank(group)
    box = [None]
    if rank == 0:
        try:
            box[0] = _ffi.comm_unique_id()
        except Exception as e:  # noqa: BLE001 - every rank has to leave the broadcast, then all of them raise
            box[0] = e
    dist.broadcast_object_list(box, src=0, group=group)
    if not isinstance(box[0], (bytes, bytearray)):
        raise RuntimeError(f"rank 0 could not create the communicator id: {box[0]!r}")
    ctx.attach_comm(box[0], world, rank)
    return world, rank


def allreduce_partials(partial, group=None, device=None):
    """Sum the 4-vector {ΣE, n_points, n_neg_var, chol_info_flag} over ranks with one collective.
    chol_info is combined as a max (any rank failing fails the step) by packing it in its own slot."""
    import torch
    import torch.distributed as dist

    t = torch.tensor([partial[0], partial[1], partial[2], 1.0 if partial[3] != 0 else 0.0], dtype=torch.float64,
                     device=device if device is not None else "cpu")
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t.cpu().numpy()


def combine(total, kl: float, num_data: float):
    """ELBO = ΣE · num_data / n_global − KL  (SVA:357-359) from the all-reduced vector."""
    sum_e, n_global, n_neg, bad_chol = (float(v) for v in total[:4])
    if bad_chol > 0:
        raise ArithmeticError("Kuu was not positive definite on at least one rank")
    if len(total) > 4 and float(total[4]) > 0:
        raise RuntimeError("a rank failed before the collective")
    if n_global <= 0:
        raise ValueError("empty global batch")
    return sum_e * (float(num_data) / n_global) - kl


def allreduce_value_and_gradient(value, grads, group=None, device=None):
    """Data-parallel value-and-gradient.  Every rank evaluated its shard with svgp_elbo_grad_shard:
    F_r = s·E_r − KL / W with s = num_data / n_global and W = world size, so the global ELBO and its gradient are the
    plain SUM over ranks — for both parametrisations (the Centered KL depends on the kernel parameters too, so a
    host-side correction of the KL gradient would not do).  ONE all-reduce of the flat vector
    [F_r, variance, lik_sigma2, mean_const, inv_lengthscale(d), z(M·d), m(M), Lq(M²)] (fp64; ≈ 8.4 MB at M = 1024:
    the only place the per-link xGMI ring bound matters)."""
    import torch
    import torch.distributed as dist

    keys = ("variance", "lik_sigma2", "mean_const", "inv_lengthscale", "z", "m", "Lq")
    parts = [np.atleast_1d(np.asarray(value, dtype=np.float64))] + [np.asarray(grads[k], dtype=np.float64).ravel(order="F") for k in keys]
    flat = torch.from_numpy(np.concatenate(parts)).to(device if device is not None else "cpu")
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat = flat.cpu().numpy()
    out, pos = {}, 1
    for k in keys:
        shape = np.shape(grads[k])
        n = int(np.prod(shape)) if shape else 1
        out[k] = flat[pos:pos + n].reshape(shape, order="F") if shape else float(flat[pos])
        pos += n
    return float(flat[0]), out


class ShardedELBO:
    """One rank's view of a minibatched, data-parallel ELBO (config C5): holds this rank's shard of the
    data on its GPU and evaluates `batch` points per step starting at a rotating offset.

    With a communicator on the model's context (`attach_comm_via_torch`) the steps are the library's collective
    svgp_elbo / svgp_elbo_grad.  Without one they fall back to the host-side combination: every rank then reaches the
    all-reduce even when its own evaluation raised (the error travels in the reduced vector and is raised on all ranks
    together, so nobody is left waiting inside the collective)."""

    def __init__(self, model, data, num_data: float, group=None, device=None):
        self.model, self.data, self.num_data, self.group, self.device = model, data, float(num_data), group, device
        info = getattr(getattr(model, "ctx", None), "comm_info", None)
        self.in_library = bool(info and info()[0] > 1) or bool(getattr(model, "force_in_library", False))

    def step(self, off: int, length: int):
        if self.in_library:
            return self.model.elbo(self.data, off, length, self.num_data)[0]
        err, kl, partial = None, 0.0, np.zeros(4)
        try:
            partial = self.model.elbo_partial(self.data, off, length)
            kl, _ = self.model.prior_kl()
        except Exception as e:   # noqa: BLE001 - re-raised after the collective
            err = e
        vec = [partial[0], partial[1], partial[2], partial[3], 0.0 if err is None else 1.0]
        total = allreduce_partials5(vec, self.group, self.device)
        if err is not None:
            raise err
        return combine(total, kl, self.num_data)

    def step_grad(self, off: int, length: int, n_global: int, world: int):
        """Value and gradient of the global minibatch ELBO; `n_global` = total points all ranks evaluate this step."""
        if self.in_library:
            val, _, g = self.model.elbo_grad(self.data, off, length, self.num_data)
            return val, g
        # The host-side mirror of the library's protocol (csrc/api.hip: grad_handshake): the gradient all-reduce is sized by the model,
        # so a rank whose evaluation raised cannot simply enter it.  Every rank first enters ONE fixed-size all-reduce of its failure
        # flag; if any rank failed, all of them leave here - the failing rank with its own exception, the others with RuntimeError -
        # and nobody enters the gradient all-reduce.
        err, val, g = None, 0.0, None
        try:
            val, _, g = self.model.elbo_grad(self.data, off, length, shard=(self.num_data / n_global, 1.0 / world))
        except Exception as e:   # noqa: BLE001 - re-raised after the flag's all-reduce
            err = e
        nfail = allreduce_failure_flag(err is not None, self.group, self.device)
        if err is not None:
            raise err
        if nfail > 0:
            raise RuntimeError("a rank failed before the gradient all-reduce (it was skipped on every rank)")
        return allreduce_value_and_gradient(val, g, self.group, self.device)


def allreduce_partials5(vec, group=None, device=None):
    """{ΣE, n_points, n_neg_var, chol flag, failure flag} summed over ranks (host-side path)."""
    import torch
    import torch.distributed as dist

    t = torch.tensor([vec[0], vec[1], vec[2], 1.0 if vec[3] != 0 else 0.0, vec[4]], dtype=torch.float64,
                     device=device if device is not None else "cpu")
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t.cpu().numpy()


def allreduce_failure_flag(failed: bool, group=None, device=None) -> float:
    """Number of ranks that failed locally: one fixed-size all-reduce every rank can enter whatever state it is in."""
    import torch
    import torch.distributed as dist

    t = torch.tensor([1.0 if failed else 0.0], dtype=torch.float64, device=device if device is not None else "cpu")
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return float(t.item())
