#!/usr/bin/env python
"""bench.py — ELBO-evaluation throughput of the MI355X SVGP path (BASELINE.json metric:
"SVGP ELBO evals/sec at N=1e6, M=1024; Kuf-assembly HBM GB/s vs roofline").

    python bench.py --gpus 1 --steps K --warmup W            # one GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step is ONE full ELBO evaluation (reference elbo(sva, lfx, y), SVA:340-360) over one batch of synthetic
data already resident in HBM: Kuu assembly, cholesky(Kuu), diagonal-block inverses / T panels, KL, then the
fused Kuf -> trsm -> trmm -> expectation pass over every point, and the read-back of the scalar.
Workload H: N = 1e6 points per GPU, M = 1024, d = 8, SE-ARD, Gaussian likelihood, fp64,
NonCentered.  With N GPUs every rank holds its own 1e6-point shard (weak scaling), evaluates its partial
sum, and ONE all-reduce (RCCL) of 4 doubles combines them; `value` is then whole-job 1e6-point ELBO
evaluations per second.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "approximategps.jl_amd"))

# kernel families / likelihood codes of the C-ABI
SE, M32, M52 = 0, 1, 2
GAUSS, BERN, POIS = 0, 1, 2

CONFIGS = {
    # name: (n_per_gpu, M, d, family, lik, dtype)
    "H": (1_000_000, 1024, 8, SE, GAUSS, "f64"),      # headline metric
    "H32": (1_000_000, 1024, 8, SE, GAUSS, "f32"),
    "H896": (1_000_000, 896, 8, SE, GAUSS, "f64"),     # leading-dimension experiments (Mp*8 not a power of two)
    "H1152": (1_000_000, 1152, 8, SE, GAUSS, "f64"),
    "C2": (100_000, 512, 8, SE, GAUSS, "f64"),
    "C3": (1_000_000, 2048, 16, M52, BERN, "f32"),
    "C4": (100_000, 8192, 8, SE, GAUSS, "f32"),
    "C5": (262_144, 1024, 8, SE, GAUSS, "f32"),        # per-GPU minibatch of the 8-GPU config
}
PEAK_TFLOPS = {"f64": 78.6, "f32": 157.3}  # MI355X dense matrix peaks (AMD spec; MI355X_MICROARCH.md for f32)
PEAK_HBM_GBS = 8000.0


def synth(config_id, n, M, d, family, lik, dtype, rank=0):
    """SURVEY §8d synthetic problem; the model is identical on every rank, the data shard is per rank."""
    rng = np.random.default_rng(20260313 + config_id)
    z = rng.standard_normal((d, M)) + 1e-3 * rng.standard_normal((d, M))
    ell = math.sqrt(d) * (0.75 + 0.5 * np.arange(d) / d)
    m = 0.1 * rng.standard_normal(M)
    Lq = np.eye(M) + 0.05 * np.tril(rng.standard_normal((M, M))) / math.sqrt(M)
    Lq[np.diag_indices(M)] = np.abs(np.diag(Lq))
    drng = np.random.default_rng(977 * (rank + 1) + config_id)
    x = drng.standard_normal((d, n))
    s = x.sum(axis=0) / math.sqrt(d)
    sigma2 = 0.3
    if lik == GAUSS:
        y = np.sin(s) + math.sqrt(sigma2) * drng.standard_normal(n)
    else:
        y = (drng.random(n) < 1.0 / (1.0 + np.exp(-2.0 * np.sin(s)))).astype(np.float64)
    np_dt = np.float64 if dtype == "f64" else np.float32
    jitter = 1e-5 if dtype == "f64" else 1e-3
    rt = lambda a: np.asarray(a, dtype=np_dt)
    return dict(x=rt(x), y=rt(y), z=rt(z), m=rt(m), Lq=rt(Lq), inv_l=1.0 / ell, variance=1.3, sigma2=sigma2,
                jitter=jitter, np_dt=np_dt)


def cpu_baseline(p, family, lik, sample, n_full):
    """The CPU restatement (oracle, reference operation order: materialise Kuf, trsm, trmm, reductions) timed on
    the host cores on a bounded sample of the same workload, extrapolated linearly in N."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import svgp_oracle as o

    kernel = o.Kernel(family, p["variance"], p["inv_l"])
    f64 = lambda a: np.asarray(a, dtype=np.float64)
    sva = o.SVA(kernel, f64(p["z"]), f64(p["m"]), f64(p["Lq"]), jitter=p["jitter"])
    xs, ys = f64(p["x"][:, :sample]), f64(p["y"][:sample])
    o.elbo(sva, xs[:, :2000], ys[:2000], lik=lik, sigma2=p["sigma2"])  # warm BLAS threads
    t_small = []
    for _ in range(2):
        t0 = time.perf_counter()
        o.elbo(sva, xs[:, :2000], ys[:2000], lik=lik, sigma2=p["sigma2"])
        t_small.append(time.perf_counter() - t0)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        o.elbo(sva, xs, ys, lik=lik, sigma2=p["sigma2"])
        ts.append(time.perf_counter() - t0)
    t_s, t_2k = float(np.median(ts)), float(min(t_small))
    per_point = max(t_s - t_2k, 1e-9) / (sample - 2000)          # data-proportional part
    fixed = max(t_2k - 2000 * per_point, 0.0)                    # cholesky(Kuu) etc.
    t_full = fixed + per_point * n_full
    return {"value": 1.0 / t_full, "unit": "evals/s", "cores": os.cpu_count(), "kind": "port",
            "sample": f"oracle/svgp_oracle.py (numpy + scipy-OpenBLAS, fp64) on {sample} of the {n_full} points, "
                      f"median of 3 = {t_s:.2f} s; extrapolated linearly in N to {t_full:.1f} s/eval"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="H", choices=sorted(CONFIGS))
    ap.add_argument("--cpu-sample", type=int, default=100000)   # ~10 s of host work at H
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kuf", action="store_true")
    ap.add_argument("--no-grad", action="store_true")
    args = ap.parse_args()

    # Everything but the final JSON line goes to stderr: RCCL prints a version banner (and warnings) on the C stdout,
    # which would otherwise surround the one line the driver reads.  fd 1 is restored just before the JSON is written.
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    from approxgp import _ffi

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the library has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1"   # the env knob exercises the RCCL path on one GPU
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    n, M, d, family, lik, dtype = CONFIGS[args.config]
    cid = sorted(CONFIGS).index(args.config)
    p = synth(cid, n, M, d, family, lik, dtype, rank)
    stream = torch.cuda.current_stream().cuda_stream
    ctx = _ffi.Context(local_rank, stream if stream else None)
    desc, keep = _ffi.make_desc(p["np_dt"], family, p["variance"], p["inv_l"], p["z"], p["m"], p["Lq"], p["jitter"],
                                likelihood=lik, lik_sigma2=p["sigma2"])
    model = _ffi.DeviceModel(ctx, desc, keep)
    data = _ffi.DeviceData(ctx, p["x"], p["y"], p["np_dt"])
    num_data = float(n * world)

    from approxgp.distributed import allreduce_partials, combine

    def step():
        part = model.elbo_partial(data, 0, n)              # prep + fused strips + read-back (HIP library)
        kl, _ = model.prior_kl()                           # cached scalars of the same prep
        if use_dist:
            part = allreduce_partials(part, device=dev)    # ONE RCCL all-reduce of 4 doubles
        else:
            part = np.array([part[0], part[1], part[2], 1.0 if part[3] else 0.0])
        return combine(part, kl, num_data)

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # Kuf assembly alone (SVA:216): M x n column-major written once -> HBM-write bound.  Measured before the ELBO loop
    # (its own launches, its own HIP events): the figure is a property of this kernel, not of what ran before it.
    kuf_roofline = None
    if rank == 0 and world == 1 and not args.no_kuf:
        es = 8 if dtype == "f64" else 4
        times = []
        for _ in range(8):
            model.kuf(data, 0, n, fetch=False)
            times.append(ctx.timing().ms_kuf)
        t_kuf = float(np.median(times[1:]))
        bytes_alg = es * (M * n + n * d + M * d)
        gbs = bytes_alg / (t_kuf * 1e-3) / 1e9
        kuf_roofline = {"kernel": "kuf_kernel", "bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS,
                        "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS, "traffic": None,
                        "bytes_per_launch": bytes_alg, "ms_per_launch": t_kuf,
                        "ms_per_launch_all": [round(t, 4) for t in times]}
        # what a plain write-only stream of the same size reaches on this box (report both)
        try:
            buf = torch.empty(M * n, dtype=torch.float64 if dtype == "f64" else torch.float32, device=dev)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ts = []
            for _ in range(6):
                ev[0].record()
                buf.fill_(1.0)
                ev[1].record()
                torch.cuda.synchronize()
                ts.append(ev[0].elapsed_time(ev[1]))
            t_fill = float(np.median(ts[1:]))
            kuf_roofline["stream_write_GBps"] = es * M * n / (t_fill * 1e-3) / 1e9
            kuf_roofline["frac_of_stream_write"] = gbs / kuf_roofline["stream_write_GBps"]
            del buf
        except Exception as e:  # noqa: BLE001
            kuf_roofline["stream_write_GBps"] = None
            kuf_roofline["stream_write_error"] = repr(e)

    for _ in range(args.warmup):
        step()
    strip_ms, prep_ms, expect_ms = [], [], []
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        val = step()
        t = ctx.timing()
        strip_ms.append(t.ms_strip)
        prep_ms.append(t.ms_prep)
        expect_ms.append(t.ms_expect)
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    ms_per_step = 1e3 * elapsed / args.steps
    value = world * args.steps / elapsed                    # 1e6-point ELBO evaluations per second, whole job
    strip_avg_ms = float(np.mean(strip_ms))
    Mp = (M + 127) // 128 * 128
    strip_w = 64 if (dtype == "f64" or Mp > 2048) else 128   # strip.hip: strip_nt()
    flops_strip = 2.0 * M * M * n                           # algorithmic: trsm + trmm
    ach = flops_strip / (strip_avg_ms * 1e-3) / 1e12
    out = {
        "metric": "SVGP ELBO evals/sec at N=1e6, M=1024" if args.config in ("H", "H32") else f"SVGP ELBO evals/sec ({args.config})",
        "value": value, "unit": "evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": dtype, "data": "synthetic",
        "config": {"workload": f"{args.config}: N={n} points per GPU, M={M}, d={d}, "
                               f"{['SE', 'Matern32', 'Matern52'][family]}-ARD, "
                               f"{['Gaussian', 'Bernoulli-logistic GH-20', 'Poisson'][lik]}, {dtype}, NonCentered; "
                               "one step = one full elbo(sva, lfx, y) incl. cholesky(Kuu)",
                   "global_points": n * world, "points_per_s": n * world * args.steps / elapsed,
                   "parallelism": f"data-parallel shards x{world}, one 4-double all-reduce per eval",
                   "elbo": val},
        "roofline": {"kernel": "strip_kernel (fused Kuf -> trsm -> trmm)", "bound": "mfma", "achieved": ach,
                     "peak": PEAK_TFLOPS[dtype], "unit": "TFLOP/s", "frac": ach / PEAK_TFLOPS[dtype], "traffic": None,
                     "flops_per_launch": flops_strip, "ms_per_launch": strip_avg_ms,
                     # MFMA work actually issued: full 128-row blocks below the diagonal + 20 of the 32 tile-steps of every
                     # (triangular) diagonal block, per phase, on whole strips (matches SQ_INSTS_MFMA x 2048 of the PMC profile)
                     "executed_flops_per_launch": 2.0 * 2.0 * 128 * 128 * ((Mp // 128) * (Mp // 128 - 1) / 2 + 0.625 * (Mp // 128))
                     * strip_w * math.ceil(n / strip_w)},
        "breakdown_ms": {"prep (Kuu, cholesky, T panels, KL)": float(np.mean(prep_ms)), "strip": strip_avg_ms,
                         "expectation + reduce": float(np.mean(expect_ms))},
    }

    # HBM-side traffic of the dominant kernel: PMC counters cannot be read from inside this process, so the
    # figure is the one measured by tools/run_profile.sh (separate rocprofv3 --pmc passes) for this workload.
    pmc_path = os.path.join(ROOT, "profiles", "round1", "H_fp64_v8_pmc.json")
    pm = {}
    if args.config == "H":
        try:   # a missing / reshaped summary must never cost the benchmark line
            pm = json.load(open(pmc_path))
        except (OSError, ValueError):
            pm = {}
    strip_pm = next((v for k, v in pm.items() if k.startswith("strip_kernel<double") and isinstance(v, dict)), None)
    if strip_pm is not None:
        out["roofline"]["traffic"] = strip_pm.get("traffic_bytes_per_launch")
        out["roofline"]["traffic_note"] = ("bytes per launch at the L2<->fabric boundary (FETCH_SIZE x2 + WRITE_SIZE, "
                                           "profiles/round1/H_fp64_v8_pmc.json); includes Infinity-Cache hits of the per-workgroup "
                                           "scratch strips; algorithmic HBM bytes are 88 MB")
    if kuf_roofline is not None:
        kuf_pm = next((v for k, v in pm.items() if k.startswith("kuf_kernel<double") and isinstance(v, dict)), None)
        if kuf_pm is not None:
            kuf_roofline["traffic"] = kuf_pm.get("traffic_bytes_per_launch")
        out["kuf_roofline"] = kuf_roofline
    if rank == 0 and world == 1 and not args.no_grad:
        # value-and-gradient evaluation (svgp_elbo_grad: what a training step costs), same workload, same residency
        try:
            model.elbo_grad(data, 0, n, num_data)
            tg = []
            for _ in range(3):
                t0 = time.perf_counter()
                model.elbo_grad(data, 0, n, num_data)
                tg.append(time.perf_counter() - t0)
            out["value_and_gradient"] = {"evals_per_s": 1.0 / min(tg), "ms_per_eval": 1e3 * min(tg),
                                         "ratio_to_forward": 1e3 * min(tg) / ms_per_step}
        except Exception as e:  # noqa: BLE001  (reported beside the headline, never part of it)
            out["value_and_gradient"] = {"error": repr(e)}
    if use_dist and not args.no_grad:
        # data-parallel training step: every rank's shard gradient (svgp_elbo_grad_shard), ONE sum all-reduce of the flat
        # [value, gradients] vector (about 8.4 MB fp64 at M = 1024) over RCCL.  Reported beside the headline, never part of it;
        # a failure here must not cost the main line.
        try:
            from approxgp.distributed import ShardedELBO
            sh = ShardedELBO(model, data, num_data=num_data, device=dev)
            sh.step_grad(0, n, n_global=n * world, world=world)
            fence()
            t0 = time.perf_counter()
            for _ in range(3):
                gval, _ = sh.step_grad(0, n, n_global=n * world, world=world)
            fence()
            tg = torch.tensor([(time.perf_counter() - t0) / 3], dtype=torch.float64, device=dev)
            dist.all_reduce(tg, op=dist.ReduceOp.MAX)
            out["distributed_value_and_gradient"] = {"ms_per_step": 1e3 * float(tg.item()), "global_points": n * world,
                                                     "value": gval}
        except Exception as e:  # noqa: BLE001
            out["distributed_value_and_gradient"] = {"error": repr(e)}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(p, family, lik, min(args.cpu_sample, n), n)
            out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
        except Exception as e:  # noqa: BLE001
            out["cpu_baseline"] = {"value": None, "unit": "evals/s", "cores": os.cpu_count(), "kind": "port", "sample": "failed: " + repr(e)}
    model.free()
    data.free()
    ctx.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    # flush what the C libraries buffered (it belongs to stderr), then give fd 1 back for the JSON line
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass
    sys.stdout.flush()
    os.dup2(saved_stdout, 1)
    os.close(saved_stdout)
    if rank == 0:
        if world > 1:
            time.sleep(1.0)   # let the other ranks finish their teardown chatter: the JSON stays the last line even in a merged capture
        sys.stdout.write(json.dumps(out) + "\n")
        sys.stdout.flush()


if __name__ == "__main__":
    main()
