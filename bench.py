#!/usr/bin/env python
"""bench.py — ELBO-evaluation throughput of the MI355X SVGP path (BASELINE.json metric:
"SVGP ELBO evals/sec at N=1e6, M=1024; Kuf-assembly HBM GB/s vs roofline").

    python bench.py --gpus 1 --steps K --warmup W            # one GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
    python bench.py --gpus N ...        # started plainly with N > 1: launches the line above itself (N fresh children, before
                                        # this process touches the GPU) and relays rank 0's JSON line

A step is ONE full ELBO evaluation (reference elbo(sva, lfx, y), SVA:340-360) over one batch of synthetic
data already resident in HBM: Kuu assembly, cholesky(Kuu), diagonal-block inverses / T panels, KL, then the
fused Kuf -> trsm -> trmm -> expectation pass over every point, and the read-back of the scalar.
Workload H: N = 1e6 points per GPU, M = 1024, d = 8, SE-ARD, Gaussian likelihood, fp64,
NonCentered.  With N GPUs every rank holds its own 1e6-point shard (weak scaling) and calls the library's COLLECTIVE
svgp_elbo: the ranks' partial sums are combined by ONE ncclAllReduce of 8 doubles issued inside the library on the
device-resident result (csrc/comm.hip; torch.distributed only carries the 128-byte communicator id and the timing
barrier); `value` is then whole-job 1e6-point ELBO evaluations per second.  The same line carries `c5_minibatch`:
BASELINE's 8-GPU configuration C5 (fp32, 2^18 points per GPU per step, num_data = 1e8) through the same path.
Inputs come from approxgp/synthetic.py, the one §8d recipe the parity tests use too (z = first M points of x + 1e-3 noise).
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
# multi-process GPU work on this pool: the host driver only supports dmabuf IPC (RCCL fails with hipIpcGetMemHandle otherwise);
# read by the HSA runtime when it initialises, so it has to be in the environment before the first HIP call
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

# kernel families / likelihood codes of the C-ABI
SE, M32, M52 = 0, 1, 2
GAUSS, BERN, POIS = 0, 1, 2

CONFIGS = {
    # name: (n_per_gpu, M, d, family, lik, dtype, §8d config id = seed offset)
    "H": (1_000_000, 1024, 8, SE, GAUSS, "f64", 6),      # headline metric
    "H32": (1_000_000, 1024, 8, SE, GAUSS, "f32", 6),
    "H896": (1_000_000, 896, 8, SE, GAUSS, "f64", 6),     # leading-dimension experiments (Mp*8 not a power of two)
    "H1152": (1_000_000, 1152, 8, SE, GAUSS, "f64", 6),
    "C2": (100_000, 512, 8, SE, GAUSS, "f64", 2),
    "C3": (1_000_000, 2048, 16, M52, BERN, "f32", 3),
    "C4": (100_000, 8192, 8, SE, GAUSS, "f32", 4),
    "C5": (262_144, 1024, 8, SE, GAUSS, "f32", 5),        # per-GPU minibatch of the 8-GPU config
    # wide inputs (round 4): the headline shape at d = 32 / 64, both precisions
    "Hd32": (1_000_000, 1024, 32, SE, GAUSS, "f64", 6), "Hd64": (1_000_000, 1024, 64, SE, GAUSS, "f64", 6),
    "H32d32": (1_000_000, 1024, 32, SE, GAUSS, "f32", 6), "H32d64": (1_000_000, 1024, 64, SE, GAUSS, "f32", 6),
    "Hd16": (1_000_000, 1024, 16, SE, GAUSS, "f64", 6), "Hd17": (1_000_000, 1024, 17, SE, GAUSS, "f64", 6), "Hd24": (1_000_000, 1024, 24, SE, GAUSS, "f64", 6), "Hd48": (1_000_000, 1024, 48, SE, GAUSS, "f64", 6),
    # minibatch shapes of the verdict's prep-overlap targets
    "MB16k": (16_384, 1024, 8, SE, GAUSS, "f64", 7), "MB4k": (4_096, 512, 8, SE, GAUSS, "f64", 7),
}
C5_NUM_DATA = 1.0e8
PEAK_TFLOPS = {"f64": 78.6, "f32": 157.3}  # MI355X dense matrix peaks (AMD spec; MI355X_MICROARCH.md for f32)
PEAK_HBM_GBS = 8000.0


def synth(config_id, n, M, d, family, lik, dtype, rank=0):
    """SURVEY §8d synthetic problem from the shared recipe (approxgp/synthetic.py): the model is identical on every
    rank (z = the first M points of shard 0 + 1e-3 noise), the data shard is per rank."""
    from approxgp.synthetic import synth_arrays

    np_dt = np.float64 if dtype == "f64" else np.float32
    a = synth_arrays(config_id, n, M, d, lik=lik, dtype=np_dt, shard=rank)
    rt = lambda v: np.asarray(v, dtype=np_dt)
    return dict(x=rt(a["x"]), y=rt(a["y"]), z=rt(a["z"]), m=rt(a["m"]), Lq=rt(a["Lq"]), inv_l=a["inv_lengthscale"],
                variance=a["variance"], sigma2=a["sigma2"], jitter=a["jitter"], np_dt=np_dt)


def mem_available_bytes():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                return int(line.split()[1]) * 1024
    except OSError:
        pass
    return 0


def cpu_baseline(p, family, lik, sample, n_full, M, force_sample=False):
    """The CPU restatement (oracle, reference operation order: materialise Kuf, trsm, trmm, reductions) timed on
    the host cores.  SURVEY §8d: at the FULL workload when the host has room for the reference's M x N temporaries
    (MemAvailable >= 5 M N 8 bytes; one evaluation, ~30 s at H), otherwise on a bounded sample extrapolated linearly in N."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import svgp_oracle as o

    kernel = o.Kernel(family, p["variance"], p["inv_l"])
    f64 = lambda a: np.asarray(a, dtype=np.float64)
    sva = o.SVA(kernel, f64(p["z"]), f64(p["m"]), f64(p["Lq"]), jitter=p["jitter"])
    avail, need = mem_available_bytes(), 5 * M * n_full * 8
    try:
        from threadpoolctl import threadpool_info
        threads = max([t.get("num_threads", 1) for t in threadpool_info()] or [os.cpu_count()])
    except Exception:  # noqa: BLE001
        threads = os.cpu_count()
    base = {"unit": "evals/s", "cores": threads, "kind": "port", "host_cpus": os.cpu_count(),
            "mem_available_GB": round(avail / 1e9, 1), "mem_needed_full_GB": round(need / 1e9, 1)}
    o.elbo(sva, f64(p["x"][:, :2000]), f64(p["y"][:2000]), lik=lik, sigma2=p["sigma2"])  # warm BLAS threads
    if avail >= need and os.environ.get("BENCH_CPU_SAMPLE_ONLY") != "1" and not force_sample:
        # the median of three warm evaluations of the full workload when three of them fit the budget (BENCH_CPU_BUDGET_S,
        # default 100 s: ~85 s at H on the GPU box's host), else as many as fit - `runs` and `t_evals_s` say which
        xs, ys = f64(p["x"]), f64(p["y"])
        budget = float(os.environ.get("BENCH_CPU_BUDGET_S", "100"))
        t_evals = []
        while len(t_evals) < 3:
            t0 = time.perf_counter()
            ref = o.elbo(sva, xs, ys, lik=lik, sigma2=p["sigma2"])
            t_evals.append(time.perf_counter() - t0)
            if sum(t_evals) + max(t_evals) > budget:   # the next one would not fit
                break
        t_full = float(np.median(t_evals))
        return dict(base, value=1.0 / t_full, oracle_elbo=float(ref), oracle_points=int(n_full), runs=len(t_evals),
                    t_evals_s=[round(t, 3) for t in t_evals],
                    sample=f"oracle/svgp_oracle.py (numpy + scipy-OpenBLAS, fp64, {threads} BLAS threads) on ALL {n_full} points, "
                           f"median of {len(t_evals)} evaluation(s) = {t_full:.1f} s (no extrapolation)")
    xs, ys = f64(p["x"][:, :sample]), f64(p["y"][:sample])
    t_small = []
    for _ in range(2):
        t0 = time.perf_counter()
        o.elbo(sva, xs[:, :2000], ys[:2000], lik=lik, sigma2=p["sigma2"])
        t_small.append(time.perf_counter() - t0)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        ref = o.elbo(sva, xs, ys, lik=lik, sigma2=p["sigma2"])
        ts.append(time.perf_counter() - t0)
    t_s, t_2k = float(np.median(ts)), float(min(t_small))
    per_point = max(t_s - t_2k, 1e-9) / max(sample - 2000, 1)    # data-proportional part
    fixed = max(t_2k - 2000 * per_point, 0.0)                    # cholesky(Kuu) etc.
    t_full = fixed + per_point * n_full
    return dict(base, value=1.0 / t_full, oracle_elbo=float(ref), oracle_points=int(sample), runs=len(ts), t_evals_s=[round(t, 3) for t in ts],
                sample=f"oracle/svgp_oracle.py (numpy + scipy-OpenBLAS, fp64, {threads} BLAS threads) on {sample} of the {n_full} "
                       f"points ({'N > 1 ranks: bounded sample of rank 0 shard' if force_sample else 'host RAM below 5 M N 8 bytes'}), "
                       f"median of 3 = {t_s:.2f} s; extrapolated linearly in N to {t_full:.1f} s/eval")


def lib_sha16():
    """sha256 (first 16 hex digits) of the library this process loaded: the build log (approximategps.jl_amd/csrc/build.log, copied to profiles/roundN/build.log by the evidence run) lists the same hash for the .so
    build.sh produced from the tree's sources."""
    import hashlib
    from approxgp import _ffi
    try:
        return hashlib.sha256(open(_ffi.LIB_PATH, "rb").read()).hexdigest()[:16]
    except OSError:
        return None


class GpuTelemetry:
    """Shader clock and socket power of one GPU, sampled from sysfs (readable without privileges) by a thread while a timed
    loop runs: /sys/class/drm/card*/device/pp_dpm_sclk (the level marked `*`) and hwmon power1_average / power1_input (uW).
    Whatever is missing on a box is reported as absent, never guessed."""

    def __init__(self, index=0, period=0.5, pci_bus_id=None):
        import glob
        self.period, self.samples, self._stop, self._thr = period, [], False, None
        cards = []
        for c in sorted(glob.glob("/sys/class/drm/card[0-9]*")):
            try:
                if open(os.path.join(c, "device", "vendor")).read().strip() == "0x1002":
                    cards.append(c)
            except OSError:
                pass
        self.card = None
        if pci_bus_id:   # the card whose PCI address is the HIP device's (a box may expose more cards than the one we were given)
            want = pci_bus_id.lower()
            for c in cards:
                try:
                    if os.path.basename(os.path.realpath(os.path.join(c, "device"))).lower() == want:
                        self.card = c
                except OSError:
                    pass
        self.card_matched_by_pci = self.card is not None
        if self.card is None:
            self.card = cards[index] if index < len(cards) else None
        self.power_file = None
        if self.card:
            for pat in ("device/hwmon/hwmon*/power1_average", "device/hwmon/hwmon*/power1_input"):
                hits = sorted(glob.glob(os.path.join(self.card, pat)))
                if hits:
                    self.power_file = hits[0]
                    break

    def _read(self):
        rec = {"t": time.perf_counter()}
        if self.card:
            try:
                for ln in open(os.path.join(self.card, "device", "pp_dpm_sclk")):
                    if "*" in ln:
                        rec["sclk_mhz"] = float(ln.split(":")[1].lower().replace("mhz", "").replace("*", "").strip())
            except (OSError, ValueError, IndexError):
                pass
            try:
                rec["busy_pct"] = float(open(os.path.join(self.card, "device", "gpu_busy_percent")).read())
            except (OSError, ValueError):
                pass
        if self.power_file:
            try:
                rec["power_w"] = float(open(self.power_file).read()) / 1e6
            except (OSError, ValueError):
                pass
        return rec

    def start(self):
        import threading

        def loop():
            while not self._stop:
                self.samples.append(self._read())
                time.sleep(self.period)
        self._thr = threading.Thread(target=loop, daemon=True)
        self._thr.start()
        return self

    def stop(self):
        self._stop = True
        if self._thr:
            self._thr.join(timeout=2.0)
        return self.samples

    @staticmethod
    def summarize(samples, t0, t1):
        sel = [r for r in samples if t0 <= r["t"] <= t1]
        out = {"samples": len(sel)}
        for k in ("sclk_mhz", "power_w", "busy_pct"):
            v = [r[k] for r in sel if k in r]
            out[k] = ({"mean": float(np.mean(v)), "min": float(np.min(v)), "max": float(np.max(v))} if v else None)
        return out


def sustained_run(step, fence, seconds, ms_per_step, flops_per_step, peak_tflops, device_index=0, pci_bus_id=None):
    """the driver's timed region is K steps (0.7 s at H); this leg runs the SAME step back to back for
    `seconds` of wall clock and reports evals/s of the first and the last 5 s, with the shader clock and socket power sampled
    beside it, so that the headline fraction can be read as a sustained figure."""
    # a FIXED step count from the measured step time (identical on every rank: it comes from the max-reduced timed region), never
    # a per-rank clock test - with a communicator every step is a collective and the ranks must agree on how many there are
    nsteps = max(1, int(math.ceil(seconds * 1e3 / ms_per_step)))
    tel = GpuTelemetry(device_index, pci_bus_id=pci_bus_id).start()
    fence()
    t_start = time.perf_counter()
    stamps = []
    for _ in range(nsteps):
        step()
        stamps.append(time.perf_counter())
    fence()
    t_end = time.perf_counter()
    samples = tel.stop()
    stamps = np.asarray(stamps)

    def window(a, b):
        k = int(np.sum((stamps > a) & (stamps <= b)))
        return {"steps": k, "evals_per_s": k / (b - a), "tflops": k * flops_per_step / (b - a) / 1e12,
                "frac_of_peak_whole_eval": k * flops_per_step / (b - a) / 1e12 / peak_tflops,
                "telemetry": GpuTelemetry.summarize(samples, a, b)}
    w = min(5.0, (t_end - t_start) / 2)
    return {"seconds": t_end - t_start, "steps": int(len(stamps)), "evals_per_s": len(stamps) / (t_end - t_start),
            "first_window": window(t_start, t_start + w), "last_window": window(t_end - w, t_end), "window_s": w,
            "telemetry_whole_run": GpuTelemetry.summarize(samples, t_start, t_end),
            "telemetry_source": {"card": tel.card, "matched_by_pci_bus_id": tel.card_matched_by_pci, "pci_bus_id": pci_bus_id,
                                 "power_file": tel.power_file, "period_s": tel.period},
            "note": "wall-clock evals/s of back-to-back svgp_elbo calls (prep + strips + reduce + read-back each); flops = 2 M^2 n + M^3/3"}


def profile_traffic(config, kernel_prefix):
    """HBM-side traffic of a kernel: PMC counters cannot be read from inside this process, so the figure is the one
    measured by tools/run_profile.sh (separate rocprofv3 --pmc passes, FETCH_SIZE doubled as MI355X_MICROARCH.md
    prescribes for gfx950) and committed under profiles/.  It is labelled with its source, and marked stale when the
    kernel sources changed after the profile was taken (the summary records their hash)."""
    import glob
    import hashlib

    h = hashlib.sha256()
    for f in ("strip.hip", "device_common.hpp"):
        h.update(open(os.path.join(ROOT, "approximategps.jl_amd", "csrc", f), "rb").read())
    cur = h.hexdigest()[:16]
    # forward profiles of this config (not the value-and-gradient ones), newest round first; a summary taken with exactly the
    # current kernel sources wins
    cands = [c for c in sorted(glob.glob(os.path.join(ROOT, "profiles", "round*", f"{config}_*_pmc.json")), reverse=True)
             if "grad" not in os.path.basename(c)]
    best = None
    for path in cands:
        try:
            pm = json.load(open(path))
        except (OSError, ValueError):
            continue
        hits = [v for k, v in pm.items() if k.startswith(kernel_prefix) and isinstance(v, dict) and v.get("traffic_bytes_per_launch")]
        if not hits:
            continue
        e = max(hits, key=lambda v: v["traffic_bytes_per_launch"])   # the main launch, not a concurrent half-width tail
        src_hash = pm.get("kernel_source_sha16")
        rec = {"traffic": e["traffic_bytes_per_launch"],
               "traffic_source": os.path.relpath(path, ROOT) + " (rocprofv3 --pmc, not measured in this run)",
               "traffic_stale": (src_hash != cur) if src_hash else None, "hbm_share_note": pm.get("hbm_share_note")}
        if src_hash == cur:
            return rec
        best = best or rec
    return best


def launcher_command(n_gpus, argv, port):
    """The command `python bench.py --gpus N` (N > 1, not under torchrun) starts: N fresh ranks, one per GPU, rendezvous on
    127.0.0.1 (the container hostname may not resolve).  `argv` = this invocation's own arguments, passed through."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(n_gpus)}",
            "--master-addr", "127.0.0.1", "--master-port", str(int(port)), os.path.abspath(__file__)] + list(argv)


def needs_self_launch(n_gpus, environ):
    """--gpus N > 1 asks for N ranks; without torchrun's environment (RANK / WORLD_SIZE) this process is not one of them."""
    return int(n_gpus) > 1 and "RANK" not in environ and int(environ.get("WORLD_SIZE", "1")) <= 1


def self_launch(n_gpus, argv):
    """Runs the N-rank job as CHILD processes (never exec; this parent has not touched HIP or torch at all) and relays rank
    0's JSON line as the last line of stdout.  Exit code = the launcher's; on failure a JSON line with value null is still
    printed and the exit code is non-zero."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = launcher_command(n_gpus, argv, port)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // int(n_gpus))))
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        ln = ln.strip()
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        elif ln:
            print(ln, file=sys.stderr)
    if line is None:
        line = json.dumps({"metric": "SVGP ELBO evals/sec at N=1e6, M=1024", "value": None, "unit": "evals/s", "n_gpus": int(n_gpus),
                           "error": f"the {n_gpus}-rank launch produced no result line (exit code {proc.returncode})",
                           "launcher": " ".join(cmd)})
    sys.stdout.write(line + "\n")
    sys.stdout.flush()
    if proc.returncode:
        return proc.returncode
    return 1 if json.loads(line).get("value") is None else 0


def julia_probe():
    """probe for the reference's own toolchain on the bench host.  Returns (path or None, version string)."""
    import shutil
    import subprocess

    exe = shutil.which("julia")
    if not exe:
        return None, "absent"
    try:
        out = subprocess.run([exe, "--version"], capture_output=True, text=True, timeout=60)
        return exe, (out.stdout or out.stderr).strip()
    except Exception as e:  # noqa: BLE001
        return exe, "present, --version failed: " + repr(e)


def julia_reference_baseline(exe, p, family, lik, n_sample):
    """If Julia AND the reference's packages are installed on this host: time the reference's own elbo (ApproximateGPs.jl) on a
    bounded sample of the same synthetic inputs (oracle/reference_julia_bench.jl reads them from an .npz).  Any failure is
    reported, never fatal - the restatement's number is always there."""
    import subprocess
    import tempfile

    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "problem.npz")
        np.savez(path, x=np.asarray(p["x"][:, :n_sample], dtype=np.float64), y=np.asarray(p["y"][:n_sample], dtype=np.float64),
                 z=np.asarray(p["z"], dtype=np.float64), m=np.asarray(p["m"], dtype=np.float64), Lq=np.asarray(p["Lq"], dtype=np.float64),
                 inv_lengthscale=np.asarray(p["inv_l"], dtype=np.float64), variance=float(p["variance"]), sigma2=float(p["sigma2"]),
                 jitter=float(p["jitter"]), family=int(family), lik=int(lik))
        try:
            out = subprocess.run([exe, "-t", "auto", os.path.join(ROOT, "oracle", "reference_julia_bench.jl"), path],
                                 capture_output=True, text=True, timeout=900)
        except Exception as e:  # noqa: BLE001
            return {"error": repr(e)}
        res = None
        for ln in out.stdout.splitlines():
            if ln.startswith("{"):
                try:
                    res = json.loads(ln)
                except ValueError:
                    pass
        if res is None:
            return {"error": "no result line", "rc": out.returncode, "stderr_tail": out.stderr[-400:]}
        if res.get("elbo") is not None:   # the pin: the oracle against the REAL reference on the very same sample
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import svgp_oracle as o

            f64 = lambda a: np.asarray(a, dtype=np.float64)
            sva = o.SVA(o.Kernel(family, p["variance"], p["inv_l"]), f64(p["z"]), f64(p["m"]), f64(p["Lq"]), jitter=p["jitter"])
            ref = o.elbo(sva, f64(p["x"][:, :n_sample]), f64(p["y"][:n_sample]), lik=lik, sigma2=p["sigma2"])
            res["oracle_elbo_same_sample"] = float(ref)
            res["oracle_vs_reference_rel_err"] = abs(float(ref) - res["elbo"]) / abs(res["elbo"])
        res["kind"] = "reference"
        res["sample_points"] = int(n_sample)
        return res


def measure_host_step(ctx, n=16384, M=1024, d=8, reps=20):
    """One minibatch TRAINING STEP as a host optimiser sees it (reference: examples/a-regression/script.jl:176-194 - new parameters
    every step, gradients consumed on the host): svgp_model_update from host memory (z, m, the M x M Lq) + svgp_elbo_grad with the
    gradients (the M x M Lq_bar among them) back in host memory.  Wall clock, median; `ms_device` is the same step's HIP-event time."""
    from approxgp import _ffi
    p = synth(7, n, M, d, SE, GAUSS, "f64")
    desc, keep = _ffi.make_desc(p["np_dt"], SE, p["variance"], p["inv_l"], p["z"], p["m"], p["Lq"], p["jitter"], lik_sigma2=p["sigma2"])
    model = _ffi.DeviceModel(ctx, desc, keep)
    data = _ffi.DeviceData(ctx, p["x"], p["y"], p["np_dt"])
    tu, tg, td, g = [], [], [], None
    try:
        for it in range(reps + 4):
            t0 = time.perf_counter()
            model.update(desc, keep)
            t1 = time.perf_counter()
            _, _, g = model.elbo_grad(data, 0, n, float(n), **({"out": g} if g is not None else {}))
            t2 = time.perf_counter()
            if it >= 4:
                tu.append(t1 - t0), tg.append(t2 - t1), td.append(ctx.timing().ms_total)
    finally:
        model.free()
        data.free()
    step = float(np.median(np.add(tu, tg))) * 1e3
    return {"workload": f"minibatch of {n} points, M={M}, d={d}, SE-ARD, Gaussian, f64; parameters from host memory every step, "
                        "gradients to host memory (pinned staging, reused host arrays)",
            "ms_step": step, "steps_per_s": 1e3 / step, "ms_model_update": float(np.median(tu)) * 1e3,
            "ms_elbo_grad": float(np.median(tg)) * 1e3, "ms_device": float(np.median(td)), "pcie_bytes_per_step": 8 * (M * M + M * d + M) + 8 * (M * (M + 1) // 2 + M * d + M),
            "pcie_note": "parameters up: dense M x M Lq as the C-ABI takes it; gradients down: Lq_bar as its packed lower triangle (round 5)"}


def timing_on(ctx=None):
    """The library's device timings exist unless the context was created with SVGP_TIMING=0 (svgp_last_timing then reports zeros).  Asked of
    the CONTEXT (what it captured at svgp_ctx_create: a timed evaluation has ms_total > 0), not of this process's present environment."""
    if ctx is not None:
        try:
            return float(ctx.timing().ms_total) > 0.0
        except Exception:  # noqa: BLE001
            pass
    return os.environ.get("SVGP_TIMING", "1").strip() != "0"


def measure_kuf(name, ctx, model, data, torch, dev):
    """Kuf assembly alone (SVA:216): M x n column-major written once -> HBM-write bound.  Its own launches, its own HIP events
    (recorded by the library on its stream around the launch), AFTER the ELBO loop so the clocks are up: median and p95 over 32
    launches; beside it what a plain write-only stream of the same size reaches on this box (report both)."""
    n, M, d, family, lik, dtype, cid = CONFIGS[name]
    es = 8 if dtype == "f64" else 4
    times = []
    for _ in range(34):
        model.kuf(data, 0, n, fetch=False)
        times.append(ctx.timing().ms_kuf)
    times = times[2:]
    t_kuf = float(np.median(times))
    if not (t_kuf > 0):   # a context without timing events (SVGP_TIMING=0): no device time to divide by
        return {"kernel": "kuf_cols_kernel", "bound": "hbm", "achieved": None, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                "note": "library timing events are off (SVGP_TIMING=0): no Kuf launch time"}
    bytes_alg = es * (M * n + n * d + M * d)
    gbs = bytes_alg / (t_kuf * 1e-3) / 1e9
    kuf_roofline = {"kernel": "kuf_cols_kernel (kuf_kernel for layouts it does not take)", "bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS,
                    "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS, "traffic": None,
                    "bytes_per_launch": bytes_alg, "ms_per_launch": t_kuf, "launches": len(times),
                    "ms_p95": float(np.percentile(times, 95)), "ms_min": float(np.min(times)),
                    "GBps_p95_launch": bytes_alg / (float(np.percentile(times, 95)) * 1e-3) / 1e9}
    if torch is not None:
        try:
            buf = torch.empty(M * n, dtype=torch.float64 if dtype == "f64" else torch.float32, device=dev)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ts = []
            for _ in range(12):
                ev[0].record()
                buf.fill_(1.0)
                ev[1].record()
                torch.cuda.synchronize()
                ts.append(ev[0].elapsed_time(ev[1]))
            t_fill = float(np.median(ts[2:]))
            kuf_roofline["stream_write_GBps"] = es * M * n / (t_fill * 1e-3) / 1e9
            kuf_roofline["frac_of_stream_write"] = gbs / kuf_roofline["stream_write_GBps"]
            del buf
        except Exception as e:  # noqa: BLE001
            kuf_roofline["stream_write_GBps"] = None
            kuf_roofline["stream_write_error"] = repr(e)
    tk = profile_traffic(name, "kuf_")   # kuf_cols_kernel (the default) or kuf_kernel
    if tk:
        kuf_roofline.update({k: v for k, v in tk.items() if k != "hbm_share_note" and (v is not None or k == "traffic_stale")})
    return kuf_roofline


def baseline_and_parity(args, name, p, model, data, res, world, value):
    """`cpu_baseline` (the fp64 oracle timed on the host cores) and `parity` (its ELBO against the device's on the same inputs).
    One GPU: the oracle evaluates the TIMED workload itself when the host has the RAM.  N > 1 ranks (rank 0 calls this for its OWN
    shard): a bounded sample of the shard, extrapolated linearly in N to the shard's size, and the device side through the local
    svgp_elbo_partial + svgp_prior_kl (svgp_elbo would be collective).  The unit stays evals/s of ONE shard-sized evaluation on
    the host, so `gpu_over_cpu` divides the per-GPU rate by it."""
    n, M, d, family, lik, dtype, cid = CONFIGS[name]
    tol = 1e-8 if dtype == "f64" else 1e-4
    jl_exe, jl_version = julia_probe()
    out = {}
    try:
        cb = cpu_baseline(p, family, lik, min(args.cpu_sample, n), n, M, force_sample=(world > 1))
        ref, npts = cb.pop("oracle_elbo"), cb.pop("oracle_points")
        cb["julia"] = jl_version
        if world > 1:
            cb["scope"] = f"rank 0's shard ({n} of the {n * world} global points); value = host evals/s of ONE shard-sized evaluation"
        out["cpu_baseline"] = cb
        if value:
            out["gpu_over_cpu"] = (value / world) / cb["value"]
        # the oracle saw the first `npts` points with num_data = npts: the GPU value for exactly that batch
        if world == 1:
            gpu_val = res["elbo"] if npts == n else model.elbo(data, 0, npts, float(npts))[0]
            via = "svgp_elbo"
        else:
            part = model.elbo_partial(data, 0, npts)
            gpu_val = float(part[0]) - model.prior_kl()[0]           # scale = num_data / n = 1
            via = "svgp_elbo_partial + svgp_prior_kl (local calls; rank 0's shard)"
        rel = abs(gpu_val - ref) / abs(ref)
        out["parity"] = {"oracle_elbo": ref, "gpu_elbo": gpu_val, "rel_err": rel, "tol": tol, "points": npts, "via": via,
                         "ok": bool(rel <= tol), "oracle": "oracle/svgp_oracle.py (fp64; parity UNPINNED: no output of the Julia "
                         "reference has been available to check it against)"}
        if jl_exe:   # the reference itself, when the host has it (never expected on the GPU box: it receives only this repo)
            out["cpu_baseline_reference_julia"] = julia_reference_baseline(jl_exe, p, family, lik, min(args.cpu_sample, n))
    except Exception as e:  # noqa: BLE001
        out["cpu_baseline"] = {"value": None, "unit": "evals/s", "cores": os.cpu_count(), "kind": "port", "julia": jl_version,
                               "sample": "failed: " + repr(e)}
    return out


def bench_config(args, name, ctx, torch, dist, dev, world, rank, use_dist, steps, warmup, num_data_override=None, host_comm=False,
                 resident=None):
    """Times `steps` full elbo evaluations of config `name` (after `warmup`), barrier + synchronize on both sides, MAX over
    ranks.  Returns (dict of measurements, model, data, p) with the model and data still resident.
    `resident` > n (C5): that many points stay in HBM and step i evaluates the minibatch window
    [(i mod W) n, (i mod W) n + n), W = resident // n - a different batch every step, as a training loop draws them."""
    from approxgp import _ffi

    n, M, d, family, lik, dtype, cid = CONFIGS[name]
    n_res = int(resident) if resident and resident > n else n
    nwin = n_res // n
    p = synth(cid, n_res, M, d, family, lik, dtype, rank)
    desc, keep = _ffi.make_desc(p["np_dt"], family, p["variance"], p["inv_l"], p["z"], p["m"], p["Lq"], p["jitter"],
                                likelihood=lik, lik_sigma2=p["sigma2"])
    model = _ffi.DeviceModel(ctx, desc, keep)
    data = _ffi.DeviceData(ctx, p["x"], p["y"], p["np_dt"])
    num_data = float(num_data_override if num_data_override else n * world)

    if host_comm:
        from approxgp.distributed import ShardedELBO
        sharded = ShardedELBO(model, data, num_data, device=dev)

    counter = [0]

    def step():
        # prep + fused strips + reduce (HIP library); with a communicator on ctx this call is the library's collective:
        # ONE ncclAllReduce of the device-resident 8-vector, every rank gets the global ELBO
        off = (counter[0] % nwin) * n
        counter[0] += 1
        if host_comm:   # fallback only: partial sums through torch.distributed
            class _T:
                n_points = n * world
            return sharded.step(off, n), _T
        return model.elbo(data, off, n, num_data)

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    strip_ms, prep_ms, expect_ms, chol_ms, overlap_ms, total_ms = [], [], [], [], [], []
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        val, terms = step()
        t = ctx.timing()
        strip_ms.append(t.ms_strip)
        prep_ms.append(t.ms_prep)
        expect_ms.append(t.ms_expect)
        chol_ms.append(t.ms_chol)
        overlap_ms.append(t.ms_overlap)
        total_ms.append(t.ms_total)
    fence()
    t_region = (t0, time.perf_counter())   # (for a telemetry thread the caller may run beside the region)
    elapsed = t_region[1] - t0
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    Mp = (M + 127) // 128 * 128
    strip_w = 64 if (dtype == "f64" or Mp > 2048) else 128   # strip.hip: strip_nt()
    strip_avg_ms = float(np.mean(strip_ms))
    flops_strip = 2.0 * M * M * n                           # algorithmic: trsm + trmm
    kernel_name = "strip_kernel (fused Kuf -> trsm -> trmm)"
    # A batch of at most one round of strips runs its phase 1 as segmented strips BESIDE the factorisation (svgp_timing.ms_overlap > 0):
    # ms_strip then covers only the launches behind the prep, while 2 M^2 n counts both phases - a "fraction" from those two exceeded 1
    # (VERDICT r4).  There is no single dominant launch in that regime: the roofline object is the WHOLE evaluation,
    # (2 M^2 n + M^3 / 3) flops over the device time of the step (events around prep + strips), which can never exceed the peak.
    overlapped = timing_on(ctx) and float(np.mean(overlap_ms)) > 0.0
    if overlapped:
        strip_avg_ms = float(np.mean(total_ms)) - float(np.mean(expect_ms))
        flops_strip = 2.0 * M * M * n + M ** 3 / 3.0
        kernel_name = "whole evaluation (segmented strip launches beside the factorisation: no single dominant launch)"
    ach = flops_strip / (strip_avg_ms * 1e-3) / 1e12 if strip_avg_ms > 0 else None
    res = {
        "elapsed": elapsed, "t_region": t_region, "ms_per_step": 1e3 * elapsed / steps, "evals_per_s": world * steps / elapsed,
        "points_per_s": n * world * steps / elapsed, "elbo": val, "n_points_global": int(terms.n_points),
        "roofline": {"kernel": kernel_name, "bound": "mfma", "achieved": ach,
                     "peak": PEAK_TFLOPS[dtype], "unit": "TFLOP/s", "frac": (ach / PEAK_TFLOPS[dtype]) if ach else None, "traffic": None,
                     "flops_per_launch": flops_strip, "ms_per_launch": strip_avg_ms, "overlapped_with_prep": bool(overlapped),
                     # MFMA work actually issued: full 128-row blocks below the diagonal + 20 of the 32 tile-steps of every
                     # (triangular) diagonal block, per phase, on whole strips (matches SQ_INSTS_MFMA x 2048 of the PMC profile)
                     "executed_flops_per_launch": 2.0 * 2.0 * 128 * 128 * ((Mp // 128) * (Mp // 128 - 1) / 2 + 0.625 * (Mp // 128))
                     * strip_w * math.ceil(n / strip_w)},
        "breakdown_ms": {"prep (Kuu, cholesky, T panels, KL)": float(np.mean(prep_ms)), "strip": strip_avg_ms,
                         "expectation + reduce (+ all-reduce)": float(np.mean(expect_ms)), "cholesky alone (inside prep)": float(np.mean(chol_ms))},
        # BASELINE config 4 / SURVEY Appendix G: "report Cholesky MFMA utilisation separately".  cholesky(Kuu) with its T panels,
        # HIP events of the library around the factorisation's launches; flops = M^3 / 3 (algorithmic).  Latency-bound by its
        # panel dependency, not a roofline kernel: the fraction says how far from the matrix peak the serial chain keeps it.
        "cholesky_roofline": {"kernels": "potf2 (in-kernel hand-over) + chol_tile / syrk128 MFMA tiles", "bound": "mfma (latency-bound chain)",
                              "flops": M ** 3 / 3.0, "ms": float(np.mean(chol_ms)),
                              "achieved": (M ** 3 / 3.0) / (float(np.mean(chol_ms)) * 1e-3) / 1e12 if np.mean(chol_ms) > 0 else None,
                              "peak": PEAK_TFLOPS[dtype], "unit": "TFLOP/s",
                              "frac": ((M ** 3 / 3.0) / (float(np.mean(chol_ms)) * 1e-3) / 1e12 / PEAK_TFLOPS[dtype]) if np.mean(chol_ms) > 0 else None},
        "workload": f"{name}: N={n} points per GPU, M={M}, d={d}, {['SE', 'Matern32', 'Matern52'][family]}-ARD, "
                    f"{['Gaussian', 'Bernoulli-logistic GH-20', 'Poisson'][lik]}, {dtype}, NonCentered; "
                    "one step = one full elbo(sva, lfx, y) incl. cholesky(Kuu)"
                    + (f"; {n_res} points resident per GPU, the batch window advances by {n} points every step ({nwin} windows)" if nwin > 1 else ""),
        "num_data": num_data, "resident_points_per_gpu": n_res, "windows": nwin, "step_fn": step, "fence_fn": fence,
    }
    return res, model, data, p


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)      # H: 100 x 34 ms: a timed region of > 3 s
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="H", choices=sorted(CONFIGS))
    ap.add_argument("--cpu-sample", type=int, default=100000)   # ~10 s of host work at H (used when RAM is short)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kuf", action="store_true")
    ap.add_argument("--no-grad", action="store_true")
    ap.add_argument("--no-c5", action="store_true")
    # builder-side runs (the driver's --steps / --warmup semantics are untouched): after the K timed steps, keep evaluating back to
    # back for this many seconds and report first / last 5 s with clock and power
    ap.add_argument("--min-seconds", type=float, default=0.0)
    ap.add_argument("--sustained-out", default=None, help="also write the `sustained` object to this JSON file")
    ap.add_argument("--c5-resident", type=int, default=int(os.environ.get("BENCH_C5_RESIDENT", "12500000")),
                    help="points resident per GPU for the C5 minibatch leg (SURVEY 8e: 1.25e7); the window moves every step")
    args = ap.parse_args()

    # `python bench.py --gpus N` with N > 1, started plainly: become the launcher of N fresh ranks (before torch / HIP are
    # even imported here), so that an N-GPU line is an N-rank RCCL run however the script was started
    if needs_self_launch(args.gpus, os.environ):
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))

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
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the library has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1"   # the env knob exercises the RCCL path on one GPU
    lib_comm_error = None
    stream = torch.cuda.current_stream().cuda_stream
    ctx = _ffi.Context(local_rank, stream if stream else None)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        from approxgp.distributed import attach_comm_via_torch
        try:
            attach_comm_via_torch(ctx)                        # the library's own RCCL communicator (ncclCommInitRank)
            assert ctx.comm_info() == (world, rank)
        except Exception as e:  # noqa: BLE001
            # never lose the scaling measurement to a communicator problem: fall back to the host-side combination
            # (svgp_elbo_partial + ONE torch.distributed all-reduce of 5 doubles) and say so in the output
            lib_comm_error = repr(e)
        # every rank must take the same path
        flag = torch.tensor([0.0 if lib_comm_error is None else 1.0], device=dev)
        dist.all_reduce(flag)
        if float(flag.item()) > 0 and lib_comm_error is None:
            ctx.detach_comm()
            lib_comm_error = "a peer rank could not attach the library communicator"

    name = args.config
    n, M, d, family, lik, dtype, cid = CONFIGS[name]
    host_comm = use_dist and lib_comm_error is not None
    res, model, data, p = bench_config(args, name, ctx, torch, dist, dev, world, rank, use_dist, args.steps, args.warmup,
                                       host_comm=host_comm)
    num_data = res["num_data"]
    out = {
        "metric": "SVGP ELBO evals/sec at N=1e6, M=1024" if name in ("H", "H32") else f"SVGP ELBO evals/sec ({name})",
        "value": res["evals_per_s"], "unit": "evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": dtype, "data": "synthetic",
        "config": {"workload": res["workload"], "global_points": n * world, "points_per_s": res["points_per_s"],
                   "parallelism": ("single GPU" if not use_dist else
                                   f"data-parallel shards x{world}; one 8-double ncclAllReduce per eval inside the library "
                                   "(svgp_elbo on a context with a communicator)" if not host_comm else
                                   f"data-parallel shards x{world}; FALLBACK: host-side all-reduce of the partial sums "
                                   f"(library communicator unavailable: {lib_comm_error})"),
                   "elbo": res["elbo"], "timed_region_s": res["elapsed"]},
        "roofline": res["roofline"], "breakdown_ms": res["breakdown_ms"], "cholesky_roofline": res["cholesky_roofline"],
        # what the library itself reports: the size of its RCCL communicator (svgp_ctx_comm_info; 1 = no communicator) and the
        # number of points the all-reduced evaluation covered (svgp_terms.n_points after the in-library ncclAllReduce)
        "rccl_world": int(ctx.comm_info()[0]), "n_points_global": res["n_points_global"],
    }
    if use_dist:
        assert res["n_points_global"] == n * world, (res["n_points_global"], n, world)
        if not host_comm:
            # cross-check of the in-library collective on this very run: every rank's LOCAL partial sums (svgp_elbo_partial,
            # never collective) all-reduced by torch.distributed must give the ELBO the library's own ncclAllReduce gave
            part, kl, err = (0.0, 0.0), 0.0, None
            try:
                part = model.elbo_partial(data, 0, n)
                kl, _ = model.prior_kl()
            except Exception as e:  # noqa: BLE001 - every rank still reaches the all-reduce below
                err = repr(e)
            tt = torch.tensor([part[0], part[1], 0.0 if err is None else 1.0], dtype=torch.float64, device=dev)
            dist.all_reduce(tt)
            if float(tt[2].item()) > 0:
                out["collective_check"] = {"error": err or "a peer rank failed its local evaluation"}
            else:
                ref = float(tt[0].item()) * (num_data / float(tt[1].item())) - kl
                out["collective_check"] = {"library_allreduce_elbo": res["elbo"], "torch_allreduce_of_local_partials": ref,
                                           "rel_err": abs(res["elbo"] - ref) / abs(ref)}
    if rank == 0:   # the Cholesky's MFMA utilisation is a PMC figure: from the committed profile of this configuration, labelled
        import glob
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "round*", f"{name}_*_pmc.json")), reverse=True):
            try:
                agg = json.load(open(path)).get("cholesky_aggregate")
            except (OSError, ValueError):
                agg = None
            if agg and "mfma_busy_frac" in agg and "grad" not in os.path.basename(path):
                out["cholesky_roofline"].update({"mfma_busy": agg["mfma_busy_frac"], "mfma_busy_source": os.path.relpath(path, ROOT) +
                                                 " (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES over every launch of the factorisation; not measured in this run)",
                                                 "ms_under_rocprof": agg["ms_per_evaluation"]})
                break
    tr = profile_traffic(name, "strip_kernel<") if rank == 0 else None
    if tr:
        out["roofline"].update({k: v for k, v in tr.items() if v is not None or k == "traffic_stale"})
        out["roofline"]["traffic_note"] = ("bytes per launch at the L2<->fabric boundary (FETCH_SIZE x2 + WRITE_SIZE); includes "
                                           "Infinity-Cache hits of the per-workgroup scratch strips; algorithmic HBM bytes are "
                                           f"{(8 if dtype == 'f64' else 4) * n * (d + 1) + 16 * n} B")

    # Kuf assembly alone (SVA:216), rank 0 only (a local call: no collective), also on N > 1
    if rank == 0 and not args.no_kuf:
        out["kuf_roofline"] = measure_kuf(name, ctx, model, data, torch, dev)

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    if args.min_seconds > 0 and not host_comm:
        flops_eval = 2.0 * M * M * n + M ** 3 / 3.0
        try:
            pr = torch.cuda.get_device_properties(local_rank)
            pci = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        except Exception:  # noqa: BLE001
            pci = None
        sus = sustained_run(res["step_fn"], res["fence_fn"], args.min_seconds, res["ms_per_step"], flops_eval, PEAK_TFLOPS[dtype], local_rank, pci)
        sus["twenty_step_region"] = {"evals_per_s": res["evals_per_s"] / world, "ms_per_step": res["ms_per_step"],
                                     "strip_frac": res["roofline"]["frac"]}
        sus["n_gpus"] = world          # evals_per_s above are per rank; every step is one collective evaluation on all ranks
        out["sustained"] = sus
        if rank == 0 and args.sustained_out:
            os.makedirs(os.path.dirname(os.path.abspath(args.sustained_out)), exist_ok=True)
            json.dump({"workload": res["workload"], "lib_sha16": lib_sha16(), **sus}, open(args.sustained_out, "w"), indent=1)
    res.pop("step_fn", None)
    res.pop("fence_fn", None)

    if not args.no_grad and not host_comm:
        # value-and-gradient evaluation (svgp_elbo_grad: what a training step costs), same workload, same residency; on N > 1
        # GPUs the library's collective form (batch size all-reduced on the device, one grouped gradient all-reduce).
        # Reported beside the headline, never part of it; a failure here must not cost the main line.
        try:
            _, _, gout = model.elbo_grad(data, 0, n, num_data)
            fence()
            t0 = time.perf_counter()
            reps = 3
            for _ in range(reps):   # out=: the gradient lands in the previous step's host arrays, as in a training loop
                gval, _, gout = model.elbo_grad(data, 0, n, num_data, out=gout)
            fence()
            tg = (time.perf_counter() - t0) / reps
            if use_dist:
                tt = torch.tensor([tg], dtype=torch.float64, device=dev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                tg = float(tt.item())
            out["value_and_gradient"] = {"evals_per_s": world / tg, "ms_per_eval": 1e3 * tg,
                                         "ratio_to_forward": 1e3 * tg / res["ms_per_step"], "value": gval}
        except Exception as e:  # noqa: BLE001
            out["value_and_gradient"] = {"error": repr(e)}
    if name == "H" and world == 1 and not args.no_grad and not host_comm:
        try:
            out["host_training_step"] = measure_host_step(ctx)
        except Exception as e:  # noqa: BLE001
            out["host_training_step"] = {"error": repr(e)}
    if name == "H" and not args.no_c5:
        # BASELINE config C5 (8 x MI355X: minibatched ELBO, N = 1e8, per-GPU batch 2^18, M = 1024, fp32) through the same
        # collective path: every rank evaluates its own 2^18-point minibatch per step, scale = 1e8 / (world * 2^18).
        try:
            # the shader clock during C5's timed region (0.1 s of back-to-back 5 ms steps, sampled every 10 ms): a short region runs on a
            # clock that is still ramping, which is what separates its roofline fraction from a profiled run's
            try:
                pr5 = torch.cuda.get_device_properties(local_rank)
                pci5 = f"{pr5.pci_domain_id:04x}:{pr5.pci_bus_id:02x}:{pr5.pci_device_id:02x}.0"
            except Exception:  # noqa: BLE001
                pci5 = None
            tel5 = GpuTelemetry(local_rank, period=0.01, pci_bus_id=pci5).start() if rank == 0 else None
            c5, m5, d5, _ = bench_config(args, "C5", ctx, torch, dist, dev, world, rank, use_dist, max(20, args.steps), 5,
                                         num_data_override=C5_NUM_DATA, host_comm=host_comm, resident=args.c5_resident)
            clk5 = GpuTelemetry.summarize(tel5.stop(), *c5["t_region"]) if tel5 else None
            nwin5 = c5["windows"]
            c5out = {"workload": c5["workload"] + f"; num_data = {C5_NUM_DATA:.0e}, global minibatch = {world} x 262144",
                     "resident_points_per_gpu": c5["resident_points_per_gpu"], "windows": nwin5,
                     "minibatch_steps_per_s": c5["evals_per_s"] / world, "ms_per_step": c5["ms_per_step"],
                     "points_per_s": c5["points_per_s"], "dtype": "f32", "roofline": c5["roofline"],
                     "breakdown_ms": c5["breakdown_ms"], "elbo": c5["elbo"], "n_points_global": c5["n_points_global"]}
            if clk5 is not None:
                c5out["shader_clock_mhz_during_timed_region"] = clk5.get("sclk_mhz")
                c5out["shader_clock_source"] = {"card": tel5.card, "matched_by_pci_bus_id": tel5.card_matched_by_pci}
                c5out["shader_clock_note"] = ("roofline.frac is flops / time / the 2.4 GHz peak: a region run at a lower (ramping or power-limited) "
                                              "clock reads proportionally lower; samples = " + str(clk5.get("samples")))
            if not args.no_grad and not host_comm:
                _, _, g5 = m5.elbo_grad(d5, 0, 262144, C5_NUM_DATA)
                fence()
                t0 = time.perf_counter()
                reps5 = 10
                for i5 in range(reps5):   # a different resident window every training step
                    _, _, g5 = m5.elbo_grad(d5, ((i5 + 1) % nwin5) * 262144, 262144, C5_NUM_DATA, out=g5)
                fence()
                tg = (time.perf_counter() - t0) / reps5
                if use_dist:
                    tt = torch.tensor([tg], dtype=torch.float64, device=dev)
                    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                    tg = float(tt.item())
                # one training step = value + gradient of the global minibatch ELBO on every rank (svgp_elbo_grad: the batch size
                # and the gradient blocks all-reduced inside the library), wall clock, max over ranks
                c5out["training_step_ms"] = 1e3 * tg
                c5out["training_steps_per_s"] = 1.0 / tg
                c5out["training_points_per_s"] = world * 262144 / tg
                c5out["training_step_ratio_to_forward"] = 1e3 * tg / c5["ms_per_step"]
            m5.free()
            d5.free()
            out["c5_minibatch"] = c5out
        except Exception as e:  # noqa: BLE001
            out["c5_minibatch"] = {"error": repr(e)}

    # CPU baseline + parity of the TIMED workload (parity is a gate before any timing counts), rank 0 only, on
    # N > 1 too (its own shard, through the LOCAL svgp_elbo_partial: the other ranks wait at the closing barrier meanwhile)
    if rank == 0 and not args.no_cpu_baseline:
        out.update(baseline_and_parity(args, name, p, model, data, res, world, out["value"]))
        if out.get("parity") and not out["parity"]["ok"]:
            out["value_withdrawn"] = out["value"]
            out["value"] = None
    out["lib_sha16"] = lib_sha16()
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
