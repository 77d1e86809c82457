"""(Ran at commit 4434fb5, the last one that holds both forms - the VALU kernels left the tree right after it:
profiles/round6/kgrad_valu_kernels_removed.patch; from the repo root: python profiles/round6/scripts/kgrad_ab.py ...)
A/B of the kernel-gradient reductions: the VALU kernels of rounds 2-5 (SVGP_KGRAD_MFMA=0) against the MFMA form (round 6), on the
EXPERIMENTS library (the knob is read once per process, so every leg is its own process).  For each bench configuration given:
value-and-gradient time (best of 3) and the gradient blocks' largest difference between the two legs relative to the block's scale.
usage: python tools/round6/kgrad_ab.py H Hd16 Hd32 Hd64 H32 H32d64 C3 C5   (leg mode: --leg <0|1> <cfg> <out.npz>)"""
import os, subprocess, sys, time
R = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(R, "..", "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "approximategps.jl_amd"))
import numpy as np


def leg(cfg, out, truth=False):
    import bench
    from approxgp import _ffi
    n, M, d, family, lik, dtype, cid = bench.CONFIGS[cfg]
    p = bench.synth(cid, n, M, d, family, lik, dtype)
    if truth:   # the fp64 library on the fp32-rounded inputs: what an exact fp32-input evaluation would return
        p = {k: (np.asarray(v, dtype=np.float64) if isinstance(v, np.ndarray) else v) for k, v in p.items()}
        p["np_dt"] = np.float64
    ctx = _ffi.Context(0)
    desc, keep = _ffi.make_desc(p["np_dt"], family, p["variance"], p["inv_l"], p["z"], p["m"], p["Lq"], p["jitter"], likelihood=lik,
                                lik_sigma2=p["sigma2"], neg_var_policy=_ffi.NEGVAR_CLAMP)
    model = _ffi.DeviceModel(ctx, desc, keep); data = _ffi.DeviceData(ctx, p["x"], p["y"], p["np_dt"])
    model.elbo_grad(data, 0, n, float(n))
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); v, _, g = model.elbo_grad(data, 0, n, float(n)); ts.append(time.perf_counter() - t0)
    np.savez(out, ms=min(ts) * 1e3, value=v, variance=g["variance"], inv_lengthscale=g["inv_lengthscale"], z=g["z"], m=g["m"], Lq=g["Lq"])


if __name__ == "__main__":
    if sys.argv[1] == "--leg":
        leg(sys.argv[3], sys.argv[4], truth=(sys.argv[2] == "t")); sys.exit(0)
    lib = os.path.join(ROOT, "approximategps.jl_amd", "csrc", "ablate", "libsvgp_experiments.so")
    import tempfile
    tmp = tempfile.mkdtemp(prefix="kgrad_ab_")   # (the legs' gradient blocks are 8 MB each: not into gpurun_out, which travels back)
    for cfg in sys.argv[1:]:
        res = {}
        import bench
        legs = ("0", "1", "t") if bench.CONFIGS[cfg][5] == "f32" else ("0", "1")
        for k in legs:
            out = os.path.join(tmp, f"kgrad_ab_{cfg}_{k}.npz")
            env = dict(os.environ, SVGP_MI355X_LIB=lib, SVGP_KGRAD_MFMA=("1" if k == "t" else k))
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--leg", k, cfg, out], env=env, capture_output=True, text=True, timeout=900)
            if r.returncode != 0:
                print(cfg, "leg", k, "FAILED", r.stderr[-400:]); break
            res[k] = np.load(out)
        if len(res) < len(legs):
            continue
        a, b = res["0"], res["1"]
        if "t" in res:   # error of each fp32 leg against the fp64 evaluation of the same (fp32-rounded) inputs
            for name, leg_ in (("valu", a), ("mfma", b)):
                err = {}
                for key in ("variance", "inv_lengthscale", "z", "m", "Lq"):
                    x0, x1 = np.atleast_1d(res["t"][key]).astype(np.float64), np.atleast_1d(leg_[key]).astype(np.float64)
                    err[key] = float(np.abs(x0 - x1).max() / max(np.abs(x0).max(), 1e-300))
                print(f"{cfg}: {name} vs fp64-on-fp32-inputs  " + "  ".join(f"{k} {v:.1e}" for k, v in err.items()), flush=True)
        dif = {}
        for key in ("variance", "inv_lengthscale", "z", "m", "Lq"):
            x0, x1 = np.atleast_1d(a[key]).astype(np.float64), np.atleast_1d(b[key]).astype(np.float64)
            dif[key] = float(np.abs(x0 - x1).max() / max(np.abs(x0).max(), 1e-300))
        print(f"{cfg}: valu {float(a['ms']):.2f} ms  mfma {float(b['ms']):.2f} ms  value diff {abs(float(a['value']) - float(b['value'])) / abs(float(a['value'])):.1e}  "
              + "  ".join(f"{k} {v:.1e}" for k, v in dif.items()), flush=True)
