"""Small problems: where does the drop-in pay?  (the reference's own workloads are tiny:
/root/reference/examples/a-regression/script.jl:33,69,176 - N = 10 000, M = 20, minibatch 100; BASELINE C1 - N = 1000, M = 32.)

For each (resident N, batch, M, d): wall time per call of
  resident    model.elbo on resident data (svgp_elbo: prep + strips + read-back)
  update      svgp_model_update + svgp_elbo (what a training loop with resident handles pays per forward evaluation)
  grad        svgp_elbo_grad (value and gradient)
  oneshot     svgp_elbo_host: model create + upload + evaluate + free (what the un-modified Julia `elbo` call reaches via the hook)
  cpu / cpu_grad   the numpy oracle on the host cores (a stand-in for the reference's CPU path; Julia itself is absent)
Writes a markdown table + JSON to gpurun_out/small_problems.{md,json}."""
import ctypes as C, json, os, sys, time
R = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(R, "..")); sys.path.insert(0, os.path.join(R, "..", "approximategps.jl_amd")); sys.path.insert(0, os.path.join(R, "..", "oracle"))  # (R = tests/)
import numpy as np
import torch  # noqa: F401  (HIP runtime load order)
import svgp_oracle as o
from approxgp import _ffi
from approxgp.synthetic import synth_arrays

def med(fn, reps):
    fn(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e6

ctx = _ffi.Context(0)
rows = []
CASES = ((1000, 1000, 32, 1), (10000, 100, 20, 1), (10000, 10000, 20, 1), (4096, 4096, 128, 8), (4096, 4096, 512, 8), (16384, 16384, 1024, 8),
         (100000, 100000, 128, 8), (100000, 2048, 512, 8))
for dt in (np.float64, np.float32):
    for (N, B, M, d) in CASES:
        a = synth_arrays(1, N, M, d, dtype=dt)
        desc, keep = _ffi.make_desc(dt, 0, a["variance"], a["inv_lengthscale"], a["z"], a["m"], a["Lq"], a["jitter"], likelihood=0, lik_sigma2=a["sigma2"])
        model = _ffi.DeviceModel(ctx, desc, keep); data = _ffi.DeviceData(ctx, a["x"], a["y"], dt)
        xb = np.asfortranarray(a["x"][:, :B] if d > 1 else a["x"][:B]).astype(dt); yb = np.ascontiguousarray(a["y"][:B]).astype(dt)
        out, terms = C.c_double(), _ffi.Terms()
        def oneshot():
            rc = ctx.lib.svgp_elbo_host(ctx.h, C.byref(desc), _ffi.COLVECS if d > 1 else _ffi.VEC, B, xb.ctypes.data_as(C.c_void_p),
                                        yb.ctypes.data_as(C.c_void_p), float(N), C.byref(out), C.byref(terms))
            assert rc == 0, rc
        r = {"dtype": np.dtype(dt).name, "N_resident": N, "batch": B, "M": M, "d": d}
        r["resident_us"] = med(lambda: model.elbo(data, 0, B, float(N)), 30)
        t = ctx.timing(); r["device_prep_us"], r["device_strip_us"] = t.ms_prep * 1e3, t.ms_strip * 1e3
        r["update_us"] = med(lambda: (model.update(desc, keep), model.elbo(data, 0, B, float(N))), 30)
        r["grad_us"] = med(lambda: model.elbo_grad(data, 0, B, float(N)), 20)
        r["oneshot_us"] = med(oneshot, 20)
        if dt == np.float64:
            f64 = lambda v: np.asarray(v, dtype=np.float64)
            sva = o.SVA(o.Kernel(0, a["variance"], a["inv_lengthscale"]), f64(a["z"]), f64(a["m"]), f64(a["Lq"]), jitter=a["jitter"])
            xs, ys = f64(a["x"][..., :B]), f64(a["y"][:B])
            reps = 5 if B * M <= 5_000_000 else 2
            r["cpu_us"] = med(lambda: o.elbo(sva, xs, ys, sigma2=a["sigma2"], num_data=float(N)), reps)
            r["cpu_grad_us"] = med(lambda: o.elbo_grad(sva, xs, ys, sigma2=a["sigma2"], num_data=float(N)), reps)
            ref = o.elbo(sva, xs, ys, sigma2=a["sigma2"], num_data=float(N))
            r["rel_err"] = abs(model.elbo(data, 0, B, float(N))[0] - ref) / abs(ref)
        rows.append(r)
        print(r, flush=True)
        model.free(); data.free()
os.makedirs(os.path.join(R, "..", "gpurun_out"), exist_ok=True)
json.dump(rows, open(os.path.join(R, "..", "gpurun_out", "small_problems.json"), "w"), indent=1)
with open(os.path.join(R, "..", "gpurun_out", "small_problems.md"), "w") as f:
    f.write("| dtype | resident N | batch | M | d | resident elbo us | update+elbo us | value+grad us | one-shot us | device prep / strip us | CPU oracle elbo us | CPU oracle grad us |\n|---|---|---|---|---|---|---|---|---|---|---|---|\n")
    for r in rows:
        f.write(f"| {r['dtype']} | {r['N_resident']} | {r['batch']} | {r['M']} | {r['d']} | {r['resident_us']:.0f} | {r['update_us']:.0f} | {r['grad_us']:.0f} | {r['oneshot_us']:.0f} | "
                f"{r['device_prep_us']:.0f} / {r['device_strip_us']:.0f} | {r.get('cpu_us', float('nan')):.0f} | {r.get('cpu_grad_us', float('nan')):.0f} |\n")
