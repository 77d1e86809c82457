"""ctypes binding of libsvgp_mi355x.so (include/svgp_mi355x.h) — the same symbols the Julia shim
`ccall`s.  There is no CPU fallback: a missing library or a missing GPU raises."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SVGP_MI355X_LIB") or os.path.join(os.path.dirname(_HERE), "csrc", "libsvgp_mi355x.so")

OK, INVALID_ARG, NOT_POSDEF, NEG_VARIANCE, UNSUPPORTED, HIP_ERROR, RCCL_ERROR, OOM = range(8)
F64, F32 = 0, 1
COLVECS, ROWVECS, VEC = 0, 1, 2
KERNEL_SE, KERNEL_MATERN32, KERNEL_MATERN52 = 0, 1, 2
LIK_GAUSSIAN, LIK_BERNOULLI_LOGISTIC, LIK_POISSON_EXP, LIK_EXPONENTIAL_EXP, LIK_GAMMA_EXP, LIK_BERNOULLI_NORMCDF = 0, 1, 2, 3, 4, 5
NONCENTERED, CENTERED = 0, 1
NEGVAR_ERROR, NEGVAR_CLAMP = 0, 1


class ModelDesc(C.Structure):
    _fields_ = [
        ("dtype", C.c_int32), ("kernel", C.c_int32), ("parametrization", C.c_int32), ("likelihood", C.c_int32),
        ("quadrature_n", C.c_int32), ("layout_z", C.c_int32), ("neg_var_policy", C.c_int32), ("d", C.c_int32),
        ("M", C.c_int64), ("variance", C.c_double), ("inv_lengthscale", C.POINTER(C.c_double)),
        ("mean_const", C.c_double), ("jitter", C.c_double), ("lik_sigma2", C.c_double),
        ("z", C.c_void_p), ("m", C.c_void_p), ("Lq", C.c_void_p),
    ]


class Terms(C.Structure):
    _fields_ = [
        ("elbo", C.c_double), ("expectation", C.c_double), ("kl", C.c_double), ("scale", C.c_double),
        ("logdet_kuu", C.c_double), ("n_points", C.c_int64), ("n_neg_var", C.c_int64),
        ("chol_info", C.c_int32), ("reserved", C.c_int32),
    ]


class Grads(C.Structure):
    _fields_ = [("variance", C.c_double), ("lik_sigma2", C.c_double), ("mean_const", C.c_double),
                ("inv_lengthscale", C.POINTER(C.c_double)), ("z", C.c_void_p), ("m", C.c_void_p), ("Lq", C.c_void_p)]


class Timing(C.Structure):
    _fields_ = [("ms_total", C.c_double), ("ms_prep", C.c_double), ("ms_strip", C.c_double), ("ms_expect", C.c_double),
                ("ms_kuf", C.c_double),
                ("strip_launches", C.c_int64),
                # appended since ABI v3 (read through svgp_last_timing_sized): v4 ms_chol, v5 ms_overlap
                ("ms_chol", C.c_double), ("ms_overlap", C.c_double)]


# every symbol include/svgp_mi355x.h declares: (restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    "svgp_version": (C.c_int32, []),
    "svgp_device_count": (C.c_int32, []),
    "svgp_ctx_create": (C.c_int32, [C.c_int32, _P, C.POINTER(_P)]),
    "svgp_ctx_destroy": (C.c_int32, [_P]),
    "svgp_last_error": (C.c_char_p, [_P]),
    "svgp_last_timing": (C.c_int32, [_P, C.POINTER(Timing)]),
    "svgp_last_timing_sized": (C.c_int32, [_P, _P, C.c_int64]),
    "svgp_data_upload": (C.c_int32, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_int64, _P, _P, C.POINTER(_P)]),
    "svgp_data_wrap_device": (C.c_int32, [_P, C.c_int32, C.c_int32, C.c_int64, C.c_int64, _P, _P, C.POINTER(_P)]),
    "svgp_data_free": (C.c_int32, [_P, _P]),
    "svgp_model_create": (C.c_int32, [_P, C.POINTER(ModelDesc), C.POINTER(_P)]),
    "svgp_model_update": (C.c_int32, [_P, _P, C.POINTER(ModelDesc)]),
    "svgp_model_free": (C.c_int32, [_P, _P]),
    "svgp_elbo": (C.c_int32, [_P, _P, _P, C.c_int64, C.c_int64, C.c_double, C.POINTER(C.c_double), C.POINTER(Terms)]),
    "svgp_elbo_partial": (C.c_int32, [_P, _P, _P, C.c_int64, C.c_int64, C.POINTER(C.c_double)]),
    "svgp_elbo_grad": (C.c_int32, [_P, _P, _P, C.c_int64, C.c_int64, C.c_double, C.POINTER(C.c_double), C.POINTER(Terms),
                                   C.POINTER(Grads)]),
    "svgp_elbo_grad_shard": (C.c_int32, [_P, _P, _P, C.c_int64, C.c_int64, C.c_double, C.c_double, C.POINTER(C.c_double),
                                         C.POINTER(Terms), C.POINTER(Grads)]),
    "svgp_marginals": (C.c_int32, [_P, _P, _P, C.c_int64, C.c_int64, _P, _P]),
    "svgp_elbo_grad_ext": (C.c_int32, [_P, _P, _P, C.c_int64, C.c_int64, C.c_double, C.c_double, _P, _P, C.POINTER(C.c_double),
                                       C.POINTER(Terms), C.POINTER(Grads)]),
    "svgp_prior_kl": (C.c_int32, [_P, _P, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "svgp_elbo_host": (C.c_int32, [_P, C.POINTER(ModelDesc), C.c_int32, C.c_int64, _P, _P, C.c_double,
                                   C.POINTER(C.c_double), C.POINTER(Terms)]),
    "svgp_posterior": (C.c_int32, [_P, _P, _P, _P, _P]),
    "svgp_predict": (C.c_int32, [_P, _P, C.c_int32, C.c_int64, _P, _P, _P, _P]),
    "svgp_predict_cross_cov": (C.c_int32, [_P, _P, C.c_int32, C.c_int64, _P, C.c_int64, _P, _P]),
    "svgp_kuf": (C.c_int32, [_P, _P, _P, C.c_int64, C.c_int64, _P]),
    "svgp_gausshermite": (C.c_int32, [C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "svgp_offload_advice": (C.c_int32, [C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32]),
    "svgp_offload_work": (C.c_double, [C.c_int64, C.c_int64, C.c_int32]),
    # multi-GPU
    "svgp_comm_unique_id": (C.c_int32, [_P]),
    "svgp_ctx_attach_comm": (C.c_int32, [_P, _P, C.c_int32, C.c_int32]),
    "svgp_ctx_detach_comm": (C.c_int32, [_P]),
    "svgp_ctx_comm_info": (C.c_int32, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "svgp_group_create": (C.c_int32, [C.c_int32, C.POINTER(C.c_int32), C.POINTER(_P)]),
    "svgp_group_destroy": (C.c_int32, [_P]),
    "svgp_group_size": (C.c_int32, [_P]),
    "svgp_group_ctx": (_P, [_P, C.c_int32]),
    "svgp_group_last_error": (C.c_char_p, [_P]),
    "svgp_group_data_upload": (C.c_int32, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_int64, _P, _P, C.POINTER(_P)]),
    "svgp_group_model_create": (C.c_int32, [_P, C.POINTER(ModelDesc), C.POINTER(_P)]),
    "svgp_group_model_update": (C.c_int32, [_P, C.POINTER(_P), C.POINTER(ModelDesc)]),
    "svgp_group_elbo": (C.c_int32, [_P, C.POINTER(_P), C.POINTER(_P), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_double,
                                    C.POINTER(C.c_double), C.POINTER(Terms)]),
    "svgp_group_elbo_grad": (C.c_int32, [_P, C.POINTER(_P), C.POINTER(_P), C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                         C.c_double, C.POINTER(C.c_double), C.POINTER(Terms), C.POINTER(Grads)]),
}
COMM_ID_BYTES = 128

_lib = None


def load_library():
    """dlopen the in-tree library and type every symbol.  Raises (never falls back) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with ./build.sh (hipcc --offload-arch=gfx950). "
            "This package has no CPU implementation."
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


# ---- exceptions mirroring the Julia ones the shim raises --------------------------
class SvgpError(RuntimeError):
    pass


class PosDefException(SvgpError):
    def __init__(self, info, msg=""):
        super().__init__(msg or f"matrix is not positive definite; Cholesky factorization failed (info={info})")
        self.info = info


class DomainError(SvgpError):
    pass


class DeclinedError(SvgpError):
    """The problem is below the library's offload threshold (svgp_offload_advice): the Julia hooks return `nothing` here and
    the reference's own method body runs; the Python mirror has no host path to fall back to and says so."""


def offload_advice(n_points: int, M: int, d: int, dtype=F64, want_gradient=False) -> bool:
    """True when a problem of this size is worth sending to the device (include/svgp_mi355x.h: svgp_offload_advice)."""
    return bool(load_library().svgp_offload_advice(int(n_points), int(M), int(d), int(dtype), int(bool(want_gradient))))


def offload_work(n_points: int, M: int, d: int) -> float:
    return float(load_library().svgp_offload_work(int(n_points), int(M), int(d)))


class UnsupportedError(SvgpError):
    pass


def np_dtype(dtype: int):
    return np.float64 if dtype == F64 else np.float32


def dtype_code(dt) -> int:
    dt = np.dtype(dt)
    if dt == np.float64:
        return F64
    if dt == np.float32:
        return F32
    raise UnsupportedError(f"unsupported element type {dt}; use float64 or float32")


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Context:
    """One GPU + one HIP stream.  `stream` may be a raw hipStream_t (e.g. torch.cuda.current_stream().cuda_stream)."""

    def __init__(self, device: int = 0, stream: int | None = None):
        self.lib = load_library()
        if self.lib.svgp_device_count() < 1:
            raise SvgpError("no HIP device visible: libsvgp_mi355x has no CPU path")
        h = C.c_void_p()
        rc = self.lib.svgp_ctx_create(device, C.c_void_p(stream) if stream else None, C.byref(h))
        if rc != OK:
            raise SvgpError(f"svgp_ctx_create failed with status {rc}")
        self.h = h
        self.device = device

    def close(self):
        if getattr(self, "h", None):
            self.lib.svgp_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc: int, terms: Terms | None = None):
        if rc == OK:
            return
        msg = (self.lib.svgp_last_error(self.h) or b"").decode()
        if rc == INVALID_ARG:
            raise ValueError(msg)  # Julia ArgumentError
        if rc == NOT_POSDEF:
            raise PosDefException(terms.chol_info if terms is not None else -1, msg)
        if rc == NEG_VARIANCE:
            raise DomainError(msg)
        if rc == UNSUPPORTED:
            raise UnsupportedError(msg)
        if rc == OOM:
            raise MemoryError(msg)
        raise SvgpError(f"status {rc}: {msg}")

    def timing(self) -> Timing:
        t = Timing()
        self.lib.svgp_last_timing_sized(self.h, C.byref(t), C.sizeof(Timing))
        return t

    # ---- multi-GPU: one process per GPU (include/svgp_mi355x.h "multi-GPU") ----
    def attach_comm(self, comm_id: bytes, world_size: int, rank: int):
        """Collective over all ranks (ncclCommInitRank).  Afterwards DeviceModel.elbo / elbo_grad on this context are
        collective and return the GLOBAL ELBO; elbo_partial / elbo_grad(shard=...) stay local."""
        if len(comm_id) != COMM_ID_BYTES:
            raise ValueError("communicator id must be SVGP_COMM_ID_BYTES long")
        buf = C.create_string_buffer(bytes(comm_id), COMM_ID_BYTES)
        self.check(self.lib.svgp_ctx_attach_comm(self.h, C.cast(buf, C.c_void_p), world_size, rank))

    def detach_comm(self):
        self.check(self.lib.svgp_ctx_detach_comm(self.h))

    def comm_info(self):
        w, r = C.c_int32(), C.c_int32()
        self.lib.svgp_ctx_comm_info(self.h, C.byref(w), C.byref(r))
        return w.value, r.value


def comm_unique_id() -> bytes:
    """ncclGetUniqueId through the library (rank 0 calls it; the host transports the bytes to the other ranks)."""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    rc = load_library().svgp_comm_unique_id(C.cast(buf, C.c_void_p))
    if rc != OK:
        raise SvgpError(f"svgp_comm_unique_id failed with status {rc} (is librccl loadable?)")
    return buf.raw


_default_ctx: Context | None = None


def default_context() -> Context:
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


class DeviceData:
    """x = lfx.fx.x and y resident in HBM (svgp_data)."""

    def __init__(self, ctx: Context, x: np.ndarray, y: np.ndarray | None, dtype, layout: int = COLVECS):
        self.ctx = ctx
        dt = np_dtype(dtype_code(dtype))
        x = np.asarray(x, dtype=dt)
        if x.ndim == 1:
            layout, d, n = VEC, 1, x.shape[0]
            xbuf = np.ascontiguousarray(x)
        elif layout == COLVECS:
            d, n = x.shape  # (d, n) numpy view of Julia's d×n column-major matrix: point-contiguous
            xbuf = np.asfortranarray(x)
        else:
            n, d = x.shape  # RowVecs: n×d column-major, feature-contiguous
            xbuf = np.asfortranarray(x)
        self._x = xbuf
        self._y = None if y is None else np.ascontiguousarray(np.asarray(y, dtype=dt))
        if self._y is not None and self._y.shape[0] != n:
            raise ValueError("x and y lengths differ")
        self.n, self.d, self.dtype = n, d, dtype_code(dt)
        h = C.c_void_p()
        ctx.check(ctx.lib.svgp_data_upload(ctx.h, self.dtype, layout, d, n, _ptr(self._x), _ptr(self._y), C.byref(h)))
        self.h = h

    @classmethod
    def wrap(cls, ctx: Context, dtype, d: int, n: int, ldx: int, x_ptr: int, y_ptr: int | None):
        self = cls.__new__(cls)
        self.ctx, self.n, self.d, self.dtype = ctx, n, d, dtype_code(dtype)
        h = C.c_void_p()
        ctx.check(ctx.lib.svgp_data_wrap_device(ctx.h, self.dtype, d, n, ldx, C.c_void_p(x_ptr),
                                                C.c_void_p(y_ptr) if y_ptr else None, C.byref(h)))
        self.h = h
        return self

    def free(self):
        if getattr(self, "h", None):
            self.ctx.lib.svgp_data_free(self.ctx.h, self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def make_desc(dtype, kernel, variance, inv_lengthscale, z, m, Lq, jitter, *, parametrization=NONCENTERED,
              likelihood=LIK_GAUSSIAN, lik_sigma2=1.0, quadrature_n=0, mean_const=0.0, neg_var_policy=NEGVAR_ERROR,
              layout_z=COLVECS):
    """Builds a svgp_model_desc plus the numpy buffers it borrows (keep the returned tuple alive)."""
    code = dtype_code(dtype)
    dt = np_dtype(code)
    z = np.asarray(z, dtype=dt)
    if z.ndim == 1:
        layout_z, d, M = VEC, 1, z.shape[0]
        zbuf = np.ascontiguousarray(z)
    elif layout_z == COLVECS:
        d, M = z.shape
        zbuf = np.asfortranarray(z)
    else:
        M, d = z.shape
        zbuf = np.asfortranarray(z)
    il = np.ascontiguousarray(np.broadcast_to(np.asarray(inv_lengthscale, dtype=np.float64), (d,)))
    mbuf = np.ascontiguousarray(np.asarray(m, dtype=dt))
    Lbuf = np.asfortranarray(np.asarray(Lq, dtype=dt))
    if mbuf.shape != (M,) or Lbuf.shape != (M, M):
        raise ValueError("m must have M entries and Lq must be M×M")
    desc = ModelDesc(code, kernel, parametrization, likelihood, quadrature_n, layout_z, neg_var_policy, d, M,
                     float(variance), il.ctypes.data_as(C.POINTER(C.c_double)), float(mean_const), float(jitter),
                     float(lik_sigma2), _ptr(zbuf), _ptr(mbuf), _ptr(Lbuf))
    return desc, (il, zbuf, mbuf, Lbuf)


class DeviceModel:
    """SparseVariationalApproximation(fz, q) + likelihood resident in HBM (svgp_model)."""

    def __init__(self, ctx: Context, desc: ModelDesc, keep):
        self.ctx = ctx
        self.M, self.d, self.dtype = desc.M, desc.d, desc.dtype
        h = C.c_void_p()
        ctx.check(ctx.lib.svgp_model_create(ctx.h, C.byref(desc), C.byref(h)))
        self.h = h
        del keep

    def update(self, desc: ModelDesc, keep):
        self.ctx.check(self.ctx.lib.svgp_model_update(self.ctx.h, self.h, C.byref(desc)))
        del keep

    def elbo(self, data: DeviceData, off=0, length=None, num_data=0.0):
        length = data.n - off if length is None else length
        out, terms = C.c_double(), Terms()
        rc = self.ctx.lib.svgp_elbo(self.ctx.h, self.h, data.h, off, length, float(num_data), C.byref(out), C.byref(terms))
        self.ctx.check(rc, terms)
        return out.value, terms

    def elbo_partial(self, data: DeviceData, off=0, length=None):
        length = data.n - off if length is None else length
        buf = (C.c_double * 4)()
        self.ctx.check(self.ctx.lib.svgp_elbo_partial(self.ctx.h, self.h, data.h, off, length, buf))
        return np.array(buf[:], dtype=np.float64)

    def marginals(self, data: DeviceData, off=0, length=None):
        """marginals(f_post(x)) of SVA:354 for the batch: (mu, v + 1e-18) as fp64 arrays (svgp_marginals)."""
        length = data.n - off if length is None else length
        mu, var = np.zeros(length), np.zeros(length)
        self.ctx.check(self.ctx.lib.svgp_marginals(self.ctx.h, self.h, data.h, off, length, _ptr(mu), _ptr(var)))
        return mu, var

    def elbo_grad(self, data: DeviceData, off=0, length=None, num_data=0.0, z_shape=None, shard=None, ext=None, out=None):
        """-> (elbo, terms, dict(variance, inv_lengthscale, z, m, Lq, lik_sigma2, mean_const)); z in the layout it was given.
        shard = (scale, kl_weight) evaluates the data-parallel shard form svgp_elbo_grad_shard instead.
        ext = (sum_e, g_mu, g_v): a likelihood the host evaluated on `marginals` (svgp_elbo_grad_ext).
        out = the gradient dict of an earlier call: its arrays are written in place instead of allocating M^2 fresh elements per step
        (a training loop that has consumed the previous gradient; at M = 2048 the first touch of 33 MB of new pages costs ~2 ms)."""
        length = data.n - off if length is None else length
        dt = np_dtype(self.dtype)
        zshape = z_shape if z_shape is not None else ((self.M,) if self.d == 1 else (self.d, self.M))
        if out is not None:
            il, zb, mb, Lb = out["inv_lengthscale"], out["z"], out["m"], out["Lq"]
            ok = (il.dtype == np.float64 and il.shape == (self.d,) and zb.dtype == dt and zb.shape == tuple(zshape) and zb.flags.f_contiguous
                  and mb.dtype == dt and mb.shape == (self.M,) and Lb.dtype == dt and Lb.shape == (self.M, self.M) and Lb.flags.f_contiguous)
            if not ok:
                raise ValueError("out= must be the gradient dict of an earlier call on a model of the same shape and dtype")
        else:
            il = np.zeros(self.d)
            zb = np.zeros(zshape, dtype=dt, order="F")
            mb = np.zeros(self.M, dtype=dt)
            Lb = np.zeros((self.M, self.M), dtype=dt, order="F")
        g = Grads(0.0, 0.0, 0.0, il.ctypes.data_as(C.POINTER(C.c_double)), _ptr(zb), _ptr(mb), _ptr(Lb))
        out, terms = C.c_double(), Terms()
        if ext is not None:
            gmu, gv = (np.ascontiguousarray(a, dtype=np.float64) for a in ext[1:])
            if gmu.shape != (length,) or gv.shape != (length,):
                raise ValueError("one point gradient per point of the batch")
            rc = self.ctx.lib.svgp_elbo_grad_ext(self.ctx.h, self.h, data.h, off, length, float(num_data), float(ext[0]),
                                                 _ptr(gmu), _ptr(gv), C.byref(out), C.byref(terms), C.byref(g))
        elif shard is None:
            rc = self.ctx.lib.svgp_elbo_grad(self.ctx.h, self.h, data.h, off, length, float(num_data), C.byref(out),
                                             C.byref(terms), C.byref(g))
        else:
            rc = self.ctx.lib.svgp_elbo_grad_shard(self.ctx.h, self.h, data.h, off, length, float(shard[0]), float(shard[1]),
                                                   C.byref(out), C.byref(terms), C.byref(g))
        self.ctx.check(rc, terms)
        return out.value, terms, dict(variance=g.variance, inv_lengthscale=il, z=zb, m=mb, Lq=Lb,
                                      lik_sigma2=g.lik_sigma2, mean_const=g.mean_const)

    def prior_kl(self):
        kl, ld = C.c_double(), C.c_double()
        self.ctx.check(self.ctx.lib.svgp_prior_kl(self.ctx.h, self.h, C.byref(kl), C.byref(ld)))
        return kl.value, ld.value

    def posterior(self):
        dt = np_dtype(self.dtype)
        Lk = np.zeros((self.M, self.M), dtype=dt, order="F")
        alpha = np.zeros(self.M, dtype=dt)
        B = np.zeros((self.M, self.M), dtype=dt, order="F")
        self.ctx.check(self.ctx.lib.svgp_posterior(self.ctx.h, self.h, _ptr(Lk), _ptr(alpha), _ptr(B)))
        return Lk, alpha, B

    def predict(self, x, want_mean=True, want_var=True, want_cov=False, layout=COLVECS):
        dt = np_dtype(self.dtype)
        x = np.asarray(x, dtype=dt)
        if x.ndim == 1:
            layout, n, xb = VEC, x.shape[0], np.ascontiguousarray(x)
        else:
            n = x.shape[1] if layout == COLVECS else x.shape[0]
            xb = np.asfortranarray(x)
        mean = np.zeros(n, dtype=dt) if want_mean else None
        var = np.zeros(n, dtype=dt) if want_var else None
        cov = np.zeros((n, n), dtype=dt, order="F") if want_cov else None
        self.ctx.check(self.ctx.lib.svgp_predict(self.ctx.h, self.h, layout, n, _ptr(xb), _ptr(mean), _ptr(var), _ptr(cov)))
        return mean, var, cov

    def cross_cov(self, x, y, layout=COLVECS):
        dt = np_dtype(self.dtype)
        x, y = np.asarray(x, dtype=dt), np.asarray(y, dtype=dt)
        if x.ndim == 1:
            layout, nx, ny = VEC, x.shape[0], y.shape[0]
            xb, yb = np.ascontiguousarray(x), np.ascontiguousarray(y)
        else:
            nx = x.shape[1] if layout == COLVECS else x.shape[0]
            ny = y.shape[1] if layout == COLVECS else y.shape[0]
            xb, yb = np.asfortranarray(x), np.asfortranarray(y)
        cov = np.zeros((nx, ny), dtype=dt, order="F")
        self.ctx.check(self.ctx.lib.svgp_predict_cross_cov(self.ctx.h, self.h, layout, nx, _ptr(xb), ny, _ptr(yb), _ptr(cov)))
        return cov

    def kuf(self, data: DeviceData, off=0, length=None, fetch=True):
        length = data.n - off if length is None else length
        out = np.zeros((self.M, length), dtype=np_dtype(self.dtype), order="F") if fetch else None
        self.ctx.check(self.ctx.lib.svgp_kuf(self.ctx.h, self.h, data.h, off, length, _ptr(out)))
        return out

    def free(self):
        if getattr(self, "h", None):
            self.ctx.lib.svgp_model_free(self.ctx.h, self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def gausshermite(n: int):
    lib = load_library()
    xs = np.zeros(n)
    ws = np.zeros(n)
    rc = lib.svgp_gausshermite(n, xs.ctypes.data_as(C.POINTER(C.c_double)), ws.ctypes.data_as(C.POINTER(C.c_double)))
    if rc != OK:
        raise ValueError(f"svgp_gausshermite({n}) failed with status {rc}")
    return xs, ws


class _Borrowed:
    """A member context of a Group seen through the Context interface (the group owns and destroys it)."""

    check = Context.check
    timing = Context.timing
    comm_info = Context.comm_info

    def __init__(self, lib, h, device):
        self.lib, self.h, self.device = lib, C.c_void_p(h), device


class Group:
    """One process driving several GPUs (svgp_group_*): data sharded over the members, the model replicated, and
    elbo / elbo_grad evaluated on all of them with one RCCL all-reduce issued by the library."""

    def __init__(self, device_ids):
        self.lib = load_library()
        ids = (C.c_int32 * len(device_ids))(*device_ids)
        h = C.c_void_p()
        rc = self.lib.svgp_group_create(len(device_ids), ids, C.byref(h))
        if rc != OK:
            raise SvgpError(f"svgp_group_create failed with status {rc}")
        self.h, self.n = h, len(device_ids)
        self.members = [_Borrowed(self.lib, self.lib.svgp_group_ctx(h, i), device_ids[i]) for i in range(self.n)]
        self._data, self._models = None, None

    def _check(self, rc, terms=None):
        if rc == OK:
            return
        msg = (self.lib.svgp_group_last_error(self.h) or b"").decode()
        if rc == INVALID_ARG:
            raise ValueError(msg)
        if rc == NOT_POSDEF:
            raise PosDefException(terms.chol_info if terms is not None else -1, msg)
        if rc == NEG_VARIANCE:
            raise DomainError(msg)
        raise SvgpError(f"status {rc}: {msg}")

    def upload(self, x, y, dtype, layout=COLVECS):
        dt = np_dtype(dtype_code(dtype))
        x = np.asarray(x, dtype=dt)
        if x.ndim == 1:
            layout, d, n, xb = VEC, 1, x.shape[0], np.ascontiguousarray(x)
        elif layout == COLVECS:
            (d, n), xb = x.shape, np.asfortranarray(x)
        else:
            (n, d), xb = x.shape, np.asfortranarray(x)
        yb = np.ascontiguousarray(np.asarray(y, dtype=dt))
        arr = (C.c_void_p * self.n)()
        self._check(self.lib.svgp_group_data_upload(self.h, dtype_code(dt), layout, d, n, _ptr(xb), _ptr(yb), arr))
        self._data = arr
        base, rem = divmod(n, self.n)
        self.shard_sizes = [base + (1 if i < rem else 0) for i in range(self.n)]
        return arr

    def create_model(self, desc, keep):
        arr = (C.c_void_p * self.n)()
        self._check(self.lib.svgp_group_model_create(self.h, C.byref(desc), arr))
        self._models, self._M, self._d, self._dtype = arr, desc.M, desc.d, desc.dtype
        del keep
        return arr

    def update_model(self, desc, keep):
        self._check(self.lib.svgp_group_model_update(self.h, self._models, C.byref(desc)))
        del keep

    def _ranges(self, offs, lens):
        offs = [0] * self.n if offs is None else list(offs)
        lens = [s - o for s, o in zip(self.shard_sizes, offs)] if lens is None else list(lens)
        return (C.c_int64 * self.n)(*offs), (C.c_int64 * self.n)(*lens)

    def elbo(self, offs=None, lens=None, num_data=0.0):
        o, l = self._ranges(offs, lens)
        out, terms = C.c_double(), Terms()
        self._check(self.lib.svgp_group_elbo(self.h, self._models, self._data, o, l, float(num_data), C.byref(out), C.byref(terms)), terms)
        return out.value, terms

    def elbo_grad(self, offs=None, lens=None, num_data=0.0):
        o, l = self._ranges(offs, lens)
        dt = np_dtype(self._dtype)
        il = np.zeros(self._d)
        zb = np.zeros((self._M,) if self._d == 1 else (self._d, self._M), dtype=dt, order="F")
        mb = np.zeros(self._M, dtype=dt)
        Lb = np.zeros((self._M, self._M), dtype=dt, order="F")
        g = Grads(0.0, 0.0, 0.0, il.ctypes.data_as(C.POINTER(C.c_double)), _ptr(zb), _ptr(mb), _ptr(Lb))
        out, terms = C.c_double(), Terms()
        self._check(self.lib.svgp_group_elbo_grad(self.h, self._models, self._data, o, l, float(num_data), C.byref(out),
                                                  C.byref(terms), C.byref(g)), terms)
        return out.value, terms, dict(variance=g.variance, lik_sigma2=g.lik_sigma2, mean_const=g.mean_const,
                                      inv_lengthscale=il, z=zb, m=mb, Lq=Lb)

    def close(self):
        if getattr(self, "h", None):
            for i in range(self.n):
                if self._models is not None:
                    self.lib.svgp_model_free(self.members[i].h, self._models[i])
                if self._data is not None:
                    self.lib.svgp_data_free(self.members[i].h, self._data[i])
            self._models = self._data = None
            self.lib.svgp_group_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
