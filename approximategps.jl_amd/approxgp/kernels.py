"""Kernel objects mirroring the KernelFunctions.jl constructs the reference's tests and examples use
(test/test_utils.jl:2, examples/a-regression/script.jl:55-59).  They only carry parameters: all
kernel-matrix arithmetic happens in the HIP library."""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from . import _ffi


@dataclass(frozen=True)
class _Base:
    family: int

    def __rmul__(self, variance):
        return ScaledKernel(TransformedKernel(self, None), float(variance))

    def compose(self, transform):
        return TransformedKernel(self, transform)

    __matmul__ = compose  # `SqExponentialKernel() @ ScaleTransform(s)` stands in for Julia's ∘


def SqExponentialKernel():
    return _Base(_ffi.KERNEL_SE)


SEKernel = SqExponentialKernel


def Matern32Kernel():
    return _Base(_ffi.KERNEL_MATERN32)


def Matern52Kernel():
    return _Base(_ffi.KERNEL_MATERN52)


@dataclass(frozen=True)
class ScaleTransform:
    s: float  # inverse lengthscale


@dataclass(frozen=True)
class ARDTransform:
    v: tuple  # inverse lengthscales

    def __init__(self, v):
        object.__setattr__(self, "v", tuple(float(t) for t in np.atleast_1d(v)))


@dataclass(frozen=True)
class TransformedKernel:
    base: _Base
    transform: object  # ScaleTransform | ARDTransform | None

    def __rmul__(self, variance):
        return ScaledKernel(self, float(variance))


@dataclass(frozen=True)
class ScaledKernel:
    kernel: TransformedKernel
    variance: float


def with_lengthscale(base: _Base, lengthscale):
    """KernelFunctions.with_lengthscale: ℓ::Real -> ScaleTransform(1/ℓ), ℓ::Vector -> ARDTransform(1 ./ ℓ)."""
    ls = np.atleast_1d(np.asarray(lengthscale, dtype=np.float64))
    if ls.size == 1:
        return TransformedKernel(base, ScaleTransform(1.0 / float(ls[0])))
    return TransformedKernel(base, ARDTransform(1.0 / ls))


def unpack_kernel(kernel, d: int):
    """-> (family, variance, inv_lengthscale[d]); what the Julia shim reads from
    ScaledKernel.σ², TransformedKernel.transform.{s,v} and the base kernel type."""
    variance = 1.0
    if isinstance(kernel, ScaledKernel):
        variance, kernel = kernel.variance, kernel.kernel
    if isinstance(kernel, _Base):
        kernel = TransformedKernel(kernel, None)
    if not isinstance(kernel, TransformedKernel):
        raise _ffi.UnsupportedError(f"unsupported kernel {kernel!r}")
    t = kernel.transform
    if t is None:
        il = np.ones(d)
    elif isinstance(t, ScaleTransform):
        il = np.full(d, float(t.s))
    elif isinstance(t, ARDTransform):
        il = np.asarray(t.v, dtype=np.float64)
        if il.shape[0] != d:
            raise ValueError(f"ARDTransform has {il.shape[0]} scales but inputs have dimension {d}")
    else:
        raise _ffi.UnsupportedError(f"unsupported input transform {t!r}")
    return kernel.base.family, float(variance), il
