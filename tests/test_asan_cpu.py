"""Host-side AddressSanitizer run (CPU only).  build_asan.sh instruments every host function of the library;
this test drives the entry points that need no GPU - version, the Golub-Welsch Gauss-Hermite rule, context / group
creation failing cleanly, argument errors of the handle-free calls, the RCCL loader - in a child process with the ASan
runtime preloaded, and fails on any ASan report.  The ASan build takes about a minute, so the test builds it only when it
is missing (``build_asan.sh``) and is skipped when hipcc is not on the PATH."""
import glob
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "approximategps.jl_amd", "csrc", "asan", "libsvgp_mi355x_asan.so")

CHILD = r"""
import ctypes as C, sys
sys.path.insert(0, %r)
from approxgp import _ffi
lib = _ffi.load_library()
assert lib.svgp_version() == 5
for n in (1, 2, 7, 20, 64, 200):
    xs, ws = (C.c_double * n)(), (C.c_double * n)()
    assert lib.svgp_gausshermite(n, xs, ws) == 0
    assert abs(sum(ws) - 1.7724538509055159) < 1e-12
assert lib.svgp_gausshermite(0, (C.c_double * 1)(), (C.c_double * 1)()) == _ffi.INVALID_ARG
assert lib.svgp_gausshermite(5, None, None) == _ffi.INVALID_ARG
if lib.svgp_device_count() == 0:
    h = C.c_void_p()
    assert lib.svgp_ctx_create(0, None, C.byref(h)) == _ffi.HIP_ERROR and not h.value
    g = C.c_void_p()
    ids = (C.c_int32 * 2)(0, 0)
    assert lib.svgp_group_create(2, ids, C.byref(g)) == _ffi.INVALID_ARG          # the same device twice
    assert lib.svgp_group_create(0, ids, C.byref(g)) == _ffi.INVALID_ARG
    assert lib.svgp_group_size(None) == 0 and lib.svgp_group_destroy(None) == 0
    assert lib.svgp_ctx_destroy(None) == 0 and lib.svgp_data_free(None, None) == 0 and lib.svgp_model_free(None, None) == 0
    assert lib.svgp_last_error(None) == b"null context"
    assert lib.svgp_elbo(None, None, None, 0, 1, 0.0, None, None) == _ffi.INVALID_ARG
    assert lib.svgp_ctx_attach_comm(None, None, 1, 0) == _ffi.INVALID_ARG
    buf = C.create_string_buffer(128)
    rc = lib.svgp_comm_unique_id(C.cast(buf, C.c_void_p))                          # dlopen of librccl + ncclGetUniqueId
    assert rc in (_ffi.OK, _ffi.RCCL_ERROR)
print("ASAN_CHILD_OK")
"""


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="needs hipcc to build the instrumented library")
def test_host_side_under_address_sanitizer():
    srcs = glob.glob(os.path.join(ROOT, "approximategps.jl_amd", "csrc", "*.h*")) + [os.path.join(ROOT, "include", "svgp_mi355x.h")]
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(f) for f in srcs):   # never test a stale build
        subprocess.run(["bash", os.path.join(ROOT, "build_asan.sh")], check=True, cwd=ROOT, stdout=subprocess.DEVNULL)
    rt = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    if not rt:
        pytest.skip("clang ASan runtime not found")
    env = dict(os.environ, SVGP_MI355X_LIB=LIB, LD_PRELOAD=rt[-1],
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=66:protect_shadow_gap=0")
    r = subprocess.run([sys.executable, "-c", CHILD % os.path.join(ROOT, "approximategps.jl_amd")], env=env,
                       capture_output=True, text=True, timeout=600)
    assert "ERROR: AddressSanitizer" not in r.stderr, r.stderr[-3000:]
    assert r.returncode == 0 and "ASAN_CHILD_OK" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-3000:])
