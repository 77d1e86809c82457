"""CPU-side checks added in round 4 (no GPU): the ADVICE r3 items that can be seen without a device, and the contract between
oracle/reference_julia.jl (the standing recipe that pins the oracle to the real reference) and the committed golden fixtures."""
import ctypes as C
import glob
import os
import re

import numpy as np
import pytest

import approxgp
from approxgp import _ffi
from approxgp import sva as sva_mod

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_golden_fixture_is_one_the_julia_recipe_can_check():
    """oracle/reference_julia.jl walks tests/golden/*.npz (readdir) and reads a fixed set of keys from each; a
    fixture added without those keys, or with a kernel / likelihood code the recipe's dispatch does not know, would make the ONE
    run that pins the oracle fail or - worse - skip it.  This test fails in that case."""
    src = open(os.path.join(ROOT, "oracle", "reference_julia.jl")).read()
    assert 'readdir(dir)' in src and 'endswith(f, ".npz")' in src, "the recipe no longer walks the golden directory"
    keys = set(re.findall(r'g\["([a-z_0-9A-Z]+)"\]', src))
    assert {"elbo", "kl", "mu", "v", "Lk", "alpha", "B", "g_z", "g_Lq", "cov9", "cov_cross"} <= keys
    fam_codes = {0, 1, 2}                      # base_kernel(fam)
    lik_codes = {0, 1, 2, 3, 4, 5}             # make_lik(lk, p): 4 is the final `GammaLikelihood` branch
    fixtures = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "*.npz")))
    assert len(fixtures) >= 11
    for path in fixtures:
        g = np.load(path)
        missing = keys - set(g.files)
        assert not missing, f"{os.path.basename(path)} lacks {sorted(missing)}: oracle/reference_julia.jl cannot check it"
        assert int(g["family"]) in fam_codes and int(g["lik"]) in lik_codes, path
        assert g["x"].shape[1] >= 16, "the recipe takes cov over points 1:9 and 10:16"
        assert int(g["centered"]) in (0, 1) and int(g["f32"]) in (0, 1)
    # ... and the generator knows every fixture by name (a hand-dropped .npz has no provenance)
    gen = open(os.path.join(ROOT, "tests", "golden", "make_golden.py")).read()
    for path in fixtures:
        assert os.path.basename(path)[:-4] in gen, f"{os.path.basename(path)} is not produced by tests/golden/make_golden.py"


def test_small_problem_rule_is_skipped_under_a_communicator():
    """with a library communicator the calls are collective; a rank whose own shard falls below the offload
    threshold must not leave for the host path while its peers wait in ncclAllReduce."""
    class FakeCtx:                      # what _ffi.Context.comm_info() reports on a 2-rank communicator
        def __init__(self, world):
            self.world = world

        def comm_info(self):
            return self.world, 0

    f = approxgp.GP(approxgp.SqExponentialKernel())
    z, x, y = np.linspace(0, 1, 20), np.linspace(0, 1, 100), np.zeros(100)
    sva = approxgp.SparseVariationalApproximation(f(z, 1e-6), approxgp.MvNormal(np.zeros(20), np.eye(20)))
    lfx = approxgp.LatentFiniteGP(f(x, 0.1), approxgp.GaussianLikelihood(0.1)) if hasattr(approxgp, "LatentFiniteGP") else None
    with pytest.raises(approxgp.DeclinedError):
        sva_mod._decline_if_small(sva, lfx, y, "decline", False, FakeCtx(1))
    sva_mod._decline_if_small(sva, lfx, y, "decline", False, FakeCtx(2))          # world 2: no decline, whatever the shard
    sva_mod._decline_if_small(sva, lfx, y[:3], "decline", True, FakeCtx(8))
    # the Julia hook: the rule sits behind the communicator test in elbo_and_grads (the collective entry points)
    src = open(os.path.join(ROOT, "integration", "julia", "src", "SVGPMI355X.jl")).read()
    body = src[src.index("function elbo_and_grads"):src.index("function try_elbo_pullback")]
    assert re.search(r"\(comm_world\(\) > 1 \|\| worth_offloading\(length\(y\)", body), "small-problem rule not guarded by comm_world()"
    assert not re.search(r"^\s*worth_offloading\(length\(y\)", body, re.M)


def test_offload_threshold_env_value_is_validated(monkeypatch):
    """ADVICE r3 (low, 5): an SVGP_OFFLOAD_MIN_WORK that does not parse used to read as 0 = always offload."""
    lib = _ffi.load_library()
    small = (100, 20, 1, 0, 0)
    monkeypatch.delenv("SVGP_OFFLOAD_MIN_WORK", raising=False)
    assert lib.svgp_offload_advice(*small) == 0
    for bad in ("abc", "", "1e6x", "-5", "nan"):
        monkeypatch.setenv("SVGP_OFFLOAD_MIN_WORK", bad)
        assert lib.svgp_offload_advice(*small) == 0, bad       # ignored: the default threshold applies
    monkeypatch.setenv("SVGP_OFFLOAD_MIN_WORK", "0")
    assert lib.svgp_offload_advice(*small) == 1
    monkeypatch.setenv("SVGP_OFFLOAD_MIN_WORK", " 1e3 ")
    assert lib.svgp_offload_advice(*small) == 1
    monkeypatch.setenv("SVGP_OFFLOAD_MIN_WORK", "1e12")
    assert lib.svgp_offload_advice(100_000, 512, 8, 0, 0) == 0


def test_timing_struct_growth_is_safe_for_old_hosts():
    """svgp_timing grew past the 48 bytes a v3 host allocates.  svgp_last_timing writes the v3 prefix only;
    the appended fields travel through svgp_last_timing_sized, which never writes more than the caller says it has."""
    header = open(os.path.join(ROOT, "include", "svgp_mi355x.h")).read()
    assert "#define SVGP_TIMING_V3_BYTES 48" in header and "svgp_last_timing_sized" in header
    assert _ffi.Timing.ms_chol.offset == 48 and C.sizeof(_ffi.Timing) == 64
    lib = _ffi.load_library()
    buf = (C.c_char * 64)()
    assert lib.svgp_last_timing_sized(None, buf, 64) == _ffi.INVALID_ARG       # no context: refused, nothing written
    assert lib.svgp_last_timing(None, C.cast(buf, C.POINTER(_ffi.Timing))) == _ffi.INVALID_ARG
    api = open(os.path.join(ROOT, "approximategps.jl_amd", "csrc", "api.hip")).read()
    body = api[api.index("int32_t svgp_last_timing(const svgp_ctx* ctx, svgp_timing* out)"):api.index("int32_t svgp_last_timing_sized")]
    assert "SVGP_TIMING_V3_BYTES" in body and "*out = ctx->timing" not in body


def test_every_environment_knob_is_documented():
    """Every SVGP_* environment variable the library (or its Python mirror) reads has a row in INTEGRATION.md's table."""
    import glob
    import re
    names = set()
    for f in glob.glob(os.path.join(ROOT, "approximategps.jl_amd", "csrc", "*.h*")) + glob.glob(os.path.join(ROOT, "approximategps.jl_amd", "approxgp", "*.py")):
        names |= set(re.findall(r'(?:getenv|env_int|environ\.get|environ\[)\(?\s*"(SVGP_[A-Z0-9_]+)"', open(f).read()))
    assert len(names) > 30, names
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    missing = sorted(n for n in names if f"`{n}`" not in doc)
    assert not missing, missing
