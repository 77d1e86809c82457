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


PRODUCT_ENV = {"SVGP_TIMING", "SVGP_OVERLAP", "SVGP_SEG_SPLIT", "SVGP_DEBUG_SYNC", "SVGP_OFFLOAD_MIN_WORK", "SVGP_RCCL_LIB", "SVGP_DISABLE_RCCL"}


def test_every_environment_knob_is_documented():
    """VERDICT r4 item 5 / ADVICE r4: the PRODUCT library reads exactly the operational variables (csrc/knobs.hpp) - plain getenv /
    env_flag calls - and every tuning knob or A/B switch goes through exp_int / exp_ll / exp_double (strip.hip: env_int), which are
    compile-time constants unless the sources are built with -DSVGP_EXPERIMENTS.  Checked on the sources (both lists complete in
    INTEGRATION.md, nothing read per call inside an evaluation) and on the built product .so (no experiment knob's name in it)."""
    import glob
    import re
    srcs = glob.glob(os.path.join(ROOT, "approximategps.jl_amd", "csrc", "*.h*"))
    product, experiments = set(), set()
    for f in srcs:
        text = open(f).read()
        product |= set(re.findall(r'(?:getenv|env_flag)\(\s*"(SVGP_[A-Z0-9_]+)"', text))
        experiments |= set(re.findall(r'(?:exp_int|exp_ll|exp_double|env_int)\(\s*"(SVGP_[A-Z0-9_]+)"', text))
    assert product == PRODUCT_ENV, product ^ PRODUCT_ENV
    assert len(experiments) > 30 and not (experiments & product), experiments & product
    py = set()
    for f in glob.glob(os.path.join(ROOT, "approximategps.jl_amd", "approxgp", "*.py")):
        py |= set(re.findall(r'(?:environ\.get|environ\[)\(?\s*"(SVGP_[A-Z0-9_]+)"', open(f).read()))
    assert py <= {"SVGP_MI355X_LIB"} | PRODUCT_ENV, py
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    head, tail = doc.split("**Experiments library**")
    missing = sorted(n for n in product | py if f"`{n}`" not in head) + sorted(n for n in experiments if f"`{n}`" not in tail)
    assert not missing, missing
    # no evaluation path reads the environment: outside knobs.hpp the only getenv calls are the process-wide statics / svgp_offload_advice
    api = open(os.path.join(ROOT, "approximategps.jl_amd", "csrc", "api.hip")).read()
    assert re.findall(r'getenv\("(SVGP_[A-Z0-9_]+)"', api) == ["SVGP_OFFLOAD_MIN_WORK"]
    from approxgp import _ffi
    if os.path.exists(_ffi.LIB_PATH) and "experiments" not in os.path.basename(_ffi.LIB_PATH):
        blob = open(_ffi.LIB_PATH, "rb").read()
        assert b"svgp_debug_experiments" not in blob
        leaked = sorted(n for n in experiments if n.encode() in blob)
        assert not leaked, leaked
        for n in PRODUCT_ENV:
            assert n.encode() in blob, n


def _code_object_notes(lib_path):
    """llvm-readelf --notes of every gfx950 code object bundled in a library's .hip_fatbin (one bundle per translation unit)."""
    import subprocess
    import tempfile
    readelf = "/opt/rocm/lib/llvm/bin/llvm-readelf"
    bundler = "/opt/rocm/lib/llvm/bin/clang-offload-bundler"
    notes = ""
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", lib_path, os.path.join(td, "x.so")], check=True)
        blob = open(fat, "rb").read()
        magic = b"__CLANG_OFFLOAD_BUNDLE__"
        starts, off = [], 0
        while True:
            i = blob.find(magic, off)
            if i < 0:
                break
            starts.append(i)
            off = i + 1
        for k, (a, b) in enumerate(zip(starts, starts[1:] + [len(blob)])):
            part = os.path.join(td, f"b{k}.bin")
            open(part, "wb").write(blob[a:b])
            co = os.path.join(td, f"co{k}.o")
            r = subprocess.run([bundler, "--unbundle", "--type=o", f"--input={part}", f"--output={co}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], capture_output=True, text=True)
            if r.returncode == 0 and os.path.exists(co) and os.path.getsize(co) > 0:
                notes += subprocess.run([readelf, "--notes", co], capture_output=True, text=True).stdout
    return notes


def test_every_kernel_the_host_side_registers_has_device_code():
    """Round 5: a product library whose strip.hip was edited while build.sh was compiling it (hipcc's device pass read the old
    text, its host pass the new one) carried host stubs for two Kuf kernels without gfx950 code, and the launch aborted on the GPU
    box with 'Cannot find Symbol'.  For the built product library and, when present, the experiments library: every kernel name the
    host side registers (the mangled names in the library outside .hip_fatbin) is a kernel of a bundled gfx950 code object."""
    import re
    import subprocess
    from approxgp import _ffi
    tools = ["/opt/rocm/lib/llvm/bin/llvm-readelf", "/opt/rocm/lib/llvm/bin/clang-offload-bundler", "/opt/rocm/lib/llvm/bin/llvm-objcopy"]
    libs = [p for p in (_ffi.LIB_PATH, os.path.join(os.path.dirname(_ffi.LIB_PATH), "ablate", "libsvgp_experiments.so")) if os.path.exists(p)]
    if not libs or not all(os.path.exists(t) for t in tools):
        import pytest
        pytest.skip("needs a built library and the ROCm llvm tools")
    import tempfile
    for lib in libs:
        device = set(re.findall(r"\.name:\s+(_Z\S+)", _code_object_notes(lib)))
        assert len(device) > 100, (lib, len(device))
        with tempfile.TemporaryDirectory() as td:   # the host side: the library with the fat binary removed
            host = os.path.join(td, "host.so")
            subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objcopy", "--remove-section", ".hip_fatbin", lib, host], check=True)
            blob = open(host, "rb").read()
        registered = {m.decode() for m in re.findall(rb"_ZN4svgp[A-Za-z0-9_]*kernel[A-Za-z0-9_]*", blob)}
        registered = {n for n in registered if not n.startswith("_ZN4svgp12_GLOBAL__N_1L") and "__device_stub__" not in n}
        kernels = {n for n in registered if re.search(r"kernelI.*E[Ev]v|kernelE", n)}
        assert len(kernels) > 100, (lib, len(kernels))
        missing = sorted(kernels - device)
        assert not missing, (lib, missing[:5], len(missing))


def test_no_product_kernel_runs_at_one_wave_per_simd_unnoticed():
    """Round 5 found the kernel-gradient reductions of wide inputs at 345-350 VGPRs - ONE wave per SIMD - and their 16-feature f64 form
    spilling 111-143 VGPRs: an H-sized value-and-gradient evaluation at d = 64 took 145 ms (116 now), at d = 16 90.8 (79.8).  Nothing had
    looked at the register counts of kernels outside strip.hip.  For the built product library: a kernel that needs more than 256 VGPRs
    + AGPRs (one wave per SIMD) or spills must be on the lists below, each entry with the reason it is acceptable."""
    import re
    from approxgp import _ffi
    tools = ["/opt/rocm/lib/llvm/bin/llvm-readelf", "/opt/rocm/lib/llvm/bin/clang-offload-bundler", "/opt/rocm/lib/llvm/bin/llvm-objcopy"]
    if not os.path.exists(_ffi.LIB_PATH) or "experiments" in os.path.basename(_ffi.LIB_PATH) or not all(os.path.exists(t) for t in tools):
        import pytest
        pytest.skip("needs the built product library and the ROCm llvm tools")
    one_wave_ok = (
        "expect_kernelI",        # per-point likelihood expectations (lgamma / Gauss-Hermite loops): 0.04-0.2 ms per evaluation
        "point_grad_kernelI",    # their adjoint: 7-31 us per 65 536-point chunk
        "chol_tile_kernelIdLi1ELi32ELb1ELb0E",   # the 256-thread fused update + block factorisation: <= 28 workgroups, latency-bound chain
    )
    spills_ok = (
        "expect_kernelI", "kgrad_kernelIdLi16E", "kgrad_kernelIfLi16E",   # kgrad 16-feature forms: measured faster WITH the two-wave bound (kgrad_minw_ab.log)
        "kuf_cols_kernelIdLi64E",                                           # 2-4 VGPRs in the prologue
        "strip_kernelI",                                                      # checked kernel by kernel in the next test
    )
    offenders = []
    for m in re.finditer(r"\.name:\s+(_Z\S+)(.*?)\.wavefront_size", _code_object_notes(_ffi.LIB_PATH), flags=re.S):
        name, body = m.group(1), m.group(2)
        g = lambda k: int((re.search(r"\." + k + r":\s+(\d+)", body) or [None, "0"])[1])
        regs, spill = g("vgpr_count") + g("agpr_count"), g("vgpr_spill_count")
        if regs > 256 and not any(t in name for t in one_wave_ok):
            offenders.append((name[:90], "registers", regs))
        if spill > 0 and not any(t in name for t in spills_ok):
            offenders.append((name[:90], "spilled", spill))
    assert not offenders, offenders


def test_product_build_has_at_most_thirty_strip_kernels_and_none_that_spills():
    """the product build instantiates strip_kernel 30 times (5 shapes x {forward, value-and-gradient} x {d <= 16,
    wide inputs} + 5 x 2 segmented), none at occupancy 1 and none with spilled registers; the 512-thread strips exist in the experiments
    build only (the in-kernel likelihood-gradient forms, five of which spilled 47-79 VGPRs at occupancy 1, left the tree in round 6).  Read from
    the kernel descriptors of the built library's code object (llvm-readelf notes), so this is the .so a GPU box loads."""
    import shutil
    import subprocess
    from approxgp import _ffi
    readelf = "/opt/rocm/lib/llvm/bin/llvm-readelf"
    bundler = "/opt/rocm/lib/llvm/bin/clang-offload-bundler"
    if not (os.path.exists(_ffi.LIB_PATH) and os.path.exists(readelf) and os.path.exists(bundler)) or "experiments" in os.path.basename(_ffi.LIB_PATH):
        import pytest
        pytest.skip("needs the built product library and the ROCm llvm tools")
    notes = _code_object_notes(_ffi.LIB_PATH)
    import re
    kernels = re.findall(r"\.name:\s+(\S*strip_kernel\S*)(.*?)\.wavefront_size", notes, flags=re.S)
    assert kernels, "no strip_kernel descriptors found"
    names = {n for n, _ in kernels}
    assert len(names) <= 30, len(names)
    # Three instantiations keep a few loop-invariant values in scratch (2-14 VGPRs: stored once in the prologue, reloaded once per
    # kernel-family branch of the Kuf pre-generation, never inside a k-loop - checked in the gfx950 assembly): the wide-input (d > 16)
    # value-and-gradient strips <double, 64> and <float, 128>, and the segmented fp32 128-point value-and-gradient strips.
    small_spillers = ("IdLi64ELi16ELi256ELi2ELi16ELb1ELb1ELb0E", "IfLi128ELi16ELi256ELi2ELi16ELb1ELb1ELb0E", "IfLi128ELi16ELi256ELi2ELi16ELb1ELb0ELb1E")   # <T, NT, BK, NTHR, MINW, PAD, GRAD, BIGD, SEG>
    for n, body in kernels:
        spill = int(re.search(r"\.vgpr_spill_count:\s+(\d+)", body).group(1))
        vg = int(re.search(r"\.vgpr_count:\s+(\d+)", body).group(1))
        assert vg <= 256, (n, vg)          # <= 256 registers: two waves per SIMD (occupancy 2), never the 512-register single-wave shape
        if any(t in n for t in small_spillers):
            assert spill <= 16, (n, spill)
        else:
            assert spill == 0, (n, spill)
