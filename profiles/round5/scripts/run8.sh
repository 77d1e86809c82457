#!/usr/bin/env bash
# round 5: the 512-thread block factorisation (seven worker waves): stamps, prep times (product vs SVGP_POTF2_WAVES=4 in the experiments
# library), accuracy, race check, the Cholesky-heavy tests.  Every step under a timeout: a barrier miscount would hang the kernel.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5
L=$PWD/approximategps.jl_amd/csrc/ablate
O=gpurun_out/r5/potf2_w8.log; : > $O
f() { grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"; }
for dt in f64 f32; do POTF2_DTYPES=$dt SVGP_MI355X_LIB=$L/libsvgp_p_w8.so timeout 120 python tools/potf2_time.py 2>&1 | f >> $O || echo "TIMEOUT/FAIL potf2_time $dt" >> $O; done
timeout 300 python tools/prep_time.py 2>&1 | f >> $O || echo "TIMEOUT/FAIL prep_time" >> $O
for w in 4 8 4 8; do echo "-- experiments library, SVGP_POTF2_WAVES=$w" >> $O; SVGP_POTF2_WAVES=$w SVGP_MI355X_LIB=$L/libsvgp_experiments.so timeout 300 python tools/prep_time.py 2>&1 | f >> $O || echo "TIMEOUT/FAIL" >> $O; done
timeout 300 python tests/chol_accuracy.py 2>&1 | f >> $O || echo "TIMEOUT/FAIL accuracy" >> $O
timeout 600 python tools/chol_check.py 2>&1 | f | tail -n 11 >> $O || echo "TIMEOUT/FAIL chol_check" >> $O
cat $O | grep -v "^  block [1-6]" | cut -c1-260
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_round4.py tests/test_gpu_parity.py -q -m gpu -x 2>&1 | f | tail -n 3
