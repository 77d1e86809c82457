#!/usr/bin/env bash
# round 5: first factor beside the block load (512-thread form): stamps (standalone and fused launch), serial prep, accuracy, races, tests
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5
L=$PWD/approximategps.jl_amd/csrc/ablate
O=gpurun_out/r5/potf2_earlystore.log; : > $O
f() { grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"; }
for dt in f64 f32; do POTF2_DTYPES=$dt SVGP_MI355X_LIB=$L/libsvgp_p_w8.so timeout 120 python tools/potf2_time.py 2>&1 | f >> $O || echo "TIMEOUT/FAIL potf2_time $dt" >> $O; done
for dt in f64 f32; do echo "-- fused launch (M = 256) $dt" >> $O; SVGP_OVERLAP=0 POTF2_M=256 POTF2_DTYPES=$dt SVGP_MI355X_LIB=$L/libsvgp_p_w4f.so timeout 120 python tools/potf2_time.py 2>&1 | f >> $O || echo "TIMEOUT/FAIL" >> $O; done
echo "-- serial prep, product library" >> $O; SVGP_OVERLAP=0 timeout 300 python tools/prep_time.py 2>&1 | f >> $O
SVGP_OVERLAP=0 timeout 300 python tools/prep_time.py 2>&1 | f >> $O
timeout 300 python tests/chol_accuracy.py 2>&1 | f >> $O || echo "TIMEOUT/FAIL accuracy" >> $O
timeout 600 python tools/chol_check.py 2>&1 | f | tail -n 11 >> $O || echo "TIMEOUT/FAIL chol_check" >> $O
cat $O | grep -v "^  block [1-6]" | cut -c1-260
timeout 1200 python -m pytest tests -q -m gpu -x 2>&1 | f | tail -n 3
