#!/usr/bin/env bash
# round 5: potf2 A/B second pass - split register factor (L | X in the two wave halves), pipelined queue with the negation at the MFMA
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5
L=$PWD/approximategps.jl_amd/csrc/ablate
: > gpurun_out/r5/potf2_ab3.log
for rep in 1 2; do
for v in p_tail7 p_r5d; do
  for dt in f64 f32; do
    POTF2_DTYPES=$dt SVGP_MI355X_LIB=$L/libsvgp_$v.so python tools/potf2_time.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" >> gpurun_out/r5/potf2_ab3.log
  done
done
done
for v in p_tail7 p_r5d; do
  SVGP_MI355X_LIB=$L/libsvgp_$v.so python tools/prep_time.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" >> gpurun_out/r5/potf2_ab3.log
done
SVGP_MI355X_LIB=$L/libsvgp_p_r5d.so python tests/chol_accuracy.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" >> gpurun_out/r5/potf2_ab3.log
SVGP_MI355X_LIB=$L/libsvgp_p_r5d.so python tools/chol_check.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" >> gpurun_out/r5/potf2_ab3.log
grep -v "^  block [1-6]" gpurun_out/r5/potf2_ab3.log | cut -c1-260
