#!/usr/bin/env bash
# round 5: potf2 A/B (round-4 tail + queue | 7-stage tail | pipelined queue | both): stamps per dtype, prep times, accuracy
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5
L=$PWD/approximategps.jl_amd/csrc/ablate
: > gpurun_out/r5/potf2_ab.log
for rep in 1 2; do
for v in p_base p_tail7 p_queue p_both; do
  for dt in f64 f32; do
    POTF2_DTYPES=$dt SVGP_MI355X_LIB=$L/libsvgp_$v.so python tools/potf2_time.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" >> gpurun_out/r5/potf2_ab.log
  done
done
done
for v in p_base p_both; do
  SVGP_MI355X_LIB=$L/libsvgp_$v.so python tools/prep_time.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" >> gpurun_out/r5/potf2_ab.log
  SVGP_MI355X_LIB=$L/libsvgp_$v.so python tests/chol_accuracy.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" >> gpurun_out/r5/potf2_ab.log
  SVGP_MI355X_LIB=$L/libsvgp_$v.so python tools/chol_check.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -n 3 >> gpurun_out/r5/potf2_ab.log
done
grep -v "^  block" gpurun_out/r5/potf2_ab.log
