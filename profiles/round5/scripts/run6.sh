#!/usr/bin/env bash
# round 5: wide-input Kuf (128-point blocks at d > 32), the final potf2; GPU suite; Kuf timings; prep times
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5
python -m pytest tests -q -m gpu -x > gpurun_out/r5/gputest_b.log 2>&1; grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" gpurun_out/r5/gputest_b.log | tail -n 4
python tools/kuf_time.py Hd64 H32d64 Hd32 Hd17 H32d32 H 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tee gpurun_out/r5/kuf_wide.log
python tools/prep_time.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tee gpurun_out/r5/prep_time_final.log
