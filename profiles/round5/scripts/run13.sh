#!/usr/bin/env bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_final; mkdir -p $O
timeout 1700 python -X faulthandler -m pytest tests -m gpu -q -v > $O/gputest_product_full.log 2>&1; echo "rc=$?" >> $O/gputest_product_full.log
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" $O/gputest_product_full.log | grep -v "PASSED" | tail -n 60 | cut -c1-300
