#!/usr/bin/env bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_final; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python tools/kuf_time.py Hd17 Hd32 H > $O/kuf_dreg20.log 2>&1; tail -n 8 $O/kuf_dreg20.log | cut -c1-300
timeout 1700 python -X faulthandler -m pytest tests -m gpu -q > $O/gputest_product.log 2>&1; echo "rc=$?" >> $O/gputest_product.log
tail -n 6 $O/gputest_product.log | cut -c1-300
