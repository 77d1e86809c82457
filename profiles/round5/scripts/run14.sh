#!/usr/bin/env bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_dbg; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -X faulthandler -m pytest tests/test_gpu_parity.py -k high_dimensional -m gpu -q -v > $O/single.log 2>&1; echo "rc=$?" >> $O/single.log
tail -n 15 $O/single.log | cut -c1-300
echo ---- gdb
timeout 900 /opt/rocm/bin/rocgdb -batch -ex "set pagination off" -ex "handle SIGSEGV nostop noprint pass" -ex run -ex bt -ex "info threads" --args python -m pytest tests/test_gpu_parity.py -k high_dimensional -m gpu -q -x > $O/gdb.log 2>&1; echo "rc=$?" >> $O/gdb.log
grep -v "^\[New Thread\|^\[Thread\|^warning" $O/gdb.log | tail -n 60 | cut -c1-300
echo ---- kuf time
timeout 600 python tools/kuf_time.py Hd17 Hd32 H > $O/kuf_time.log 2>&1; tail -n 12 $O/kuf_time.log | cut -c1-300
