#!/bin/bash
# round-4 closing run: CPU-free GPU suite, the overlap soak, the headline profile of the final sources, the driver's bench command (timed)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_final; mkdir -p $O
(timeout 1700 python -m pytest tests -m gpu -q 2>&1 | grep -E "^FAILED|passed|failed|error" | tail -5) > $O/pytest_final.log 2>&1; cat $O/pytest_final.log
timeout 900 python tests/soak_overlap.py 300 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|^RCCL\|^HIP" | tee $O/soak_overlap.log
timeout 600 python tests/soak.py 2>&1 | tail -3 | tee $O/soak.log
bash tools/run_profile.sh r4f_H H > $O/prof_H.log 2>&1; tail -1 $O/prof_H.log | cut -c1-200
bash tools/run_profile.sh r4f_Hgrad H grad > $O/prof_Hgrad.log 2>&1
/usr/bin/time -v python bench.py > $O/bench_H_final.json 2> $O/bench_H_final.err; grep "Elapsed (wall" $O/bench_H_final.err; cut -c1-250 $O/bench_H_final.json
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke | tee $O/smoke.log
