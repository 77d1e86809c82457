#!/usr/bin/env bash
# closing run of round 5 on the final sources: GPU suite on both libraries, the default bench line, the build log
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_final; mkdir -p $O
f() { grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"; }
(timeout 1700 python -m pytest tests -m gpu -q 2>&1 | f | tail -n 3) > $O/gputest_product.log 2>&1; cat $O/gputest_product.log
(SVGP_MI355X_LIB=$PWD/approximategps.jl_amd/csrc/ablate/libsvgp_experiments.so timeout 1700 python -m pytest tests -m gpu -q 2>&1 | f | tail -n 3) > $O/gputest_experiments.log 2>&1; cat $O/gputest_experiments.log
timeout 900 python bench.py > $O/bench_H_final.json 2> $O/bench_H_final.err; cut -c1-400 $O/bench_H_final.json
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | f | tail -n 3 | tee $O/smoke.log
cp approximategps.jl_amd/csrc/build.log $O/build.log
