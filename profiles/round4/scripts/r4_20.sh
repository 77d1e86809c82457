cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_20; mkdir -p $O
A=$PWD/approximategps.jl_amd/csrc/ablate/libsvgp_exptab.so
for rep in 1 2 3; do for c in H C2; do python tools/ablate_time.py $c 2>/dev/null | sed "s/^/base   /"; SVGP_MI355X_LIB=$A python tools/ablate_time.py $c 2>/dev/null | sed "s/^/exptab /"; done; done | tee $O/exptab.log
SVGP_MI355X_LIB=$A timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | grep -E "passed|failed" | tee $O/exptab_parity.log
BENCH_FORCE_DIST=1 timeout 900 python bench.py --gpus 1 --steps 10 --warmup 2 --no-c5 > $O/bench_forcedist.json 2> $O/bench_forcedist.err; cut -c1-200 $O/bench_forcedist.json
