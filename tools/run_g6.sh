cd $GRAFT_REPO_ROOT
(timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -15) > gpurun_out/g6_tests.log 2>&1; grep -E "passed|failed|FAILED" gpurun_out/g6_tests.log
A=approximategps.jl_amd/csrc/ablate
for cfg in H H32; do for nt in 64 128; do
  SVGP_STRIP_NT=$nt python tools/ablate_time.py $cfg 2>/dev/null
  SVGP_STRIP_NT=$nt SVGP_MI355X_LIB=$PWD/$A/libsvgp_ablate_7.so python tools/ablate_time.py $cfg 2>/dev/null
done; done
SVGP_STRIP_NT=128 SVGP_STRIP_BK=32 python tools/ablate_time.py H 2>/dev/null
SVGP_STRIP_NT=128 SVGP_STRIP_BK=32 python tools/ablate_time.py H32 2>/dev/null
