cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4_54; mkdir -p $O
SVGP_EVENT_FENCE=1 SVGP_ROW_EVENT_EXT=0 timeout 1500 python -m pytest tests/test_gpu_round4.py tests/test_gpu_grad.py tests/test_gpu_parity.py -m gpu -q 2>&1 | grep -E "passed|failed" | sed "s/^/fence=1 ext=0: /" | tee $O/knobs.log
SVGP_OVERLAP=0 timeout 1500 python -m pytest tests/test_gpu_grad.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q 2>&1 | grep -E "passed|failed" | sed "s/^/overlap=0: /" | tee -a $O/knobs.log
SVGP_SEG_SPLIT=0 SVGP_CHOL_CHAIN=1 timeout 1500 python -m pytest tests/test_gpu_grad.py tests/test_gpu_parity.py -m gpu -q 2>&1 | grep -E "passed|failed" | sed "s/^/split=0 chain=1: /" | tee -a $O/knobs.log
