cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
CFGS="H C2 C4" bash tools/run_ab.sh
python tools/grad_time.py 2>/dev/null | tail -3
SVGP_MI355X_LIB=$PWD/approximategps.jl_amd/csrc/ablate/libsvgp_prev.so python tools/grad_time.py 2>/dev/null | tail -3
