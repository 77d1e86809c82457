cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_12; mkdir -p $O
for rep in 1 2; do
python tools/ab_r3.py 2>&1 | grep "n="
SVGP_MI355X_LIB=$PWD/approximategps.jl_amd/csrc/ablate/libsvgp_r3.so python tools/ab_r3.py 2>&1 | grep "n="
done | tee $O/ab_r3.log
