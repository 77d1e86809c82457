cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_tenth; mkdir -p $O
for rep in 1 2; do for a in 1 0; do for k in 1 0; do SVGP_CHOL_ASYNC=$a SVGP_CHOL_TWO_LEVEL=$k timeout 600 python tests/chol2_check.py 2>&1 | grep "two_level" | sed "s/^/async=$a /"; done; done; done | tee $O/chol2.log
