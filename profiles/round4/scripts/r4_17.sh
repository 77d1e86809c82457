cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_17; mkdir -p $O
for r in 0 8 16 32 64; do SVGP_STREAM2_RESERVE=$r timeout 300 python tools/overlap_time.py f64 2>&1 | grep "n=" | sed "s/^/reserve=$r /"; done | tee $O/reserve_f64.log
for r in 16 32; do SVGP_OVERLAP_P2CKPT=1 SVGP_STREAM2_RESERVE=$r timeout 300 python tools/overlap_time.py f64 2>&1 | grep "n=" | sed "s/^/ckpt reserve=$r /"; done | tee $O/reserve_ckpt_f64.log
