cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4_26; mkdir -p $O
for rep in 1 2; do
for cfg in MB16k; do
  timeout 300 python tools/grad_time.py $cfg 2>/dev/null | sed "s/^/default      /"
  for r in 16 32 64; do SVGP_STREAM2_RESERVE=$r timeout 300 python tools/grad_time.py $cfg 2>/dev/null | sed "s/^/reserve=$r   /"; done
  SVGP_OVERLAP_P2CKPT=1 timeout 300 python tools/grad_time.py $cfg 2>/dev/null | sed "s/^/p2ckpt       /"
  SVGP_STREAM2_LOW_PRIO=0 timeout 300 python tools/grad_time.py $cfg 2>/dev/null | sed "s/^/sameprio     /"
done; done | tee $O/ab.log
timeout 600 python tools/overlap_time.py f64 2>/dev/null | tee $O/overlap_f64.log
timeout 600 python tools/overlap_time.py f32 2>/dev/null | tee $O/overlap_f32.log
