cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4_35; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_round4.py tests/test_gpu_grad.py -m gpu -q -x 2>&1 | grep -E "passed|failed|Error|assert" | tee $O/pytest.log
for rep in 1 2; do for sp in 1 0; do for sh in "1024 1024" "4096 1024" "8192 1024" "4096 2048" "2048 768"; do
  SVGP_SEG_SPLIT=$sp python tools/mb_one.py $sh 2>/dev/null | awk -v s="$sh" -v sp=$sp '{n=split($0,a," "); m=1e9; for(i=3;i<=n;i++) if(a[i]<m) m=a[i]; print "split=" sp, s, "min ms", m}'
done; done; done | tee $O/ab.log
