cd $GRAFT_REPO_ROOT
for rep in 1 2; do for cfg in H32 C3 C5; do
  SVGP_F32_BK=32 python tools/ablate_time.py $cfg 2>/dev/null | sed 's/^/bk32 /'
  SVGP_F32_BK=16 python tools/ablate_time.py $cfg 2>/dev/null | sed 's/^/bk16 /'
done; done
