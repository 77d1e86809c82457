cd $GRAFT_REPO_ROOT
for rep in 1 2; do for cfg in H32 C3 C5; do for w in 2 3 4; do
  SVGP_WG_PER_CU=$w python tools/ablate_time.py $cfg 2>/dev/null | sed "s/^/wg$w /"
done; done; done
