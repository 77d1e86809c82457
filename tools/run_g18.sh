cd $GRAFT_REPO_ROOT
for rep in 1 2; do for cfg in H C2; do
  python tools/ablate_time.py $cfg 2>/dev/null | sed "s/^/default      /"
  SVGP_STRIP_NT=128 SVGP_F64_THREADS=256 SVGP_WG_PER_CU=1 python tools/ablate_time.py $cfg 2>/dev/null | sed "s/^/w64x64_1wg /"
done; done
