cd $GRAFT_REPO_ROOT
A=approximategps.jl_amd/csrc/ablate
for nt in 64 128; do
  SVGP_STRIP_NT=$nt python tools/ablate_time.py H 2>/dev/null
  for v in 7 15 31; do SVGP_STRIP_NT=$nt SVGP_MI355X_LIB=$PWD/$A/libsvgp_ablate_$v.so python tools/ablate_time.py H 2>/dev/null; done
done
