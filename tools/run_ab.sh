# same-box A/B: working-tree library vs csrc/ablate/libsvgp_prev.so, interleaved
cd $GRAFT_REPO_ROOT
P=$PWD/approximategps.jl_amd/csrc/ablate/libsvgp_prev.so
for rep in 1 2; do for cfg in ${CFGS:-H H32 C3}; do
  python tools/ablate_time.py $cfg 2>/dev/null
  SVGP_MI355X_LIB=$P python tools/ablate_time.py $cfg 2>/dev/null
done; done
