cd $GRAFT_REPO_ROOT
A=$PWD/approximategps.jl_amd/csrc/ablate
for rep in 1 2; do
python tools/ablate_time.py H 2>/dev/null
for v in 1 2 3; do SVGP_MI355X_LIB=$A/libsvgp_nt$v.so python tools/ablate_time.py H 2>/dev/null; done
done
for v in 0 3; do L=$A/libsvgp_nt$v.so; [ $v = 0 ] && L=$PWD/approximategps.jl_amd/csrc/libsvgp_mi355x.so; SVGP_MI355X_LIB=$L python tools/ablate_time.py H32 2>/dev/null; SVGP_MI355X_LIB=$L python tools/ablate_time.py C4 2>/dev/null; done
