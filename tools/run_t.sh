cd $GRAFT_REPO_ROOT
SVGP_MI355X_LIB=$PWD/approximategps.jl_amd/csrc/ablate/libsvgp_stamps.so python tools/potf2_time.py
