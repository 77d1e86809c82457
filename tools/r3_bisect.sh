#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in default nodpp now0 neither; do
  if [ $v = default ]; then python tools/chol_check.py 2>&1 | tail -12; else SVGP_MI355X_LIB=$PWD/approximategps.jl_amd/csrc/ablate/libsvgp_$v.so python tools/chol_check.py 2>&1 | tail -12; fi
done
