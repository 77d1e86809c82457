#!/usr/bin/env bash
# same-box, interleaved A/B of several library builds on value-and-gradient (and forward) times: LIBS="a.so b.so" tools/round6/libs_ab.sh H C5 ...
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for cfg in "$@"; do for L in $LIBS; do
  echo "$(basename $L) $(SVGP_MI355X_LIB=$PWD/$L python3 tools/grad_time.py $cfg 2>&1 | tr '\n' ' ')"
done; done; done
