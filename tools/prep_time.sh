#!/usr/bin/env bash
# prep (Kuu, Cholesky, T panels, KL) ms per config; optional first argument: another library build to compare with
cd "$(dirname "$0")/.."
for r in 1 2; do for cfg in H C2 C5 C4 C3; do
  echo -n "this  "; python tools/ablate_time.py $cfg 2>&1 | tail -1
  [ -n "$1" ] && { echo -n "other "; SVGP_MI355X_LIB=$1 python tools/ablate_time.py $cfg 2>&1 | tail -1; }
done; done
