#!/usr/bin/env bash
# same-box A/B of two library builds (value-and-gradient ms): tools/lib_ab.sh <other.so> [configs...]
cd "$(dirname "$0")/.."
OTHER=$1; shift
for r in 1 2; do for cfg in "$@"; do
  echo -n "this  "; python tools/grad_time.py $cfg 2>&1 | tail -1
  echo -n "other "; SVGP_MI355X_LIB=$OTHER python tools/grad_time.py $cfg 2>&1 | tail -1
done; done
