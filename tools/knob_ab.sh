#!/usr/bin/env bash
# same-box A/B of one environment knob: tools/knob_ab.sh NAME "v1 v2 ..." cfg...   (strip / prep ms via tools/ablate_time.py)
cd "$(dirname "$0")/.."
K=$1; VALS=$2; shift 2
for r in 1 2; do for cfg in "$@"; do for v in $VALS; do
  echo -n "$K=$v "; env $K=$v python tools/ablate_time.py $cfg 2>&1 | tail -1
done; done; done
