#!/bin/bash
# potf2 variants: correctness (chol_check), stamps per dtype, M = 128 prep time against a reference build, prep per config
A=approximategps.jl_amd/csrc/ablate
timeout 300 python3 tools/chol_check.py 2>&1 | tail -1
POTF2_DTYPES=f64 SVGP_MI355X_LIB=$A/libsvgp_stamps.so python3 tools/potf2_time.py
POTF2_DTYPES=f32 SVGP_MI355X_LIB=$A/libsvgp_stamps.so python3 tools/potf2_time.py
for r in 1 2; do
  echo "this:  $(python3 tools/potf2_time.py | head -1)"
  [ -f $A/libsvgp_ref.so ] && echo "ref:   $(SVGP_MI355X_LIB=$A/libsvgp_ref.so python3 tools/potf2_time.py | head -1)"
done
[ -f $A/libsvgp_ref.so ] && bash tools/prep_time.sh $A/libsvgp_ref.so 2>&1 | tail -10
