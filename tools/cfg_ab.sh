# same-box A/B of strip-kernel knobs: forward strip ms for the configs given (default C4 C2); KNOB=ENVVAR toggles 1 / 0
K=${KNOB:-SVGP_CTAIL}
for rep in 1 2; do for c in ${CFGS:-C4 C2}; do
env $K=1 python tools/ablate_time.py $c 2>/dev/null | sed "s/^/$K=1 /"
env $K=0 python tools/ablate_time.py $c 2>/dev/null | sed "s/^/$K=0 /"
done; done
