# same-box A/B of strip-schedule knobs: forward strip ms for the configs given (default C2 C4 C5 H32)
for rep in 1 2; do for c in ${CFGS:-C2 C4 C5 H32}; do
python tools/ablate_time.py $c 2>/dev/null | sed 's/^/ctail=1 /'
SVGP_CTAIL=0 python tools/ablate_time.py $c 2>/dev/null | sed 's/^/ctail=0 /'
done; done
