#!/usr/bin/env bash
# rocprofv3 per-kernel averages of kgrad / strips / SYRK in a value-and-gradient evaluation, for the configs given, once per setting of
# an experiments-build knob:  KNOB=SVGP_KGRAD_WG_PER_CU VALUES="2 4" bash tools/round6/kgrad_stats.sh H C5
# (LIB=product uses the product library and ignores the knob)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
[ "${LIB:-exp}" = exp ] && export SVGP_MI355X_LIB=$GRAFT_REPO_ROOT/approximategps.jl_amd/csrc/ablate/libsvgp_experiments.so
for C in "$@"; do
  for V in ${VALUES:-default}; do
    [ -n "${KNOB:-}" ] && export $KNOB=$V
    O=/tmp/kgs_${C}_$V; rm -rf $O; mkdir -p $O
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 tools/grad_time.py $C > $O/out.txt 2> $O/err.txt
    f=$(ls -S $O/stats/*/*kernel_stats.csv | head -1)
    echo "== $C ${KNOB:-}=$V  $(grep elbo_grad $O/out.txt)"
    python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].replace("void svgp::(anonymous namespace)::", "")
    if any(k in n for k in ("kgrad", "syrk", "strip_kernel")) and int(r["Calls"]) >= 4:
        print(f'   {n[:64]:66s} calls {r["Calls"]:>4s} avg {float(r["AverageNs"])/1e3:9.1f} us')
PY
  done
done
