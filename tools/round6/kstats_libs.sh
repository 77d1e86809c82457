#!/usr/bin/env bash
# rocprofv3 per-kernel averages (gradient strips / SYRK / kgrad) of a value-and-gradient evaluation for several library builds: LIBS="a.so b.so" ... H C5
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for C in "$@"; do for L in $LIBS; do
  O=/tmp/ksl_$(basename $L .so)_$C; rm -rf $O; mkdir -p $O
  SVGP_MI355X_LIB=$GRAFT_REPO_ROOT/$L rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 tools/grad_time.py $C > $O/out.txt 2> $O/err.txt
  f=$(ls -S $O/stats/*/*kernel_stats.csv | head -1)
  echo "== $(basename $L) $C $(grep elbo_grad $O/out.txt | cut -c1-40)"
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].replace("void svgp::(anonymous namespace)::", "")
    if any(k in n for k in ("kgrad_mfma", "syrk", "strip_kernel")) and int(r["Calls"]) >= 4:
        print(f'   {n[:64]:66s} calls {r["Calls"]:>4s} avg {float(r["AverageNs"])/1e3:9.1f} us')
PY
done; done
