#!/usr/bin/env bash
# same-box A/B of library builds through bench.py itself (one line per config and library): LIBS="a.so b.so" tools/round6/bench_ab.sh C2 C5 ...
cd $GRAFT_REPO_ROOT
for r in 1 2; do for c in "$@"; do for L in $LIBS; do
  SVGP_MI355X_LIB=$PWD/$L timeout 600 python bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline --no-c5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=d.get('value_and_gradient',{}); b=d['breakdown_ms']
print('$(basename $L)', '$c', 'ms/step', round(d['ms_per_step'],3), 'device sum', round(sum(v for k,v in b.items() if not k.startswith('cholesky')),3), 'grad ms', round(g.get('ms_per_eval',0),3))"
done; done; done
