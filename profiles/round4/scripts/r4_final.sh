#!/bin/bash
# round-4 closing run on the final sources: the evidence run (tools/r4_profiles.sh) + the overlap soak + smoke
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/r4_profiles.sh
O=gpurun_out/r4_prof
timeout 900 python tests/soak_overlap.py 300 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|^RCCL\|^HIP" | tee $O/soak_overlap.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke | tee $O/smoke.log
