cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
A=$PWD/approximategps.jl_amd/csrc/ablate
export SVGP_STRIP_NT=64
for v in default 7; do
  if [ $v = 7 ]; then export SVGP_MI355X_LIB=$A/libsvgp_ablate_7.so; fi
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/g9/$v -- python3 tools/ablate_time.py H > gpurun_out/g9_$v.log 2>&1
done
