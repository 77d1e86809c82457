cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
A=$PWD/approximategps.jl_amd/csrc/ablate
export SVGP_STRIP_NT=64
export SVGP_MI355X_LIB=$A/libsvgp_ablate_7.so
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA --kernel-trace --output-format csv -d gpurun_out/g8/a -- python3 tools/ablate_time.py H > gpurun_out/g8_a.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU --kernel-trace --output-format csv -d gpurun_out/g8/b -- python3 tools/ablate_time.py H > gpurun_out/g8_b.log 2>&1
rocprofv3 --pmc SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_LDS SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_VALU_MFMA_COEXEC_CYCLES --kernel-trace --output-format csv -d gpurun_out/g8/c -- python3 tools/ablate_time.py H > gpurun_out/g8_c.log 2>&1
tail -1 gpurun_out/g8_a.log gpurun_out/g8_b.log gpurun_out/g8_c.log
