cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_19; mkdir -p $O
export SVGP_MI355X_LIB=$PWD/approximategps.jl_amd/csrc/ablate/libsvgp_stripstamps.so
python tools/strip_stamps_grad.py H 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|^RCCL\|^HIP" | tee $O/stamps_grad_H.log
python tools/strip_stamps.py H 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|^RCCL\|^HIP" | tee $O/stamps_fwd_H.log
python tools/strip_stamps_grad.py H32 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|^RCCL\|^HIP" | tee $O/stamps_grad_H32.log
