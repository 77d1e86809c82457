cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_eighth; mkdir -p $O
A=$PWD/approximategps.jl_amd/csrc/ablate
for v in base wstep0 wstep2 syrknow; do
  L=$A/libsvgp_$v.so; [ $v = base ] && L=$PWD/approximategps.jl_amd/csrc/libsvgp_mi355x.so
  export SVGP_MI355X_LIB=$L
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_$v -- python3 tools/grad_time.py H > /dev/null 2> $O/st_$v.err
  f=$(find $O/st_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v"; grep -i "syrk_async\|strip_kernel\|kgrad" $f | cut -d, -f1-5 | cut -c1-160
done | tee $O/syrk_variants.log
unset SVGP_MI355X_LIB
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
timeout 900 python -m pytest tests/test_gpu_round4.py -m gpu -x -q > $O/pytest_r4.log 2>&1; grep -n "passed\|failed" $O/pytest_r4.log | tail -2
