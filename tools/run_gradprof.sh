# kernel-time breakdown of value-and-gradient evaluations (rocprofv3 --stats) for one config
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
C=${1:-H}; O=gpurun_out/gradprof_$C; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --config $C --steps 1 --warmup 0 --no-kuf --no-cpu-baseline > $O/bench.json 2> $O/err.txt
find $O -name "*agent_info.csv" -delete; find $O -name "*domain_stats.csv" -delete; find $O -name "*kernel_trace.csv" -delete
f=$(ls -S $O/stats/*/*kernel_stats.csv | head -1); head -25 $f | cut -c1-200
