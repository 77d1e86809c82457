# rocprofv3 kernel stats of value-and-gradient evaluations for the configs given (stats only)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for C in "$@"; do
  O=gpurun_out/gradstats_$C; mkdir -p $O
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 tools/grad_time.py $C > $O/out.txt 2> $O/err.txt
  find $O -name "*agent_info.csv" -delete; find $O -name "*domain_stats.csv" -delete; find $O -name "*kernel_trace.csv" -delete
  f=$(ls -S $O/stats/*/*kernel_stats.csv | head -1); echo "== $C"; head -12 $f | cut -d, -f1-4 | cut -c1-150
done
