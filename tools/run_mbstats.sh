# rocprofv3 kernel stats of minibatch value-and-gradient evaluations: run_mbstats.sh N M d [f32]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/mbstats_$1_$2; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 tools/mb_grad.py "$@" > $O/out.txt 2> $O/err.txt
find $O -name "*agent_info.csv" -delete; find $O -name "*domain_stats.csv" -delete; find $O -name "*kernel_trace.csv" -delete
f=$(ls -S $O/stats/*/*kernel_stats.csv | head -1); cp $f gpurun_out/mbstats_$1_$2.csv; cat $O/out.txt
