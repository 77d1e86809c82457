# round profile of the default bench: kernel stats + HBM traffic counters (separate passes, as the guide prescribes)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_$1; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-grad > $O/bench_stats.json 2> $O/stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-grad > $O/bench_fetch.json 2> $O/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-grad > $O/bench_write.json 2> $O/write.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/sq -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kuf --no-grad > $O/bench_sq.json 2> $O/sq.err
timeout 600 python bench.py > $O/bench_plain.json 2> $O/plain.err
find $O -name "*.csv" | wc -l; cat $O/bench_plain.json | cut -c1-200
