# round 4, first GPU call: baseline of this box (tests, driver-like bench), the sustained leg, the strip-grid A/B with
# FETCH / WRITE and the list of counters rocprofv3 offers here
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_first; mkdir -p $O
(rocprofv3 -L > $O/counters_list.txt 2>&1 || rocprofv3 --list-avail > $O/counters_list.txt 2>&1) ; wc -l $O/counters_list.txt
timeout 1700 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 5 --min-seconds 30 --sustained-out $O/sustained_H.json --no-c5 --no-grad --no-cpu-baseline --no-kuf > $O/bench_sustained.json 2> $O/bench_sustained.err
for rep in 1 2; do for g in 512 480 448 384; do SVGP_STRIP_GRID=$g python tools/ablate_time.py H 2>/dev/null | sed "s/^/grid=$g /"; done; done > $O/strip_grid_ab.log; cat $O/strip_grid_ab.log
for g in 512 448; do
  SVGP_STRIP_GRID=$g rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/grid${g}_fetch -- python3 tools/ablate_time.py H > /dev/null 2> $O/grid${g}_fetch.err
  SVGP_STRIP_GRID=$g rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/grid${g}_write -- python3 tools/ablate_time.py H > /dev/null 2> $O/grid${g}_write.err
  SVGP_STRIP_GRID=$g rocprofv3 --pmc TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --kernel-trace --output-format csv -d $O/grid${g}_dram -- python3 tools/ablate_time.py H > /dev/null 2> $O/grid${g}_dram.err
done
find $O -name "*agent_info.csv" -delete; find $O -name "*domain_stats.csv" -delete
timeout 900 python bench.py > $O/bench_H.json 2> $O/bench_H.err; cut -c1-400 $O/bench_H.json
