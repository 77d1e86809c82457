# profile of one bench configuration: kernel stats + HBM traffic and SQ counters in separate passes (as the guide prescribes)
# usage (on the GPU box): bash tools/run_profile.sh <tag> [config] [grad]      -> gpurun_out/prof_<tag>/
#   third argument "grad": profile value-and-gradient evaluations (tools/grad_time.py) instead of the forward bench loop
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_$1; C=${2:-H}; mkdir -p $O
cat approximategps.jl_amd/csrc/strip.hip approximategps.jl_amd/csrc/device_common.hpp | sha256sum | cut -c1-16 > $O/kernel_source_sha16.txt   # what was profiled
if [ "${3:-}" = "grad" ]; then
  B="tools/grad_time.py $C"; S=""; P=""
else
  B="bench.py --config $C --no-cpu-baseline --no-grad --no-c5"; S="--steps 5 --warmup 1"; P="--steps 3 --warmup 1"
fi
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $B $S > $O/bench_stats.json 2> $O/stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 $B $P > $O/bench_fetch.json 2> $O/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 $B $P > $O/bench_write.json 2> $O/write.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/sq -- python3 $B $P > $O/bench_sq.json 2> $O/sq.err
timeout 900 python bench.py --config $C --no-c5 --steps 20 > $O/bench_plain.json 2> $O/plain.err
# keep the merge-back small: only the per-kernel tables
find $O -name "*agent_info.csv" -delete; find $O -name "*domain_stats.csv" -delete
find $O -name "*.csv" | wc -l; cut -c1-300 $O/bench_plain.json
