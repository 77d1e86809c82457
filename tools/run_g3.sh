mkdir -p gpurun_out/g3; cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
(timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -5) > gpurun_out/g3/tests.log 2>&1; grep -E "passed|failed" gpurun_out/g3/tests.log
pr() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d['value'],3), 'evals/s', round(d['ms_per_step'],2), 'ms strip TF', round(d['roofline']['achieved'],1), 'frac', round(d['roofline']['frac'],3), d['breakdown_ms'])" "$1"; }
for c in H H32 C2; do timeout 600 python bench.py --config $c --steps 5 --warmup 1 --no-cpu-baseline --no-kuf 2>/dev/null | pr "$c bk16"; SVGP_STRIP_BK=32 timeout 600 python bench.py --config $c --steps 5 --warmup 1 --no-cpu-baseline --no-kuf 2>/dev/null | pr "$c bk32"; done
rocprofv3 -L 2>/dev/null | grep -o -E "\b(SQ_[A-Z_0-9]+|GRBM_[A-Z_0-9]+|TCC_[A-Z_0-9]+|FETCH_SIZE|WRITE_SIZE|TCP_[A-Z_0-9]+)\b" | sort -u > gpurun_out/g3/counters.txt; wc -l gpurun_out/g3/counters.txt
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/g3/pmc1 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kuf > gpurun_out/g3/pmc1.json 2> gpurun_out/g3/pmc1.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/g3/pmc2 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kuf > gpurun_out/g3/pmc2.json 2> gpurun_out/g3/pmc2.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/g3/pmc3 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kuf > gpurun_out/g3/pmc3.json 2> gpurun_out/g3/pmc3.err
find gpurun_out/g3 -name "*.csv" | head -20; du -sh gpurun_out/g3
