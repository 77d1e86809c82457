cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q 2>&1 | tail -15) > gpurun_out/g13_tests.log 2>&1; grep -E "passed|failed|FAILED|Error|assert" gpurun_out/g13_tests.log | head
BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kuf 2>&1 | tail -2 | cut -c1-400
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-kuf 2>&1 | tail -2 | cut -c1-300
