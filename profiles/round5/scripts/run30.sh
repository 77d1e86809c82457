#!/usr/bin/env bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5_sus
timeout 900 python bench.py --no-cpu-baseline --no-c5 --no-grad --min-seconds 30 --sustained-out gpurun_out/r5_sus/sustained_H.json > gpurun_out/r5_sus/line.json 2> gpurun_out/r5_sus/err.log
python3 -c "
import json; d=json.load(open('gpurun_out/r5_sus/sustained_H.json')); print(json.dumps(d)[:900])"
