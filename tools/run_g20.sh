#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_grad.py -x -q 2>&1 | tail -25
