#!/bin/bash
out=/root/repo/gpurun_out/r2e
rm -rf $out; mkdir -p $out
cd /root/repo
timeout 1200 python -m pytest tests/test_gpu_matching.py -x -q -m gpu > $out/matching.log 2>&1
tail -25 $out/matching.log
