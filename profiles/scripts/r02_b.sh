#!/bin/bash
# round 2, second GPU call: the sharded exchange kernels and the sharded model mirrors
out=/root/repo/gpurun_out/r2b
rm -rf $out; mkdir -p $out
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_shard.py -x -q -m gpu > $out/shard.log 2>&1
tail -25 $out/shard.log
timeout 900 python -m pytest tests/test_gpu_sharded_world2.py -x -q -m gpu > $out/world2.log 2>&1
tail -25 $out/world2.log
