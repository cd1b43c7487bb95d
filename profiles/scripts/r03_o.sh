#!/bin/bash
cd /root/repo
python -m pytest tests/test_gpu_shard.py tests/test_gpu_sharded_world2.py -x -q -m gpu 2>&1 | tail -25
