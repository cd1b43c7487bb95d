#!/bin/bash
cd /root/repo
python -m pytest tests/test_gpu_sharded_world2.py -x -q -m gpu -k "sync_batch_norm" 2>&1 | tail -25
python -m pytest tests/test_gpu_matching.py -x -q -m gpu -k "batch_norm or mlp" 2>&1 | tail -5
