#!/bin/bash
cd /root/repo
python -m pytest tests/test_gpu_matching.py -x -q -m gpu -k "prelu or batch_norm or dssm or mlp" 2>&1 | tail -25
