#!/bin/bash
cd /root/repo
python -m pytest tests/test_gpu_optim.py -x -q -m gpu 2>&1 | tail -25
