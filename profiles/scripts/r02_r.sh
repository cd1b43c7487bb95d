#!/bin/bash
out=/root/repo/gpurun_out/r2r
rm -rf $out; mkdir -p $out
cd /root/repo
timeout 2400 python -m pytest tests -x -q -m gpu > $out/tests.log 2>&1
tail -5 $out/tests.log
