#!/bin/bash
# round 2, third GPU call: remaining shard tests + the new bench configs on one GPU (single-GPU form and world-of-one sharded form)
out=/root/repo/gpurun_out/r2c
rm -rf $out; mkdir -p $out
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_shard.py -x -q -m gpu > $out/shard.log 2>&1
tail -5 $out/shard.log
for cfg in youtubednn deepfm sasrec; do
  timeout 600 python bench.py --config $cfg --steps 20 --warmup 5 > $out/bench_$cfg.json 2> $out/bench_$cfg.err
  tail -c 1800 $out/bench_$cfg.json; tail -3 $out/bench_$cfg.err
done
for cfg in youtubednn deepfm; do
  timeout 600 python bench.py --config $cfg --steps 20 --warmup 5 --force-sharded --no-cpu-baseline > $out/bench_${cfg}_sharded1.json 2> $out/bench_${cfg}_sharded1.err
  tail -c 1500 $out/bench_${cfg}_sharded1.json; tail -3 $out/bench_${cfg}_sharded1.err
done
