#!/bin/bash
# the new bench configs on one GPU (single-GPU form and world-of-one sharded form)
out=/root/repo/gpurun_out/r2d
rm -rf $out; mkdir -p $out
cd /root/repo
for cfg in youtubednn deepfm sasrec; do
  timeout 600 python bench.py --config $cfg --steps 20 --warmup 5 > $out/bench_$cfg.json 2> $out/bench_$cfg.err
  tail -c 1800 $out/bench_$cfg.json; tail -3 $out/bench_$cfg.err | cut -c1-300
done
for cfg in youtubednn deepfm; do
  timeout 600 python bench.py --config $cfg --steps 20 --warmup 5 --force-sharded --no-cpu-baseline > $out/bench_${cfg}_sharded1.json 2> $out/bench_${cfg}_sharded1.err
  tail -c 1500 $out/bench_${cfg}_sharded1.json; tail -3 $out/bench_${cfg}_sharded1.err | cut -c1-300
done
