#!/bin/bash
# HBM traffic of the FM forward kernel and the FM segmented reduce from the PMC counters, separate passes (as r02_h.sh)
out=/root/repo/gpurun_out/r2pmcfm
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
cd /root/repo
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $c -d $out/pmc_fm_$c -o b -- python /root/repo/bench.py --no-cpu-baseline --eager --steps 5 --warmup 3 > /dev/null 2>&1)
  python profiles/pmc.py $(find $out/pmc_fm_$c -name "*.db" | head -1) $c > $out/pmc_fm_$c.txt
  rm -rf $out/pmc_fm_$c
done
grep -h -E "fm_fused_fwd|segment_reduce" $out/pmc_*.txt | cut -c1-160
