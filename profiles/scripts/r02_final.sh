#!/bin/bash
# end-of-round evidence: full GPU suite, the four bench lines (N = 1) + the world-of-one sharded forms, rocprofv3 kernel
# stats of the same commands, one replay timeline of the FM step, ATen operator breakdown of the model configs
out=/root/repo/gpurun_out/r2final
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
cd /root/repo
timeout 2400 python -m pytest tests -x -q -m gpu > $out/tests.log 2>&1
tail -3 $out/tests.log
timeout 600 python bench.py > $out/bench_fm.json 2>$out/bench_fm.err; cut -c1-300 $out/bench_fm.json
for c in youtubednn deepfm sasrec; do timeout 900 python bench.py --config $c > $out/bench_$c.json 2>$out/bench_$c.err; cut -c1-300 $out/bench_$c.json; done
timeout 600 python bench.py --config youtubednn --force-sharded --no-cpu-baseline > $out/bench_youtubednn_sharded1.json 2>/dev/null
timeout 600 python bench.py --config deepfm --force-sharded --no-cpu-baseline > $out/bench_deepfm_sharded1.json 2>/dev/null
timeout 600 python bench.py --force-sharded --no-cpu-baseline > $out/bench_fm_sharded1.json 2>/dev/null
timeout 600 python bench.py --rotate-by-copy --no-cpu-baseline > $out/bench_fm_rotate_by_copy.json 2>/dev/null
timeout 600 python bench.py --rotate 1 --no-cpu-baseline > $out/bench_fm_onebatch.json 2>/dev/null
timeout 600 python bench.py --dist zipf --no-cpu-baseline > $out/bench_fm_zipf.json 2>/dev/null
for f in youtubednn_sharded1 deepfm_sharded1 fm_sharded1 fm_rotate_by_copy fm_onebatch fm_zipf; do python -c "
import json
d=json.loads(open('$out/bench_$f.json').readline()); print('$f', round(d['ms_per_step'],4))"; done
prof() { # name, bench args
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $out/prof -o b -- python /root/repo/bench.py --no-cpu-baseline $2 > $out/prof_$1.log 2>&1)
  python profiles/topk.py $(find $out/prof -name "*.db" | head -1) 48 > $out/$1_kernel_stats.txt
  if [ $1 = fm ]; then python profiles/timeline.py $(find $out/prof -name "*.db" | head -1) rezero_rows 30 > $out/fm_replay_timeline.txt 2>&1; fi
  if [ $1 = fm ]; then python profiles/kernel_slice.py $(find $out/prof -name "*.db" | head -1) fm_fused_fwd 24 60 > $out/fm_fwd_kernel_by_phase.txt 2>&1; fi
  rm -rf $out/prof
}
prof fm ""
prof youtubednn "--config youtubednn --steps 20 --warmup 5"
prof deepfm "--config deepfm --steps 20 --warmup 5"
prof sasrec "--config sasrec --steps 20 --warmup 5"
for c in youtubednn deepfm sasrec; do timeout 300 python profiles/scripts/op_breakdown.py $c 2>/dev/null > $out/op_breakdown_$c.txt; done
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
