#!/bin/bash
# round 5: quad forward with explicitly rounded arithmetic in both kernels: FM tests; timeline of the replayed step
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05e
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ranking.py -x -q -m gpu -k "quad or fm or FM or bench_configuration" > $O/tests.log 2>&1
echo "tests exit $?"; tail -5 $O/tests.log
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -f csv -d $GRAFT_REPO_ROOT/$O/prof -o fm -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-extra-configs --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench_fm.json 2> $GRAFT_REPO_ROOT/$O/bench_fm.err)
ls $O/prof; python profiles/timeline.py $(find $O/prof -name "*kernel_trace.csv" | head -1) > $O/fm_replay_timeline.txt 2>&1; cat $O/fm_replay_timeline.txt | head -40
find $O/prof -name "*.csv" -size +4000k -delete
