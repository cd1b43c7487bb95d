# scratch: timeline of one hipGraph replay of the default bench (profiles/timeline.py) + kernel stats
mkdir -p gpurun_out/rz
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof
rocprofv3 --kernel-trace --stats -d /tmp/prof -o tl -- python $R/bench.py --no-cpu-baseline "$@" > /dev/null 2>&1
cd $R
DB=$(find /tmp/prof -name "tl_results.db" | head -1)
python profiles/timeline.py $DB ${ANCHOR:-rezero_rows} ${WHICH:-30} > gpurun_out/rz/timeline${TAG}.txt 2>&1
python profiles/topk.py $DB > gpurun_out/rz/stats${TAG}.txt
