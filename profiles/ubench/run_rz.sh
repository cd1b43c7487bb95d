# bench + kernel stats of the default run (scratch helper): bash profiles/ubench/run_rz.sh [extra bench flags]
mkdir -p gpurun_out/rz
python bench.py --no-cpu-baseline "$@" > gpurun_out/rz/bench.json 2> gpurun_out/rz/bench.err
cut -c1-330 gpurun_out/rz/bench.json
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof -o rz -- python $R/bench.py --no-cpu-baseline "$@" > /dev/null 2>&1
cd $R
python profiles/topk.py $(find /tmp/prof -name "rz_results.db" | head -1) > gpurun_out/rz/stats.txt
head -24 gpurun_out/rz/stats.txt | cut -c1-130
