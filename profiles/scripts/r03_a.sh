#!/bin/bash
# round 3, first GPU call: the tier-A FM backward -- parity tests that exercise it, then the headline bench both ways
mkdir -p gpurun_out/r03
python -m pytest tests/test_gpu_ranking.py tests/test_gpu_cabi_vs_c_oracle.py tests/test_gpu_edge_cases.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r03/a_tests.log
cat gpurun_out/r03/a_tests.log
python bench.py > gpurun_out/r03/a_bench_tier.json 2> gpurun_out/r03/a_bench_tier.err
cat gpurun_out/r03/a_bench_tier.json
RBX_FM_TIER_A=0 python bench.py > gpurun_out/r03/a_bench_notier.json 2> gpurun_out/r03/a_bench_notier.err
cat gpurun_out/r03/a_bench_notier.json
