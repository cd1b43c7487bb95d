#!/bin/bash
# round 5: runtime settings against the ~20 us between two graph replays of the FM step (bench.py, 100 replayed steps, two processes each)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05x
mkdir -p $O
export TMPDIR=/tmp
run() { # name, env...
  n=$1; shift
  for rep in 1 2; do
    env "$@" timeout 200 python bench.py --steps 100 --warmup 10 --no-extra-configs --no-cpu-baseline > $O/bench_${n}_$rep.json 2> $O/bench_${n}_$rep.err
    python - <<PY | tee -a $O/ab.txt
import json
try:
    d=json.loads([l for l in open('$O/bench_${n}_$rep.json') if l.startswith('{')][-1]); r=d['roofline']
    print('%-40s ms_per_step %.4f  fwd %.1f us' % ('${n}_$rep', d['ms_per_step'], r['kernel_ms']*1e3))
except Exception as e:
    print('${n}_$rep', 'failed', e); print(open('$O/bench_${n}_$rep.err').read()[-800:])
PY
  done
}
run default X=1
run dev_kernarg HIP_FORCE_DEV_KERNARG=1
run dev_kernarg0 HIP_FORCE_DEV_KERNARG=0
run hwq2 GPU_MAX_HW_QUEUES=2
run hwq8 GPU_MAX_HW_QUEUES=8
run no_interrupt HSA_ENABLE_INTERRUPT=0
run sdma0 HSA_ENABLE_SDMA=0
run graph_mempool HIP_MEM_POOL_USE_VM=0
run default_again X=1
