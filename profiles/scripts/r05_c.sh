#!/bin/bash
# round 5, second GPU call: the FM forward laboratory in its arena form (8 batches in rotation): quad, packed rows, small tables
# in LDS, ablations of the miss classes; L1-miss counters (requests to L2 and their latency) per form.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05c
mkdir -p $O
export TMPDIR=/tmp
LAB=$GRAFT_REPO_ROOT/profiles/ubench/fm_fwd_lab
timeout 200 $LAB -1 40 > $O/lab.txt 2>&1; echo "lab exit $?"; cat $O/lab.txt
pass() {  # name, variant, counters...
  n=$1; v=$2; shift 2
  (cd /tmp && timeout 120 rocprofv3 --kernel-trace --pmc "$@" -f csv -d $GRAFT_REPO_ROOT/$O/pmc_v${v}_$n -o p -- $LAB $v 16 > $GRAFT_REPO_ROOT/$O/pmc_v${v}_$n.log 2>&1)
  echo "pass $n variant $v exit $?"
}
for v in 7 9; do
  pass tcp $v TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum
  pass sq $v SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS
done
for v in 7 9; do echo "== variant $v"; python profiles/pmc_csv.py $O/pmc_v${v}_tcp fm_; python profiles/pmc_csv.py $O/pmc_v${v}_sq fm_; done > $O/pmc_summary.txt 2>&1
cat $O/pmc_summary.txt
find $O -name "*.csv" -size +200k -delete
