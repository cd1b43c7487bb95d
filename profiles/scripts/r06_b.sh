#!/bin/bash
# round 6: does address translation (UTCL1 / UTCL2) add to the latency of the FM forward's misses?  Counter passes on the lab's g4 form.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06b
mkdir -p $O
export TMPDIR=/tmp
LAB=$GRAFT_REPO_ROOT/profiles/ubench/fm_fwd_lab
(cd /tmp && timeout 120 rocprofv3 -L > $GRAFT_REPO_ROOT/$O/counters_list.txt 2>&1)
grep -o -E "\b(SQ|SQC|TA|TCP|TCC|TD|GRBM|TCA|UTCL2|ATC|VM|MC)_[A-Z0-9_a-z]+" $O/counters_list.txt | sort -u > $O/counter_names.txt; wc -l $O/counter_names.txt
grep -i "utcl\|tlb\|trans" $O/counter_names.txt
pass() {  # name, variant, counters...
  n=$1; v=$2; shift 2
  (cd /tmp && timeout 180 rocprofv3 --kernel-trace --pmc "$@" -f csv -d $GRAFT_REPO_ROOT/$O/pmc_v${v}_$n -o p -- $LAB $v 20 > $GRAFT_REPO_ROOT/$O/pmc_v${v}_$n.log 2>&1)
  echo "pass $n variant $v exit $?"
}
v=9
pass utcl1 $v TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_PERMISSION_MISS_sum
pass utcl2 $v TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_LRU_INFLIGHT_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_LFIFO_FULL_sum
pass utcl3 $v TCP_UTCL1_STALL_LFIFO_NOT_RES_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_STALL_MISSFIFO_FULL_sum TCP_GATE_EN1_sum
python profiles/pmc_csv.py $O fm_ > $O/pmc_summary.txt 2>&1
cat $O/pmc_summary.txt
grep -l -i "error\|invalid\|fail" $O/pmc_*.log | head -20
find $O -name "*.csv" -size +200k -delete
