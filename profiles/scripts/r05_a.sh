#!/bin/bash
# round 5, first GPU call: the FM forward laboratory (profiles/ubench/fm_fwd_lab.hip) -- every form, uniform and Zipf ids --
# and counter passes on the round-4 form (variant 0) and the quad form (variant 1) that name the limiter.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05a
mkdir -p $O
export TMPDIR=/tmp
LAB=$GRAFT_REPO_ROOT/profiles/ubench/fm_fwd_lab
timeout 120 $LAB -1 20 > $O/lab.txt 2>&1; echo "lab exit $?"; cat $O/lab.txt
(cd /tmp && timeout 120 rocprofv3 -L > $GRAFT_REPO_ROOT/$O/counters_list.txt 2>&1)
grep -o -E "\b(SQ|TA|TCP|TCC|TD|GRBM)_[A-Z0-9_a-z]+" $O/counters_list.txt | sort -u > $O/counter_names.txt; wc -l $O/counter_names.txt
pass() {  # name, variant, counters...
  n=$1; v=$2; shift 2
  (cd /tmp && timeout 180 rocprofv3 --kernel-trace --pmc "$@" -f csv -d $GRAFT_REPO_ROOT/$O/pmc_v${v}_$n -o p -- $LAB $v 20 > $GRAFT_REPO_ROOT/$O/pmc_v${v}_$n.log 2>&1)
  echo "pass $n variant $v exit $?"
}
for v in 0 1; do
  pass sq1 $v SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU
  pass sq2 $v SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAIT_INST_VMEM SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE
  pass ta $v TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum
  pass ta2 $v TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WAVEFRONTS_sum
  pass tcp1 $v TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
  pass tcp2 $v TCP_TA_TCP_STATE_READ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum
  pass tcp3 $v TCP_TOTAL_ACCESSES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
  pass tcc1 $v TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
  pass tcc2 $v TCC_TAG_STALL_sum TCC_BUSY_avr TCC_EA0_RDREQ_32B_sum TCC_READ_sum
done
python profiles/pmc_csv.py $O fm_ > $O/pmc_summary.txt 2>&1
cat $O/pmc_summary.txt
grep -l -i "error\|invalid\|fail" $O/pmc_*.log | head -20
find $O -name "*.csv" -size +200k -delete
