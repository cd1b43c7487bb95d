#!/bin/bash
# round 5: counter passes on gemm_bxp_kernel (best shape 16384 x 4096 x 4096 and cfg 4's layer 1): what the wavefronts wait for
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05gp
mkdir -p $O
export TMPDIR=/tmp
(cd /tmp && timeout 120 rocprofv3 -L > $GRAFT_REPO_ROOT/$O/counters_list.txt 2>&1)
grep -o -E "\b(SQ|TA|TCP|TCC|TD|GRBM)_[A-Z0-9_a-z]+" $O/counters_list.txt | sort -u > $O/counter_names.txt; wc -l $O/counter_names.txt
grep -i "lds\|mfma\|barrier\|wait\|VALU" $O/counter_names.txt | tr '\n' ' '
pass() {  # name, shape tag, M K N, counters...
  n=$1; tag=$2; M=$3; K=$4; N=$5; shift 5
  (cd /tmp && timeout 180 rocprofv3 --kernel-trace --pmc "$@" -f csv -d $GRAFT_REPO_ROOT/$O/pmc_${tag}_$n -o p -- python $GRAFT_REPO_ROOT/profiles/ubench/gemm_one.py $M $K $N > $GRAFT_REPO_ROOT/$O/pmc_${tag}_$n.log 2>&1)
  echo "pass $n $tag exit $?"
}
for shape in "big 16384 4096 4096" "layer1 65536 1677 400"; do
  set -- $shape
  pass sq1 $1 $2 $3 $4 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
  pass sq2 $1 $2 $3 $4 SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE
  pass sq3 $1 $2 $3 $4 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC
  pass sq4 $1 $2 $3 $4 SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_IFETCH SQ_WAIT_IFETCH SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_MFMA_F32
  pass tcp $1 $2 $3 $4 TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum
  pass tcc $1 $2 $3 $4 TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum
  echo "== $1"; python profiles/pmc_csv.py $O/pmc_$1_sq1 gemm_bxp; python profiles/pmc_csv.py $O/pmc_$1_sq2 gemm_bxp; python profiles/pmc_csv.py $O/pmc_$1_sq3 gemm_bxp; python profiles/pmc_csv.py $O/pmc_$1_sq4 gemm_bxp; python profiles/pmc_csv.py $O/pmc_$1_tcp gemm_bxp; python profiles/pmc_csv.py $O/pmc_$1_tcc gemm_bxp
done > $O/pmc_summary.txt 2>&1
cat $O/pmc_summary.txt
grep -l -i "error\|invalid\|fail" $O/pmc_*.log | head -20
find $O -name "*.csv" -size +200k -delete
