#!/bin/bash
# Round 3 diagnostics: shader clock + VALU issue cost from in-kernel counters, and SQ / GRBM
# counters of the SINGLE-VIEW step only (scripts/cbench ... single), so every kernel row is the headline step.
#   gpurun -- bash scripts/gpu_r3_diag.sh TAG [lib]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r03a}
LIB=${2:-r2_gaussian_amd/libr2hip.so}
O=gpurun_out/diag_$TAG
mkdir -p $O
echo "== ubench_clock"; timeout 120 scripts/ubench_clock | tee $O/ubench_clock.txt
echo "== cbench single+stages"; timeout 200 scripts/cbench 300 $LIB single,stages | tee $O/cbench_single.txt
echo "== kernel trace (no counters)"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- scripts/cbench 100 $LIB single > $O/kt.log 2>&1; tail -1 $O/kt.log
SQ1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY"
SQ2="SQ_BUSY_CU_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
SQ3="GRBM_GUI_ACTIVE GRBM_COUNT SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU"
i=0
for SET in "$SQ1" "$SQ2" "$SQ3" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $O/pmc$i -o p -- scripts/cbench 30 $LIB single > $O/pmc$i.log 2>&1
  tail -1 $O/pmc$i.log
done
python3 scripts/summarize_diag.py $O | tee $O/summary.txt
find $O -name "*.csv" -size +6M -delete
