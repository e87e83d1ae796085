#!/bin/bash
# rocprofv3 counter passes (one run per pass: <=8 SQ, <=4 TCC, <=2 GRBM counters each) over
# tools/prof_run.py (or PT_PMC_PROG=tools/ref_gfx950_time.py).  Usage: tools/pmc_passes.sh <tag> [program args...];
# CSVs land in gpurun_out/pmc_<tag>/.
set -u
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/pmc_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
passes=(
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAIT_ANY GRBM_GUI_ACTIVE"
 "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32"
 "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_MUL_F64 SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_INSTS_LDS SQ_INSTS_VMEM"
 "FETCH_SIZE"
 "WRITE_SIZE"
)
# PT_PMC_EXTRA=1: the memory-side passes (L2 hit/miss, L1 -> L2 requests and their latency, LDS activity)
if [ "${PT_PMC_EXTRA:-0}" = "1" ]; then
passes+=(
 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum"
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum"
 "SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_LDS_LOAD"
)
fi
i=0
for p in "${passes[@]}"; do
  i=$((i+1))
  echo "== pass $i: $p"
  timeout -s KILL 120 rocprofv3 --pmc $p -d "$out/p$i" -o p$i --output-format csv -- python3 "$root/${PT_PMC_PROG:-tools/prof_run.py}" "$@" > "$out/p$i.log" 2>&1 || { echo "pass $i failed"; tail -5 "$out/p$i.log"; exit 1; }
  tail -1 "$out/p$i.log"
done
find "$out" -name "*counter_collection.csv" | head
