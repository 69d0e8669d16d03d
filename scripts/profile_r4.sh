#!/bin/bash
# Round 4 (profile_r3.sh for round 4).  Runs on the GPU box (via gpurun): rocprofv3 passes of one bench.py command, each in its own run
# (kernel-trace --stats; then --pmc passes WITHOUT any trace domain besides kernel-trace: the pool refuses more).
#   profile_r3.sh <tag> [bench.py arguments]      outputs under gpurun_out/prof_r4_<tag>/
set -o pipefail
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_r4_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$REPO/bench.py --steps 10 --warmup 2 --prewarm-ms 0 --no-cpu-baseline --no-latency --no-host-io --no-parity --no-literal-mac $@"
echo "bench.py --steps 10 --warmup 2 --prewarm-ms 0 --no-cpu-baseline --no-latency --no-host-io --no-parity --no-literal-mac $@" > $OUT/command.txt
# (the duration pass with the clocks warm, like the bench itself: 100 ms of untimed steps first; the counter passes without)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 ${ARGS/--prewarm-ms 0/--prewarm-ms 100} > $OUT/stats.log 2>&1 || { tail -5 $OUT/stats.log; exit 1; }
pass() {  # name, counters
    rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $OUT/$1 -- python3 $ARGS > $OUT/$1.log 2>&1 || { echo "pass $1 failed"; tail -3 $OUT/$1.log; }
    # keep the merged output small: only the per-dispatch counter table
    find $OUT/$1 -name "*_kernel_trace.csv" -delete
}
pass pmc_fetch "FETCH_SIZE"
pass pmc_write "WRITE_SIZE"
pass pmc_sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
pass pmc_sq2 "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM"
pass pmc_sq3 "SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_BUSY_CU_CYCLES"
pass pmc_tcc "TCC_HIT_sum TCC_MISS_sum"
# round 3: how a CU requests memory - L1 -> L2 requests and their latency, the cycles the L1 could not
# take another request, how busy the address units are, and the L2's requests split by where they went
pass pmc_mem1 "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum"
pass pmc_mem2 "TA_BUSY_avr TA_TA_BUSY_sum TCP_GATE_EN1_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
pass pmc_mem3 "TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum"
pass pmc_mem4 "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_64B_sum"
find $OUT -name "*_kernel_trace.csv" -size +2M -delete
du -sh $OUT
