#!/bin/bash
# rocprofv3 passes of the headline bench in the overlap-save form: durations, then HBM-side byte counters (separate --pmc passes)
set -o pipefail
TAG=${1:-os}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_r4_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$REPO/bench.py --steps 10 --warmup 2 --prewarm-ms 0 --no-cpu-baseline --no-latency --no-host-io --no-parity"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 ${ARGS/--prewarm-ms 0/--prewarm-ms 100} > $OUT/stats.log 2>&1 || { tail -5 $OUT/stats.log; exit 1; }
pass() {
    rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $OUT/$1 -- python3 $ARGS > $OUT/$1.log 2>&1 || { echo "pass $1 failed"; tail -3 $OUT/$1.log; }
    find $OUT/$1 -name "*_kernel_trace.csv" -delete
}
pass pmc_fetch "FETCH_SIZE"
pass pmc_write "WRITE_SIZE"
pass pmc_tcc "TCC_HIT_sum TCC_MISS_sum"
pass pmc_mem1 "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum"
pass pmc_mem3 "TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum"
pass pmc_mem4 "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_64B_sum"
pass pmc_sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for d in sorted(glob.glob(out + '/pmc_*/')):
    fs = glob.glob(d + '/*/*counter_collection.csv')
    if not fs: continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')[:24]
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, cs in acc.items():
        if not k.startswith('k_os') : continue
        print(d.split('/')[-2], k, ' '.join('%s=%.4g(n%d)' % (c, sum(v) / len(v), len(v)) for c, v in cs.items()))
PY
du -sh $OUT
