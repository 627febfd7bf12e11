#!/bin/bash
# Runs ON THE GPU BOX: memory-pipeline counters of one configuration's kernels (four per pass, one block per pass).
#   usage: tools/collect_mem_counters.sh <config> ; output gpurun_out/memctr/<config>/<pass>/ ; print with tools/pmc_summary.py
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
c=${1:-cfg1}
OUT=$R/gpurun_out/memctr/$c
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; timeout -k 10 240 rocprofv3 --pmc "$@" -d $OUT/$name -o p --output-format csv -- python3 $R/tools/prof_cfg.py $c --steps 3 > $OUT/$name.log 2>&1 || echo "$name failed"; }
run tcp1 TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum
run tcp2 TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum
run ta1 TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum
run tcc1 TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum
run tcc2 TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_NORMAL_WRITEBACK_sum
run sq3 SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY
echo done
