#!/bin/bash
# Runs ON THE GPU BOX (gpurun): rocprofv3 passes of every BASELINE configuration through tools/prof_cfg.py.
#   one --kernel-trace --stats pass, and counter passes in runs of their own (WRITE_SIZE, FETCH_SIZE, two SQ groups),
#   the program directly after `--`.  Output: gpurun_out/prof_<round>/<config>/<pass>/ ; summarised by tools/profiles_summary.py.
#   usage: tools/collect_profiles.sh [config ...]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof_${FCPP_ROUND:-r05}
CFGS=${*:-cfg1 cfg1_clothoid cfg2_ref cfg2_0.5 cfg2_0.1 cfg3 cfg5}
cd /tmp && export TMPDIR=/tmp
for c in $CFGS; do
    mkdir -p $OUT/$c
    rocprofv3 --kernel-trace --stats -d $OUT/$c/stats -o p --output-format csv -- python3 $R/tools/prof_cfg.py $c --steps 10 > $OUT/$c/stats.log 2>&1
    rocprofv3 --pmc WRITE_SIZE -d $OUT/$c/pmc_write -o p --output-format csv -- python3 $R/tools/prof_cfg.py $c --steps 3 > $OUT/$c/pmc_write.log 2>&1
    rocprofv3 --pmc FETCH_SIZE -d $OUT/$c/pmc_fetch -o p --output-format csv -- python3 $R/tools/prof_cfg.py $c --steps 3 > $OUT/$c/pmc_fetch.log 2>&1
    rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/$c/pmc_sq1 -o p --output-format csv -- python3 $R/tools/prof_cfg.py $c --steps 3 > $OUT/$c/pmc_sq1.log 2>&1
    rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU -d $OUT/$c/pmc_sq2 -o p --output-format csv -- python3 $R/tools/prof_cfg.py $c --steps 3 > $OUT/$c/pmc_sq2.log 2>&1
    echo "$c done: $(tail -1 $OUT/$c/stats.log | cut -c1-150)"
done
