#!/bin/bash
# CPU only: builds and runs tools/native/fragile_tally.cpp (the library's host planner against the C oracle on the integers that hinge on the last bit of a sine).
#   usage: tools/fragile_tally.sh [fields] [seed] [threads]   -> stdout (kept as profiles/r05_fragile_tally.txt)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/build
gcc -O2 -ffp-contract=off -fno-fast-math -std=c11 -c -o $R/build/fcpp_oracle_tally.o $R/oracle/fcpp_oracle.c
g++ -O2 -std=c++17 -ffp-contract=off -fno-fast-math -pthread -Wno-unknown-pragmas -o $R/build/fragile_tally $R/tools/native/fragile_tally.cpp \
    $R/field_coverage_path_planning_amd/csrc/fcpp_host.cpp $R/build/fcpp_oracle_tally.o -lm
$R/build/fragile_tally ${1:-1200000} ${2:-1} ${3:-8}
