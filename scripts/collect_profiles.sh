#!/bin/bash
# Collects what profiles/ is built from (run through gpurun from the repository root):
#   kernel-trace statistics and the two HBM PMC passes of the default bench workload, in separate
#   rocprofv3 runs (counters never together with other trace domains), the program right after `--`.
# AGX_QUEUE_AHEAD=0: no speculative (empty) derivative-pass launches, so per-kernel averages are those
# of real launches only.  AGX_K1_FUSED=0: the running-node derivative kernel as its own launch.
set -e
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export AGX_QUEUE_AHEAD=0
export AGX_K1_FUSED=0   # running / terminal nodes as separate launches: the kernel the roofline is quoted on appears alone
rocprofv3 --kernel-trace --stats -d $OUT/trace -- python3 $ROOT/bench.py --no-cpu-baseline --no-batch1 > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -- python3 $ROOT/bench.py --no-cpu-baseline --no-batch1 > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -- python3 $ROOT/bench.py --no-cpu-baseline --no-batch1 > $OUT/write.log 2>&1
cd $ROOT
python scripts/rocpd_stats.py $(find $OUT/trace -name "*.db" | head -1) > $OUT/kernel_stats.txt
python scripts/pmc_summary.py $(find $OUT/fetch -name "*.db" | head -1) > $OUT/pmc_fetch.txt
python scripts/pmc_summary.py $(find $OUT/write -name "*.db" | head -1) > $OUT/pmc_write.txt
