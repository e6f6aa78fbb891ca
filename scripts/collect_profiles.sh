#!/bin/bash
# Collects what profiles/ is built from (run through gpurun from the repository root):
#   scripts/collect_profiles.sh [workload] [extra bench.py arguments]
# kernel-trace statistics and the two HBM PMC passes of one bench workload, in separate rocprofv3 runs (counters
# never together with other trace domains), the program right after `--`.  The humanoid workload adds the fp64
# matrix-core counter pass.
# AGX_NO_EMPTY_LAUNCHES=1: the trial launches of the line search are skipped when the head of the step finished every
# instance, so per-kernel averages are those of working launches only.  AGX_K1_FUSED=0: the running-node derivative kernel as
# its own launch.
set -e
W=${1:-sine}
shift || true
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prof_$W
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export AGX_NO_EMPTY_LAUNCHES=1
export AGX_K1_FUSED=0   # running / terminal nodes as separate launches: the kernel the roofline is quoted on appears alone
ARGS="--workload $W --no-cpu-baseline --no-batch1 --steps 20 --warmup 3 $@"
rocprofv3 --kernel-trace --stats -d $OUT/trace -- python3 $ROOT/bench.py $ARGS > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -- python3 $ROOT/bench.py $ARGS > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -- python3 $ROOT/bench.py $ARGS > $OUT/write.log 2>&1
if [ "$W" = "humanoid" ]; then
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d $OUT/mfma -- python3 $ROOT/bench.py $ARGS > $OUT/mfma.log 2>&1 || true
fi
cd $ROOT
python scripts/rocpd_stats.py $(find $OUT/trace -name "*.db" | head -1) > $OUT/kernel_stats.txt
python scripts/pmc_summary.py $(find $OUT/fetch -name "*.db" | head -1) > $OUT/pmc_fetch.txt
python scripts/pmc_summary.py $(find $OUT/write -name "*.db" | head -1) > $OUT/pmc_write.txt
if [ "$W" = "humanoid" ]; then python scripts/pmc_summary.py $(find $OUT/mfma -name "*.db" | head -1) > $OUT/pmc_mfma.txt || true; fi
tail -n 3 $OUT/trace.log | cut -c1-400
