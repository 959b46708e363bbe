#!/bin/bash
# PMC pass over a short bench run: per-kernel dynamic instruction mix and wait breakdown.
# usage: scripts/pmc_run.sh <chunks> <out-subdir> [counters...]
CH=${1:-2048}; OUT=${2:-pmc}; shift; shift
CTRS=${@:-SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY}
export TMPDIR=/tmp
ROOT=$PWD
mkdir -p $ROOT/gpurun_out/$OUT
cd /tmp
rm -rf /tmp/pmc_out
rocprofv3 --pmc $CTRS -d /tmp/pmc_out -o pmc -- python $ROOT/bench.py --steps 1 --warmup 1 --chunks $CH --no-cpu-baseline --no-others --verify-chunks 0 ${BENCH_ARGS} > $ROOT/gpurun_out/$OUT/run.log 2>&1
DB=$(find /tmp/pmc_out -name "*.db" | head -1)
python3 $ROOT/scripts/pmc_summary.py "$DB" > $ROOT/gpurun_out/$OUT/pmc.txt
cat $ROOT/gpurun_out/$OUT/pmc.txt
