#!/bin/bash
# Round profile on the GPU box: bench line, rocprofv3 kernel stats, and HBM traffic counters (separate PMC passes).
# usage: scripts/profile_round.sh <tag> [chunks]     outputs under gpurun_out/<tag>/
TAG=${1:-r02}; CH=${2:-8192}; WL=${3:-c2}
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp
python $ROOT/bench.py --workload $WL --steps 10 --warmup 2 --chunks $CH > $OUT/bench.json 2> $OUT/bench.err
tail -c 3000 $OUT/bench.json
rm -rf /tmp/prof_kt; rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt -- python $ROOT/bench.py --workload $WL --steps 5 --warmup 1 --chunks $CH --no-cpu-baseline --no-others --verify-chunks 0 > $OUT/kt_run.log 2>&1
DB=$(find /tmp/prof_kt -name "*.db" | head -1)
python3 $ROOT/scripts/rocprof_summary.py "$DB" $OUT/kernel_stats.md "rocprofv3 --kernel-trace --stats -- python bench.py --workload $WL --steps 5 --warmup 1 --chunks $CH --no-cpu-baseline" > /dev/null
head -30 $OUT/kernel_stats.md
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/prof_$C; rocprofv3 --pmc $C -d /tmp/prof_$C -o pmc -- python $ROOT/bench.py --workload $WL --steps 2 --warmup 1 --chunks $CH --no-cpu-baseline --no-others --verify-chunks 0 > $OUT/pmc_$C.log 2>&1
  DB=$(find /tmp/prof_$C -name "*.db" | head -1)
  python3 $ROOT/scripts/pmc_summary.py "$DB" > $OUT/pmc_$C.txt
  cat $OUT/pmc_$C.txt
done
python3 $ROOT/scripts/make_traffic_json.py $OUT $CH $WL $OUT/traffic.json $OUT/pmc_hbm_traffic.txt
