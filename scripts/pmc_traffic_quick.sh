#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of a short run (2 PMC passes), printing the kernels above 0.5 GB: scripts/pmc_traffic_quick.sh <workload> <chunks> <tag>
WL=${1:-c2}; CH=${2:-8192}; TAG=${3:-tq}
export TMPDIR=/tmp; ROOT=$PWD; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/prof_$C; rocprofv3 --pmc $C -d /tmp/prof_$C -o pmc -- python $ROOT/bench.py --workload $WL --steps 2 --warmup 1 --chunks $CH --no-cpu-baseline --no-others --verify-chunks 0 > $OUT/pmc_$C.log 2>&1
  DB=$(find /tmp/prof_$C -name "*.db" | head -1)
  python3 $ROOT/scripts/pmc_summary.py "$DB" > $OUT/pmc_$C.txt
done
python3 $ROOT/scripts/make_traffic_json.py $OUT $CH $WL $OUT/traffic.json $OUT/pmc_hbm_traffic.txt
grep -E "fetch +[0-9.]+ GB" $OUT/pmc_hbm_traffic.txt | awk '$3 > 0.5 || $6 > 0.5'
