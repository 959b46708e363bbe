export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/prof_auto
rocprofv3 --kernel-trace --stats -d /tmp/prof_auto -o kt -- python /root/repo/bench.py --steps 3 --warmup 1 --chunks 2048 --workload c2auto --no-cpu-baseline > /tmp/auto_run.log 2>&1
DB=$(find /tmp/prof_auto -name "*.db" | head -1)
python3 /root/repo/scripts/rocprof_summary.py "$DB" /tmp/auto_stats.md "auto" > /dev/null
grep pcogfx /tmp/auto_stats.md | cut -c1-110
tail -1 /tmp/auto_run.log | cut -c1-200
