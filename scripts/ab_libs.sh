#!/bin/bash
# A/B of library builds on the GPU box: scripts/ab_libs.sh "<workload> <chunks>" <lib.so> [<lib.so> ...]   ("base" = the in-tree library)
# Ablation builds decode garbage on purpose: PCO_BENCH_NO_VERIFY=1 is set for every lib whose name contains "no".
W=$1; shift
set -- $W "$@"; WL=$1; CH=$2; shift; shift
for LIB in "$@"; do
  if [ "$LIB" = base ]; then unset PCO_GFX_LIB; else export PCO_GFX_LIB=$PWD/$LIB; fi
  case "$LIB" in *no*) export PCO_BENCH_NO_VERIFY=1;; *) unset PCO_BENCH_NO_VERIFY;; esac
  python bench.py --workload $WL --chunks $CH --steps ${STEPS:-5} --warmup 1 --no-cpu-baseline --no-others --verify-chunks 64 2>/dev/null | python scripts/bench_brief.py 0.05 "$(basename $LIB) $WL $CH:"
done
