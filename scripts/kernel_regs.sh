#!/bin/bash
# Registers / LDS / scratch of the kernels in a built library: scripts/kernel_regs.sh [name-substring] [lib]
# (reads the code object's metadata notes: .vgpr_count, .agpr_count, .sgpr_count, .group_segment_fixed_size, .private_segment_fixed_size = scratch bytes)
PAT=${1:-}; LIB=${2:-pcodec_amd/libpco_gfx.so}
B=/opt/rocm/lib/llvm/bin; T=$(mktemp -d)
$B/llvm-objcopy --dump-section .hip_fatbin=$T/fat.bin $LIB 2>/dev/null
$B/clang-offload-bundler --unbundle --type=o --input=$T/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/dev.co
$B/llvm-readelf --notes $T/dev.co | python3 -c "
import sys,re
txt=sys.stdin.read()
pat=sys.argv[1]
for blk in re.split(r'\n\s+- \.agpr_count', txt)[1:]:
    blk='.agpr_count'+blk
    g=lambda k:(re.search(r'\.'+k+r':\s+(\S+)',blk) or [None,'?'])[1]
    name=g('name')
    if pat and pat not in name: continue
    print(f\"{name[:90]:90s} vgpr {g('vgpr_count'):>4s} agpr {g('agpr_count'):>4s} sgpr {g('sgpr_count'):>4s} lds {g('group_segment_fixed_size'):>6s} scratch {g('private_segment_fixed_size'):>5s}\")
" "$PAT"
rm -rf $T
