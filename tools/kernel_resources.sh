#!/bin/bash
# VGPRs / SGPRs / scratch / LDS / occupancy of every kernel in the built library (no GPU needed):
#   bash tools/kernel_resources.sh [library] [filter]
LIB=${1:-metal-pathtracer-arm64_amd/libptr_hip.so}
FILTER=${2:-.}
LLVM=/opt/rocm/lib/llvm/bin
TMP=$(mktemp -d)
$LLVM/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$LIB --output=$TMP/dev.co --unbundle 2>/dev/null || \
  $LLVM/llvm-objcopy --dump-section .hip_fatbin=$TMP/fat.bin $LIB && $LLVM/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$TMP/fat.bin --output=$TMP/dev.co --unbundle
$LLVM/llvm-readelf --notes $TMP/dev.co | python3 -c "
import sys, re
txt = sys.stdin.read()
for m in re.finditer(r'\.name:\s+(\S+).*?(?=\n\s+- \.a|\Z)', txt, re.S):
    blk = m.group(0)
    def f(k):
        r = re.search(r'\.' + k + r':\s+(\d+)', blk)
        return int(r.group(1)) if r else -1
    name = m.group(1)
    if not re.search(sys.argv[1], name): continue
    print('%-90s vgpr %3d  sgpr %3d  scratch %5d  lds %6d  spill v%d s%d' % (name[:90], f('vgpr_count'), f('sgpr_count'), f('private_segment_fixed_size'), f('group_segment_fixed_size'), f('vgpr_spill_count'), f('sgpr_spill_count')))
" "$FILTER" | c++filt | sort
rm -rf $TMP
