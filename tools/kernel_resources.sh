#!/bin/bash
# Register / scratch use of every kernel of one source of the library (device-only compile + ELF notes).
#   tools/kernel_resources.sh mlp.hip [pattern]
set -e
SRC=${1:-mlp.hip}; PAT=${2:-.}
OUT=$(mktemp -d)
EXTRA=""; [ "$SRC" = "mlp.hip" ] && EXTRA="-mllvm -amdgpu-kernarg-preload-count=14"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -x hip $EXTRA --offload-device-only \
  -c "$(dirname "$0")/../curious_amd/csrc/$SRC" -o $OUT/k.co
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$OUT/k.co --output=$OUT/k.elf --unbundle
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $OUT/k.elf | python3 -c "
import re,sys
for b in sys.stdin.read().split('- .agpr_count')[1:]:
    g=lambda k: (re.search(r'\.%s:\s+(\S+)'%k,b) or [None,None])[1]
    n=g('name')
    if n and re.search('$PAT', n): print('%-72s vgpr %s sgpr %s lds %s scratch %s spill %s' % (n[:72], g('vgpr_count'), g('sgpr_count'), g('group_segment_fixed_size'), g('private_segment_fixed_size'), g('vgpr_spill_count')))
"
