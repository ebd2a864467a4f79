#!/bin/bash
# usage: tools/dev/kstats.sh <object built by uforecon_amd.build, e.g. uforecon_amd/lib/view_transformer.o> [disasm-out.s]
# prints registers / spills / LDS of every gfx950 kernel in the object (code-object metadata); optional disassembly
set -e
LLVM=/opt/rocm/lib/llvm/bin
T=$(mktemp -d)
$LLVM/llvm-objcopy --dump-section .hip_fatbin=$T/fb.bin "$1"
$LLVM/clang-offload-bundler --unbundle --type=o --input=$T/fb.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/k.co
$LLVM/llvm-readelf --notes $T/k.co | grep -E "^\s+\.name:|\.vgpr_count|\.agpr_count|vgpr_spill|sgpr_spill|group_segment_fixed|private_segment_fixed" | \
  awk '/\.name:/{name=$2} /agpr_count/{a=$2} /group_segment/{l=$2} /private_segment/{p=$2} /sgpr_spill/{ss=$2} /vgpr_count/{v=$2} /vgpr_spill/{printf "%-70s vgpr %s agpr %s spill_v %s spill_s %s lds %s scratch %s\n", substr(name,1,70), v, a, $2, ss, l, p}'
if [ -n "$2" ]; then $LLVM/llvm-objdump -d $T/k.co > "$2"; fi
rm -rf $T
