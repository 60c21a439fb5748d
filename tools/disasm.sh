#!/bin/bash
# Device ISA of one kernel of a built object: tools/disasm.sh pyramid blur_chain_kernel > /tmp/k.s
L=/opt/rocm/lib/llvm/bin; T=$(mktemp -d); O=affnet_amd/csrc/obj/$1.o
$L/llvm-objcopy --dump-section .hip_fatbin=$T/fat.bin $O $T/copy.o && $L/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/fat.bin --output=$T/dev.co && $L/llvm-objdump -d $T/dev.co | awk -v pat="$2" '/^[0-9a-f]+ </{f = index($0, pat) > 0; if (f) print; next} f'
rm -rf $T
