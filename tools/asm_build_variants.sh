#!/bin/bash
# (CPU) usage: tools/asm_build_variants.sh name:"-Dflags" ...  -- srba_assemble.hip alone with extra flags, linked with the built objects of the other units into
# srba_amd/lib/variants/libsrba_hip_<name>.so (seconds per variant); tools/asm_variants.sh <name> ... runs tools/diag_assemble.py per variant on the GPU box
cd "$(dirname "$0")/.."; mkdir -p srba_amd/lib/variants
for v in "$@"; do n=${v%%:*}; f=${v#*:}; hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -pthread -Wno-unused-value $f -c srba_amd/csrc/srba_assemble.hip -o /tmp/asm_$n.o 2>&1 | grep -v "argument unused"
  hipcc --offload-arch=gfx950 -fPIC -shared -pthread srba_amd/lib/srba_hip.o srba_amd/lib/srba_big.o /tmp/asm_$n.o -o srba_amd/lib/variants/libsrba_hip_$n.so && echo "built $n ($f)"; done
