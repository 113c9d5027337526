#!/bin/bash
# usage: tools/quick_build.sh <name> [-Dflag ...]   -> srba_amd/lib/variants/libsrba_hip_<name>.so : the headline family only (compiles in seconds), for A/B runs with tools/r4_variants.sh
n=$1; shift; mkdir -p /tmp/qb srba_amd/lib/variants
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -pthread -Wno-unused-value -mllvm -disable-machine-licm -mllvm -sink-insts-to-avoid-spills -DSRBA_ONLY_RELPOSE2D "$@" \
	-c srba_amd/csrc/srba_hip.hip -o /tmp/qb/srba_hip_$n.o 2>&1 | grep -E "error|warning: .*spill"
hipcc --offload-arch=gfx950 -fPIC -shared -pthread /tmp/qb/srba_hip_$n.o srba_amd/lib/srba_big.o srba_amd/lib/srba_assemble.o -o srba_amd/lib/variants/libsrba_hip_$n.so && echo "built $n"
