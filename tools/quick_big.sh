#!/bin/bash
# usage: tools/quick_big.sh <name> [-Dflag ...] -> srba_amd/lib/variants/libsrba_hip_<name>.so with srba_big.hip rebuilt (20 s; every family) beside the product's other objects
n=$1; shift; mkdir -p /tmp/qb srba_amd/lib/variants
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -pthread -Wno-unused-value -mllvm -disable-machine-licm -mllvm -sink-insts-to-avoid-spills "$@" \
	-c srba_amd/csrc/srba_big.hip -o /tmp/qb/srba_big_$n.o 2>&1 | grep -E "error|warning: .*spill"
hipcc --offload-arch=gfx950 -fPIC -shared -pthread srba_amd/lib/srba_hip.o /tmp/qb/srba_big_$n.o srba_amd/lib/srba_assemble.o -o srba_amd/lib/variants/libsrba_hip_$n.so && echo "built $n"
