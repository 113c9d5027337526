#!/bin/bash
# Register need of every phase of the workgroup landmark kernel, each compiled as a kernel of its own (SRBA_PROBE_KERNELS in srba_hip.hip): usage tools/probe_phases.sh [SRBA_SE3_STEREO]
fam=${1:-SRBA_SE3_STEREO}; mkdir -p /tmp/qb
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -pthread -Wno-unused-value -DSRBA_ONLY_FAMILY=$fam -DSRBA_PROBE_KERNELS=$fam -c srba_amd/csrc/srba_hip.hip -o /tmp/qb/probe.o 2>&1 | grep -E "error" | head
t=$(mktemp -d); objcopy -O binary --only-section=.hip_fatbin /tmp/qb/probe.o $t/u.fat
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$t/u.fat --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$t/u.co
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $t/u.co | python3 -c '
import sys, re
for blk in sys.stdin.read().split("- .agpr_count:")[1:]:
    g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
    if "kp_" in g("name") or "k_lm_wg" in g("name") or "k_solve_wg" in g("name"):
        print("%-60s vgpr %3s sgpr-spill %4s vgpr-spill %4s scratch %5s" % (g("name")[:60], g("vgpr_count"), g("sgpr_spill_count"), g("vgpr_spill_count"),
                g("private_segment_fixed_size")))
' | c++filt
rm -rf $t
