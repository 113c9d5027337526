#!/bin/bash
# (CPU) usage: tools/asm_regs.sh [-Dflags]  -- registers / spills / scratch of every k_assemble_se2rel instantiation of srba_assemble.hip built with the given flags
cd "$(dirname "$0")/.."; hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value "$@" -S --cuda-device-only srba_amd/csrc/srba_assemble.hip -o /tmp/asm_regs.s 2>&1 | grep -v "argument unused"
python3 - <<'PY'
import re
t=open('/tmp/asm_regs.s').read()
for m in re.finditer(r'\.name:\s+(\S+)\n(?:.*\n)*?\s+\.sgpr_count:\s+(\d+)\n\s+\.sgpr_spill_count:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)', t):
    print(m.group(1)[:60], "sgpr", m.group(2), "spilled", m.group(3), "vgpr", m.group(4), "spilled", m.group(5))
PY
