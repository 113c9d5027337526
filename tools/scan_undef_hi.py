"""Static scan of a gfx950 code object for the miscompile behind round 2's memory-aperture violation: a 32-bit value (an index loaded with global_load_dword /
ds_read_b32) is widened to 64 bits for address arithmetic WITHOUT its high half being written -- the pair v[N:N+1] is then read while v(N+1) still holds whatever an earlier,
unrelated instruction left there (clang 22 / ROCm 7.2 drops the zero-extension of an int it has proved non-negative and treats the high half as don't-care).
Heuristic, linear in program order (branches ignored): a finding is a read of the PAIR v[N:N+1] by a 64-bit integer instruction where vN was last written by a 32-bit load and
v(N+1) was last written BEFORE that load by an instruction that is not a move of 0 / an arithmetic shift producing the sign.
usage: scan_undef_hi.py <libsrba_hip.so> [kernel-name-substring]"""
import re, subprocess, sys, tempfile, os
LLVM = "/opt/rocm/lib/llvm/bin"
so = sys.argv[1]; want = sys.argv[2] if len(sys.argv) > 2 else ""
tmp = tempfile.mkdtemp(); fat = os.path.join(tmp, "fat.bin"); co = os.path.join(tmp, "co.o")
subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", so, fat], check=True)
subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat, "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True)
dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], capture_output=True, text=True, check=True).stdout
reg = re.compile(r"\bv(\d+)\b"); pair = re.compile(r"\bv\[(\d+):(\d+)\]")
LOAD32 = ("global_load_dword ", "global_load_ubyte ", "global_load_sbyte ", "global_load_ushort ", "ds_read_b32 ", "ds_read_u8 ", "flat_load_dword ", "scratch_load_dword ", "buffer_load_dword ")
WIDE_INT = ("v_lshl_add_u64", "v_lshlrev_b64", "v_mad_u64_u32", "v_mad_i64_i32", "v_mov_b64", "v_add_co", "v_lshl_add_u64", "global_load", "global_store", "global_atomic", "flat_load", "flat_store")
findings = 0; kernel = None
for line in dis.splitlines():
    m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
    if m:
        kernel = m.group(1); lastdef = {}; kind = {}; idx = 0; continue
    if kernel is None or want not in kernel or not line.startswith("\t"): continue
    ins = line.strip().split("//")[0].strip(); idx += 1
    op = ins.split()[0] if ins else ""
    ops = ins[len(op):]
    parts = [p.strip() for p in ops.split(",")]
    dst = parts[0] if parts else ""
    # reads: every operand but the first (stores / cmp have no dst but this heuristic only needs pair reads)
    srcs = ",".join(parts[1:]) if not (op.startswith("global_store") or op.startswith("flat_store") or op.startswith("v_cmp")) else ops
    if op.startswith(WIDE_INT) or op.startswith("v_mov_b64"):
        for a, b in pair.findall(srcs):
            a, b = int(a), int(b)
            if b != a + 1: continue
            if kind.get(a) == "load32" and lastdef.get(b, -1) < lastdef.get(a, -1) and kind.get(b) not in ("zero", "sign"):
                findings += 1; print("%s: [%d] %s   <- v%d from a 32-bit load at [%d], v%d last written at [%d] (%s)" % (kernel[:60], idx, ins, a, lastdef[a], b, lastdef.get(b, -1), kind.get(b)))
    # defs
    dm = pair.match(dst)
    if dm:
        wz = op.startswith("v_mov_b64") and parts[1:] and parts[1] == "0"
        for r in range(int(dm.group(1)), int(dm.group(2)) + 1): lastdef[r] = idx; kind[r] = "zero" if wz else "wide"
    else:
        dm = re.match(r"^v(\d+)$", dst)
        if dm:
            r = int(dm.group(1)); lastdef[r] = idx
            if op.startswith(LOAD32.__class__(x.strip() for x in LOAD32)): kind[r] = "load32"
            elif op.startswith("v_mov_b32") and parts[1:] and parts[1] == "0": kind[r] = "zero"
            elif op.startswith("v_ashrrev_i32") and parts[1:] and parts[1] == "31": kind[r] = "sign"
            else: kind[r] = "other"
print("findings:", findings)
