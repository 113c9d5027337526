"""Static scan of a gfx950 code object for the miscompile behind round 2's memory-aperture violation (profiles/r03_fault_root_cause.md): a 32-bit value (an index
loaded with global_load_dword / ds_read_b32) is widened to 64 bits for address arithmetic WITHOUT its high half being written -- the pair v[N:N+1] is then used while v(N+1) still
holds whatever an earlier, unrelated instruction left there (clang 22 / ROCm 7.2 drops the zero-extension of an int it has proved non-negative and mis-tracks the sub-register
that stood for its zero in kernels with AGPR spill traffic).

Heuristic, linear in program order (branches ignored). A pair v[N:N+1] is STALE when vN was last written by a 32-bit load and v(N+1) was last written BEFORE that load by an
instruction that is not a move of 0 (directly,
        through another register or through an accumulation register the zero was parked in) / the arithmetic shift that produces a sign word; staleness travels through v_mov_b64 (the compiler shuffles such pairs around, which by
itself is harmless: it later writes the high half). A FINDING is a stale pair consumed as a 64-bit integer: the address operand of a global / flat memory instruction or an operand
of v_lshl_add_u64 / v_lshlrev_b64 / v_mad_u64_u32 / v_mad_i64_i32. tests/test_kernel_resources.py runs it over every k_lm_run and kb_ kernel (round 4).
usage: scan_undef_hi.py <libsrba_hip.so | object file> [kernel-name-substring]"""
import os, re, subprocess, sys, tempfile
LLVM = "/opt/rocm/lib/llvm/bin"
_pair = re.compile(r"\bv\[(\d+):(\d+)\]")
LOAD32 = ("global_load_dword ", "global_load_ubyte ", "global_load_sbyte ", "global_load_ushort ", "global_load_sshort ", "ds_read_b32 ", "ds_read_u8 ", "ds_read_u16 ", "flat_load_dword ",
        "scratch_load_dword ", "buffer_load_dword ")
CONSUME = ("v_lshl_add_u64", "v_lshlrev_b64", "v_mad_u64_u32", "v_mad_i64_i32")
MEM = ("global_load", "global_store", "global_atomic", "flat_load", "flat_store", "flat_atomic")


def _vreg(t):
    m = re.match(r"^v(\d+)$", t); return int(m.group(1)) if m else None


def disassemble(path):
    """gfx950 disassembly of the device code bundled in a host object / shared library"""
    tmp = tempfile.mkdtemp(); fat = os.path.join(tmp, "fat.bin"); co = os.path.join(tmp, "co.o")
    subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", path, fat], check=True)
    subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat, "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True)
    return subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], capture_output=True, text=True, check=True).stdout


def scan(dis, want=""):
    """[(kernel, instruction index, instruction, explanation)] for the kernels whose mangled name contains `want`; also returns the names of the kernels scanned"""
    findings = []; scanned = []; kernel = None
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            kernel = m.group(1); lastdef = {}; kind = {}; akind = {}; idx = 0
            if want in kernel: scanned.append(kernel)
            continue
        if kernel is None or want not in kernel or not line.startswith("\t"): continue
        ins = line.strip().split("//")[0].strip(); idx += 1
        op = ins.split()[0] if ins else ""; ops = ins[len(op):]; parts = [p.strip() for p in ops.split(",")]; dst = parts[0] if parts else ""
        def stale(a):
            return kind.get(a + 1) == "stale" or (kind.get(a) == "load32" and lastdef.get(a + 1, -1) < lastdef.get(a, -1) and kind.get(a + 1) not in ("zero", "sign"))
        is_store = op.startswith(("global_store", "flat_store")); is_mem = op.startswith(MEM)
        if is_mem:   # address operand: first operand of a store / atomic without return, second of a load
            addr = parts[0] if (is_store or (op.startswith(("global_atomic", "flat_atomic")) and not _pair.match(parts[1] if len(parts) > 1 else ""))) else (parts[1] if len(parts) > 1 else "")
            m2 = _pair.match(addr)
            if m2 and int(m2.group(2)) == int(m2.group(1)) + 1 and stale(int(m2.group(1))):
                findings.append((kernel, idx, ins, "address pair v[%s:%s] with a stale high half" % (m2.group(1), m2.group(2))))
        elif op.startswith(CONSUME):
            for a, b in _pair.findall(",".join(parts[1:])):
                a, b = int(a), int(b)
                if b == a + 1 and stale(a): findings.append((kernel, idx, ins, "64-bit integer operand v[%d:%d] with a stale high half" % (a, b)))
        # definitions
        if op.startswith("v_accvgpr_write") and len(parts) > 1: akind[dst] = "zero" if (parts[1] == "0" or kind.get(_vreg(parts[1])) == "zero") else "other"
        dm = _pair.match(dst) if not is_store else None
        if dm and not op.startswith("v_cmp"):
            lo, hi = int(dm.group(1)), int(dm.group(2))
            if op.startswith("v_mov_b64") and len(parts) > 1 and _pair.match(parts[1]) and hi == lo + 1 and stale(int(_pair.match(parts[1]).group(1))):
                lastdef[lo] = lastdef[hi] = idx; kind[lo] = "other"; kind[hi] = "stale"
            else:
                wz = op.startswith("v_mov_b64") and len(parts) > 1 and parts[1] == "0"
                for r in range(lo, hi + 1): lastdef[r] = idx; kind[r] = "zero" if wz else "wide"
        else:
            dm = re.match(r"^v(\d+)$", dst) if not is_store else None
            if dm and not op.startswith("v_cmp"):
                r = int(dm.group(1)); lastdef[r] = idx
                if op.startswith(tuple(x.strip() for x in LOAD32)) and (op + " ").startswith(LOAD32): kind[r] = "load32"
                elif op.startswith("v_mov_b32") and len(parts) > 1 and (parts[1] == "0" or kind.get(_vreg(parts[1])) == "zero"): kind[r] = "zero"
                elif op.startswith("v_accvgpr_read") and len(parts) > 1 and akind.get(parts[1]) == "zero": kind[r] = "zero"   # a zero kept in (or spilled to) an accumulation register"
                elif op.startswith("v_ashrrev_i32") and len(parts) > 1 and parts[1] == "31": kind[r] = "sign"
                else: kind[r] = "other"
    return findings, scanned


if __name__ == "__main__":
    f, names = scan(disassemble(sys.argv[1]), sys.argv[2] if len(sys.argv) > 2 else "")
    for k, i, ins, why in f: print("%s: [%d] %s   <- %s" % (k[:60], i, ins, why))
    print("kernels scanned: %d, findings: %d" % (len(names), len(f)))
