"""Register / scratch budget of the headline kernel, read from the gfx950 code objects of the translation units libsrba_hip.so is linked from (no GPU needed).

k_lm_run<SE2_RELPOSE2D> runs two wavefronts per SIMD only while it needs at most 256 VGPRs and no scratch; one more live pose pushes it over and the 30 000-key-frame
benchmark drops from 45 ms to 74 ms per step (measured twice in round 2). The numbers come from the AMDGPU metadata note of the code object."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LLVM = "/opt/rocm/lib/llvm/bin"
SGPR_SPILLS = {}   # kernel -> spilled scalar registers (filled by kernel_resources)


def kernel_resources(tmp_path):
    import __graft_entry__ as ge
    ge.build()
    # one code object per translation unit: read them from the objects the library is linked from (build() keeps them beside it)
    out = {}
    for unit in ("srba_hip", "srba_big", "srba_assemble"):
        obj = os.path.join(ROOT, "srba_amd", "lib", unit + ".o"); fat = str(tmp_path / (unit + ".fat")); co = str(tmp_path / (unit + ".co"))
        subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat], check=True)
        subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat, "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True)
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True, check=True).stdout
        for blk in notes.split("- .agpr_count:")[1:]:
            name = re.search(r"\.name:\s+(\S+)", blk); v = re.search(r"\.vgpr_count:\s+(\d+)", blk); sc = re.search(r"\.private_segment_fixed_size:\s+(\d+)",
                    blk); ss = re.search(r"\.sgpr_spill_count:\s+(\d+)", blk)
            if name and v and sc:
                out[name.group(1)] = (int(v.group(1)), int(sc.group(1))); SGPR_SPILLS[name.group(1)] = int(ss.group(1)) if ss else 0
    return out


@pytest.mark.skipif(not os.path.exists(os.path.join(LLVM, "clang-offload-bundler")), reason="ROCm LLVM tools not found")
def test_headline_kernel_keeps_two_wavefronts_per_simd(tmp_path):
    res = kernel_resources(tmp_path)
    lm = {k: v for k, v in res.items() if "k_lm_runILi" in k}
    assert len(lm) == 9                                                      # one instantiation per model family
    one = lambda i: [v for k, v in lm.items() if "k_lm_runILi%dE" % i in k][0]
    vgpr, scratch = one(0)
    assert vgpr <= 256 and scratch == 0, (vgpr, scratch)
    # no kernel may copy the Batch argument block into scratch (a phase left out of line does that: ~1.4 KB)
    assert all(s <= 256 for _, s in lm.values()), lm
    # private arrays indexed at run time live in scratch (the full-pivot inverse of the landmark blocks did: 48..80 bytes per lane): only the two kernels that
    # run out of their 512 registers (stereo, range-bearing 3D) may use any, for spills
    assert all(one(i)[1] == 0 for i in (0, 1, 2, 4, 5, 7, 8)), lm


@pytest.mark.skipif(not os.path.exists(os.path.join(LLVM, "clang-offload-bundler")), reason="ROCm LLVM tools not found")
def test_round4_instantiations_keep_three_wavefronts_per_simd(tmp_path):
    """k_lm_run_lean (the small size classes of a big relative-pose batch: twelve wavefronts per CU) and k_lm_run2 (two wavefronts per capsule: six workgroups per CU) are sized for three
    wavefronts per SIMD: at most 168 VGPRs, and what they spill stays small (measured: 24 and 31 dwords; a kernel that starts spilling its hot loops shows hundreds)."""
    res = kernel_resources(tmp_path)
    lean = [v for k, v in res.items() if "k_lm_run_leanILi0" in k]; two = [v for k, v in res.items() if "k_lm_run2ILi0" in k]; spec = [v for k, v in res.items() if "k_lm_specILi0" in k]
    assert len(lean) == 1 and len(two) == 1 and len(spec) == 1, (lean, two, spec)
    for vgpr, scratch in lean + two + spec:  # (k_lm_spec: the replicas of a batch of one capsule, same workgroup shape as k_lm_run2)
        assert vgpr <= 168 and scratch <= 256, (vgpr, scratch)


@pytest.mark.skipif(not os.path.exists(os.path.join(LLVM, "clang-offload-bundler")), reason="ROCm LLVM tools not found")
def test_fused_lm_kernels_keep_the_batch_pointers_out_of_vgpr_lanes(tmp_path):
    """Round 5: the ~100 array pointers of the batch record used to arrive as kernel arguments, be hoisted to the top of the LM loop and live in VGPR lanes (221 - 289
    spilled scalars; 1 450 of 12 000 static instructions of k_lm_run_lean were the v_readlane bringing one back). Every phase now reads them through its own laundered reference to a device copy
    of the record (srba_device.hpp lnd): 101 / 87 / 84 spilled scalars. Built without machine-level loop-invariant hoisting (__graft_entry__.build: the literals, lane predicates and addresses of
    every phase were hoisted to the top of the trial loop and held across all of it) they spill 62 / 37 / 48: fenced at 75."""
    res = kernel_resources(tmp_path)
    for tag in ("k_lm_runILi0E", "k_lm_run_leanILi0E", "k_lm_run2ILi0E"):
        ks = [k for k in res if tag in k]; assert len(ks) == 1, (tag, ks)
        assert SGPR_SPILLS[ks[0]] <= 75, (tag, SGPR_SPILLS[ks[0]])


@pytest.mark.skipif(not os.path.exists(os.path.join(LLVM, "clang-offload-bundler")), reason="ROCm LLVM tools not found")
def test_workgroup_kernels_of_the_landmark_families_keep_two_wavefronts_per_simd(tmp_path):
    """Round 5: the SE3 landmark families ran on k_lm_run<3..6> at 504 - 512 registers, one wavefront per SIMD. Their windows now take k_lm_wg<FAM, 128 | 256 | 512>
    (one workgroup per capsule, U_Ap in LDS, tile Cholesky on the matrix cores): at most 256 registers -- two wavefronts per SIMD, i.e. a 512-thread workgroup per CU -- in all twelve
    instantiations, and no scratch for three of the four families (the build without machine-level hoisting removed 60 - 376 bytes per lane; family 3 keeps 8 spilled registers = 36 bytes)."""
    res = kernel_resources(tmp_path)
    wg = {k: v for k, v in res.items() if "k_lm_wgILi" in k}
    assert len(wg) == 12, sorted(wg)
    assert all(v <= 256 and s <= 64 for v, s in wg.values()), wg
    assert all(s == 0 for k, (v, s) in wg.items() if "k_lm_wgILi3E" not in k), wg


@pytest.mark.skipif(not os.path.exists(os.path.join(LLVM, "clang-offload-bundler")), reason="ROCm LLVM tools not found")
def test_dense_factorisation_kernels_do_not_spill_their_broadcasts(tmp_path):
    """k_chol_step / k_chol_panel (srba_big.hpp) hand the pivot column round by v_readlane: 31 SGPR pairs per pivot step. Left to itself the compiler issued them all first and spilled
    them to VGPR lanes (538 SGPR spills in k_chol_panel, 984 in k_chol_step: three instructions per broadcast instead of one, 20 % of the factorisation's time); the loops are written
    so that it cannot (branch-free pivots, update pairs pinned to their broadcast). Keep it so: no SGPR or VGPR spill, no scratch."""
    obj = os.path.join(ROOT, "srba_amd", "lib", "srba_big.o"); fat = str(tmp_path / "u.fat"); co = str(tmp_path / "u.co")
    subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat], check=True)
    subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat, "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True)
    notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True, check=True).stdout
    seen = {}
    for blk in notes.split("- .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        if "k_chol_step" in name or "k_chol_panel" in name:
            seen[name] = tuple(int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1)) for k in ("sgpr_spill_count", "vgpr_spill_count", "private_segment_fixed_size"))
    assert len(seen) == 2 and all(v == (0, 0, 0) for v in seen.values()), seen


@pytest.mark.skipif(not os.path.exists(os.path.join(LLVM, "clang-offload-bundler")), reason="ROCm LLVM tools not found")
def test_fused_normal_equations_kernel_keeps_three_wavefronts_per_simd(tmp_path):
    """k_assemble_se2rel (srba_assemble.hip, observation-major since round 6): three bins of four wavefronts per CU, i.e. three per SIMD -- at most 168 VGPRs, no scratch, in its three Lambda
    instantiations x three bin widths (1, 2, 4 wavefronts)."""
    res = kernel_resources(tmp_path)
    ks = {k: v for k, v in res.items() if "k_assemble_se2rel" in k}
    assert len(ks) == 9, ks
    assert all(v <= 168 and s == 0 for v, s in ks.values()), ks


@pytest.mark.skipif(not os.path.exists(os.path.join(LLVM, "clang-offload-bundler")), reason="ROCm LLVM tools not found")
def test_no_index_is_widened_with_a_stale_high_half_in_the_lm_kernels():
    """The compiler error behind round 2's memory-aperture fault (profiles/r03_fault_root_cause.md: a table index proved non-negative is widened to 64 bits by pairing it with a
    register that no longer holds its zero) as a static fence over the device code that ships: tools/scan_undef_hi.py on all nine k_lm_run instantiations and on every kernel of the
    multi-workgroup path (kb_*). Before Worker::wide was hardened in round 4 (the value is made opaque BEFORE it is widened) the scan reported six such pairs in the stereo kernel."""
    import __graft_entry__ as ge
    ge.build()
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import scan_undef_hi
    dis = scan_undef_hi.disassemble(os.path.join(ROOT, "srba_amd", "lib", "srba_hip.o")); dis_big = scan_undef_hi.disassemble(os.path.join(ROOT, "srba_amd", "lib", "srba_big.o"))
    f_lm, k_lm = scan_undef_hi.scan(dis, "k_lm_runILi"); f_kb, k_kb = scan_undef_hi.scan(dis_big, "kb_"); f_r4, k_r4 = scan_undef_hi.scan(dis, "k_lm_run_lean"); f_2, k_2 = scan_undef_hi.scan(dis,
            "k_lm_run2"); f_sp, k_sp = scan_undef_hi.scan(dis, "k_lm_spec")
    assert len(k_lm) == 9 and len(k_kb) >= 100 and len(k_r4) == 1 and len(k_2) == 1 and len(k_sp) == 1, (len(k_lm), len(k_kb), len(k_r4), len(k_2), len(k_sp))
    assert not f_lm and not f_kb and not f_r4 and not f_2 and not f_sp, (f_lm + f_kb + f_r4 + f_2 + f_sp)[:4]
