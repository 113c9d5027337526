/*
 * srba_assemble.hpp / srba_assemble.hip -- the normal equations of a batch of relative-pose SE(2) capsules in ONE fused launch that never writes a Jacobian block to HBM
 * (K2 + K5 + K6 of the stepwise API for <SE2, RelativePoses2D>: BASELINE configs[1], the 30k-key-frame graph-SLAM batch).
 *
 * What it computes (reference): every dh_dAp block of the window (jacobians.h:645-744, closed form below), the upper Hessian blocks H_ij = sum_k J_ki^t Lambda J_kj
 * (sparse_hessian_update_numeric.h:26-58), minus_grad_i = sum_k J_ki^t Lambda r_k (compute_minus_gradient.h:20-91) and the initial lambda
 * (optimize_edges.h:366-390: 1e-3 * the largest diagonal entry).
 *
 * How (round 6: OBSERVATION-major). Every Hessian term pairs two Jacobian blocks of ONE observation row, and a row of this family has at most three blocks (the unknown
 * edges on the spanning-tree path between its two key-frames, tree depth 3: 49 % of the rows of the benchmark batch have none, 6 % one, 33 % two, 8 % three). So a LANE
 * owns an observation row: it gathers the row's residual and the key-frame-relative pose D of each of its blocks, forms the blocks in registers -- a block of this family is
 *          J = sg * K,  K = [ c  s  x s - y c ;  -s  c  x c + y s ;  0 0 1 ]      (c, s, x, y of D' = D or p (+) D; the device keeps cos / sin next to every pose)
 * i.e. FOUR numbers and a sign -- and adds J^t Lambda r, the three self products J_a^t Lambda J_a and the up to three cross products J_a^t Lambda J_b into the image of
 * its capsule in LDS with ds_add_f64. A wavefront per capsule, `rounds` = ceil(rows with blocks / 64) passes; the records of the next pass and the gathers of the one after
 * travel under the arithmetic of the current one. The image is the OUTPUT only (72 B per Hessian block + gradient + the poses of the unknown edges: 11 KB for the median
 * window), not the Jacobian blocks: twelve capsules fit a CU instead of eight, and nothing is handed from one lane to another (the block-major kernel of rounds 3-5 kept 32 B per
 * block in LDS for a second phase that paired blocks across lanes, ran two prefix scans per capsule and was bound by capsules-resident x latency: 0.37 ms, HISTORY 4b).
 * One wavefront alone adds to an image and the LDS serves the lanes of an instruction in a fixed order: sums are reproducible run to run (tools/probes/lds_atomic_rate.hip
 * measures both: 2.7 ns per conflict-free wavefront instruction at eight wavefronts per CU, x (lanes on one address) within a group of 16 lanes). The host deals the rows of a
 * capsule to the lanes so that rows on the same unknowns land in different 16-lane groups.
 * ONE launch: capsules are packed into bins (workgroups of four wavefronts sharing 52 KB of LDS); the bins of the largest windows -- few capsules for their LDS -- are spread over
 * the first 70 % of the launch with bins of small windows between them (asm_plan).
 * HBM sees: 16 B of record per row with blocks, its residual row (24 B), one pose gather (32 of 40 B) per block, 72 B per Hessian block, 24 B per unknown out.
 * The Jacobian array is not touched: srba_hip_debug_read(1) materialises it on demand with the unfused kernel.
 * Capsules whose image exceeds a bin, with a row of more than three blocks or whose indices do not fit the packed records take k_linearize.
 */
#pragma once
#include <cstdint>

namespace srbadev {

// per wavefront of every bin (workgroup), in launch order
struct AsmDesc { int pidx /* -1: this wavefront of the bin has no capsule */, n_rec /* row records (a multiple of 16) */, n_hap, nK, lds_off /* bytes: its image inside the bin */, pad;
	long long o_rec /* records */, o_pose /* doubles */, o_edge /* doubles */, o_res /* doubles */, o_hap /* blocks */, o_scal /* doubles */, o_unk /* unknowns: hap_diag */; };
// one observation row with m <= 3 blocks (a = 0..2 in block order = ascending unknown), 16 bytes:
//   w0 = (D pose index of block 0) + 1 | (block 1) << 14 | m << 28                           (14 bits each; 0: D = identity)
//   w1 = (block 2) + 1 | residual row << 14 (11 bits) | unknown slot of block 0 << 25 (7 bits)
//   w2 = unknown of block 1 | of block 2 << 7 | Hessian block of cross term (0,1) << 14 (11 bits; 0x7ff: the plan has no such term) | flags << 25
//        flags: bit a = block a belongs to an edge taken in its inverse direction (its sign; a cross term carries the product of the two)
//   w3 = Hessian block of cross term (0,2) | of cross term (1,2) << 11
// (the diagonal Hessian block of an unknown comes from a table in LDS). An all-zero record (m = 0) is a lane without a row.
struct AsmRec { uint32_t w[4]; };
struct AsmTables { const AsmDesc *desc; const AsmRec *rec; };
constexpr int ASM_MAX_WPW = 4;          // wavefronts (capsules) per bin: 1, 2 or 4 (SRBA_HIP_ASM_WPW, default 4)
constexpr int ASM_DEFAULT_BIN_KB = 52;  // three bins per CU: the LDS is handed out in granules, 3 x 53 KB does not fit the 160 KB of a CU (SRBA_HIP_ASM_BIN_KB)
constexpr int ASM_MIX_F = 50, ASM_MIX_S = 70; // dispatch order (asm_plan): the F % largest bins spread over the first S % of the launch (SRBA_HIP_ASM_MIX="F,S"; "0,0": largest first)
constexpr int ASM_MAX_NK = 127, ASM_MAX_POSE = 16382, ASM_MAX_ROW = 2047, ASM_MAX_HAP = 2046; // what the record fields hold

// LDS image of a capsule: Hessian blocks | gradient | poses of the unknown edges (5 doubles each) | diagonal block of every unknown (int), rounded to 64 bytes
inline size_t asm_image_bytes(int n_hap, int nK) { return ((size_t)8 * (9 * (size_t)n_hap + 8 * (size_t)nK + 2) + 4 * (size_t)nK + 63) & ~(size_t)63; }
// room for the records of a capsule (an upper bound known before they are built)
inline long long asm_rec_room(int n_obs, int n_bp) { const int a = n_obs < n_bp ? n_obs : n_bp; return 16LL * ((a + 15) / 16) + 32; } // (rows of three, two, one blocks are padded to 16 each)

} // namespace srbadev
struct srba_problem_capsule;
namespace srbadev {
struct Batch; struct DevParams;
// host: the packed records of one capsule into dst (asm_rec_room of them, cleared by the caller); returns the number of records used (a multiple of 16), 0 if the capsule does not fit the kernel
int asm_pack(const srba_problem_capsule &k, AsmRec *dst);
// host: launch geometry (environment or defaults)
void asm_config(int &waves_per_bin, int &bin_bytes);
// host: packs the capsules into bins. dsc[p]: descriptor of capsule p (lds_off unset), rounds[p] == 0: does not fit the kernel; cap: largest image taken. Writes waves_per_bin descriptors
// per bin into out (room: ASM_MAX_WPW * n), the capsules left over into rest[0 .. n_rest); returns the number of bins
int asm_plan(int n, const AsmDesc *dsc, const int *rounds, size_t cap, int waves_per_bin, int bin_bytes, AsmDesc *out, int32_t *rest, int &n_rest);
// host: sizeof(Batch) * 10007 + sizeof(DevParams) as this translation unit sees them (srba_hip_create compares: srba_ctx.hpp)
unsigned long long asm_layout_signature();
// host entry of the translation unit that holds the kernels (srba_assemble.hip): ONE launch, a workgroup per bin
int asm_launch(int lambda_mode /* 0 identity, 1 diagonal, 2 full matrix */, int waves_per_bin, int n_bins, size_t lds_bytes, hipStream_t stream, const Batch &B, const DevParams &prm,
	const AsmTables &T);

} // namespace srbadev
