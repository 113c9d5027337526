/*
 * srba_assemble.hpp / srba_assemble.hip -- the normal equations of a batch of relative-pose SE(2) capsules in ONE fused launch that never writes a Jacobian block to HBM
 * (K2 + K5 + K6 of the stepwise API for <SE2, RelativePoses2D>: BASELINE configs[1], the 30k-key-frame graph-SLAM batch).
 *
 * What it computes (reference): every dh_dAp block of the window (jacobians.h:645-744, closed form below), the upper Hessian blocks H_ij = sum_k J_ki^t Lambda J_kj
 * (sparse_hessian_update_numeric.h:26-58), minus_grad_i = sum_k J_ki^t Lambda r_k (compute_minus_gradient.h:20-91) and the initial lambda
 * (optimize_edges.h:366-390: 1e-3 * the largest diagonal entry).
 *
 * How (one wavefront per capsule; two capsules share a workgroup = a bin of 40 KB of LDS packed at upload -- the largest remaining image with the smallest that fits beside
 * it --, ONE launch for the batch, largest bins first; per capsule the chip holds
 * 32 bytes per block, its Hessian blocks, its gradient and the poses of its unknown edges in LDS):
 *   A  every lane owns cb = ceil(n_bp / 64) CONSECUTIVE blocks (the capsule lists its blocks unknown by unknown), four in flight: it reads their packed records (8 bytes:
 *      D pose, unknown slot, residual row, direction, diagonal Hessian block), gathers the keyframe-relative pose D (and the edge's own pose for an inverse edge) and
 *      the residual row, and forms the block in registers. A block of this family is
 *          J = sg * K,  K = [ c  s  x s - y c ;  -s  c  x c + y s ;  0 0 1 ]      (c, s, x, y of D' = D or p (+) D; the device keeps cos / sin next to every pose)
 *      i.e. FOUR numbers and a sign: the numbers go to LDS (four planes: no bank conflicts), the sign stays in the records. J^t Lambda r (gradient) and J^t Lambda J (the term this block adds to the
 	diagonal Hessian block of its unknown) are summed
 *      over the run of blocks of the unknown: serially inside a lane, and -- for a run that crosses lanes -- through ONE prefix scan over the wavefront per capsule;
 *      the lane that holds the last block of the run puts the unknown's gradient and diagonal block into the LDS image.
 *   B  every lane owns ct consecutive OFF-DIAGONAL terms (the list is sorted by Hessian block; its first records were requested before phase A): J1^t Lambda J2 from
 *      the two blocks in LDS (times the product of their signs), the same run sums, blocks into the LDS image.
 *   C  the Hessian blocks and the gradient leave the image as contiguous spans, 16 bytes per lane and request (windows whose image with the Hessian blocks does not fit a
 *      bin store every block from the lane that summed it and keep half the image); lambda guess out.
 * The run sums replace the per-pass segmented reductions of an earlier version of this kernel (54 cross-lane double moves per 64 terms: bound by VALU and
 * LDS-crossbar issue, 0.8 ms for the benchmark batch) and the one-lane-per-Hessian-block form before it (lanes idle behind the longest list, 0.6 ms); DESIGN 4b has the history.
 * HBM sees: 8 B of record, one pose gather (40 B; 80 B for inverse edges) and one residual row (24 B) per block, 8 B per off-diagonal term, 72 B per Hessian block,
 * 24 B per unknown, 96 B of descriptor per capsule. The Jacobian array is not touched: srba_hip_debug_read(1) materialises it on demand with the unfused kernel.
 *
 * Sums are formed in a fixed tree order (not the reference's sequential order): results are reproducible run to run and agree with the oracle to rounding.
 * Capsules whose image exceeds a bin even without its Hessian blocks, or whose indices do not fit the packed records, take k_linearize.
 */
#pragma once

namespace srbadev {

// per wavefront of every bin (workgroup), in launch order. cb / ct: consecutive blocks / off-diagonal terms per lane (ceil(n / 64)); block b lives in LDS slot (b % cb) * 64 + b / cb
struct AsmDesc { int pidx /* -1: this wavefront of the bin has no capsule */, n_bp, n_terms /* off-diagonal */, cb, ct, n_hap, nK, stage /* its Hessian blocks are staged in LDS */,
	lds_off /* bytes: its image inside the bin */, pad; long long o_bp, o_hapt, o_pose /* doubles */, o_edge /* doubles */, o_res /* doubles */, o_hap, o_scal; };
// blk : per Jacobian block, sorted by unknown   lo = (D pose index + 1) | unknown slot << 16 | inverse << 29 | first block of its unknown << 30 | last << 31
//                                               hi = residual row | index of the unknown's diagonal Hessian block << 16
// term: per OFF-DIAGONAL U_Ap term, sorted by Hessian block   lo = LDS slot of block t1 | (the two blocks have opposite directions) << 15 | slot of t2 << 16 ;
	// hi = Hessian block | first term of its block << 30 | last << 31
//       (the terms of a diagonal block pair every Jacobian block of the unknown with itself: they are formed with the blocks, in phase A)
struct AsmTables { const AsmDesc *desc; const unsigned long long *blk, *term; };
constexpr int ASM_WAVES_PER_WG = 2, ASM_BIN_BYTES = 40 * 1024; // four bins per CU (160 KB of LDS), eight wavefronts

// host entry of the translation unit that holds the kernels (srba_assemble.hip): ONE launch, a workgroup per bin
int asm_launch(int lambda_mode /* 0 identity, 1 diagonal, 2 full matrix */, int n_bins, size_t lds_bytes, hipStream_t stream, const Batch &B, const DevParams &prm, const AsmTables &T);

} // namespace srbadev
