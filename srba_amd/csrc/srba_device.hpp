/*
 * srba_device.hpp -- gfx950 device code of the SRBA local-optimisation hot path (FP64 throughout).
 *
 * One problem capsule (= one reference optimize_edges() call, include/srba/impl/optimize_edges.h:44-793) is owned by ONE wavefront
 * (a 64-lane workgroup: no cross-wave barriers).  Work items inside the capsule (spanning-tree pairs, Jacobian blocks, observations,
 * Hessian blocks, landmarks) are spread over the lanes; reductions are DPP trees (deterministic, no FP64 atomics).  The system of the
 * LM step is a block-sparse LL^t on 3x3 blocks held in LDS (symbolic factorisation on the host); capsules too large for that run on
 * the multi-workgroup path of srba_big.hpp with a dense blocked factorisation in HBM.
 *
 * Device restatement of the reference hot loops:
 *   K1  phase_spantree      impl/spantree_update_numeric.h:19-81
 *   K2  jac_dh_dp           impl/jacobians.h:207-354 (+ :364-494 SE3 points, :501-634 SE2 points, :645-744 SE2 relative poses)
 *   K3  jac_dh_df           impl/jacobians.h:888-1013
 *   K4  phase_residuals     impl/reprojection_residuals.h:16-81, RbaEngine.h:810-813 (pseudo-Huber)
 *   K5  phase_gradient      impl/compute_minus_gradient.h:20-91
 *   K6  phase_hessian       impl/sparse_hessian_update_numeric.h:22-60, srba_options_noise.h:44-71,102-131
 *   K7,K8,K10 schur_*       impl/schur.h:180-311
 *   K9  sp_factor_fsub_rows / sp_bsub_rows   impl/lev-marq_solvers.h:80-187,279-381,474-568 (all three solvers solve the same SPD
 *                                 system; here always by block-sparse LL^t -- "not positive definite" == a pivot <= 0 in every variant)
 *   K11,K12 phase_apply / phase_restore + LM control   impl/optimize_edges.h:361-696
 * The SE2 Jacobians are evaluated in closed form (products J0*J1*J2 of jacobians.h:720-736 multiplied out), which needs one
 * sincos instead of four; results agree with the reference formula to rounding.
 */
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include "../../include/srba_hip.h"

#define SRBA_WG 64   /* one wavefront per capsule */

namespace srbadev {

// ------------------------------------------------------------------------------------------------ problem descriptor
// A pointer into device (global) memory that says so in its type. The array pointers of `Batch` are read from a copy of the record in device memory (round 5: lnd below); a plain `double *`
// loaded from memory is a GENERIC pointer to the compiler -- flat_load / flat_store instead of global_load / global_store, and every flat operation also counts as an LDS operation for
// s_waitcnt. The wrapper holds an address_space(1) pointer and converts to the plain pointer on use: the conversion is an address-space cast the optimiser traces back, so the accesses are
// global again. Same size and layout as a pointer, trivially copyable (the host fills the record with memset / assignments and copies it as bytes).
template <class T> struct gptr {
	typedef __attribute__((address_space(1))) T *gp_t;
	gp_t p;
	__host__ __device__ __forceinline__ operator T *() const { return (T *)p; }
	__host__ __device__ __forceinline__ gptr &operator=(T *q) { p = (gp_t)q; return *this; }
	__host__ __device__ __forceinline__ T *operator+(long long i) const { return (T *)p + i; }
	__host__ __device__ __forceinline__ T &operator[](long long i) const { return ((T *)p)[i]; }
};
struct ProbDesc {
	int n_edges, nK, nF, n_klm, n_pairs, n_obs, n_valid, n_bp, n_bf, n_hap, n_hf, n_hapf, n_sch, n_req, n_sys, n_scal, nb, nnzoff, n_hapt /* U_Ap terms */;
	// element offsets into the batch-wide arrays
	long long o_edge, o_unk, o_ulm, o_klm, o_pair, o_ppoff, o_path, o_obs, o_valid, o_bp, o_colp, o_bf, o_colf;
	int n_fill; long long o_spfill; // blocks of the factor that no Hessian block maps onto (fill-in): the only ones the assembly has to zero
	int n_need, need_flat; // pairs whose pose is re-evaluated inside the LM loop (pair_needed != 0); need_flat: all their paths have <= 4 edges (need_rec usable)
	long long o_hap, o_hapoff, o_hapt, o_hf, o_hfoff, o_hft, o_hapf, o_hapfoff, o_hapft, o_sch, o_lmoff, o_req, o_scal, o_yw, o_spcol, o_sprow, o_spitem, o_spperm, o_dense;
	int hapt_split; // two wavefronts per capsule: the second one takes the U_Ap terms from this index on (the first term of a Hessian block at or after the middle of the list: no block is summed by
		// both)
	int n_items, aligned; // block updates per factorisation; 1 if every Hessian block maps onto whole 3x3 blocks (always, except L==2 with the no-Schur solver)
	int n_panel; long long o_ptab; // ... in n_panel passes over block ranges that fit the LDS of the workgroup (one for most windows): Batch::ptab holds, from o_ptab, the n_panel + 1 block bounds,
		// then the bounds of the panels' K6 terms, then those of their Schur terms (both lists sorted by panel first)
	int hs_lds; long long o_hapo, o_schl; // workgroup path, U_Ap accumulators in LDS (Solver::phase_hessian_lds / schur_reduce_lds): the U_Ap terms sorted by observation {t1, t2, block}
		// from o_hapo (x3), the Schur terms sorted by landmark {lm, b1, b2, block | edge << 16 | diagonal << 31} from o_schl (x4)
	int n_hrec, hap_chunked /* unused since the workgroup path sums U_Ap in LDS */; long long o_hrec; // K6 work records {U_Ap block, first term, end term} (Batch::hap_rec), one per block
	int n_vb; long long o_vb; // multi-workgroup path, Schur reduction with a wavefront per U_Ap block (kb_schur_reduce_wave): n_vb work records {first term, end term, block, 0} from o_vb (x4) in
		// Batch::sch_vb, longest list first; the terms themselves as packed records {landmark, W block 1, W block 2, Y slot} from o_sch (x4) in Batch::sch_rec
	int od_ok; // every observation of the capsule appears in at most three residual rows: Batch::od_tab holds one record per DISTINCT observation (Worker::residuals_distinct)
	int dense_in_lds, dense_blocks; // dense_blocks: the LDS image holds ALL blocks of the lower triangle (column-major), no symbolic structure (mid-size, nearly dense systems)
};

struct Batch {
	int n_prob; int max_lds_doubles; int hess_terms; int dense_left; // hess_terms: the fused kernel accumulates U_Ap term-parallel in LDS;
		// dense_left: left-looking sweeps on the HBM-resident dense layout
	gptr<const ProbDesc> desc; gptr<const int> order; // order: capsule indices grouped by LDS size class (one launch per class)
	// inputs
	gptr<const double> edge0, ulm0, klm, obs_z;
	gptr<const int> pair_path_off, path_edge, obs_pose, obs_lm, obs_valid;
	gptr<const int> obs_rec; // per observation {-2: identity | -1, pose index: a pose of the table no trial changes | 0 / 1 = which pose of its pair, four path entries (edge << 1 | inverse,
		// -1 = none)}: phase_residuals_fused
	gptr<const int> bp_col, bp_res, bp_A, bp_D, bp_lm, colp_off, bf_col, bf_res, bf_pose, colf_off;
	gptr<const int> hap_i, hap_j, hap_term_off, hap_t1, hap_t2, hap_tblk /* block of every U_Ap term */, hf_i, hf_j, hf_term_off, hf_t1, hf_t2;
	gptr<const int> hapf_i, hapf_j, hapf_term_off, hapf_t1, hapf_t2, hap_diag, hf_diag;
	gptr<const int> sch_term_off, sch_b1, sch_b2, sch_lm, sch_yw, sch_tblk /* U_Ap block of every Schur term */, sch_vb, sch_rec /* ProbDesc::n_vb */, lm_hapf_off, lm_hapf_idx, req_idx, need_idx, need_rec;
		// need_rec: per needed pair {pair, 4 path entries (edge<<1|inv, -1 = none)}
	gptr<const unsigned char> pair_needed, bp_normal;
	gptr<const int> sp_fill; // unified block indices (diag k -> k, off-diagonal i -> nb+i)
	gptr<const int> hapo, schl, ptab; // see ProbDesc::hs_lds, n_panel
	gptr<const int> od_tab; // per DISTINCT observation (validity slot; ProbDesc::o_valid, x12): {representative row, second row | -1, third row | -1, pose index | -1, the five words of obs_rec of the
		// representative row, 0, 0, 0}. The reference lists an observation once per Jacobian block that refers to it (optimize_edges.h:177-193: involved_obs with duplicates): the rows
		// of one observation are copies, their residual is evaluated once and stored to each of them
	gptr<const int> hap_rec; // K6 work records, sorted by decreasing term count (longest first: balances the lanes of K6): {block, first term, end term}; ProbDesc::n_hrec of them from o_hrec
	const int *sp_col_off, *sp_row, *sp_item_off, *sp_tgt /* packed update items: unified target block << 18 | a << 9 | b (packed at upload) */, *sp_rptr,
		*sp_rcol /* packed row-view entries: column << 14 | off-diagonal block */, *sp_perm; // symbolic factorisation of every capsule's system
	const int *hap_dst, *hapf_dst, *hf_dst; // destination 3x3 block of every aligned sub-block of the Hessian blocks (see symbolic_factor)
	// state + workspace
	double *edge, *ulm, *pose, *Jp, *Jf, *resid, *resid2, *HAp, *HAp0, *Hf, *HApf, *grad, *grad0 /* SRBA_EXT_SCHUR_KEEPS_GRADIENT: the gradient as K5 produced it */, *delta, *Hfinv, *YW,
		*Yh /* workgroup path: Y = W Hf^-1 of every U_Apf block, written once per solve (schur_reduce) */;
	double *old_edge, *old_ulm, *old_pose, *dense, *ulm_inf;
	double *edge1, *ulm1, *pose1; const unsigned char *pose_req; // second copy of the unknowns and of the spanning-tree poses (double-buffered LM loop);
		// per pose: a Jacobian block reads it (list_of_required_num_poses)
	int *valid, *first_fail, *hf_ok;
	unsigned char *bp_ok, *bf_ok; // per Jacobian block: its observation row is valid (set by phase_jacobians, read by phase_hessian)
	unsigned char *ulm_inf_valid;
	srba_lm_result *results;
	double *lambda_io, *chi2; int *notpd;
	long long *phase_cycles; // [n_prob*16] when phase timing is on (SRBA_HIP_PHASE_TIMING=1), else NULL
};

struct DevParams {
	int solver, noise, sensor_pose, max_iters, use_robust_kernel, cov_recovery, ext /* srba_hip_params::extensions */;
	double inv_sigma, lambda[36], kernel_param, max_err, max_rho, max_lambda, min_relin;
	double SPt[3], SPR[9];      // sensor pose on the robot
	double camL[4], camR[4];    // fx fy cx cy
	double R2Lt[3], R2LR[9];    // (-)rightCameraPose
};

// ------------------------------------------------------------------------------------------------ family traits
// PD here is the DEVICE pose stride: SE2 poses are kept as [x y phi cos(phi) sin(phi)] so that composing / inverting poses (K1), the
// residuals (K4) and the closed-form SE2 Jacobians (K2) need no trigonometric function at all; sincos is evaluated once per unknown edge
// per LM update (and once per edge at upload, on the host).  The ABI layout stays [x y phi] (srba_hip.h); conversion happens in
// srba_hip_upload_problems / srba_hip_download_state.  SE3: [t R] on both sides.
template <int FAM> struct Tr;
template <> struct Tr<SRBA_SE2_RELPOSE2D> { static constexpr int P = 3, L = 3, O = 3, PD = 5; static constexpr bool SE3 = false, REL = true; };
template <> struct Tr<SRBA_SE2_RB2D>      { static constexpr int P = 3, L = 2, O = 2, PD = 5; static constexpr bool SE3 = false, REL = false; };
template <> struct Tr<SRBA_SE2_CART2D>    { static constexpr int P = 3, L = 2, O = 2, PD = 5; static constexpr bool SE3 = false, REL = false; };
template <> struct Tr<SRBA_SE3_STEREO>    { static constexpr int P = 6, L = 3, O = 4, PD = 12; static constexpr bool SE3 = true, REL = false; };
template <> struct Tr<SRBA_SE3_MONO>      { static constexpr int P = 6, L = 3, O = 2, PD = 12; static constexpr bool SE3 = true, REL = false; };
template <> struct Tr<SRBA_SE3_CART3D>    { static constexpr int P = 6, L = 3, O = 3, PD = 12; static constexpr bool SE3 = true, REL = false; };
template <> struct Tr<SRBA_SE3_RB3D>      { static constexpr int P = 6, L = 3, O = 3, PD = 12; static constexpr bool SE3 = true, REL = false; };
template <> struct Tr<SRBA_SE3_RELPOSE3D> { static constexpr int P = 6, L = 6, O = 6, PD = 12; static constexpr bool SE3 = true, REL = true; };
template <> struct Tr<SRBA_SE2_STEREO>    { static constexpr int P = 3, L = 3, O = 4, PD = 5; static constexpr bool SE3 = false, REL = false; };  // SE(2) key-frames, 3D points

// ------------------------------------------------------------------------------------------------ poses
__device__ __forceinline__ double wrap_pi(double a) { // mrpt::math::wrapToPi up to rounding (and the sign of an exact +-pi)
	return a - (2.0 * M_PI) * rint(a * (0.5 / M_PI));
}
struct P2 { double x, y, phi, c, s; };
struct P3 { double t[3]; double R[9]; };
__device__ __forceinline__ P2 ident2() { P2 r; r.x = 0; r.y = 0; r.phi = 0; r.c = 1; r.s = 0; return r; }
__device__ __forceinline__ P3 ident3() { P3 r; r.t[0] = r.t[1] = r.t[2] = 0; r.R[0] = 1; r.R[1] = 0; r.R[2] = 0; r.R[3] = 0; r.R[4] = 1; r.R[5] = 0; r.R[6] = 0; r.R[7] = 0; r.R[8] = 1; return r; }
// Small fixed-size records (poses, Jacobian / Hessian blocks) are gathered by lanes that each read their own record: the vector memory pipeline then handles one
// request per lane and per instruction, whatever its width. Records are therefore moved 16 bytes at a time (global_load / store_dwordx4; gfx950 only asks for dword
// alignment of the address, and the records are 8-byte aligned): a 72-byte block costs 5 requests instead of 9.
typedef double f64x2u __attribute__((ext_vector_type(2), aligned(8)));
template <int N> __device__ __forceinline__ void ldn(double *dst, const double *src) {
#pragma unroll
	for (int k = 0; k + 1 < N; k += 2) { const f64x2u v = *(const f64x2u *)(src + k); dst[k] = v.x; dst[k + 1] = v.y; }
	if (N & 1) dst[N - 1] = src[N - 1];
}
template <int N> __device__ __forceinline__ void stn(double *dst, const double *src) {
#pragma unroll
	for (int k = 0; k + 1 < N; k += 2) { f64x2u v; v.x = src[k]; v.y = src[k + 1]; *(f64x2u *)(dst + k) = v; }
	if (N & 1) dst[N - 1] = src[N - 1];
}
__device__ __forceinline__ P2 ld2(const double *p) { double v[5]; ldn<5>(v, p); P2 r; r.x = v[0]; r.y = v[1]; r.phi = v[2]; r.c = v[3]; r.s = v[4]; return r; }
__device__ __forceinline__ void st2(double *p, const P2 &a) { const double v[5] = {a.x, a.y, a.phi, a.c, a.s}; stn<5>(p, v); }
__device__ __forceinline__ P3 ld3(const double *p) { double v[12]; ldn<12>(v, p); P3 r; for (int i = 0; i < 3; i++) r.t[i] = v[i]; for (int i = 0; i < 9; i++) r.R[i] = v[3 + i]; return r; }
__device__ __forceinline__ void st3(double *p, const P3 &a) { double v[12]; for (int i = 0; i < 3; i++) v[i] = a.t[i]; for (int i = 0; i < 9; i++) v[3 + i] = a.R[i]; stn<12>(p, v); }
__device__ __forceinline__ P2 comp(const P2 &A, const P2 &B) { P2 r; r.x = A.x + B.x * A.c - B.y * A.s; r.y = A.y + B.x * A.s + B.y * A.c; r.phi = wrap_pi(A.phi + B.phi); r.c = A.c * B.c - A.s * B.s;
	r.s = A.s * B.c + A.c * B.s; return r; }
__device__ __forceinline__ P2 inv(const P2 &A) { P2 r; r.x = -A.x * A.c - A.y * A.s; r.y = A.x * A.s - A.y * A.c; r.phi = -A.phi; r.c = A.c; r.s = -A.s; return r; }
__device__ __forceinline__ P3 comp(const P3 &A, const P3 &B) {
	P3 r;
#pragma unroll
	for (int i = 0; i < 3; i++) {
#pragma unroll
		for (int j = 0; j < 3; j++) r.R[3 * i + j] = A.R[3 * i] * B.R[j] + A.R[3 * i + 1] * B.R[3 + j] + A.R[3 * i + 2] * B.R[6 + j];
		r.t[i] = A.t[i] + A.R[3 * i] * B.t[0] + A.R[3 * i + 1] * B.t[1] + A.R[3 * i + 2] * B.t[2];
	}
	return r;
}
__device__ __forceinline__ P3 inv(const P3 &A) {
	P3 r;
#pragma unroll
	for (int i = 0; i < 3; i++) {
#pragma unroll
		for (int j = 0; j < 3; j++) r.R[3 * i + j] = A.R[3 * j + i];
		r.t[i] = -(A.R[i] * A.t[0] + A.R[3 + i] * A.t[1] + A.R[6 + i] * A.t[2]);
	}
	return r;
}
__device__ __forceinline__ P3 exp_se3(const double *v) { // SE_traits<3>::pseudo_exp: t=v[0:3], R=Rodrigues(v[3:6])
	P3 r; r.t[0] = v[0]; r.t[1] = v[1]; r.t[2] = v[2];
	const double wx = v[3], wy = v[4], wz = v[5], th2 = wx * wx + wy * wy + wz * wz, th = sqrt(th2);
	double a, b;
	if (th < 1e-8) { a = 1.0 - th2 * (1.0 / 6.0); b = 0.5 - th2 * (1.0 / 24.0); } else { double s, c; sincos(th, &s, &c); a = s / th; b = (1.0 - c) / th2; }
	r.R[0] = 1.0 - b * (wy * wy + wz * wz); r.R[1] = -a * wz + b * wx * wy; r.R[2] = a * wy + b * wx * wz;
	r.R[3] = a * wz + b * wx * wy; r.R[4] = 1.0 - b * (wx * wx + wz * wz); r.R[5] = -a * wx + b * wy * wz;
	r.R[6] = -a * wy + b * wx * wz; r.R[7] = a * wx + b * wy * wz; r.R[8] = 1.0 - b * (wx * wx + wy * wy);
	return r;
}
template <bool SE3> struct PoseOps;
template <> struct PoseOps<false> { typedef P2 T; static __device__ __forceinline__ T ident() { return ident2(); } static __device__ __forceinline__ T ld(const double *p) { return ld2(p); }
	static __device__ __forceinline__ void st(double *p, const T &a) { st2(p, a); }
	static __device__ __forceinline__ T from(const double *v) { P2 r; r.x = v[0]; r.y = v[1]; r.phi = v[2]; r.c = v[3]; r.s = v[4]; return r; } static __device__ __forceinline__ void to(double *v,
		const T &a) { v[0] = a.x; v[1] = a.y; v[2] = a.phi; v[3] = a.c; v[4] = a.s; }
	static __device__ __forceinline__ T expm(const double *v) { T r; r.x = v[0]; r.y = v[1]; r.phi = v[2]; sincos(v[2], &r.s, &r.c); return r; } };
template <> struct PoseOps<true> { typedef P3 T; static __device__ __forceinline__ T ident() { return ident3(); } static __device__ __forceinline__ T ld(const double *p) { return ld3(p); }
	static __device__ __forceinline__ void st(double *p, const T &a) { st3(p, a); }
	static __device__ __forceinline__ T from(const double *v) { P3 r; for (int i = 0; i < 3; i++) r.t[i] = v[i]; for (int i = 0; i < 9; i++) r.R[i] = v[3 + i]; return r; }
		static __device__ __forceinline__ void to(double *v, const T &a) { for (int i = 0; i < 3; i++) v[i] = a.t[i]; for (int i = 0; i < 9; i++) v[3 + i] = a.R[i]; }
	static __device__ __forceinline__ T expm(const double *v) { return exp_se3(v); } };

// ------------------------------------------------------------------------------------------------ block reductions (deterministic)
// DPP cross-lane moves (no LDS traffic, ~8 cycles): row_shr within rows of 16 lanes, then row_bcast15 / row_bcast31 across rows; the
// total lands in lane 63 and is broadcast with v_readlane. Fixed order -> deterministic.
template <int CTRL, int ROW_MASK> __device__ __forceinline__ double dpp_shift0(double v) { // lanes without a source read 0.0
	int lo = __double2loint(v), hi = __double2hiint(v);
	lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false); hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
	return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane63(double v) {
	const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
	return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v) {
	v += dpp_shift0<0x111, 0xf>(v); v += dpp_shift0<0x112, 0xf>(v); v += dpp_shift0<0x114, 0xf>(v); v += dpp_shift0<0x118, 0xf>(v); // row_shr:1,2,4,8
	v += dpp_shift0<0x142, 0xa>(v); v += dpp_shift0<0x143, 0xc>(v); // row_bcast:15 (rows 1,3), row_bcast:31 (rows 2,3)
	return readlane63(v);
}
__device__ __forceinline__ double wave_max(double v) { // operands are >= 0 here (|g|, Hessian diagonals): 0-fill is neutral
	v = fmax(v, dpp_shift0<0x111, 0xf>(v)); v = fmax(v, dpp_shift0<0x112, 0xf>(v)); v = fmax(v, dpp_shift0<0x114, 0xf>(v)); v = fmax(v, dpp_shift0<0x118, 0xf>(v));
	{ const double t = dpp_shift0<0x142, 0xa>(v); v = fmax(v, t); } { const double t = dpp_shift0<0x143, 0xc>(v); v = fmax(v, t); }
	return readlane63(v);
}
// One wavefront per capsule: "block" reductions are wave reductions; every lane gets the same result.
__device__ __forceinline__ double block_sum(double v, double *) { return wave_sum(v); }
__device__ __forceinline__ double block_max(double v, double *) { return wave_max(v); }
// ---- capsule GROUPS (round 4): the lanes that work on one capsule. G = 64: one wavefront (every kernel of rounds 1-3). G = 128: TWO wavefronts per capsule (k_lm_run2: the big,
// LDS-bound windows of a big batch -- their wavefronts sit alone on a SIMD and the launch waits for their latency: the lane-parallel phases go twice as wide, the block-sparse
// solver stays on the first wavefront). Reductions over a group: per wavefront (DPP tree), then the two totals through two doubles of LDS (`red`) in a fixed order: deterministic.
template <int G> __device__ __forceinline__ double grp_sum(double v, double *red) { // red: G / 64 doubles of LDS (at most 12)
	v = wave_sum(v);
	if constexpr (G > 64) { static_assert(G % 64 == 0 && G <= 768, "groups of up to twelve wavefronts");
		if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v; __syncthreads(); v = red[0] + red[1];
		if constexpr (G > 128) v = v + (red[2] + red[3]);
		if constexpr (G > 256) v = v + ((red[4] + red[5]) + (red[6] + red[7]));
		if constexpr (G > 512) v = v + ((red[8] + red[9]) + (red[10] + red[11]));
		__syncthreads(); }
	return v;
}
template <int G> __device__ __forceinline__ double grp_max(double v, double *red) {
	v = wave_max(v);
	if constexpr (G > 64) { if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v; __syncthreads(); v = fmax(red[0], red[1]);
		if constexpr (G > 128) v = fmax(v, fmax(red[2], red[3]));
		if constexpr (G > 256) v = fmax(v, fmax(fmax(red[4], red[5]), fmax(red[6], red[7])));
		if constexpr (G > 512) v = fmax(v, fmax(fmax(red[8], red[9]), fmax(red[10], red[11])));
		__syncthreads(); }
	return v;
}
// hand-off through LDS inside a phase: one wavefront needs no barrier (its LDS instructions execute in order), two do
template <int G> __device__ __forceinline__ void grp_lds_sync() { if constexpr (G > 64) __syncthreads(); else { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } }

// ------------------------------------------------------------------------------------------------ full-pivot LU inverse (schur.h:200-206)
// Every index below is a compile-time constant (the loops are fully unrolled and the pivot position acts through selects): the matrices stay in registers. With
// run-time row / column indices the arrays lived in scratch memory (48..80 bytes per lane in every landmark-family kernel).
template <int N> __device__ __forceinline__ bool fullpiv_inverse(const double *A, double *Ai) {
	double lu[N * N]; int rp[N], cp[N]; double piv[N];
#pragma unroll
	for (int i = 0; i < N * N; i++) lu[i] = A[i];
#pragma unroll
	for (int i = 0; i < N; i++) { rp[i] = i; cp[i] = i; piv[i] = 0; }
	double maxpiv = 0; bool live = true; // live: no exactly-zero remainder met yet (the reference's loop ends there)
#pragma unroll
	for (int k = 0; k < N; k++) {
		int br = k, bc = k; double best = -1;
#pragma unroll
		for (int c = k; c < N; c++)
#pragma unroll
			for (int r = k; r < N; r++) { const double v = fabs(lu[r * N + c]); const bool up = v > best; best = up ? v : best; br = up ? r : br; bc = up ? c : bc; }
		live = live && !(best == 0.0);
		if (live) {
			if (best > maxpiv) maxpiv = best;
#pragma unroll
			for (int r = k + 1; r < N; r++) { const bool sw = (br == r); // rows k <-> br
#pragma unroll
				for (int c = 0; c < N; c++) { const double a = lu[k * N + c], b = lu[r * N + c]; lu[k * N + c] = sw ? b : a; lu[r * N + c] = sw ? a : b; }
				const int a = rp[k], b = rp[r]; rp[k] = sw ? b : a; rp[r] = sw ? a : b; }
#pragma unroll
			for (int c = k + 1; c < N; c++) { const bool sw = (bc == c); // columns k <-> bc
#pragma unroll
				for (int r = 0; r < N; r++) { const double a = lu[r * N + k], b = lu[r * N + c]; lu[r * N + k] = sw ? b : a; lu[r * N + c] = sw ? a : b; }
				const int a = cp[k], b = cp[c]; cp[k] = sw ? b : a; cp[c] = sw ? a : b; }
			piv[k] = lu[k * N + k];
#pragma unroll
			for (int r = k + 1; r < N; r++) lu[r * N + k] /= lu[k * N + k];
#pragma unroll
			for (int r = k + 1; r < N; r++)
#pragma unroll
				for (int c = k + 1; c < N; c++) lu[r * N + c] -= lu[r * N + k] * lu[k * N + c];
		}
	}
	const double thr = 2.220446049250313e-16 * N * fabs(maxpiv);
	int rank = 0;
#pragma unroll
	for (int k = 0; k < N; k++) if (fabs(piv[k]) > thr) rank++;
	if (rank != N) return false;
#pragma unroll
	for (int col = 0; col < N; col++) {
		double b[N];
#pragma unroll
		for (int r = 0; r < N; r++) b[r] = (rp[r] == col) ? 1.0 : 0.0;
#pragma unroll
		for (int r = 0; r < N; r++)
#pragma unroll
			for (int c = 0; c < r; c++) b[r] -= lu[r * N + c] * b[c];
#pragma unroll
		for (int r = N - 1; r >= 0; r--) {
#pragma unroll
			for (int c = r + 1; c < N; c++) b[r] -= lu[r * N + c] * b[c];
			b[r] /= lu[r * N + r]; }
#pragma unroll
		for (int i = 0; i < N; i++) { double v = 0; // Ai[cp[r]][col] = b[r]
#pragma unroll
			for (int r = 0; r < N; r++) v = (cp[r] == i) ? b[r] : v;
			Ai[i * N + col] = v; }
	}
	return true;
}

// ------------------------------------------------------------------------------------------------ block-sparse Cholesky by one wavefront
// The SPD system (H + lambda I) of a capsule is block-sparse (a depth-3 window of sub-maps gives an almost banded pattern: ~110 non-zero
// 3x3 blocks of ~300, and practically no fill).  The host computes the symbolic factorisation once per capsule at upload time
// (srba_hip.hip: symbolic_factor) -- the analogue of CSparse's cs_schol inside mrpt::math::CSparseMatrix::CholeskyDecomp that the
// reference builds once per optimize_edges() call (lev-marq_solvers.h:164-166) -- and the device runs the numeric right-looking
// factorisation over that fixed pattern, in LDS, for every LM trial:
//   storage : diag[nb][9] | off[nnzoff][9] (column-compressed, rows ascending) | rhs[nb][3]
//   step k  : every lane refactors the 3x3 diagonal block (no broadcast); update items (a>=b) of column k recompute the two panel
//             blocks they need and subtract L_ak L_bk^t from their precomputed target block; then the panel blocks are overwritten
//             by L_ak and the right-hand side is eliminated (forward substitution fused).  Two wave barriers per step.
// "Not positive definite" == a scalar pivot <= 0 (Eigen LLT / cs_chol criterion), decided identically by all lanes.
struct SparseSys { // per-capsule symbolic structure (LDS copy of the host's symbolic factorisation) + numeric storage, all in LDS
	int nb, nnzoff, dense; // dense: every off-diagonal block (r > c) is stored, column after column: index c (nb-1) - c (c-1)/2 + (r-c-1); no index arrays
	const int *col_off, *row;  // col_off[nb+1], row[nnzoff] (rows ascending inside a column)
	const int *item;           // update items of all columns, column after column (cn(cn+1)/2 each, a>=b row positions inside the column):
	                           //   one packed word  u<<18 | a<<9 | b  (u = unified block index: diag k -> k, off-diag i -> nb+i)
	const int *rptr, *rent;    // row view for the backward substitution: entries of block-row a = rent[rptr[a]..rptr[a+1]) = col<<14 | off-diag index
	const int *perm;           // perm[original 3-row block] = position in the fill-reducing elimination order
	double *diag, *off, *rhs;
	double *row_lds;           // HBM-resident dense layout with left-looking sweeps: 21 nb doubles of LDS (rows k, k+1 of the factor | y), else null
	double *tiles, *linv; int nt; // workgroup path (srba_wg.hpp, ProbDesc::dense_blocks == 3): lower triangle of 16 x 16 frag tiles + the right-hand-side tile row | inverse diagonal factors;
		// rhs then points at x in LDS, natural order (no permutation)
	__device__ __forceinline__ double sol(int q) const { return tiles ? rhs[q] : rhs[3 * perm[q / 3] + q % 3]; } // component q (original numbering) of the solved right-hand side
};
// Cross-lane hand-off inside the solver. LDS instructions of one wavefront execute in issue order, so a ds_write followed by a ds_read of
// another lane's data needs no s_waitcnt -- only the compiler must keep the program order (wavefront-scope fence = no instructions).
__device__ __forceinline__ void solver_sync() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
struct Chol3 { double l10, l20, l21, r0, r1, r2, l00, l11, l22; };
// Branch-free 3x3 Cholesky of the lower triangle (a00 a10 a11 a20 a21 a22): the three pivots are tested together at the end, so the six
// loads and the whole dependent chain are issued without a scalar branch in between. Garbage in c when it returns false.
__device__ __forceinline__ bool chol3v(double a00, double a10, double a11, double a20, double a21, double a22, Chol3 &c) {
	c.r0 = rsqrt(a00); c.l00 = a00 * c.r0; c.l10 = a10 * c.r0; c.l20 = a20 * c.r0;
	const double d1 = a11 - c.l10 * c.l10;
	c.r1 = rsqrt(d1); c.l11 = d1 * c.r1; c.l21 = (a21 - c.l20 * c.l10) * c.r1;
	const double d2 = a22 - c.l20 * c.l20 - c.l21 * c.l21;
	c.r2 = rsqrt(d2); c.l22 = d2 * c.r2;
	return (a00 > 0.0) & (d1 > 0.0) & (d2 > 0.0); // NaN-safe: any non-positive or NaN pivot fails
}
__device__ __forceinline__ bool chol3(const double *D, Chol3 &c) { return chol3v(D[0], D[3], D[4], D[6], D[7], D[8], c); }
// ---- LDS solver, lane-per-block-row form. The one-lane-per-3x3-block loops above keep 1..10 of 64 lanes busy and every wave instruction
// costs an issue slot whatever the number of live lanes, so the factorisation was issue-bound at ~250 instructions per column. Here a 3x3
// block is worked on by THREE lanes, one per block row (lane = 3*block + row; lane 63 idles in these phases):
//   panel   : row r of L_ak = A_ak L_kk^-t is a 3-term forward substitution of row r of A_ak alone      (3 loads, 6 flops, 3 stores per lane)
//   rhs     : rhs_a[r] -= L_ak[r,:] . y_k                                                                (same lane, 1 load, 1 store)
//   update  : row r of T -= L_ak L_bk^t needs row r of L_ak, all of L_bk and row r of T                  (15 loads, 9 fma, 3 stores per lane)
//   backward: y_k[q] -= (L_ak^t x_a)[q] = sum_r L_ak[r][q] x_a[r], one lane per (entry, q)               (4 loads, 3 fma, 1 store per lane)
// so a column with 4 panel blocks and 10 update items occupies 12 / 30 lanes instead of 4 / 10 and the dependent chain per lane is a third as
// long; the instruction count per column drops to about half. The diagonal 3x3 Cholesky stays redundant in every lane (no broadcast).
// Same arithmetic per scalar as the lane-per-block form (same operation order inside every dot product): results are bit-identical.
#ifndef SRBA_SOLVER_LDL
#define SRBA_SOLVER_LDL 1 /* 1: square-root-free block LDL^t sweeps (round 5), 0: the block LL^t sweeps of rounds 2-4 */
#endif
#if !SRBA_SOLVER_LDL
__device__ __forceinline__ bool sp_factor_fsub_rows(const SparseSys &S) {
	const int lane = threadIdx.x, nb = S.nb;
	const int grp = (lane * 171) >> 9, sub = lane - 3 * grp; // lane / 3, lane % 3 for lane < 64
	const bool worker = lane < 63;
	int cb = S.col_off[0], ce = nb > 0 ? S.col_off[1] : cb, ib = 0;
	int ra = (worker && cb + grp < ce) ? S.row[cb + grp] : 0; // block-row of this lane's panel block in the coming column
	for (int k = 0; k < nb; k++) {
		const int cn = ce - cb, nitems = cn * (cn + 1) / 2;
		double *D = S.diag + 9 * k;
		const double a00 = D[0], a10 = D[3], a11 = D[4], a20 = D[6], a21 = D[7], a22 = D[8];
		const double b0 = S.rhs[3 * k], b1 = S.rhs[3 * k + 1], b2 = S.rhs[3 * k + 2];
		const bool pl = worker && grp < cn;
		double *Arow = S.off + 9 * (cb + grp) + 3 * sub; double *rr = S.rhs + 3 * ra + sub;
		double A0 = 0, A1 = 0, A2 = 0, rv = 0;
		if (pl) { A0 = Arow[0]; A1 = Arow[1]; A2 = Arow[2]; rv = *rr; }
		// index loads for the next column
		const int ce_n = (k + 2 <= nb) ? S.col_off[k + 2] : ce;
		const int ra_n = (worker && ce + grp < ce_n) ? S.row[ce + grp] : 0;
		Chol3 c;
		if (!chol3v(a00, a10, a11, a20, a21, a22, c)) return false;
		const double y0 = b0 * c.r0, y1 = (b1 - c.l10 * y0) * c.r1, y2 = (b2 - c.l20 * y0 - c.l21 * y1) * c.r2;
		if (pl) {
			const double x0 = A0 * c.r0, x1 = (A1 - x0 * c.l10) * c.r1, x2 = (A2 - x0 * c.l20 - x1 * c.l21) * c.r2;
			Arow[0] = x0; Arow[1] = x1; Arow[2] = x2;
			*rr = rv - (x0 * y0 + x1 * y1 + x2 * y2);
		}
		if (worker) for (int p = grp + 21; p < cn; p += 21) { // columns with more than 21 blocks
			double *Ax = S.off + 9 * (cb + p) + 3 * sub; double *rx = S.rhs + 3 * S.row[cb + p] + sub;
			const double x0 = Ax[0] * c.r0, x1 = (Ax[1] - x0 * c.l10) * c.r1, x2 = (Ax[2] - x0 * c.l20 - x1 * c.l21) * c.r2;
			Ax[0] = x0; Ax[1] = x1; Ax[2] = x2; *rx -= x0 * y0 + x1 * y1 + x2 * y2;
		}
		if (lane == SRBA_WG - 1) { // L_kk, reciprocal diagonal in the unused upper part, y_k
			D[0] = c.l00; D[3] = c.l10; D[4] = c.l11; D[6] = c.l20; D[7] = c.l21; D[8] = c.l22; D[1] = c.r0; D[2] = c.r1; D[5] = c.r2;
			S.rhs[3 * k] = y0; S.rhs[3 * k + 1] = y1; S.rhs[3 * k + 2] = y2;
		}
		solver_sync();
		if (worker) for (int t = grp; t < nitems; t += 21) { // trailing update, row `sub` of target -= L_ak L_bk^t
			const unsigned w = (unsigned)S.item[ib + t];
			const double *La = S.off + 9 * (cb + ((w >> 9) & 511)) + 3 * sub, *Lb = S.off + 9 * (cb + (w & 511)); double *T = S.diag + 9 * (w >> 18) + 3 * sub;
			const double la0 = La[0], la1 = La[1], la2 = La[2];
			double lb[9];
#pragma unroll
			for (int q = 0; q < 9; q++) lb[q] = Lb[q];
			const double t0 = T[0], t1 = T[1], t2 = T[2];
			T[0] = t0 - (la0 * lb[0] + la1 * lb[1] + la2 * lb[2]);
			T[1] = t1 - (la0 * lb[3] + la1 * lb[4] + la2 * lb[5]);
			T[2] = t2 - (la0 * lb[6] + la1 * lb[7] + la2 * lb[8]);
		}
		solver_sync();
		cb = ce; ce = ce_n; ib += nitems; ra = ra_n;
	}
	return true;
}
__device__ __forceinline__ void sp_bsub_rows(const SparseSys &S) {
	const int lane = threadIdx.x, nb = S.nb;
	if (nb <= 0) return;
	const int grp = (lane * 171) >> 9, sub = lane - 3 * grp; const bool worker = lane < 63;
	int re = S.rptr[nb], rb = S.rptr[nb - 1], rb_n = nb >= 2 ? S.rptr[nb - 2] : 0;
	unsigned w = (worker && rb + grp < re) ? (unsigned)S.rent[rb + grp] : 0u; // col << 14 | off-diagonal block
	for (int a = nb - 1; a >= 0; a--) {
		const bool act = worker && rb + grp < re;
		const double *D = S.diag + 9 * a; const double *Lb = S.off + 9 * (w & 0x3fff) + sub; double *y = S.rhs + 3 * (w >> 14) + sub;
		const double r0 = S.rhs[3 * a], r1 = S.rhs[3 * a + 1], r2 = S.rhs[3 * a + 2], d5 = D[5], d7 = D[7], d2 = D[2], d3 = D[3], d6 = D[6], d1 = D[1];
		double l0 = 0, l1 = 0, l2 = 0, yv = 0;
		if (act) { l0 = Lb[0]; l1 = Lb[3]; l2 = Lb[6]; yv = *y; }
		const unsigned w_n = (worker && a > 0 && rb_n + grp < rb) ? (unsigned)S.rent[rb_n + grp] : 0u;
		const int rb_nn = a >= 2 ? S.rptr[a - 2] : 0;
		const double x2 = r2 * d5, x1 = (r1 - d7 * x2) * d2, x0 = (r0 - d3 * x1 - d6 * x2) * d1;
		if (act) *y = yv - (l0 * x0 + l1 * x1 + l2 * x2);
		if (worker) for (int j = rb + grp + 21; j < re; j += 21) { // rows with more than 21 blocks
			const unsigned wx = (unsigned)S.rent[j]; const double *Lx = S.off + 9 * (wx & 0x3fff) + sub; double *yx = S.rhs + 3 * (wx >> 14) + sub;
			*yx -= Lx[0] * x0 + Lx[3] * x1 + Lx[6] * x2;
		}
		if (lane == SRBA_WG - 1) { S.rhs[3 * a] = x0; S.rhs[3 * a + 1] = x1; S.rhs[3 * a + 2] = x2; }
		solver_sync();
		re = rb; rb = rb_n; rb_n = rb_nn; w = w_n;
	}
}
#else
// ---- Round 5: the same two sweeps as a square-root-free BLOCK LDL^t (H + lambda I = L D L^t, D = 3x3 diagonal blocks, L unit block-lower-triangular).
// Why: a column of the LL^t form is a dependent chain of ~70 FP64 instructions (14-16 cycles each for a lone wavefront), 45 of them the redundant 3x3 Cholesky with three
// refined rsqrt (71 cycles each), then the panel solve, an LDS hand-off, the update. Here the pivot block is INVERTED by cofactors (two products deep, then det: three, one
// refined v_rcp_f64: five) and the panel is never written back: with W_ak = the (updated) block (a,k) as it stands when column k is reached,
//     L_ak = W_ak D_k^-1,   target(a,b) -= L_ak W_bk^t = (W_ak D_k^-1) W_bk^t,   z_a -= W_ak (D_k^-1 z_k),   x_k = D_k^-1 (z_k - sum_a W_ak^t x_a)
// so the update reads the blocks of column k as the previous columns left them -- no panel store, and ONE LDS hand-off per column instead of two. The factor kept in the
// image is {D_k^-1 (lower triangle, in the diagonal block's place), W_ak (in place)}; the right-hand side keeps z_k (not D_k^-1 z_k) until the backward sweep.
// "Not positive definite" == a leading minor of a pivot block <= 0 (a00, a00 a11 - a10^2, det): in exact arithmetic the same verdict as a Cholesky pivot <= 0
// (Sylvester), decided identically by all lanes. Rounding differs from the LL^t sweeps (and from the oracle's) -- parity is the decision-replay check, not bit identity.
// A floating-point VALU instruction with eight or fewer live lanes costs three times one with nine or more (tools/probes/valu_mask_rate.hip: v_fma_f64 21 against 7 shader cycles per
// wavefront, two wavefronts per SIMD; the same for FP32; integer 1.25 x). The panel / update / substitution arithmetic of a column with one or two off-diagonal blocks has 3 .. 6 live
// lanes when it sits under its store's condition: it is formed by every lane instead (idle lanes work on block 0 of the image, their loads are unconditional anyway) and only the store
// is conditional. The empty asm pins the value where all lanes are live, so that the compiler cannot sink the arithmetic back under the condition. Same numbers, bit for bit.
#ifndef SRBA_SOLVER_ALL_LANES
#define SRBA_SOLVER_ALL_LANES 1
#endif
#if SRBA_SOLVER_ALL_LANES
#define SRBA_ALL_LANES(x) asm volatile("" : "+v"(x))
#else
#define SRBA_ALL_LANES(x) do { } while (0)
#endif
struct Inv3 { double i00, i10, i11, i20, i21, i22; };
__device__ __forceinline__ double rcp_refined(double d) { // 1/d: v_rcp_f64 + two Newton steps (the reciprocal of the division expansion without its numerator steps)
	double x = __builtin_amdgcn_rcp(d); double e = fma(-d, x, 1.0); x = fma(x, e, x); e = fma(-d, x, 1.0); return fma(x, e, x);
}
__device__ __forceinline__ bool inv3v(double a00, double a10, double a11, double a20, double a21, double a22, Inv3 &v) {
	const double c00 = fma(a11, a22, -(a21 * a21)), c10 = fma(a20, a21, -(a10 * a22)), c20 = fma(a10, a21, -(a20 * a11));
	const double c11 = fma(a00, a22, -(a20 * a20)), c21 = fma(a10, a20, -(a00 * a21)), c22 = fma(a00, a11, -(a10 * a10));
	const double det = fma(a20, c20, fma(a10, c10, a00 * c00));
	const double r = rcp_refined(det);
	v.i00 = c00 * r; v.i10 = c10 * r; v.i11 = c11 * r; v.i20 = c20 * r; v.i21 = c21 * r; v.i22 = c22 * r;
	return (a00 > 0.0) & (c22 > 0.0) & (det > 0.0); // NaN-safe: any non-positive or NaN minor fails
}
__device__ __forceinline__ bool sp_factor_fsub_rows(const SparseSys &S) {
	const int lane = threadIdx.x, nb = S.nb;
	const int grp = (lane * 171) >> 9, sub = lane - 3 * grp; // lane / 3, lane % 3 for lane < 64
	const bool worker = lane < 63;
	int cb = S.col_off[0], ce = nb > 0 ? S.col_off[1] : cb, ib = 0;
	int ra = (worker && cb + grp < ce) ? S.row[cb + grp] : 0; // block-row of this lane's panel block in the coming column
	unsigned w0 = (worker && nb > 0 && grp < (ce - cb) * (ce - cb + 1) / 2) ? (unsigned)S.item[grp] : 0u; // this lane's first update item of the coming column
	for (int k = 0; k < nb; k++) {
		const int cn = ce - cb, nitems = cn * (cn + 1) / 2;
		double *D = S.diag + 9 * k;
		const double a00 = D[0], a10 = D[3], a11 = D[4], a20 = D[6], a21 = D[7], a22 = D[8];
		const double z0 = S.rhs[3 * k], z1 = S.rhs[3 * k + 1], z2 = S.rhs[3 * k + 2];
		const bool pl = worker && grp < cn, up = worker && grp < nitems;
		// (loads are unconditional -- an idle lane reads block 0 of the column / of the image, inside the LDS image -- only the stores are guarded: no zero-fill, no exec juggling around them)
		const double *Arow = S.off + 9 * (cb + (pl ? grp : 0)) + 3 * sub; double *rr = S.rhs + 3 * ra + sub;
		const double A0 = Arow[0], A1 = Arow[1], A2 = Arow[2], rv = *rr;
		// operands of this lane's first update item (w0 = 0 for an idle lane): none of them depends on the inverse, their loads travel with the pivot block's
		const double *La = S.off + 9 * (cb + ((w0 >> 9) & 511)) + 3 * sub, *Lb = S.off + 9 * (cb + (w0 & 511)); double *T = S.diag + 9 * (w0 >> 18) + 3 * sub;
		const double la0 = La[0], la1 = La[1], la2 = La[2], t0 = T[0], t1 = T[1], t2 = T[2]; double lb[9];
#pragma unroll
		for (int q = 0; q < 9; q++) lb[q] = Lb[q];
		// index loads for the next column
		const int ce_n = (k + 2 <= nb) ? S.col_off[k + 2] : ce;
		const int ra_n = (worker && ce + grp < ce_n) ? S.row[ce + grp] : 0;
		const int cn_n = ce_n - ce; const unsigned w0_n = (worker && k + 1 < nb && grp < cn_n * (cn_n + 1) / 2) ? (unsigned)S.item[ib + nitems + grp] : 0u;
		Inv3 v;
		if (!inv3v(a00, a10, a11, a20, a21, a22, v)) return false;
		const double u0 = fma(v.i20, z2, fma(v.i10, z1, v.i00 * z0)), u1 = fma(v.i21, z2, fma(v.i11, z1, v.i10 * z0)), u2 = fma(v.i22, z2, fma(v.i21, z1, v.i20 * z0)); // D_k^-1 z_k
		{ double nr = rv - fma(A2, u2, fma(A1, u1, A0 * u0)); SRBA_ALL_LANES(nr); if (pl) *rr = nr; }
		if (worker) for (int p = grp + 21; p < cn; p += 21) { // columns with more than 21 blocks
			const double *Ax = S.off + 9 * (cb + p) + 3 * sub; double *rx = S.rhs + 3 * S.row[cb + p] + sub;
			*rx -= fma(Ax[2], u2, fma(Ax[1], u1, Ax[0] * u0));
		}
		{ // row `sub` of target -= (W_ak D_k^-1) W_bk^t: formed by EVERY lane (an idle lane works on block 0 of the image), stored by the lanes that own an item -- see SRBA_ALL_LANES
			const double l0 = fma(la2, v.i20, fma(la1, v.i10, la0 * v.i00)), l1 = fma(la2, v.i21, fma(la1, v.i11, la0 * v.i10)), l2 = fma(la2, v.i22, fma(la1, v.i21, la0 * v.i20));
			double n0 = t0 - fma(l2, lb[2], fma(l1, lb[1], l0 * lb[0])), n1 = t1 - fma(l2, lb[5], fma(l1, lb[4], l0 * lb[3])), n2 = t2 - fma(l2, lb[8], fma(l1, lb[7], l0 * lb[6]));
			SRBA_ALL_LANES(n0); SRBA_ALL_LANES(n1); SRBA_ALL_LANES(n2);
			if (up) { T[0] = n0; T[1] = n1; T[2] = n2; }
		}
		if (worker) for (int t = grp + 21; t < nitems; t += 21) { // the other update items of a column with more than 21
			const unsigned w = (unsigned)S.item[ib + t];
			const double *Lx = S.off + 9 * (cb + ((w >> 9) & 511)) + 3 * sub, *Ly = S.off + 9 * (cb + (w & 511)); double *Tx = S.diag + 9 * (w >> 18) + 3 * sub;
			const double x0 = Lx[0], x1 = Lx[1], x2 = Lx[2];
			double ly[9];
#pragma unroll
			for (int q = 0; q < 9; q++) ly[q] = Ly[q];
			const double s0 = Tx[0], s1 = Tx[1], s2 = Tx[2];
			const double l0 = fma(x2, v.i20, fma(x1, v.i10, x0 * v.i00)), l1 = fma(x2, v.i21, fma(x1, v.i11, x0 * v.i10)), l2 = fma(x2, v.i22, fma(x1, v.i21, x0 * v.i20));
			Tx[0] = s0 - fma(l2, ly[2], fma(l1, ly[1], l0 * ly[0]));
			Tx[1] = s1 - fma(l2, ly[5], fma(l1, ly[4], l0 * ly[3]));
			Tx[2] = s2 - fma(l2, ly[8], fma(l1, ly[7], l0 * ly[6]));
		}
		if (lane == SRBA_WG - 1) { D[0] = v.i00; D[3] = v.i10; D[4] = v.i11; D[6] = v.i20; D[7] = v.i21; D[8] = v.i22; } // D_k^-1 in the pivot block's place (z_k stays in the right-hand side)
		solver_sync();
		cb = ce; ce = ce_n; ib += nitems; ra = ra_n; w0 = w0_n;
	}
	return true;
}
__device__ __forceinline__ void sp_bsub_rows(const SparseSys &S) {
	const int lane = threadIdx.x, nb = S.nb;
	if (nb <= 0) return;
	const int grp = (lane * 171) >> 9, sub = lane - 3 * grp; const bool worker = lane < 63;
	int re = S.rptr[nb], rb = S.rptr[nb - 1], rb_n = nb >= 2 ? S.rptr[nb - 2] : 0;
	unsigned w = (worker && rb + grp < re) ? (unsigned)S.rent[rb + grp] : 0u; // col << 14 | off-diagonal block
	for (int a = nb - 1; a >= 0; a--) {
		const bool act = worker && rb + grp < re;
		const double *D = S.diag + 9 * a; const double *Lb = S.off + 9 * (w & 0x3fff) + sub; double *y = S.rhs + 3 * (w >> 14) + sub;
		const double r0 = S.rhs[3 * a], r1 = S.rhs[3 * a + 1], r2 = S.rhs[3 * a + 2], i00 = D[0], i10 = D[3], i11 = D[4], i20 = D[6], i21 = D[7], i22 = D[8];
		double l0 = 0, l1 = 0, l2 = 0, yv = 0;
		if (act) { l0 = Lb[0]; l1 = Lb[3]; l2 = Lb[6]; yv = *y; }
		const unsigned w_n = (worker && a > 0 && rb_n + grp < rb) ? (unsigned)S.rent[rb_n + grp] : 0u;
		const int rb_nn = a >= 2 ? S.rptr[a - 2] : 0;
		const double x0 = fma(i20, r2, fma(i10, r1, i00 * r0)), x1 = fma(i21, r2, fma(i11, r1, i10 * r0)), x2 = fma(i22, r2, fma(i21, r1, i20 * r0)); // x_a = D_a^-1 (z_a - sum_{r > a} W_ra^t x_r)
		{ double ny = yv - fma(l2, x2, fma(l1, x1, l0 * x0)); SRBA_ALL_LANES(ny); if (act) *y = ny; }
		if (worker) for (int j = rb + grp + 21; j < re; j += 21) { // rows with more than 21 blocks
			const unsigned wx = (unsigned)S.rent[j]; const double *Lx = S.off + 9 * (wx & 0x3fff) + sub; double *yx = S.rhs + 3 * (wx >> 14) + sub;
			*yx -= fma(Lx[6], x2, fma(Lx[3], x1, Lx[0] * x0));
		}
		if (lane == SRBA_WG - 1) { S.rhs[3 * a] = x0; S.rhs[3 * a + 1] = x1; S.rhs[3 * a + 2] = x2; }
		solver_sync();
		re = rb; rb = rb_n; rb_n = rb_nn; w = w_n;
	}
}
#endif
// ---- the same two sweeps for the DENSE block layout (SparseSys::dense): column k holds the blocks of rows k+1 .. nb-1, update item t = a(a+1)/2 + b of column k
// targets block (k+1+a, k+1+b); every index is arithmetic, the LDS image carries numbers only. Used for mid-size systems whose factor is (nearly) full -- the
// Schur-reduced systems of landmark windows -- where the item list of the sparse form (~nb^3/6 words) would not fit next to the numbers.
__device__ __forceinline__ int dense_col_start(int nb, int c) { return c * (nb - 1) - c * (c - 1) / 2; }
// GLOBAL: the numbers live in HBM (ProbDesc::dense_blocks == 2): the hand-off between the sub-steps waits for the memory operations instead of relying on LDS order
template <bool GLOBAL> __device__ __forceinline__ void dense_sync() { if constexpr (GLOBAL) __syncthreads(); else solver_sync(); }
template <bool GLOBAL> __device__ __forceinline__ bool sp_factor_fsub_dense(const SparseSys &S) {
	const int lane = threadIdx.x, nb = S.nb;
	const int grp = (lane * 171) >> 9, sub = lane - 3 * grp; const bool worker = lane < 63;
	for (int k = 0; k < nb; k++) {
		const int cb = dense_col_start(nb, k), cn = nb - 1 - k, nitems = cn * (cn + 1) / 2;
		double *D = S.diag + 9 * k;
		const double a00 = D[0], a10 = D[3], a11 = D[4], a20 = D[6], a21 = D[7], a22 = D[8];
		const double b0 = S.rhs[3 * k], b1 = S.rhs[3 * k + 1], b2 = S.rhs[3 * k + 2];
		Chol3 c;
		if (!chol3v(a00, a10, a11, a20, a21, a22, c)) return false;
		const double y0 = b0 * c.r0, y1 = (b1 - c.l10 * y0) * c.r1, y2 = (b2 - c.l20 * y0 - c.l21 * y1) * c.r2;
		if (worker) for (int p = grp; p < cn; p += 21) { // row `sub` of panel block (k+1+p, k) and its share of the forward substitution
			double *Ax = S.off + 9 * (cb + p) + 3 * sub; double *rx = S.rhs + 3 * (k + 1 + p) + sub;
			const double x0 = Ax[0] * c.r0, x1 = (Ax[1] - x0 * c.l10) * c.r1, x2 = (Ax[2] - x0 * c.l20 - x1 * c.l21) * c.r2;
			Ax[0] = x0; Ax[1] = x1; Ax[2] = x2; *rx -= x0 * y0 + x1 * y1 + x2 * y2;
		}
		if (lane == SRBA_WG - 1) { // L_kk, reciprocal diagonal in the unused upper part, y_k
			D[0] = c.l00; D[3] = c.l10; D[4] = c.l11; D[6] = c.l20; D[7] = c.l21; D[8] = c.l22; D[1] = c.r0; D[2] = c.r1; D[5] = c.r2;
			S.rhs[3 * k] = y0; S.rhs[3 * k + 1] = y1; S.rhs[3 * k + 2] = y2;
		}
		dense_sync<GLOBAL>();
		// trailing update, row `sub` of block (k+1+a, k+1+b) -= L_ak L_bk^t. With the numbers in HBM two items are in flight per lane and pass (their loads are
		// independent): the pass is a memory round trip, and a column of a 60..90-row system has thousands of items
		constexpr int U = GLOBAL ? 2 : 1;
		if (worker) for (int t0 = grp; t0 < nitems; t0 += 21 * U) {
			double la[U][3], lb[U][9], tv[U][3]; double *T[U]; bool live[U];
#pragma unroll
			for (int u = 0; u < U; u++) {
				const int t = t0 + 21 * u; live[u] = t < nitems; const int tt = live[u] ? t : 0;
				int a = (int)((sqrtf(8.0f * (float)tt + 1.0f) - 1.0f) * 0.5f); a += ((a + 1) * (a + 2) / 2 <= tt) ? 1 : 0; a -= (a * (a + 1) / 2 > tt) ? 1 : 0; const int b = tt - a * (a + 1) / 2;
				const double *La = S.off + 9 * (cb + a) + 3 * sub, *Lb = S.off + 9 * (cb + b);
				T[u] = (a == b ? S.diag + 9 * (k + 1 + a) : S.off + 9 * (dense_col_start(nb, k + 1 + b) + (a - b - 1))) + 3 * sub;
				if (live[u]) {
#pragma unroll
					for (int q = 0; q < 3; q++) { la[u][q] = La[q]; tv[u][q] = T[u][q]; }
#pragma unroll
					for (int q = 0; q < 9; q++) lb[u][q] = Lb[q];
				}
			}
#pragma unroll
			for (int u = 0; u < U; u++) if (live[u]) {
				T[u][0] = tv[u][0] - (la[u][0] * lb[u][0] + la[u][1] * lb[u][1] + la[u][2] * lb[u][2]);
				T[u][1] = tv[u][1] - (la[u][0] * lb[u][3] + la[u][1] * lb[u][4] + la[u][2] * lb[u][5]);
				T[u][2] = tv[u][2] - (la[u][0] * lb[u][6] + la[u][1] * lb[u][7] + la[u][2] * lb[u][8]);
			}
		}
		dense_sync<GLOBAL>();
	}
	return true;
}
template <bool GLOBAL> __device__ __forceinline__ void sp_bsub_dense(const SparseSys &S) {
	const int lane = threadIdx.x, nb = S.nb;
	const int grp = (lane * 171) >> 9, sub = lane - 3 * grp; const bool worker = lane < 63;
	for (int a = nb - 1; a >= 0; a--) {
		const double *D = S.diag + 9 * a;
		const double r0 = S.rhs[3 * a], r1 = S.rhs[3 * a + 1], r2 = S.rhs[3 * a + 2], d5 = D[5], d7 = D[7], d2 = D[2], d3 = D[3], d6 = D[6], d1 = D[1];
		const double x2 = r2 * d5, x1 = (r1 - d7 * x2) * d2, x0 = (r0 - d3 * x1 - d6 * x2) * d1;
		if (worker) for (int cidx = grp; cidx < a; cidx += 21) { // y_c[sub] -= (L_ac^t x_a)[sub] for every column c < a
			const double *Lx = S.off + 9 * (dense_col_start(nb, cidx) + (a - cidx - 1)) + sub; double *yx = S.rhs + 3 * cidx + sub;
			*yx -= Lx[0] * x0 + Lx[3] * x1 + Lx[6] * x2;
		}
		if (lane == SRBA_WG - 1) { S.rhs[3 * a] = x0; S.rhs[3 * a + 1] = x1; S.rhs[3 * a + 2] = x2; }
		dense_sync<GLOBAL>();
	}
}
// ---- HBM-resident dense layout (ProbDesc::dense_blocks == 2), LEFT-looking. The right-looking sweep above, run on numbers in HBM, is a read-modify-write of every
// target block per column: ~nb^3/6 dependent round trips shared by 64 lanes, two barriers with a store drain per column. Here column k is finished in one go:
//   row k of L (blocks (k,j), j<k, written by earlier columns) is staged in LDS (9 doubles per block);
//   the lane that owns row `sub` of block (r,k), r = k..nb-1, keeps that row in registers and subtracts L_rj L_kj^t for j = 0..k-1 -- its own operand streams from HBM
//   (loads only, eight columns requested before the first is used; the 21 blocks x 3 rows of a pass are one contiguous span of column j), the shared one comes from LDS;
//   the three lanes of the diagonal block hand their rows to everybody (v_readlane), every lane factors it, the panel rows are scaled and stored once.
// The right-hand side follows the same pattern (y_k = L_kk^-1 (b_k - sum_j L_kj y_j), y kept in LDS) and the backward sweep is the column-dot form
// x_a = L_aa^-t (y_a - sum_{r>a} L_ra^t x_r) on the LDS copy: no read-modify-write of HBM anywhere, one barrier with a store drain per column of the factor, none in the
// backward sweep. Same operations in the same order per scalar of the factor as the right-looking form (updates of a block arrive in increasing j in both).
// nb <= 168 (eight passes of 21 blocks).
template <int LN> __device__ __forceinline__ double readlane_c(double v) {
	const int lo = __builtin_amdgcn_readlane(__double2loint(v), LN), hi = __builtin_amdgcn_readlane(__double2hiint(v), LN);
	return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_v(double v, int l) { // l uniform
	const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
	return __hiloint2double(hi, lo);
}
// Two columns (k, k+1) per step: every operand L_rj streamed from HBM serves both (the sweep is bound by that stream: hundreds of wavefronts each re-read their
// 0.1 .. 0.8 MB factor nb/3 times), then column k is finished, applied to column k+1 inside the registers (L_{k+1,k} is handed over by the three lanes that hold it)
// and column k+1 is finished. rowk: two rows of L (18 nb doubles of LDS), yl: 3 nb doubles.
typedef __attribute__((address_space(3))) double lds_f64; // explicit LDS operands: ds_read / ds_write whatever the optimiser can or cannot infer
__device__ __forceinline__ bool sp_factor_fsub_dense_left(const SparseSys &S, lds_f64 *rowk, lds_f64 *yl) {
	const int lane = threadIdx.x, nb = S.nb;
	const int grp = (lane * 171) >> 9, sub = lane - 3 * grp; const bool worker = lane < 63;
	constexpr int U = 8;
	lds_f64 *row0 = rowk, *row1 = rowk + 9 * nb;
	for (int k = 0; k < nb; k += 2) {
		const bool two = k + 1 < nb;
		const int cn = nb - 1 - k, cb = dense_col_start(nb, k), cb1 = dense_col_start(nb, k + 1);
		const double bk = (lane < 3) ? S.rhs[3 * k + lane] : ((lane < 6 && two) ? S.rhs[3 * k + lane] : 0.0); // b_k, b_k+1 (assembled long ago)
		for (int h = 0; 84 * h < k; h++) { // (1) rows k, k+1 of L (columns j < k) -> LDS, 84 columns per round trip
			double st[4][6];
#pragma unroll
			for (int i = 0; i < 4; i++) { const int j = grp + 21 * (4 * h + i); if (worker && j < k) { const double *src = S.off + 9 * (dense_col_start(nb, j) + (k - j - 1)) + 3 * sub;
				st[i][0] = src[0]; st[i][1] = src[1]; st[i][2] = src[2]; if (two) { st[i][3] = src[9]; st[i][4] = src[10]; st[i][5] = src[11]; } } }
#pragma unroll
			for (int i = 0; i < 4; i++) { const int j = grp + 21 * (4 * h + i); if (worker && j < k) { lds_f64 *d0 = row0 + 9 * j + 3 * sub, *d1 = row1 + 9 * j + 3 * sub;
				d0[0] = st[i][0]; d0[1] = st[i][1]; d0[2] = st[i][2]; if (two) { d1[0] = st[i][3]; d1[1] = st[i][4]; d1[2] = st[i][5]; } } }
		}
		solver_sync();
		// (2) b - sum_j L_kj y_j for both rows, columns j spread over the lanes
		double s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0;
		for (int j = lane; j < k; j += SRBA_WG) {
			const lds_f64 *L0 = row0 + 9 * j, *L1 = row1 + 9 * j; const double q0 = yl[3 * j], q1 = yl[3 * j + 1], q2 = yl[3 * j + 2];
			s0 += L0[0] * q0 + L0[1] * q1 + L0[2] * q2; s1 += L0[3] * q0 + L0[4] * q1 + L0[5] * q2; s2 += L0[6] * q0 + L0[7] * q1 + L0[8] * q2;
			if (two) { s3 += L1[0] * q0 + L1[1] * q1 + L1[2] * q2; s4 += L1[3] * q0 + L1[4] * q1 + L1[5] * q2; s5 += L1[6] * q0 + L1[7] * q1 + L1[8] * q2; }
		}
		s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2);
		const double b0 = readlane_c<0>(bk) - s0, b1 = readlane_c<1>(bk) - s1, b2 = readlane_c<2>(bk) - s2;
		double e0 = 0, e1 = 0, e2 = 0; // b_k+1 - ...
		if (two) { s3 = wave_sum(s3); s4 = wave_sum(s4); s5 = wave_sum(s5); e0 = readlane_c<3>(bk) - s3; e1 = readlane_c<4>(bk) - s4; e2 = readlane_c<5>(bk) - s5; }
		// (3) the blocks of the two columns, 21 block rows per pass: item p is block row r = k + p; pass 0 holds the diagonal blocks (p = 0 of column k, p = 1 of column k+1)
		Chol3 c, c1; double y0 = 0, y1 = 0, y2 = 0, z0 = 0, z1 = 0, z2 = 0, m[9];
		for (int i = 0; 21 * i <= cn; i++) {
			const int p = grp + 21 * i, r = k + p; const bool act = worker && p <= cn, act1 = act && two && p >= 1;
			double *A = (p == 0 ? S.diag + 9 * k : S.off + 9 * (cb + p - 1)) + 3 * sub;
			double *E = (p == 1 ? S.diag + 9 * (k + 1) : S.off + 9 * (cb1 + p - 2)) + 3 * sub;
			double a0 = 0, a1 = 0, a2 = 0, f0 = 0, f1 = 0, f2 = 0;
			if (act) {
				a0 = A[0]; a1 = A[1]; a2 = A[2]; if (act1) { f0 = E[0]; f1 = E[1]; f2 = E[2]; }
				for (int j = 0; j < k; j += U) {
					double la[U][3];
#pragma unroll
					for (int u = 0; u < U; u++) if (j + u < k) { const double *src = S.off + 9 * (dense_col_start(nb, j + u) + (r - j - u - 1)) + 3 * sub; la[u][0] = src[0]; la[u][1] = src[1];
						la[u][2] = src[2]; }
#pragma unroll
					for (int u = 0; u < U; u++) if (j + u < k) { const lds_f64 *lb = row0 + 9 * (j + u), *lc = row1 + 9 * (j + u);
						a0 -= la[u][0] * lb[0] + la[u][1] * lb[1] + la[u][2] * lb[2]; a1 -= la[u][0] * lb[3] + la[u][1] * lb[4] + la[u][2] * lb[5];
							a2 -= la[u][0] * lb[6] + la[u][1] * lb[7] + la[u][2] * lb[8];
						if (two) { f0 -= la[u][0] * lc[0] + la[u][1] * lc[1] + la[u][2] * lc[2]; f1 -= la[u][0] * lc[3] + la[u][1] * lc[4] + la[u][2] * lc[5];
							f2 -= la[u][0] * lc[6] + la[u][1] * lc[7] + la[u][2] * lc[8]; } }
				}
			}
			if (i == 0) { // rows 0,1,2 of the updated diagonal block of column k sit in lanes 0,1,2
				const double d00 = readlane_c<0>(a0), d10 = readlane_c<1>(a0), d11 = readlane_c<1>(a1), d20 = readlane_c<2>(a0), d21 = readlane_c<2>(a1), d22 = readlane_c<2>(a2);
				if (!chol3v(d00, d10, d11, d20, d21, d22, c)) return false;
				y0 = b0 * c.r0; y1 = (b1 - c.l10 * y0) * c.r1; y2 = (b2 - c.l20 * y0 - c.l21 * y1) * c.r2;
			}
			if (act && p > 0) { const double x0 = a0 * c.r0, x1 = (a1 - x0 * c.l10) * c.r1, x2 = (a2 - x0 * c.l20 - x1 * c.l21) * c.r2; A[0] = x0; A[1] = x1; A[2] = x2; a0 = x0; a1 = x1; a2 = x2; }
			if (two) {
				if (i == 0) { // L_{k+1,k}: rows in lanes 3,4,5
					m[0] = readlane_c<3>(a0); m[1] = readlane_c<3>(a1); m[2] = readlane_c<3>(a2); m[3] = readlane_c<4>(a0); m[4] = readlane_c<4>(a1); m[5] = readlane_c<4>(a2);
					m[6] = readlane_c<5>(a0); m[7] = readlane_c<5>(a1); m[8] = readlane_c<5>(a2);
					e0 -= m[0] * y0 + m[1] * y1 + m[2] * y2; e1 -= m[3] * y0 + m[4] * y1 + m[5] * y2; e2 -= m[6] * y0 + m[7] * y1 + m[8] * y2;
				}
				f0 -= a0 * m[0] + a1 * m[1] + a2 * m[2]; f1 -= a0 * m[3] + a1 * m[4] + a2 * m[5]; f2 -= a0 * m[6] + a1 * m[7] + a2 * m[8];
				if (i == 0) { // the updated diagonal block of column k+1: lanes 3,4,5
					const double d00 = readlane_c<3>(f0), d10 = readlane_c<4>(f0), d11 = readlane_c<4>(f1), d20 = readlane_c<5>(f0), d21 = readlane_c<5>(f1), d22 = readlane_c<5>(f2);
					if (!chol3v(d00, d10, d11, d20, d21, d22, c1)) return false;
					z0 = e0 * c1.r0; z1 = (e1 - c1.l10 * z0) * c1.r1; z2 = (e2 - c1.l20 * z0 - c1.l21 * z1) * c1.r2;
				}
				if (act1 && p > 1) { const double x0 = f0 * c1.r0, x1 = (f1 - x0 * c1.l10) * c1.r1, x2 = (f2 - x0 * c1.l20 - x1 * c1.l21) * c1.r2; E[0] = x0; E[1] = x1; E[2] = x2; }
			}
		}
		if (lane == SRBA_WG - 1) { // L_kk, reciprocal diagonal in the unused upper part, y_k (and the same for column k+1)
			double *D = S.diag + 9 * k;
			D[0] = c.l00; D[3] = c.l10; D[4] = c.l11; D[6] = c.l20; D[7] = c.l21; D[8] = c.l22; D[1] = c.r0; D[2] = c.r1; D[5] = c.r2;
			yl[3 * k] = y0; yl[3 * k + 1] = y1; yl[3 * k + 2] = y2;
			if (two) { D += 9; D[0] = c1.l00; D[3] = c1.l10; D[4] = c1.l11; D[6] = c1.l20; D[7] = c1.l21; D[8] = c1.l22; D[1] = c1.r0; D[2] = c1.r1; D[5] = c1.r2;
				yl[3 * k + 3] = z0; yl[3 * k + 4] = z1; yl[3 * k + 5] = z2; }
		}
		__syncthreads(); // the panel stores of these columns are read (rows k+2, k+3) by the next step
	}
	return true;
}
// backward sweep on the LDS copy of y (in: y, out: x, also written to S.rhs)
__device__ __forceinline__ void sp_bsub_dense_left(const SparseSys &S, lds_f64 *yl) {
	const int lane = threadIdx.x, nb = S.nb;
	const int grp = (lane * 171) >> 9, sub = lane - 3 * grp; const bool worker = lane < 63;
	for (int a = nb - 1; a >= 0; a--) {
		const double *D = S.diag + 9 * a; const int cb = dense_col_start(nb, a), cn = nb - 1 - a;
		const double d5 = D[5], d7 = D[7], d2 = D[2], d3 = D[3], d6 = D[6], d1 = D[1];
		double s = 0; // component `sub` of sum_{r>a} L_ra^t x_r, rows r spread over the lane groups
		if (worker) for (int p = grp; p < cn; p += 21) { const double *Lx = S.off + 9 * (cb + p) + sub; const lds_f64 *x = yl + 3 * (a + 1 + p); s += Lx[0] * x[0] + Lx[3] * x[1] + Lx[6] * x[2]; }
		const double t0 = wave_sum((worker && sub == 0) ? s : 0.0), t1 = wave_sum((worker && sub == 1) ? s : 0.0), t2 = wave_sum((worker && sub == 2) ? s : 0.0);
		const double r0 = yl[3 * a] - t0, r1 = yl[3 * a + 1] - t1, r2 = yl[3 * a + 2] - t2;
		const double x2 = r2 * d5, x1 = (r1 - d7 * x2) * d2, x0 = (r0 - d3 * x1 - d6 * x2) * d1;
		solver_sync(); // every lane has read y_a
		if (lane == SRBA_WG - 1) { yl[3 * a] = x0; yl[3 * a + 1] = x1; yl[3 * a + 2] = x2; }
		solver_sync();
	}
	for (int k = lane; k < 3 * nb; k += SRBA_WG) S.rhs[k] = yl[k];
	__syncthreads();
}
// location of scalar element (r,c), r>=c (block-permutation already applied); returns nullptr if the block is structurally absent
__device__ __forceinline__ double *sp_elem(const SparseSys &S, int r, int c) {
	const int br = r / 3, bc = c / 3; // (already permuted, r >= c)
	if (br == bc) return S.diag + 9 * br + (r % 3) * 3 + (c % 3);
	if (S.dense) return S.off + 9 * (bc * (S.nb - 1) - bc * (bc - 1) / 2 + (br - bc - 1)) + (r % 3) * 3 + (c % 3);
	int lo = S.col_off[bc], hi = S.col_off[bc + 1] - 1;
	while (lo <= hi) { const int mid = (lo + hi) >> 1; const int v = S.row[mid]; if (v == br) return S.off + 9 * mid + (r % 3) * 3 + (c % 3); if (v < br) lo = mid + 1; else hi = mid - 1; }
	return nullptr;
}

// ------------------------------------------------------------------------------------------------ the per-problem worker
// LEAN (round 4): the instantiation for the small, wave-slot-bound size classes of a big batch -- fewer loads in flight per lane (pairs, Jacobian blocks, Hessian terms) so that the kernel
// fits 168 registers and three wavefronts share a SIMD (k_lm_run_lean): the memory-level parallelism that the per-lane prefetches buy is bought with wavefronts instead. Same arithmetic.
template <int FAM, bool LEAN = false, int G = 64>
struct Worker {
	static constexpr int GRP = G; // lanes per capsule (64, or 128 = two wavefronts)
	typedef Tr<FAM> T; typedef PoseOps<T::SE3> PO; typedef typename PO::T pose_t;
	static constexpr int P = T::P, L = T::L, O = T::O, PD = T::PD;
	const Batch &B; const ProbDesc &d; const DevParams &prm; int tid;
	const int cp; // which copy of the unknowns / spanning-tree poses this worker reads and writes as "the state" (double-buffered LM loop: 0 = edge / ulm / pose, 1 = edge1 / ulm1 / pose1)
	__device__ __forceinline__ double *E() const { return cp ? B.edge1 : B.edge; }    // the state ...
	__device__ __forceinline__ double *U() const { return cp ? B.ulm1 : B.ulm; }
	__device__ __forceinline__ double *Pz() const { return cp ? B.pose1 : B.pose; }
	__device__ __forceinline__ double *Eo() const { return cp ? B.edge : B.edge1; }   // ... and the other copy (where a trial goes)
	__device__ __forceinline__ double *Uo() const { return cp ? B.ulm : B.ulm1; }
	// Re-materialise the lane id at the head of every phase: it stops the compiler from hoisting the per-lane address arithmetic of ALL
	// phases out of the LM loop (which costs >100 VGPRs of loop-invariant addresses and halves the occupancy).
	__device__ __forceinline__ void fresh() { int t = threadIdx.x; asm volatile("" : "+v"(t)); tid = t; }
	__device__ Worker(const Batch &B_, const ProbDesc &d_, const DevParams &p_, int cp_ = 0) : B(B_), d(d_), prm(p_), tid(threadIdx.x), cp(cp_) {}

	// A table index that widens to 64 bits for address arithmetic is made an opaque 64-bit value first. Why: clang 22 (ROCm 7.2) proves `idx >= 0` inside the guarded branch, drops the
	// extension and, in the 400..512-VGPR landmark kernels, built the register pair v[N:N+1] of the widened index from the loaded dword and a register that had meanwhile been reused for
	// the high half of a double -- a wild address, HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION (round 2's shelved fault; root cause with the debugger transcript: profiles/r03_fault_root_cause.md).
	// The empty asm takes the sign-extended value as a 64-bit register operand, so both halves exist before the address is formed. (The relative-pose SE2 kernel, 22 VGPRs under its
	// two-wavefront budget and never affected, is left as it is.)
	// Round 4: the 32-bit value is made opaque BEFORE it is widened (the compiler then knows nothing about its sign and has to compute the high word from it: v_ashrrev_i32 hi, 31, lo); with the
	// barrier after the widening the pair could still be formed from a register the compiler believed to hold the zero extension (tools/scan_undef_hi.py found six such pairs in the stereo kernel).
	static __device__ __forceinline__ long long wide(int i) { if constexpr (FAM != SRBA_SE2_RELPOSE2D) asm volatile("" : "+v"(i)); long long w = i;
		if constexpr (FAM != SRBA_SE2_RELPOSE2D) asm volatile("" : "+v"(w)); return w; }
	__device__ __forceinline__ pose_t pose_at(int idx) const { return idx >= 0 ? PO::ld(Pz() + (d.o_pair * 2 + wide(idx)) * PD) : PO::ident(); }
	__device__ __forceinline__ const double *lm_ptr(int ref) const { return ref >= 0 ? U() + (d.o_ulm + wide(ref)) * L : B.klm + (d.o_klm + wide(-1 - ref)) * L; }

	// ---- K1
	// edge_lds: optional copy of ALL edge poses of the capsule in LDS (stride PD, local edge order) -- the in-loop refresh then composes from LDS instead of
	// waiting for the global stores of the update it follows
	__device__ __forceinline__ void phase_spantree(bool only_needed, const double *edge_lds = nullptr, double *pose2 = nullptr /* a second copy of every pose written (all-pairs pass of the
		double-buffered loop) */) { fresh();
#ifdef SRBA_K1SMALL
		constexpr int U = 2, V = 1;
#else
		constexpr int U = LEAN ? 2 : (T::SE3 ? 2 : 4); // path edges fetched together (their loads do not depend on the running composition)
		constexpr int V = LEAN ? 1 : (T::SE3 ? 1 : 2); // pairs per lane and pass (all their loads are issued before the first store)
#endif
		if (only_needed && d.need_flat) { // in-loop refresh: one flat record per pair -> two dependent memory levels (record, edges) instead of four
			// (two copies of the loop, LDS source / HBM source: with the choice inside, the compiler merges the last load of both into a flat_load on a selected pointer)
			auto refresh = [&](auto from_lds) { constexpr bool FROM_LDS = decltype(from_lds)::value;
			for (int q0 = tid; q0 < d.n_need; q0 += V * G) {
				int p[V], pe[V][4]; pose_t acc[V];
#pragma unroll
				for (int v = 0; v < V; v++) {
					const int q = q0 + v * G; const int *rec = B.need_rec + (d.o_pair + (q < d.n_need ? q : 0)) * 5;
					p[v] = q < d.n_need ? rec[0] : -1;
#pragma unroll
					for (int u = 0; u < 4; u++) pe[v][u] = q < d.n_need ? rec[1 + u] : -1;
				}
#pragma unroll
				for (int v = 0; v < V; v++) acc[v] = PO::ident();
#pragma unroll
				for (int u0 = 0; u0 < 4; u0 += U) {
					pose_t ed[V][U];
#pragma unroll
					for (int v = 0; v < V; v++)
#pragma unroll
						for (int u = 0; u < U; u++) if (pe[v][u0 + u] >= 0) {
							if constexpr (FROM_LDS) { double t[PD]; const double *src = edge_lds + (pe[v][u0 + u] >> 1) * PD;
#pragma unroll
								for (int k = 0; k < PD; k++) t[k] = src[k];
								ed[v][u] = PO::from(t); }
							else ed[v][u] = PO::ld(E() + (d.o_edge + (pe[v][u0 + u] >> 1)) * PD);
						}
#pragma unroll
					for (int v = 0; v < V; v++)
#pragma unroll
						for (int u = 0; u < U; u++) if (pe[v][u0 + u] >= 0) acc[v] = (pe[v][u0 + u] & 1) ? comp(acc[v], inv(ed[v][u])) : comp(acc[v], ed[v][u]);
				}
#pragma unroll
				for (int v = 0; v < V; v++) if (p[v] >= 0) {
					PO::st(Pz() + (d.o_pair + p[v]) * 2 * PD, acc[v]);
					PO::st(Pz() + ((d.o_pair + p[v]) * 2 + 1) * PD, inv(acc[v]));
				}
			} };
			if (edge_lds) refresh(std::true_type()); else refresh(std::false_type());
			return;
		}
		const int cnt = only_needed ? d.n_need : d.n_pairs;
		for (int q0 = tid; q0 < cnt; q0 += V * G) {
			int p[V], pe[V][U], b[V], e[V]; pose_t ed[V][U], acc[V];
#pragma unroll
			for (int v = 0; v < V; v++) { const int q = q0 + v * G; p[v] = q < cnt ? (only_needed ? B.need_idx[d.o_pair + q] : q) : -1; }
#pragma unroll
			for (int v = 0; v < V; v++) { b[v] = e[v] = 0; if (p[v] >= 0) { b[v] = B.pair_path_off[d.o_ppoff + p[v]]; e[v] = B.pair_path_off[d.o_ppoff + p[v] + 1]; } }
#pragma unroll
			for (int v = 0; v < V; v++)
#pragma unroll
				for (int u = 0; u < U; u++) pe[v][u] = (b[v] + u < e[v]) ? B.path_edge[d.o_path + b[v] + u] : -1;
#pragma unroll
			for (int v = 0; v < V; v++)
#pragma unroll
				for (int u = 0; u < U; u++) if (pe[v][u] >= 0) ed[v][u] = PO::ld(E() + (d.o_edge + (pe[v][u] >> 1)) * PD);
#pragma unroll
			for (int v = 0; v < V; v++) {
				acc[v] = PO::ident();
#pragma unroll
				for (int u = 0; u < U; u++) if (pe[v][u] >= 0) acc[v] = (pe[v][u] & 1) ? comp(acc[v], inv(ed[v][u])) : comp(acc[v], ed[v][u]);
				for (int k = b[v] + U; k < e[v]; k++) {
					const int pk = B.path_edge[d.o_path + k];
					const pose_t ek = PO::ld(E() + (d.o_edge + (pk >> 1)) * PD);
					acc[v] = (pk & 1) ? comp(acc[v], inv(ek)) : comp(acc[v], ek);
				}
			}
#pragma unroll
			for (int v = 0; v < V; v++) if (p[v] >= 0) {
				const pose_t ia = inv(acc[v]);
				PO::st(Pz() + (d.o_pair + p[v]) * 2 * PD, acc[v]);
				PO::st(Pz() + ((d.o_pair + p[v]) * 2 + 1) * PD, ia);
				if (pose2) { PO::st(pose2 + (d.o_pair + p[v]) * 2 * PD, acc[v]); PO::st(pose2 + ((d.o_pair + p[v]) * 2 + 1) * PD, ia); }
			}
		}
	}

	// ---- sensor helpers
	__device__ __forceinline__ void to_sensor_point(double *x) const { // point_robot2sensor (srba_options_sensor_pose.h:117-120)
		if constexpr (T::SE3 || FAM == SRBA_SE2_STEREO) if (prm.sensor_pose == SRBA_SENSOR_POSE_SE3) {
			const double dx = x[0] - prm.SPt[0], dy = x[1] - prm.SPt[1], dz = x[2] - prm.SPt[2];
			x[0] = prm.SPR[0] * dx + prm.SPR[3] * dy + prm.SPR[6] * dz; x[1] = prm.SPR[1] * dx + prm.SPR[4] * dy + prm.SPR[7] * dz; x[2] = prm.SPR[2] * dx + prm.SPR[5] * dy + prm.SPR[8] * dz;
		}
	}
	// h-Jacobian wrt the point in the sensor frame, already multiplied by R_S^t (jacob_dh_dx_rotate). Returns false if invalid.
	__device__ __forceinline__ bool dh_dx(double *H, const double *x) const {
		if constexpr (FAM == SRBA_SE3_RELPOSE3D) return true; // identity, never multiplied out (sensors.h:905-913; jacobians.h:748-873 does not use dh_dx)
		else if constexpr (FAM == SRBA_SE2_RELPOSE2D || FAM == SRBA_SE2_CART2D) { for (int i = 0; i < O * L; i++) H[i] = 0; for (int i = 0; i < O; i++) H[i * L + i] = 1; return true; }
		else if constexpr (FAM == SRBA_SE2_RB2D) {
			const double r = hypot(x[0], x[1]); if (r == 0) return false;
			const double ri = 1.0 / r, ri2 = ri * ri; H[0] = x[0] * ri; H[1] = x[1] * ri; H[2] = -x[1] * ri2; H[3] = x[0] * ri2; return true;
		} else {
			double Hs[O * 3];
			if constexpr (FAM == SRBA_SE3_CART3D) { for (int i = 0; i < 9; i++) Hs[i] = (i % 4 == 0) ? 1.0 : 0.0; }
			else if constexpr (FAM == SRBA_SE3_RB3D) { // d(range, yaw, pitch)/d(x,y,z), pitch = -asin(z/range)  (sensors.h:588-608, [EXT] CPose3DQuat::sphericalCoordinates)
				const double x2y2 = x[0] * x[0] + x[1] * x[1], r2 = x2y2 + x[2] * x[2], r = sqrt(r2), rxy = sqrt(x2y2);
				Hs[0] = x[0] / r; Hs[1] = x[1] / r; Hs[2] = x[2] / r;
				Hs[3] = -x[1] / x2y2; Hs[4] = x[0] / x2y2; Hs[5] = 0;
				Hs[6] = x[0] * x[2] / (r2 * rxy); Hs[7] = x[1] * x[2] / (r2 * rxy); Hs[8] = -rxy / r2;
			}
			else {
				if (x[2] <= 0) return false;
				{ const double zi = 1.0 / x[2], zi2 = zi * zi; Hs[0] = prm.camL[0] * zi; Hs[1] = 0; Hs[2] = -prm.camL[0] * x[0] * zi2; Hs[3] = 0; Hs[4] = prm.camL[1] * zi;
					Hs[5] = -prm.camL[1] * x[1] * zi2; }
				if constexpr (FAM == SRBA_SE3_STEREO || FAM == SRBA_SE2_STEREO) {
					double xr[3];
					for (int i = 0; i < 3; i++) xr[i] = prm.R2Lt[i] + prm.R2LR[3 * i] * x[0] + prm.R2LR[3 * i + 1] * x[1] + prm.R2LR[3 * i + 2] * x[2];
					const double zi = 1.0 / xr[2], zi2 = zi * zi; Hs[6] = prm.camR[0] * zi; Hs[7] = 0; Hs[8] = -prm.camR[0] * xr[0] * zi2; Hs[9] = 0; Hs[10] = prm.camR[1] * zi;
						Hs[11] = -prm.camR[1] * xr[1] * zi2;
				}
			}
			if (prm.sensor_pose == SRBA_SENSOR_POSE_SE3) {
				for (int i = 0; i < O; i++) for (int j = 0; j < 3; j++) H[i * 3 + j] = Hs[i * 3] * prm.SPR[3 * j] + Hs[i * 3 + 1] * prm.SPR[3 * j + 1] + Hs[i * 3 + 2] * prm.SPR[3 * j + 2];
			} else for (int i = 0; i < O * 3; i++) H[i] = Hs[i];
			return true;
		}
	}

	// z - h for the stereo pair: left pinhole, right pinhole behind (-)rightCameraPose (sensors.h:175-211)
	__device__ __forceinline__ void project_stereo(const double *l, const double *z, double *r) const {
		r[0] = z[0] - (prm.camL[2] + prm.camL[0] * l[0] / l[2]); r[1] = z[1] - (prm.camL[3] + prm.camL[1] * l[1] / l[2]);
		double rr[3]; for (int k = 0; k < 3; k++) rr[k] = prm.R2Lt[k] + prm.R2LR[3 * k] * l[0] + prm.R2LR[3 * k + 1] * l[1] + prm.R2LR[3 * k + 2] * l[2];
		r[2] = z[2] - (prm.camR[2] + prm.camR[0] * rr[0] / rr[2]); r[3] = z[3] - (prm.camR[3] + prm.camR[1] * rr[1] / rr[2]);
	}
	// ---- K4 : one residual row
	__device__ __forceinline__ double residual_row(int i, double *r) const { return residual_row_at(i, r, pose_at(B.obs_pose[d.o_obs + i])); }
	__device__ __forceinline__ double residual_row_at(int i, double *r, const pose_t &bp) const { // r[O] <- (robustified) residual of row i whose base-from-observer pose is bp ; returns its chi2 term
		const int gi = d.o_obs + i;
		const double *z = B.obs_z + (long long)gi * O; const double *lm = lm_ptr(B.obs_lm[gi]);
		if constexpr (FAM == SRBA_SE2_RELPOSE2D) { // r = P(z) (-) pose (sensors.h:780-784)
			const double s = bp.s, c = bp.c, dx = z[0] - bp.x, dy = z[1] - bp.y;
			r[0] = dx * c + dy * s; r[1] = -dx * s + dy * c; r[2] = wrap_pi(z[2] - bp.phi);
		} else if constexpr (FAM == SRBA_SE2_STEREO) { // the 2D pose moves x,y and leaves z (landmarks.h Euclidean3D::composePosePoint with a CPose2D), then robot -> sensor, then the stereo model
			const double s = bp.s, c = bp.c; double l[3] = {bp.x + lm[0] * c - lm[1] * s, bp.y + lm[0] * s + lm[1] * c, lm[2]};
			to_sensor_point(l); project_stereo(l, z, r);
		} else if constexpr (FAM == SRBA_SE3_RELPOSE3D) { // r = pseudo_ln( P(z) (-) pose ) (sensors.h:873-879): h = pose^-1 (+) P(z), z = (x y z yaw pitch roll)
			double sy, cy, sp, cp, sr, cr; sincos(z[3], &sy, &cy); sincos(z[4], &sp, &cp); sincos(z[5], &sr, &cr);
			const double Rz[9] = {cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr, sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr, -sp, cp * sr, cp * cr};
			const double dt[3] = {z[0] - bp.t[0], z[1] - bp.t[1], z[2] - bp.t[2]};
			double Rh[9];
#pragma unroll
			for (int i = 0; i < 3; i++) {
				r[i] = bp.R[i] * dt[0] + bp.R[3 + i] * dt[1] + bp.R[6 + i] * dt[2];
#pragma unroll
				for (int j = 0; j < 3; j++) Rh[3 * i + j] = bp.R[i] * Rz[j] + bp.R[3 + i] * Rz[3 + j] + bp.R[6 + i] * Rz[6 + j];
			}
			const double ct = fmin(1.0, fmax(-1.0, 0.5 * (Rh[0] + Rh[4] + Rh[8] - 1.0))), th = acos(ct), f = th < 1e-8 ? 0.5 : th / (2.0 * sin(th));
				// [EXT] CPose3D::ln_rotation as restated in include/mrpt_lite.h
			r[3] = f * (Rh[7] - Rh[5]); r[4] = f * (Rh[2] - Rh[6]); r[5] = f * (Rh[3] - Rh[1]);
		} else if constexpr (!T::SE3) {
			const double s = bp.s, c = bp.c, lx = bp.x + lm[0] * c - lm[1] * s, ly = bp.y + lm[0] * s + lm[1] * c;
			if constexpr (FAM == SRBA_SE2_RB2D) { r[0] = z[0] - hypot(lx, ly); r[1] = z[1] - atan2(ly, lx); } else { r[0] = z[0] - lx; r[1] = z[1] - ly; }
		} else {
			double l[3];
			for (int k = 0; k < 3; k++) l[k] = bp.t[k] + bp.R[3 * k] * lm[0] + bp.R[3 * k + 1] * lm[1] + bp.R[3 * k + 2] * lm[2];
			to_sensor_point(l); // == (pose (-) S) (+) lm
			if constexpr (FAM == SRBA_SE3_CART3D) { for (int k = 0; k < 3; k++) r[k] = z[k] - l[k]; }
			else if constexpr (FAM == SRBA_SE3_RB3D) { // sensors.h:545-566: plain differences, angles not wrapped
				const double rg = sqrt(l[0] * l[0] + l[1] * l[1] + l[2] * l[2]);
				r[0] = z[0] - rg; r[1] = z[1] - atan2(l[1], l[0]); r[2] = z[2] + asin(l[2] / rg);
			}
			else {
				r[0] = z[0] - (prm.camL[2] + prm.camL[0] * l[0] / l[2]); r[1] = z[1] - (prm.camL[3] + prm.camL[1] * l[1] / l[2]);
				if constexpr (FAM == SRBA_SE3_STEREO) {
					double rr[3]; for (int k = 0; k < 3; k++) rr[k] = prm.R2Lt[k] + prm.R2LR[3 * k] * l[0] + prm.R2LR[3 * k + 1] * l[1] + prm.R2LR[3 * k + 2] * l[2];
					r[2] = z[2] - (prm.camR[2] + prm.camR[0] * rr[0] / rr[2]); r[3] = z[3] - (prm.camR[3] + prm.camR[1] * rr[1] / rr[2]);
				}
			}
		}
		double sum2 = 0;
#pragma unroll
		for (int k = 0; k < O; k++) sum2 += r[k] * r[k];
		double contrib = sum2;
		if (prm.use_robust_kernel) {
			const double nrm = fmax(1e-11, sqrt(sum2)), q = nrm / prm.kernel_param;
			const double hub = fabs(2.0 * prm.kernel_param * prm.kernel_param * (sqrt(1.0 + q * q) - 1.0));
			const double w = sqrt(hub) / nrm;
#pragma unroll
			for (int k = 0; k < O; k++) r[k] *= w;
			contrib = (w * w) * sum2;
		}
		return contrib;
	}
	// K4 over the DISTINCT observations of the capsule (round 6): an observation with m Jacobian blocks owns m identical residual rows -- 293 rows for 144 observations in the benchmark
	// windows. One evaluation per observation, its (robustified) residual stored to every row of it, its chi2 term added once per row (the reference sums the duplicates too:
	// reprojection_residuals.h:26-78 over involved_obs). FUSED: the pose is composed from the trial's edges in LDS (phase_residuals_fused), else read from the pose table.
	template <bool FUSED> __device__ __forceinline__ double residuals_distinct(double *out, double *red, const double *edge_lds) {
		double acc = 0; const int nv = d.n_valid; typedef int i32x4 __attribute__((ext_vector_type(4)));
		const i32x4 *tab = (const i32x4 *)(B.od_tab + d.o_valid * 12);
		auto store_rows = [&](const i32x4 &a, const double (&r)[O], double ch) __attribute__((always_inline)) { const int rows[3] = {a.x, a.y, a.z};
#pragma unroll
			for (int q = 0; q < 3; q++) if (rows[q] >= 0) {
#pragma unroll
				for (int k = 0; k < O; k++) out[(long long)(d.o_obs + rows[q]) * O + k] = r[k];
				acc += ch; } };
		if constexpr (!FUSED) { // two observations per lane and pass: the loads of both are issued before either's stores (the first 16 bytes of a record are all this form needs)
			for (int v = tid; v < nv; v += 2 * G) {
				const int w = v + G; const bool two = w < nv; const i32x4 a0 = tab[3 * v], a1 = tab[3 * (two ? w : v)];
				double r0[O], r1[O]; const double c0 = residual_row_at(a0.x, r0, pose_at(a0.w)); double c1 = 0; if (two) c1 = residual_row_at(a1.x, r1, pose_at(a1.w));
				store_rows(a0, r0, c0); if (two) store_rows(a1, r1, c1);
			}
		} else {
			const int v0 = tid < nv ? tid : 0; i32x4 a = tab[3 * v0], b = tab[3 * v0 + 1], c = tab[3 * v0 + 2];
			for (int v = tid; v < nv; v += G) {
				const int w = v + G < nv ? v + G : v; const i32x4 an = tab[3 * w], bn = tab[3 * w + 1], cn = tab[3 * w + 2]; // the next record travels under this observation's work
				const int m = b.x; pose_t bp;
				if (m >= 0) {
					pose_t acc_p = PO::ident(); const int pe[4] = {b.y, b.z, b.w, c.x};
#pragma unroll
					for (int u = 0; u < 4; u++) if (pe[u] >= 0) { double t[PD]; const double *src = edge_lds + (pe[u] >> 1) * PD;
#pragma unroll
						for (int k = 0; k < PD; k++) t[k] = src[k];
						const pose_t ed = PO::from(t); acc_p = (pe[u] & 1) ? comp(acc_p, inv(ed)) : comp(acc_p, ed); }
					bp = m ? inv(acc_p) : acc_p;
				} else bp = pose_at(a.w);
				double r[O]; const double ch = residual_row_at(a.x, r, bp);
				store_rows(a, r, ch);
				a = an; b = bn; c = cn;
			}
		}
		return grp_sum<G>(acc, red);
	}
	// by_rows: every row evaluated, copies included -- the batch-wide launch of the stepwise API streams the residual array (consecutive lanes write consecutive rows: full lines) and is
	// bound by that traffic, not by the evaluations: over distinct observations it measured 0.232 ms against 0.201
	__device__ __forceinline__ double phase_residuals(double *out, double *red, bool by_rows = false) { fresh();
		if (d.od_ok && !by_rows) return residuals_distinct<false>(out, red, nullptr);
		double acc = 0;
		for (int i = tid; i < d.n_obs; i += 2 * G) { // two rows per lane and pass: both rows' loads are issued before either row's stores
			double r0[O], r1[O]; const int j = i + G; const bool two = j < d.n_obs;
			const double c0 = residual_row(i, r0); double c1 = 0; if (two) c1 = residual_row(j, r1);
#pragma unroll
			for (int k = 0; k < O; k++) out[(long long)(d.o_obs + i) * O + k] = r0[k];
			if (two) {
#pragma unroll
				for (int k = 0; k < O; k++) out[(long long)(d.o_obs + j) * O + k] = r1[k];
			}
			acc += c0; if (two) acc += c1;
		}
		return grp_sum<G>(acc, red);
	}

	// K4 of a trial WITHOUT the spanning-tree refresh before it (round 4): the pose of a refreshed pair is composed on the spot from the edge poses of the trial staged in LDS
	// (obs_rec: per observation either its pose index in the table -- a pose no trial changes -- or the path of its pair, at most four edges, and which of the pair's two poses it reads),
	// in the order phase_spantree composes it: the same bits. A trial's evaluation then has no global store -> barrier -> gather between the update and the residuals, and a REJECTED
	// trial (60 % of them) never writes the pose table; an accepted one runs the refresh afterwards (lm_one). One row per lane and pass, the next row's record requested ahead.
	__device__ __forceinline__ double phase_residuals_fused(double *out, double *red, const double *edge_lds) { fresh();
		if (d.od_ok) return residuals_distinct<true>(out, red, edge_lds);
		double acc = 0;
		const int *rec0 = B.obs_rec + (d.o_obs + (tid < d.n_obs ? tid : 0)) * 5; int m = rec0[0], e0 = rec0[1], e1 = rec0[2], e2 = rec0[3], e3 = rec0[4];
		for (int i = tid; i < d.n_obs; i += G) {
			const int j = i + G; const int *recn = B.obs_rec + (d.o_obs + (j < d.n_obs ? j : i)) * 5; const int mn = recn[0], n0 = recn[1], n1 = recn[2], n2 = recn[3], n3 = recn[4];
			pose_t bp;
			if (m >= 0) {
				pose_t a = PO::ident(); const int pe[4] = {e0, e1, e2, e3};
#pragma unroll
				for (int u = 0; u < 4; u++) if (pe[u] >= 0) { double t[PD]; const double *src = edge_lds + (pe[u] >> 1) * PD;
#pragma unroll
					for (int k = 0; k < PD; k++) t[k] = src[k];
					const pose_t ed = PO::from(t); a = (pe[u] & 1) ? comp(a, inv(ed)) : comp(a, ed); }
				bp = m ? inv(a) : a;
			} else bp = pose_at(m == -1 ? e0 : -1);
			double r[O]; const double c = residual_row_at(i, r, bp);
#pragma unroll
			for (int k = 0; k < O; k++) out[(long long)(d.o_obs + i) * O + k] = r[k];
			acc += c;
			m = mn; e0 = n0; e1 = n1; e2 = n2; e3 = n3;
		}
		return grp_sum<G>(acc, red);
	}

	// ---- K2
	__device__ __forceinline__ void jac_dh_dp(int b) {
		const int gb = d.o_bp + b;
		const int row = B.bp_res[gb], vs = B.obs_valid[d.o_obs + row];
		double *J = B.Jp + (long long)gb * O * P;
		const int iA = B.bp_A[gb]; const bool normal = B.bp_normal[gb] != 0;
		pose_t D = pose_at(B.bp_D[gb]); pose_t A = pose_at(iA); bool hasA = iA >= 0;
		const double *xi = lm_ptr(B.bp_lm[gb]);
		double Jl[O * P]; bool ok = true;
		if constexpr (!T::SE3) {
			if (!normal) { // D' = p (+) D ; A' = A (+) (-)p (jacobians.h:565-587,684-711)
				const P2 p = ld2(E() + (d.o_edge + B.bp_col[gb]) * PD);
				D = comp(p, D);
				if constexpr (!T::REL) { A = hasA ? comp(A, inv(p)) : inv(p); hasA = true; } // the relative-pose block depends on D' only
			}
			const double sg = normal ? 1.0 : -1.0;
			if constexpr (FAM == SRBA_SE2_STEREO) { // SE(2) poses, 3D points (jacobians.h:501-641 with POINT_DIMS = 3): J = dh_dx * dPx_P * dAD_deps
				const P2 AD = hasA ? comp(A, D) : D;
				const double sa = hasA ? A.s : 0.0, ca = hasA ? A.c : 1.0, sad = AD.s, cad = AD.c;
				double xl[3] = {AD.x + xi[0] * cad - xi[1] * sad, AD.y + xi[0] * sad + xi[1] * cad, xi[2]};
				to_sensor_point(xl);
				double H[O * 3]; ok = dh_dx(H, xl);
				if (ok) {
					// dPx_P = [1 0 px; 0 1 py; 0 0 1] -- the reference sets d z / d phi = 1 (jacobians.h:546, sic) -- and dAD_deps = [ca -sa qx; sa ca qy; 0 0 1]
					const double px = -xi[0] * sad - xi[1] * cad, py = xi[0] * cad - xi[1] * sad, qx = -sa * D.x - ca * D.y, qy = ca * D.x - sa * D.y;
					for (int i = 0; i < O; i++) {
						const double h0 = H[i * 3], h1 = H[i * 3 + 1], h2 = H[i * 3 + 2], m2 = h0 * px + h1 * py + h2; // row i of dh_dx*dPx_P = [h0 h1 m2]
						Jl[i * 3] = sg * (h0 * ca + h1 * sa); Jl[i * 3 + 1] = sg * (-h0 * sa + h1 * ca); Jl[i * 3 + 2] = sg * (h0 * qx + h1 * qy + m2);
					}
				}
			} else if constexpr (T::REL) { // closed form of dh_dx*J0*J1*J2: depends on D only
				const double sd = D.s, cd = D.c;
				Jl[0] = sg * cd; Jl[1] = sg * sd; Jl[2] = sg * (D.x * sd - D.y * cd);
				Jl[3] = -sg * sd; Jl[4] = sg * cd; Jl[5] = sg * (D.x * cd + D.y * sd);
				Jl[6] = 0; Jl[7] = 0; Jl[8] = sg;
			} else {
				const P2 AD = hasA ? comp(A, D) : D;
				const double sa = hasA ? A.s : 0.0, ca = hasA ? A.c : 1.0, sad = AD.s, cad = AD.c;
				double xl[2] = {AD.x + xi[0] * cad - xi[1] * sad, AD.y + xi[0] * sad + xi[1] * cad};
				double H[O * L]; ok = dh_dx(H, xl);
				if (ok) {
					const double m02 = (-sa * D.x - ca * D.y) + (-xi[0] * sad - xi[1] * cad), m12 = (ca * D.x - sa * D.y) + (xi[0] * cad - xi[1] * sad);
					for (int i = 0; i < O; i++) { Jl[i * 3] = sg * (H[i * 2] * ca + H[i * 2 + 1] * sa); Jl[i * 3 + 1] = sg * (-H[i * 2] * sa + H[i * 2 + 1] * ca);
						Jl[i * 3 + 2] = sg * (H[i * 2] * m02 + H[i * 2 + 1] * m12); }
				}
			}
		} else if constexpr (FAM == SRBA_SE3_RELPOSE3D) { // jacobians.h:748-873: J = [d pseudo_ln / d (R,t)] (6x12) * [d (A e^eps D) / d eps] (12x6)
			double RA[9];
			if (!normal) { // D' = p (+) D ; A' = A (+) (-)p
				const P3 p = ld3(E() + (d.o_edge + B.bp_col[gb]) * PD); const P3 pin = inv(p);
				D = comp(p, D); const P3 Ap = hasA ? comp(A, pin) : pin;
				for (int k = 0; k < 9; k++) RA[k] = Ap.R[k]; A = Ap; hasA = true;
			} else if (hasA) { for (int k = 0; k < 9; k++) RA[k] = A.R[k]; }
			else { for (int k = 0; k < 9; k++) RA[k] = (k % 4 == 0) ? 1.0 : 0.0; }
			const P3 AD = hasA ? comp(A, D) : D;
			// d ln(R) / d vec(R), vec = stacked columns ([EXT] CPose3D::ln_rot_jacob: omega = theta / (2 sin theta) * vee(R - R^t), theta = acos((tr R - 1) / 2))
			const double dd = 0.5 * (AD.R[0] + AD.R[4] + AD.R[8] - 1.0); double a0 = 0, a1 = 0, a2 = 0, bb = 0.5;
			if (!(dd > 0.99999)) { const double th = acos(dd), sq = sqrt(1.0 - dd * dd), kk = (dd * th - sq) / (4.0 * sq * sq * sq); bb = th / (2.0 * sq); a0 = kk * (AD.R[7] - AD.R[5]);
				a1 = kk * (AD.R[2] - AD.R[6]); a2 = kk * (AD.R[3] - AD.R[1]); }
			const double M[27] = {a0, 0, 0, 0, a0, bb, 0, -bb, a0,   a1, 0, -bb, 0, a1, 0, bb, 0, a1,   a2, bb, 0, -bb, a2, 0, 0, 0, a2};
			// G_i = -R(A) [c_i]_x for the three columns c_i of R(D) and for t(D): rows 3i..3i+2 of the right half of d(A e^eps D)/d eps
			double Jr[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, Gt[9];
#pragma unroll
			for (int i = 0; i < 4; i++) {
				const double v0 = i < 3 ? D.R[i] : D.t[0], v1 = i < 3 ? D.R[3 + i] : D.t[1], v2 = i < 3 ? D.R[6 + i] : D.t[2];
				const double sk[9] = {0, -v2, v1, v2, 0, -v0, -v1, v0, 0}; double Gm[9];
#pragma unroll
				for (int r = 0; r < 3; r++)
#pragma unroll
					for (int q = 0; q < 3; q++) Gm[3 * r + q] = -(RA[3 * r] * sk[q] + RA[3 * r + 1] * sk[3 + q] + RA[3 * r + 2] * sk[6 + q]);
				if (i < 3) {
#pragma unroll
					for (int m = 0; m < 3; m++)
#pragma unroll
						for (int q = 0; q < 3; q++) Jr[3 * m + q] += M[9 * m + 3 * i] * Gm[q] + M[9 * m + 3 * i + 1] * Gm[3 + q] + M[9 * m + 3 * i + 2] * Gm[6 + q];
				} else for (int k = 0; k < 9; k++) Gt[k] = Gm[k];
			}
			const double sg = normal ? 1.0 : -1.0;
#pragma unroll
			for (int r = 0; r < 3; r++)
#pragma unroll
				for (int q = 0; q < 3; q++) {
					Jl[6 * r + q] = sg * (RA[3 * q] * D.R[r] + RA[3 * q + 1] * D.R[3 + r] + RA[3 * q + 2] * D.R[6 + r]); // ((R(A) R(D))^t)[r][q] (jacobians.h:842, sic: the comment there says R(A))
					Jl[6 * r + 3 + q] = sg * Gt[3 * r + q];
					Jl[6 * (3 + r) + q] = 0; Jl[6 * (3 + r) + 3 + q] = sg * Jr[3 * r + q];
				}
		} else {
			double RA[9]; bool haveRA = hasA;
			if (hasA) for (int k = 0; k < 9; k++) RA[k] = A.R[k];
			// landmark in the observer frame uses the ORIGINAL A (+) D (jacobians.h:259-270)
			const P3 AD = hasA ? comp(A, D) : D;
			double xl[3]; for (int k = 0; k < 3; k++) xl[k] = AD.t[k] + AD.R[3 * k] * xi[0] + AD.R[3 * k + 1] * xi[1] + AD.R[3 * k + 2] * xi[2];
			to_sensor_point(xl);
			double H[O * 3]; ok = dh_dx(H, xl);
			if (ok) {
				if (!normal) { // D' = p (+) D ; R(A') = R(A) R(p)^t (jacobians.h:436-461)
					const P3 p = ld3(E() + (d.o_edge + B.bp_col[gb]) * PD);
					D = comp(p, D);
					double T2[9];
					if (hasA) { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) T2[3 * i + j] = A.R[3 * i] * p.R[3 * j] + A.R[3 * i + 1] * p.R[3 * j + 1] + A.R[3 * i + 2] * p.R[3 * j + 2]; }
					else { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) T2[3 * i + j] = p.R[3 * j + i]; }
					for (int k = 0; k < 9; k++) RA[k] = T2[k]; haveRA = true;
				}
				double HR[O * 3];
				if (haveRA) { for (int i = 0; i < O; i++) for (int j = 0; j < 3; j++) HR[i * 3 + j] = H[i * 3] * RA[j] + H[i * 3 + 1] * RA[3 + j] + H[i * 3 + 2] * RA[6 + j]; }
				else for (int k = 0; k < O * 3; k++) HR[k] = H[k];
				double v[3]; for (int k = 0; k < 3; k++) v[k] = -D.t[k] - (D.R[3 * k] * xi[0] + D.R[3 * k + 1] * xi[1] + D.R[3 * k + 2] * xi[2]);
				const double sg = normal ? 1.0 : -1.0;
				for (int i = 0; i < O; i++) {
					const double h0 = HR[i * 3], h1 = HR[i * 3 + 1], h2 = HR[i * 3 + 2];
					Jl[i * 6] = sg * h0; Jl[i * 6 + 1] = sg * h1; Jl[i * 6 + 2] = sg * h2;
					Jl[i * 6 + 3] = sg * (h1 * v[2] - h2 * v[1]); Jl[i * 6 + 4] = sg * (-h0 * v[2] + h2 * v[0]); Jl[i * 6 + 5] = sg * (h0 * v[1] - h1 * v[0]);
				}
			}
		}
		if (ok) stn<O * P>(J, Jl);
		else atomicMin(&B.first_fail[d.o_valid + vs], b); // sweep index of a dh_dAp block = b
	}
	// ---- K3
	__device__ __forceinline__ void jac_dh_df(int b) {
		if constexpr (!T::REL) {
			const int gb = d.o_bf + b;
			const int row = B.bf_res[gb], vs = B.obs_valid[d.o_obs + row];
			double *J = B.Jf + (long long)gb * O * L;
			const int ip = B.bf_pose[gb];
			const pose_t bp = pose_at(ip);
			const double *xi = U() + (d.o_ulm + B.bf_col[gb]) * L;
			double Jl[O * L]; bool ok;
			if constexpr (FAM == SRBA_SE2_STEREO) { // dh_dx * R(base <- obs), R = the 3x3 rotation about z of the 2D pose (jacobians.h:984-989)
				const double s = bp.s, c = bp.c;
				double xl[3] = {bp.x + xi[0] * c - xi[1] * s, bp.y + xi[0] * s + xi[1] * c, xi[2]};
				to_sensor_point(xl);
				double H[O * 3]; ok = dh_dx(H, xl);
				if (ok) for (int i = 0; i < O; i++) { Jl[i * 3] = H[i * 3] * c + H[i * 3 + 1] * s; Jl[i * 3 + 1] = -H[i * 3] * s + H[i * 3 + 1] * c; Jl[i * 3 + 2] = H[i * 3 + 2]; }
			} else if constexpr (!T::SE3) {
				const double s = bp.s, c = bp.c;
				double xl[2] = {bp.x + xi[0] * c - xi[1] * s, bp.y + xi[0] * s + xi[1] * c};
				double H[O * L]; ok = dh_dx(H, xl);
				if (ok) for (int i = 0; i < O; i++) { Jl[i * 2] = H[i * 2] * c + H[i * 2 + 1] * s; Jl[i * 2 + 1] = -H[i * 2] * s + H[i * 2 + 1] * c; }
			} else {
				double xl[3]; for (int k = 0; k < 3; k++) xl[k] = bp.t[k] + bp.R[3 * k] * xi[0] + bp.R[3 * k + 1] * xi[1] + bp.R[3 * k + 2] * xi[2];
				to_sensor_point(xl);
				double H[O * 3]; ok = dh_dx(H, xl);
				if (ok) for (int i = 0; i < O; i++) for (int j = 0; j < 3; j++) Jl[i * 3 + j] = H[i * 3] * bp.R[j] + H[i * 3 + 1] * bp.R[3 + j] + H[i * 3 + 2] * bp.R[6 + j];
			}
			if (ok) stn<O * L>(J, Jl);
			else atomicMin(&B.first_fail[d.o_valid + vs], d.n_bp + b); // dh_df blocks are swept after all dh_dAp blocks
		}
	}
	// Jacobians of all blocks + validity semantics of jacobians.h:215-216,321-327 (see DESIGN.md "invalid rows")
	__device__ __forceinline__ void phase_jacobians() { fresh();
		for (int i = tid; i < d.n_valid; i += G) { B.valid[d.o_valid + i] = 1; B.first_fail[d.o_valid + i] = 0x7fffffff; }
		__syncthreads();
		for (int b = tid; b < d.n_bp; b += G) jac_dh_dp(b);
		for (int b = tid; b < d.n_bf; b += G) jac_dh_df(b);
		__syncthreads();
		for (int i = tid; i < d.n_valid; i += G) if (B.first_fail[d.o_valid + i] != 0x7fffffff) B.valid[d.o_valid + i] = 0;
		// the first failing block of a row (in sweep order) is zeroed; later ones keep stale values
		for (int b = tid; b < d.n_bp; b += G) {
			const int ff = B.first_fail[d.o_valid + B.obs_valid[d.o_obs + B.bp_res[d.o_bp + b]]]; B.bp_ok[d.o_bp + b] = (ff == 0x7fffffff);
			if (ff == b) { double *J = B.Jp + (long long)(d.o_bp + b) * O * P; for (int k = 0; k < O * P; k++) J[k] = 0; }
		}
		for (int b = tid; b < d.n_bf; b += G) {
			const int ff = B.first_fail[d.o_valid + B.obs_valid[d.o_obs + B.bf_res[d.o_bf + b]]]; B.bf_ok[d.o_bf + b] = (ff == 0x7fffffff);
			if (ff == d.n_bp + b) { double *J = B.Jf + (long long)(d.o_bf + b) * O * L; for (int k = 0; k < O * L; k++) J[k] = 0; }
		}
		__syncthreads();
	}

	// ---- K6: H_ij = sum J1^t Lambda J2
	template <int M1, int M2>
	__device__ __forceinline__ void hess_term(double *H, const double *A, const double *Bm) const {
		if (prm.noise == SRBA_NOISE_CONSTANT_MATRIX) {
			for (int i = 0; i < M1; i++) {
				double jl[O];
				for (int j = 0; j < O; j++) { double s = 0; for (int k = 0; k < O; k++) s += A[k * M1 + i] * prm.lambda[k * O + j]; jl[j] = s; }
				for (int j = 0; j < M2; j++) { double s = 0; for (int k = 0; k < O; k++) s += jl[k] * Bm[k * M2 + j]; H[i * M2 + j] += s; }
			}
		} else {
			for (int i = 0; i < M1; i++) for (int j = 0; j < M2; j++) { double s = 0; for (int k = 0; k < O; k++) s += A[k * M1 + i] * Bm[k * M2 + j]; H[i * M2 + j] += s; }
		}
	}
	// one ROW i of J1^t Lambda J2 (the term above, row by row: the caller adds each row to its accumulator as soon as it is formed -- six live sums instead of thirty-six)
	template <int M1, int M2>
	__device__ __forceinline__ void hess_row(double *row, const double *A, const double *Bm, int i) const {
		if (prm.noise == SRBA_NOISE_CONSTANT_MATRIX) {
			double jl[O];
#pragma unroll
			for (int j = 0; j < O; j++) { double s = 0;
#pragma unroll
				for (int k = 0; k < O; k++) s += A[k * M1 + i] * prm.lambda[k * O + j];
				jl[j] = s; }
#pragma unroll
			for (int j = 0; j < M2; j++) { double s = 0;
#pragma unroll
				for (int k = 0; k < O; k++) s += jl[k] * Bm[k * M2 + j];
				row[j] = s; }
		} else {
#pragma unroll
			for (int j = 0; j < M2; j++) { double s = 0;
#pragma unroll
				for (int k = 0; k < O; k++) s += A[k * M1 + i] * Bm[k * M2 + j];
				row[j] = s; }
		}
	}
	template <int M1, int M2, bool ATOMIC = false>
	__device__ __forceinline__ int hess_block(double *Hout, double *Hlatch, const int *t1, const int *t2, int tb, int te, const double *J1, const double *J2, const unsigned char *ok1,
		const unsigned char *ok2) {
		double H[M1 * M2];
#pragma unroll
		for (int k = 0; k < M1 * M2; k++) H[k] = 0;
		int ninv = 0;
		#ifdef SRBA_NOPAIR
		constexpr bool PAIR = false;
#else
		constexpr bool PAIR = !LEAN && ((O * (M1 + M2) <= 24) || (T::SE3 && O * (M1 + M2) <= 48)); // (the SE3 kernels run one wavefront per SIMD anyway: registers buy memory-level parallelism)
#endif // two terms in flight when their Jacobian blocks fit the register budget
		int t = tb;
		if constexpr (PAIR) {
			for (; t + 1 < te; t += 2) { // the loads of both terms are independent of the accumulator: issue them together (same summation order)
				const int a1 = t1[t], a2 = t2[t], c1 = t1[t + 1], c2 = t2[t + 1];
				const bool oka = ok1[a1] && ok2[a2], okc = ok1[c1] && ok2[c2];
				double A[O * M1], Bm[O * M2], C[O * M1], Dm[O * M2];
				ldn<O * M1>(A, J1 + (long long)a1 * O * M1); ldn<O * M1>(C, J1 + (long long)c1 * O * M1);
				ldn<O * M2>(Bm, J2 + (long long)a2 * O * M2); ldn<O * M2>(Dm, J2 + (long long)c2 * O * M2);
				if (oka) hess_term<M1, M2>(H, A, Bm); else ninv++;
				if (okc) hess_term<M1, M2>(H, C, Dm); else ninv++;
			}
		}
		for (; t < te; t++) {
			const int b1 = t1[t], b2 = t2[t];
			double A[O * M1], Bm[O * M2]; ldn<O * M1>(A, J1 + (long long)b1 * O * M1); ldn<O * M2>(Bm, J2 + (long long)b2 * O * M2);
			if (ok1[b1] && ok2[b2]) hess_term<M1, M2>(H, A, Bm); else ninv++;
		}
		const double sc = (prm.noise == SRBA_NOISE_IDENTITY) ? prm.inv_sigma : 1.0;
#pragma unroll
		for (int k = 0; k < M1 * M2; k++) H[k] *= sc;
		if constexpr (ATOMIC) { // a partial sum of the block (its term list is cut into records of a few terms: ProbDesc::hap_chunked): added into the cleared block
#pragma unroll
			for (int k = 0; k < M1 * M2; k++) unsafeAtomicAdd(Hout + k, H[k]);
		} else { stn<M1 * M2>(Hout, H); if (Hlatch) stn<M1 * M2>(Hlatch, H); }
		return ninv;
	}
	__device__ __forceinline__ int phase_hessian() { fresh(); // returns the per-thread invalid count (to be reduced by the caller if wanted)
		int ninv = 0;
		const double *Jp = B.Jp + d.o_bp * O * P, *Jf = B.Jf + d.o_bf * O * L;
		const unsigned char *rp = B.bp_ok + d.o_bp, *rf = B.bf_ok + d.o_bf;
		const bool latch = prm.solver != SRBA_SOLVER_NO_SCHUR_SPARSE_CHOL;
		for (int bi = tid; bi < d.n_hrec; bi += G) {
			const int *rec = B.hap_rec + (d.o_hrec + bi) * 3; const int b = rec[0]; const long long g = d.o_hap + b; // {block, first term, end term}: longest lists first
			// the Schur complement works on HAp in place and restores it from the latch for every lambda (schur.h:38,165-168,188)
			ninv += hess_block<P, P>(B.HAp + g * P * P, latch ? B.HAp0 + g * P * P : nullptr, B.hap_t1 + d.o_hapt, B.hap_t2 + d.o_hapt, rec[1], rec[2], Jp, Jp, rp, rp);
		}
		if constexpr (!T::REL) {
			for (int b = tid; b < d.n_hf; b += G)
				ninv += hess_block<L, L>(B.Hf + (d.o_hf + b) * L * L, nullptr, B.hf_t1 + d.o_hft, B.hf_t2 + d.o_hft, B.hf_term_off[d.o_hfoff + b], B.hf_term_off[d.o_hfoff + b + 1], Jf, Jf, rf, rf);
			for (int b = tid; b < d.n_hapf; b += G)
				ninv += hess_block<P, L>(B.HApf + (d.o_hapf + b) * P * L, nullptr, B.hapf_t1 + d.o_hapft, B.hapf_t2 + d.o_hapft, B.hapf_term_off[d.o_hapfoff + b],
					B.hapf_term_off[d.o_hapfoff + b + 1], Jp, Jf, rp, rf);
		}
		return ninv;
	}

	// K6, U_Ap blocks, TERM-parallel: the per-block form above walks the term lists one lane per block, so a pass lasts as long as its longest list (18..30 terms for
	// the diagonal blocks against 2..4 for most); here the lanes stride over the flat term list of the capsule (11 passes of 64 for the typical window instead of
	// ~36 sequential terms), every term adds its M x M product into the block's accumulator in LDS with ds_add_f64 (conflicting lanes of one instruction are
	// serialised by the LDS in a fixed order and the passes are in program order: reproducible), and the finished blocks go out as one contiguous span.
	// acc: n_hap * P * P doubles of LDS. Returns the per-lane count of skipped terms.
	__device__ __forceinline__ int phase_hessian_terms(double *acc) { fresh();
		const int n_acc = d.n_hap * P * P, n_terms = B.hap_term_off[d.o_hapoff + d.n_hap];
		for (int k = tid; k < n_acc; k += G) acc[k] = 0;
		grp_lds_sync<G>();
		const double *Jp = B.Jp + d.o_bp * O * P; const unsigned char *rp = B.bp_ok + d.o_bp; const int *t1 = B.hap_t1 + d.o_hapt, *t2 = B.hap_t2 + d.o_hapt, *tb = B.hap_tblk + d.o_hapt;
		int ninv = 0;
		// (two wavefronts: each takes a contiguous part of the list, cut between two Hessian blocks -- the additions into one accumulator all come from one wavefront, in program order)
		const int t_first = G > 64 ? ((threadIdx.x >> 6) ? d.hapt_split : 0) + (tid & 63) : tid, t_end = (G > 64 && !(threadIdx.x >> 6)) ? d.hapt_split : n_terms, t_step = G > 64 ? 64 : G;
		for (int t = t_first; t < t_end; t += t_step) {
			const int b1 = t1[t], b2 = t2[t], blk = tb[t];
			double A[O * P], Bm[O * P]; ldn<O * P>(A, Jp + (long long)b1 * O * P); ldn<O * P>(Bm, Jp + (long long)b2 * O * P);
			if (rp[b1] && rp[b2]) {
				double H[P * P];
#pragma unroll
				for (int k = 0; k < P * P; k++) H[k] = 0;
				hess_term<P, P>(H, A, Bm);
				double *dst = acc + blk * P * P;
#pragma unroll
				for (int k = 0; k < P * P; k++) atomicAdd(dst + k, H[k]);
			} else ninv++;
		}
		grp_lds_sync<G>();
		const double sc = (prm.noise == SRBA_NOISE_IDENTITY) ? prm.inv_sigma : 1.0; const bool latch = prm.solver != SRBA_SOLVER_NO_SCHUR_SPARSE_CHOL;
		double *Hg = B.HAp + d.o_hap * P * P, *H0 = B.HAp0 + d.o_hap * P * P;
		for (int k = 2 * tid; k < n_acc; k += 2 * G) { // n_acc is a multiple of 9 or 36; an odd tail is a single double
			if (k + 1 < n_acc) { f64x2u v; v.x = acc[k] * sc; v.y = acc[k + 1] * sc; *(f64x2u *)(Hg + k) = v; if (latch) *(f64x2u *)(H0 + k) = v; }
			else { const double v = acc[k] * sc; Hg[k] = v; if (latch) H0[k] = v; }
		}
		return ninv;
	}
	// U_f and U_Apf blocks, one lane per block (landmark families)
	__device__ __forceinline__ int phase_hessian_landmark_blocks() { fresh();
		int ninv = 0;
		if constexpr (!T::REL) {
			const double *Jp = B.Jp + d.o_bp * O * P, *Jf = B.Jf + d.o_bf * O * L; const unsigned char *rp = B.bp_ok + d.o_bp, *rf = B.bf_ok + d.o_bf;
			for (int b = tid; b < d.n_hf; b += G)
				ninv += hess_block<L, L>(B.Hf + (d.o_hf + b) * L * L, nullptr, B.hf_t1 + d.o_hft, B.hf_t2 + d.o_hft, B.hf_term_off[d.o_hfoff + b], B.hf_term_off[d.o_hfoff + b + 1], Jf, Jf, rf, rf);
			for (int b = tid; b < d.n_hapf; b += G)
				ninv += hess_block<P, L>(B.HApf + (d.o_hapf + b) * P * L, nullptr, B.hapf_t1 + d.o_hapft, B.hapf_t2 + d.o_hapft, B.hapf_term_off[d.o_hapfoff + b],
					B.hapf_term_off[d.o_hapfoff + b + 1], Jp, Jf, rp, rf);
		}
		return ninv;
	}

	// ---- K5
	// S lanes share one column (blocks dealt round-robin, partial sums combined in a fixed butterfly order -> deterministic)
	template <int M>
	__device__ __forceinline__ void grad_cols(double *g, int ncol, const double *J, const int *res, const int *col_off, const double *resid) {
		// lanes per column: 8 while a single pass covers all columns, else 4 -- ceil(4*ncol/64) quarter-length passes beat ceil(ncol/64) full-length ones
		// whenever ncol is not a multiple of 64 (67 columns: 1.25 instead of 2 column-times)
		const int S = 8 * ncol <= G ? 8 : 4;
		const int per = G / S, sub = tid % S;
		const double sc = (prm.noise == SRBA_NOISE_IDENTITY) ? prm.inv_sigma : 1.0;
		for (int base = 0; base < ncol; base += per) {
			const int i = base + tid / S; const bool live = i < ncol;
			double acc[M];
#pragma unroll
			for (int k = 0; k < M; k++) acc[k] = 0;
			if (live) {
				const int bb = col_off[i], be = col_off[i + 1];
#ifdef SRBA_GRADU2
				constexpr int U = 2;
#else
				constexpr int U = LEAN ? 2 : ((O * M <= 9) ? 4 : (O * M <= 24 ? 2 : 1)); // blocks in flight per lane (their loads do not depend on the running sum)
#endif
				for (int b0 = bb + sub; b0 < be; b0 += U * S) {
					double A[U][O * M], lr[U][O];
#pragma unroll
					for (int u = 0; u < U; u++) {
						const int b = b0 + u * S;
						if (b < be) {
							const double *Ab = J + (long long)b * O * M, *r = resid + (long long)(d.o_obs + res[b]) * O;
							ldn<O * M>(A[u], Ab); ldn<O>(lr[u], r);
						}
					}
#pragma unroll
					for (int u = 0; u < U; u++) {
						if (b0 + u * S < be) {
							if (prm.noise == SRBA_NOISE_CONSTANT_MATRIX) { double t[O]; for (int k = 0; k < O; k++) { double q = 0; for (int j = 0; j < O; j++) q += prm.lambda[k * O + j] * lr[u][j];
								t[k] = q; } for (int k = 0; k < O; k++) lr[u][k] = t[k]; }
							for (int q = 0; q < M; q++) { double sm = 0; for (int k = 0; k < O; k++) sm += A[u][k * M + q] * lr[u][k]; acc[q] += sm; }
						}
					}
				}
			}
			for (int m = 1; m < S; m *= 2) {
#pragma unroll
				for (int k = 0; k < M; k++) acc[k] += __shfl_xor(acc[k], m);
			}
			if (live && sub == 0) {
#pragma unroll
				for (int k = 0; k < M; k++) g[i * M + k] = acc[k] * sc;
			}
		}
	}
	__device__ __forceinline__ void phase_gradient(const double *resid) { fresh();
		double *g = B.grad + d.o_scal;
		grad_cols<P>(g, d.nK, B.Jp + d.o_bp * O * P, B.bp_res + d.o_bp, B.colp_off + d.o_colp, resid);
		if constexpr (!T::REL) grad_cols<L>(g + d.nK * P, d.nF, B.Jf + d.o_bf * O * L, B.bf_res + d.o_bf, B.colf_off + d.o_colf, resid);
	}
	__device__ __forceinline__ double lambda_guess(double *red) { fresh(); // optimize_edges.h:366-390
		double mx = 0;
		for (int i = tid; i < d.nK; i += G) { const double *H = B.HAp + (d.o_hap + B.hap_diag[d.o_unk + i]) * P * P; double m = H[0]; for (int k = 1; k < P; k++) m = fmax(m, H[k * P + k]);
			mx = fmax(mx, m); }
		if constexpr (!T::REL) for (int i = tid; i < d.nF; i += G) { const double *H = B.Hf + (d.o_hf + B.hf_diag[d.o_ulm + i]) * L * L; double m = H[0]; for (int k = 1; k < L; k++) m = fmax(m,
			H[k * L + k]); mx = fmax(mx, m); }
		return 1e-3 * grp_max<G>(mx, red);
	}
};

// the batch with its second copy of the unknowns / spanning-tree poses in place of the first (double-buffered LM loop)
// ---- lambda-ladder speculation for a batch of ONE capsule (the per-key-frame use of the engine: define_new_keyframe -> optimize_local_area -> one optimize_edges call).
// A rejected trial leaves the state as it was and only moves lambda (lambda *= nu, nu *= 2: optimize_edges.h:685-687), so the trials of a run of rejections all start from the
// same accepted state with a lambda known in advance: W workgroups ("replicas", each with its own copy of the work arena: shift_work) evaluate the W next steps of the ladder at
// once, exchange {solved, rho, chi2} through memory, and every replica then walks the SAME control flow as the sequential loop, taking the outcome of trial j from replica j
// instead of computing it. On the first accepted step the other replicas adopt the winner's increment (re-applied to their copy of the accepted state: the same arithmetic on
// the same numbers, bit-identical) and all of them relinearise redundantly. Results are those of the sequential loop bit for bit; the chain of ~26 dependent trials of a
// key-frame becomes ~12 rounds (the floor tail of a run, up to lambda > max_lambda, is one or two rounds).
struct SpecCtl {
	int w, W;          // this replica, replicas
	int *flag;         // [W]: the last round replica j has published
	double *box;       // [2][W][4]: per round parity and replica {code (1 = not positive definite, 2 = evaluated), rho, chi2 of the trial point, lambda}
	double *xdelta;    // [2][W][xstride]: the increment replica j solved for in that round
	int xstride;
	long long stride;  // bytes between the work arenas of two consecutive replicas
	int round0;        // the rounds of this launch are numbered round0 + 1 ...: above every round of the launches before it (at most 8192 rounds per launch: one per trial -- the host only speculates
		// when max_iters keeps a run below that), so `flag` needs no clearing
	double *edge_backup; // [nK * PD]: the unknown edges as the launch found them (written by replica 0 before anything else): what the host restores before it re-runs a capsule whose replicas lost
		// step (status 2) on the plain path
};
// the work arena of a replica: every state / workspace pointer of the batch moved by `bytes` (the arenas of the replicas lie one after the other: srba_hip_upload_problems)
__device__ __forceinline__ Batch shift_work(Batch B, long long bytes) {
#define SRBA_SH(f) B.f = (decltype(B.f))((char *)B.f + bytes)
	SRBA_SH(edge); SRBA_SH(ulm); SRBA_SH(pose); SRBA_SH(Jp); SRBA_SH(Jf); SRBA_SH(resid); SRBA_SH(resid2); SRBA_SH(HAp); SRBA_SH(HAp0); SRBA_SH(Hf); SRBA_SH(HApf); SRBA_SH(grad); SRBA_SH(grad0);
		SRBA_SH(delta); SRBA_SH(Hfinv); SRBA_SH(YW); SRBA_SH(Yh);
	SRBA_SH(old_edge); SRBA_SH(old_ulm); SRBA_SH(old_pose); SRBA_SH(dense); SRBA_SH(ulm_inf); SRBA_SH(edge1); SRBA_SH(ulm1); SRBA_SH(pose1); SRBA_SH(valid); SRBA_SH(first_fail); SRBA_SH(hf_ok);
		SRBA_SH(bp_ok); SRBA_SH(bf_ok); SRBA_SH(ulm_inf_valid);
	SRBA_SH(results); SRBA_SH(lambda_io); SRBA_SH(chi2); SRBA_SH(notpd); if (B.phase_cycles) SRBA_SH(phase_cycles);
#undef SRBA_SH
	return B;
}
// publish this replica's outcome of the round and wait for all the others (every replica is resident: W workgroups on 256 CUs). Release / acquire at agent scope carry the
// outcome and the increment across the L2s of the XCDs.
__device__ __forceinline__ bool spec_exchange(const SpecCtl &sc, int round, int code, double rho, double chi2, double lam) { // false: a replica did not answer within the spin bound
	__threadfence(); __syncthreads();
	if (threadIdx.x == 0) { double *b = sc.box + ((round & 1) * sc.W + sc.w) * 4; b[0] = (double)code; b[1] = rho; b[2] = chi2; b[3] = lam; __hip_atomic_store(sc.flag + sc.w, round, __ATOMIC_RELEASE,
		__HIP_MEMORY_SCOPE_AGENT); }
	int late = 0;
	if ((int)threadIdx.x < sc.W) { long long spins = 0; while (__hip_atomic_load(sc.flag + threadIdx.x, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < round) { __builtin_amdgcn_s_sleep(1);
		if (++spins > (1ll << 19)) { late = 1; break; } /* (a replica that never comes -- not resident because something else holds the CUs: give up after about a second instead of hanging the
		device; the caller sets status 2 and the host re-runs the capsule on the sequential path) */ } }
	const int any_late = __syncthreads_or(late); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
	return any_late == 0;
}
__device__ __forceinline__ double spec_ld(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// A reference to an object in (read-only) global memory through a pointer the optimiser knows nothing about. Why: the ~100 array pointers of `Batch` and the ~50 sizes / offsets of a
// `ProbDesc` are uniform, loop-invariant loads; the compiler hoists every one of them to the top of the LM loop and keeps them live across it -- far more than the 102 scalar registers hold,
// so they live in VGPR lanes (220 - 530 spilled scalars per kernel) and every use inside a hot loop is a v_readlane first. Each PHASE of the loop is handed its own laundered reference
// (lm_one: a fresh Solver per phase call): its loads cannot be merged with another phase's, so a pointer is live for one phase -- a scalar load when the phase starts instead of a
// register for the whole run. The memory is read-only for the duration of the kernel: constant address space, scalar loads.
template <class T> __device__ __forceinline__ const T &lnd(const T &r) {
	unsigned long long a = (unsigned long long)&r; // (uniform in fact -- the capsule index comes from a work counter -- but not always to the divergence analysis: v_readfirstlane makes it a scalar
		// either way)
	const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
	a = ((unsigned long long)hi << 32) | lo; asm volatile("" : "+s"(a));
	return *(const T *)(const __attribute__((address_space(4))) T *)a;
}
__device__ __forceinline__ Batch copy_view(const Batch &B, int copy) { Batch V = B; if (copy) { V.edge = B.edge1; V.ulm = B.ulm1; V.pose = B.pose1; } return V; }

} // namespace srbadev
