/*
 * srba_wg.hpp -- one WORKGROUP per capsule for the landmark families (round 5): the dense Cholesky of the Schur-reduced system on the matrix cores.
 *
 * Rounds 2-4 ran a landmark window on ONE wavefront (k_lm_run<3..6>: 504-512 registers, one wavefront per SIMD). The Schur-reduced system of such a window is (nearly)
 * full, n = 6 x unknown edges = 60 .. 360 scalars, and its factorisation was a 3x3-block sweep on the vector ALU with the numbers in HBM: 5.5 of the 15.5 ms of a trial of
 * the 40-59-edge stereo windows, at ~11 live lanes. That contraction is what `north_star` calls MFMA-worthy: here the system is a lower triangle of 16 x 16 TILES and one
 * workgroup of NW = 2 or 4 wavefronts factors it left-looking with v_mfma_f64_16x16x4_f64 (reference: the dense LL^t of lev-marq_solvers.h:474-568, Eigen::LLT).
 *
 * Tile storage = the MFMA fragment itself ("frag"): tile[4 l + r] = M[l & 15][(l >> 4) + 4 r], lane l = 0 .. 63, r = 0 .. 3. Why it is the right layout:
 *   - a lane's four doubles are contiguous: a tile moves as two 16-byte requests per lane, the wavefront covers the 2 KB tile as one contiguous span;
 *   - as the A operand of step kk of D = A B (A[i][k]: i = l & 15, k = (l >> 4) + 4 kk) the lane feeds tile[4 l + kk]: the tile is M;
 *     as the B operand (B[k][j]: k = (l >> 4) + 4 kk, j = l & 15) the same double reads as M^t: frag(M1) x frag(M2) computes M1 M2^t without a transpose anywhere;
 *   - the accumulator D[(l >> 4) + 4 r][l & 15] of lane l, read as a frag, is D^t -- so the sweep keeps TRANSPOSED accumulators: C_ik^t = A_ik^t - sum_j L_kj L_ij^t starts
 *     from frag(A_ik) as loaded, the accumulators ARE the B operand of the triangular solve L_ik^t = L_kk^-1 C_ik^t (A operand: frag(L_kk^-1) from LDS), and the result,
 *     dumped as it lies in the registers, is frag(L_ik): nothing is ever re-laid-out, every global access is a full-line span.
 * Step k of the sweep (one workgroup barrier per step): wavefront 0 takes tile row k + 1 -- finishes L_{k+1,k}, then the diagonal tile of step k + 1, its Cholesky and the
 * inverse of its factor (one lane per row, columns handed round by v_readlane; the only serial part, ~5 us) -- WHILE the other wavefronts finish the rows below. The
 * right-hand side rides along as tile row nt (forward substitution for free); the backward substitution runs on the LDS copy of y with DPP row sums.
 * "Not positive definite" == a non-positive pivot of a diagonal tile's Cholesky (Eigen::LLT's criterion), seen by all wavefronts through an LDS flag.
 */
#pragma once
#include "srba_device.hpp"

namespace srbadev {

typedef double f64x4w __attribute__((ext_vector_type(4)));
typedef double f64x2w __attribute__((ext_vector_type(2), aligned(16)));
constexpr int WT = 16;                             // tile edge = the shape of v_mfma_f64_16x16x4_f64
constexpr int WG_NT_MAX = 32;                      // tile rows of the largest system this path takes: n <= 512 scalars
constexpr int WG_LDS_DOUBLES = 256 + 512 + 16 * WG_NT_MAX + 16; // diagonal tile staging | two inverse factors (double-buffered) | y / x | group-reduction scratch (12) | flag, work-counter slot
constexpr int WG_RED = 768 + 16 * WG_NT_MAX;        // offset of the reduction scratch; the solver's flag is the int at double WG_RED + 12, the kernels' work-counter slot at WG_RED + 13
constexpr int WG_GACC = 0;                          // U_Ap-in-LDS path: accumulators of the Schur gradient correction (one per scalar of the reduced system, <= 16 WG_NT_MAX = 512): they share the
                                                    // staging area of the factorisation (768 doubles; the correction is added to the gradient before the system is assembled) ...
constexpr int WG_HS = WG_LDS_DOUBLES;               // ... and the U_Ap blocks themselves, n_hap x (P x P + 1) doubles from here
__host__ __device__ inline int wg_tile(int i, int j) { return i * (i + 1) / 2 + j; } // tile (i, j), j <= i; tile row nt = the right-hand side
__host__ __device__ inline long long wg_ws_doubles(int nt) { return 256LL * ((long long)(nt + 1) * (nt + 2) / 2 + nt + 4); } // tiles of rows 0 .. nt | nt inverse diagonal factors | 2 x 2 look-ahead
	// partial sums
// offset of element (r, c) inside a frag tile
__host__ __device__ inline int wg_frag_off(int r, int c) { return ((r + 16 * (c & 3)) << 2) + (c >> 2); }

__device__ __forceinline__ f64x4w wg_ld(const double *tile, int l) { const f64x2w a = *(const f64x2w *)(tile + 4 * l), b = *(const f64x2w *)(tile + 4 * l + 2); f64x4w v; v.x = a.x; v.y = a.y;
	v.z = b.x; v.w = b.y; return v; }
__device__ __forceinline__ void wg_st(double *tile, int l, const f64x4w &v) { f64x2w a, b; a.x = v.x; a.y = v.y; b.x = v.z; b.y = v.w; *(f64x2w *)(tile + 4 * l) = a;
	*(f64x2w *)(tile + 4 * l + 2) = b; }
// acc += M1 M2^t for two frag tiles (four matrix instructions, K = 16)
__device__ __forceinline__ f64x4w wg_mma(const f64x4w &m1, const f64x4w &m2, f64x4w acc) {
	acc = __builtin_amdgcn_mfma_f64_16x16x4f64(m1.x, m2.x, acc, 0, 0, 0); acc = __builtin_amdgcn_mfma_f64_16x16x4f64(m1.y, m2.y, acc, 0, 0, 0);
	acc = __builtin_amdgcn_mfma_f64_16x16x4f64(m1.z, m2.z, acc, 0, 0, 0); acc = __builtin_amdgcn_mfma_f64_16x16x4f64(m1.w, m2.w, acc, 0, 0, 0); return acc;
}
__device__ __forceinline__ double wg_bcast(double v, int lane /* uniform */) { return readlane_v(v, lane); }
__device__ __forceinline__ double wg_rsqrt(double d) { // 1 / sqrt(d): v_rsq_f64 + two Newton steps
	double y = __builtin_amdgcn_rsq(d); double h = 0.5 * d * y, e = fma(-h, y, 0.5); y = fma(y, e, y); h = 0.5 * d * y; e = fma(-h, y, 0.5); return fma(y, e, y);
}
// sum over the 16 lanes of a row (DPP row_shr 1, 2, 4, 8: the total lands in the last lane of the row)
__device__ __forceinline__ double wg_rowsum16(double v) { v += dpp_shift0<0x111, 0xf>(v); v += dpp_shift0<0x112, 0xf>(v); v += dpp_shift0<0x114, 0xf>(v); v += dpp_shift0<0x118, 0xf>(v); return v; }

// Wavefront 0: Cholesky of a diagonal tile given as accumulators (frag of the symmetric C_kk) and the inverse of its factor, one lane per row. The inverse goes to LDS
// (frag order: the A operand of this step's triangular solves) and to `LIk` in memory (the backward substitution reads it). Returns false on a non-positive pivot.
__device__ __forceinline__ bool wg_diag(const f64x4w &c, lds_f64 *smC, lds_f64 *smL, double *LIk, int l) {
	smC[4 * l] = c.x; smC[4 * l + 1] = c.y; smC[4 * l + 2] = c.z; smC[4 * l + 3] = c.w;
	solver_sync();
	int row = l & 15; // lanes 16 .. 63 mirror lanes 0 .. 15 (their copies are never read)
	asm volatile("" : "+v"(row)); // opaque: the ~50 lane predicates below (row == j, row > j) were otherwise formed once at kernel entry and kept, spilled, across every phase (100 scalars)
	double a[16], x[16];
#pragma unroll
	for (int q = 0; q < 16; q++) a[q] = smC[wg_frag_off(row, q)];
	bool ok = true; double rrow = 0;
#pragma unroll
	for (int q = 0; q < 16; q++) x[q] = (row == q) ? 1.0 : 0.0; // x[q] = delta_{row,q} - sum_{m < j} L[row][m] X[m][q]: row `row` of X = L^-1 is x[.] / L[row][row] once its turn has come
	// Column j of L (lane i >= j ends with a[j] = L[i][j], lanes i < j with 0) and, in the same pass, row j of X: it needs L[j][0 .. j] and rows 0 .. j-1 of X, all final once
	// column j is -- lane j hands its row round and the lanes below subtract L[i][j] X[j][q]. The two chains (next column of L, this row of X) are independent: they overlap.
#pragma unroll
	for (int j = 0; j < 16; j++) {
		const double dj = wg_bcast(a[j], j);
		ok &= (dj > 0.0);
		const double r = wg_rsqrt(dj);
		const double lj = (row == j) ? dj * r : ((row > j) ? a[j] * r : 0.0), lbelow = (row > j) ? lj : 0.0;
		a[j] = lj; rrow = (row == j) ? r : rrow;
#pragma unroll
		for (int q = j + 1; q < 16; q++) { const double lq = wg_bcast(lj, q); a[q] = fma(-lj, lq, a[q]); asm volatile("" : "+v"(a[q]));
			/* (pinned where its broadcast is: no pile of spilled v_readlane pairs, cf. chol_block_regs) */ }
#pragma unroll
		for (int q = 0; q <= j; q++) {
			const double xb = wg_bcast(x[q] * r, j);   // X[j][q] (r = 1 / L[j][j], the same in every lane)
			x[q] = fma(-lbelow, xb, x[q]);             // lanes i > j subtract L[i][j] X[j][q]; lane j and the lanes above keep theirs
			asm volatile("" : "+v"(x[q]));
		}
		__builtin_amdgcn_sched_barrier(0);
	}
#pragma unroll
	for (int q = 0; q < 16; q++) x[q] *= rrow;
	if (l < 16) {
#pragma unroll
		for (int q = 0; q < 16; q++) smL[wg_frag_off(row, q)] = (q <= row) ? x[q] : 0.0;
	}
	solver_sync();
	f64x4w v; v.x = smL[4 * l]; v.y = smL[4 * l + 1]; v.z = smL[4 * l + 2]; v.w = smL[4 * l + 3];
	wg_st(LIk, l, v);
	return ok;
}

// Factorisation + both substitutions of the tile system T (nt tile rows + the right-hand-side row) by the NW wavefronts of the workgroup. On return x (16 nt doubles) is in
// sm + 768; false: not positive definite (uniform over the workgroup). All threads of the workgroup call it.
template <int NW>
__device__ __forceinline__ bool wg_chol_solve(double *T, double *LI, const int nt, lds_f64 *sm) {
	static_assert(NW >= 2, "wavefront 0 runs the diagonal chain beside the panel wavefronts");
	const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
	lds_f64 *smC = sm, *smL = sm + 256, *yb = sm + 768; int __attribute__((address_space(3))) *flag = (int __attribute__((address_space(3))) *)(sm + WG_RED + 12);
	if (threadIdx.x == 0) *flag = 0;
	__syncthreads();
	if (w == 0) { const f64x4w c = wg_ld(T + 256 * (size_t)wg_tile(0, 0), l); if (!wg_diag(c, smC, smL, LI, l) && l == 0) *flag = 1; }
	__syncthreads();
	for (int k = 0; k < nt; k++) {
		{ const int f = *flag; if (f != 0 && f <= k + 1) return false; } // the verdict is stamped with the step from which it counts: wavefront 0 may write the one of tile k + 1 (= k + 2)
			// while a late wavefront still reads here, in step k -- that one must go on like the others did, or the barrier counts part
		const lds_f64 *Lk = smL + 256 * (k & 1);
		f64x4w li; li.x = Lk[4 * l]; li.y = Lk[4 * l + 1]; li.z = Lk[4 * l + 2]; li.w = Lk[4 * l + 3]; // frag(L_kk^-1): A operand of the triangular solves
		const double *rowk = T + 256 * (size_t)wg_tile(k, 0);
		const f64x4w zero4 = {0, 0, 0, 0};
		auto solve_store = [&](int i, const f64x4w &a0, const f64x4w &s) __attribute__((always_inline)) -> f64x4w { // L_ik = (A_ik - s) L_kk^-t, stored as it lies in the accumulators
			const f64x4w ct = a0 - s; // C_ik^t as accumulators == the B operand of the solve
			f64x4w dd = zero4;
			dd = __builtin_amdgcn_mfma_f64_16x16x4f64(li.x, ct.x, dd, 0, 0, 0); dd = __builtin_amdgcn_mfma_f64_16x16x4f64(li.y, ct.y, dd, 0, 0, 0);
			dd = __builtin_amdgcn_mfma_f64_16x16x4f64(li.z, ct.z, dd, 0, 0, 0); dd = __builtin_amdgcn_mfma_f64_16x16x4f64(li.w, ct.w, dd, 0, 0, 0);
			wg_st(T + 256 * (size_t)wg_tile(i, k), l, dd);
			if (i == nt && (l & 15) == 0) { const int c0 = 16 * k + (l >> 4); yb[c0] = dd.x; yb[c0 + 4] = dd.y; yb[c0 + 8] = dd.z; yb[c0 + 12] = dd.w; } // row 0 of the right-hand-side tile: y_k
			return dd;
		};
		// L_ik = (A_ik - sum_{j<k} L_ij L_kj^t) L_kk^-t for one tile row; diag_sum (row k + 1 only): += sum_{j<k} L_ij L_ij^t. One column j per pass, the tiles of the next pass
		// requested before this pass's matrix instructions (the tiles come from L2 / the Infinity Cache: hundreds of cycles)
		auto finish_row = [&](int i, f64x4w *diag_sum) __attribute__((always_inline)) -> f64x4w {
			const double *rowi = T + 256 * (size_t)wg_tile(i, 0);
			f64x4w s = zero4;
			const f64x4w a0 = wg_ld(rowi + 256 * (size_t)k, l);
			f64x4w A0 = zero4, B0 = zero4, nA0 = zero4, nB0 = zero4;
			if (k > 0) { A0 = wg_ld(rowk, l); B0 = wg_ld(rowi, l); }
			for (int j = 0; j < k; j++) {
				if (j + 1 < k) { const size_t o = 256 * (size_t)(j + 1); nA0 = wg_ld(rowk + o, l); nB0 = wg_ld(rowi + o, l); }
				s = wg_mma(A0, B0, s); if (diag_sum) *diag_sum = wg_mma(B0, B0, *diag_sum);
				A0 = nA0; B0 = nB0;
			}
			return solve_store(i, a0, s);
		};
		// the same for TWO tile rows at once: the tiles of row k serve both, five tiles per pass in flight instead of four for twice the work (the sweep waits for its tiles)
		auto finish_rows2 = [&](int i0, int i1) __attribute__((always_inline)) {
			const double *r0 = T + 256 * (size_t)wg_tile(i0, 0), *r1 = T + 256 * (size_t)wg_tile(i1, 0);
			f64x4w s0 = zero4, s1 = zero4;
			const f64x4w a00 = wg_ld(r0 + 256 * (size_t)k, l), a01 = wg_ld(r1 + 256 * (size_t)k, l);
			f64x4w A0 = zero4, P0 = zero4, Q0 = zero4, nA0 = zero4, nP0 = zero4, nQ0 = zero4; // (one column per pass, the next one's three tiles on their way: six tiles of registers)
			if (k > 0) { A0 = wg_ld(rowk, l); P0 = wg_ld(r0, l); Q0 = wg_ld(r1, l); }
			for (int j = 0; j < k; j++) {
				if (j + 1 < k) { const size_t o = 256 * (size_t)(j + 1); nA0 = wg_ld(rowk + o, l); nP0 = wg_ld(r0 + o, l); nQ0 = wg_ld(r1 + o, l); }
				s0 = wg_mma(A0, P0, s0); s1 = wg_mma(A0, Q0, s1);
				A0 = nA0; P0 = nP0; Q0 = nQ0;
			}
			solve_store(i0, a00, s0); solve_store(i1, a01, s1);
		};
		// Look-ahead: tile row k + 2 is wavefront 0's row of the NEXT step, whose serial chain (row -> diagonal tile -> Cholesky -> inverse) bounds a step. The panel wavefront that
		// finishes row k + 2 now (wavefront 1: it holds the row's tiles anyway) also forms what that chain needs from the columns before k: s' = sum_{j<k} L_{k+2,j} L_{k+1,j}^t and
		// ds' = sum_{j<=k} L_{k+2,j} L_{k+2,j}^t, left in the workspace (two tiles per step parity); wavefront 0 then adds the one column that was missing (k) and goes on.
		double *LA = LI + 256 * (size_t)nt; // [parity][s', ds'][256]
		auto finish_row_la = [&](int i) __attribute__((always_inline)) { // i = k + 2 <= nt - 1
			const double *rowi = T + 256 * (size_t)wg_tile(i, 0), *rown = T + 256 * (size_t)wg_tile(k + 1, 0);
			f64x4w s = zero4, sn = zero4, dsn = zero4;
			const f64x4w a0 = wg_ld(rowi + 256 * (size_t)k, l);
			f64x4w A0 = zero4, B0 = zero4, C0 = zero4, nA0 = zero4, nB0 = zero4, nC0 = zero4;
			if (k > 0) { A0 = wg_ld(rowk, l); B0 = wg_ld(rowi, l); C0 = wg_ld(rown, l); }
			for (int j = 0; j < k; j++) {
				if (j + 1 < k) { const size_t o = 256 * (size_t)(j + 1); nA0 = wg_ld(rowk + o, l); nB0 = wg_ld(rowi + o, l); nC0 = wg_ld(rown + o, l); }
				s = wg_mma(A0, B0, s); sn = wg_mma(C0, B0, sn); dsn = wg_mma(B0, B0, dsn);
				A0 = nA0; B0 = nB0; C0 = nC0;
			}
			const f64x4w dd = solve_store(i, a0, s);
			dsn = wg_mma(dd, dd, dsn);
			double *la = LA + 512 * (size_t)((k + 1) & 1); wg_st(la, l, sn); wg_st(la + 256, l, dsn);
		};
		if (w == 0) {
			if (k + 1 < nt) {
				f64x4w dd, ds = zero4; const f64x4w ckk = wg_ld(T + 256 * (size_t)wg_tile(k + 1, k + 1), l); // (everything this chain reads is requested up front: one round trip)
				if (k == 0) { f64x4w ds0 = zero4; dd = finish_row(1, &ds0); } // (nothing before column 0)
				else { // row k + 1: s = s' + L_{k+1,k-1} L_{k,k-1}^t with s', ds' as the look-ahead of step k - 1 left them
					const double *la = LA + 512 * (size_t)(k & 1);
					const f64x4w sp = wg_ld(la, l), a0 = wg_ld(T + 256 * (size_t)wg_tile(k + 1, k), l);
					const f64x4w Ak = wg_ld(rowk + 256 * (size_t)(k - 1), l), Bk = wg_ld(T + 256 * (size_t)wg_tile(k + 1, k - 1), l);
					ds = wg_ld(la + 256, l);
					dd = solve_store(k + 1, a0, wg_mma(Ak, Bk, sp));
				}
				ds = wg_mma(dd, dd, ds);
				const f64x4w c = ckk - ds;
				if (!wg_diag(c, smC, smL + 256 * ((k + 1) & 1), LI + 256 * (size_t)(k + 1), l) && l == 0) *flag = k + 2;
			} else finish_row(nt, nullptr);
		} else {
			int i = k + 1 + w;
			if (w == 1 && k + 2 <= nt - 1) { finish_row_la(k + 2); i += NW - 1; }
			for (; i + (NW - 1) <= nt; i += 2 * (NW - 1)) finish_rows2(i, i + (NW - 1));
			if (i <= nt) finish_row(i, nullptr);
		}
		__syncthreads();
	}
	if (*flag) return false;
	// backward substitution x = L^-t y on the LDS copy: x_k = L_kk^-t y_k (wavefront 0), then y_j -= L_kj^t x_k for the tiles of row k (all wavefronts)
	for (int k = nt - 1; k >= 0; k--) {
		if (w == 0) {
			const f64x4w v = wg_ld(LI + 256 * (size_t)k, l); const double yk = yb[16 * k + (l & 15)];
			const double p0 = wg_rowsum16(v.x * yk), p1 = wg_rowsum16(v.y * yk), p2 = wg_rowsum16(v.z * yk), p3 = wg_rowsum16(v.w * yk); // sum_r Linv[r][c] y[r], c = (l >> 4) + 4 q
			solver_sync(); // every lane has read y_k
			if ((l & 15) == 15) { const int c0 = 16 * k + (l >> 4); yb[c0] = p0; yb[c0 + 4] = p1; yb[c0 + 8] = p2; yb[c0 + 12] = p3; }
		}
		__syncthreads();
		const double xk = yb[16 * k + (l & 15)];
		for (int j = w; j < k; j += NW) {
			const f64x4w v = wg_ld(T + 256 * (size_t)wg_tile(k, j), l);
			const double p0 = wg_rowsum16(v.x * xk), p1 = wg_rowsum16(v.y * xk), p2 = wg_rowsum16(v.z * xk), p3 = wg_rowsum16(v.w * xk);
			if ((l & 15) == 15) { const int c0 = 16 * j + (l >> 4); yb[c0] -= p0; yb[c0 + 4] -= p1; yb[c0 + 8] -= p2; yb[c0 + 12] -= p3; }
		}
		__syncthreads();
	}
	return true;
}

} // namespace srbadev
