/*
 * srba_big.hpp -- the "large window" path: ONE problem capsule spread over the whole chip.
 *
 * A deep local area (BASELINE config 4: monocular SE3, max_tree_depth = max_optimize_depth = 8, ~10^2 unknown edges x 10^4 unknown landmarks, 10^4..10^5
 * observations, a dense Schur-reduced system of 10^3 unknowns) has plenty of parallelism INSIDE the capsule, the opposite of the one-wavefront-per-capsule kernels
 * of srba_hip.hip. Every reference hot loop becomes a grid-wide kernel over the items of that one capsule (same per-item device functions as the fused kernel:
 * Worker<FAM>::residual_row / jac_dh_dp / jac_dh_df / hess_block), the Levenberg-Marquardt control flow of optimize_edges.h:454-696 runs on the host
 * (srba_hip.hip: big_lm_run), and the reduced system (lev-marq_solvers.h:474-568: dense LL^t of the Schur complement) is factored by a blocked right-looking
 * Cholesky across workgroups:
 *     panel  k : every workgroup re-factors the 32x32 diagonal block in LDS, workgroup b solves its 64 rows of the panel  X = A_bk L_kk^-t  and eliminates them
 *                from the right-hand side (forward substitution fused)
 *     update k : trailing tiles  C_ij -= X_i X_j^t  with v_mfma_f64_16x16x4_f64 (the one MFMA-worthy contraction of the path), 64x64 tile per
 *                workgroup, operands staged through LDS
 *     bsub     : L^t x = y
 * "Not positive definite" = a pivot <= 0 in a diagonal block, as in Eigen::LLT.
 */
#pragma once

namespace srbadev {

constexpr int CB = 32;   // Cholesky block size
constexpr int CT = 64;   // trailing-update tile

struct BigSys { double *A; double *Ldiag; double *rhs; double *y; int *flag; int n, ld; }; // A: ld x ld row-major (lower triangle used), Ldiag: [ld][CB] factored diagonal blocks

// A GANG of large capsules in lock-step (round 4): every grid-wide phase below is launched ONCE for up to kGang capsules, blockIdx.y = slot of the gang; `mask` says which
// slots take part in this launch (the host knows which windows need a trial, which an accepted step's relinearisation, which a restore). A slot owns a region of the
// partial-sum / scalar / flag buffers (slot w: part + w * 3 * kBigPart, scal + w * 16, iscal + w * 8) and its dense system lies where its capsule's does (A[w]).
// Every slot uses the grid it would have alone for the phases that reduce over workgroups (fixed partition of the sums): a window's numbers do not depend on its gang.
constexpr int kGang = 32, kBigPart = 4096;
struct Gang { int p[kGang]; int ld[kGang]; int nsys[kGang]; double *A[kGang]; unsigned mask; int n; double *part; double *scal; int *iscal; };
enum { BS_CHI2 = 0, BS_MAXDIAG = 1, BS_DEN = 2, BS_NINF = 3, BS_LAMBDA = 4 }; // scal[w * 16 + .]; iscal[w * 8 + .] = {invalid Jacobians, not-positive-definite flag}
struct GangLambda { double v[kGang]; };
#ifndef SRBA_BIG_DECLS_ONLY /* (srba_hip.hip sizes buffers with the constants above; the device code below belongs to srba_big.hip) */
__device__ __forceinline__ BigSys gang_sys(const Gang &G, int w) {
	BigSys S; const int ld = G.ld[w]; S.A = G.A[w]; S.Ldiag = S.A + (size_t)ld * ld; S.rhs = S.Ldiag + (size_t)ld * CB; S.y = S.rhs + ld; S.flag = G.iscal + w * 8 + 1; S.n = G.nsys[w]; S.ld = ld;
		return S;
}
__device__ __forceinline__ int big_grid_dev(long long items, int block) { const long long g = (items + block - 1) / block; return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g)); }
#define BIG_ENTER() const int gw = blockIdx.y; if (!((G.mask >> gw) & 1u)) return; const int p = G.p[gw]
#define BIG_FLAG() (G.iscal[gw * 8 + 1])

typedef double f64x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double lane_bcast(double v, int l) { // l is wave-uniform
	const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
	return __hiloint2double(hi, lo);
}
// 1/sqrt(d) in full double precision without the sqrt + divide sequences: v_rsq_f64 seed, two Newton steps
__device__ __forceinline__ double rsqrt_nr(double d) {
	double r = __builtin_amdgcn_rsq(d);
	r = r * (1.5 - 0.5 * d * r * r); r = r * (1.5 - 0.5 * d * r * r);
	return r;
}
// Cholesky of a CB x CB block held one ROW PER LANE in registers (a[c] = A[lane][c]): the pivot and the column of L travel by v_readlane, no LDS round trip on
// the critical path. Lanes >= CB may carry extra rows b^t of an augmented matrix [A b; b^t .]: they come out as (L^-1 b)^t, i.e. the forward substitution of a
// right-hand side rides along for free. rinv[j] = 1 / L_jj (every lane). Uniform result.
__device__ __forceinline__ bool chol_block_regs(double (&a)[CB], double (&rinv)[CB], int lane) {
	bool ok = true; // no branch inside the chain: with one per pivot the compiler sinks the updates of later columns past it to their uses and keeps every broadcast L_kj alive (SGPRs spilled to VGPR
		// lanes); a bad pivot turns the rest into NaNs, which nobody reads
#pragma unroll
	for (int j = 0; j < CB; j++) {
		const double d = lane_bcast(a[j], j);
		ok &= (d > 0.0);
		const double r = rsqrt_nr(d); rinv[j] = r;
		const double l = (lane == j) ? d * r : ((lane > j) ? a[j] * r : 0.0);
		a[j] = l;
#pragma unroll
		for (int k = j + 1; k < CB; k++) { a[k] -= l * lane_bcast(l, k); /* only the lower part (lane >= k) is ever read back */ if ((k & 3) == 3) __builtin_amdgcn_sched_barrier(0); }
		__builtin_amdgcn_sched_barrier(0); // (and every four columns above) the scheduler otherwise issues all v_readlane of a pivot step first: 62 SGPRs live,
			// spilled to VGPR lanes and read back -- three instructions per value instead of one
	}
	return ok;
}

// Panel step k0: every workgroup factors the diagonal block itself (registers, lane CB carries the right-hand side); workgroup 0 publishes L_kk (Ldiag) and
// y_k, workgroup b >= 1 solves rows k0+CB+64(b-1) .. +63 of the panel against it and eliminates them from the right-hand side.
__global__ void __launch_bounds__(64) k_chol_panel(const Gang G, int k0) {
	__shared__ double Ls[(CB + 1) * (CB + 1)]; __shared__ double ri[CB];
	BIG_ENTER(); (void)p; const BigSys S = gang_sys(G, gw);
	const int lane = threadIdx.x, ld = S.ld;
	if (k0 >= ld || (int)blockIdx.x > (ld - k0 - CB + 63) / 64) return; // this window is smaller than the largest of the gang
	if (*S.flag) return; // an earlier panel met a non-positive pivot
	const int row = k0 + CB + 64 * ((int)blockIdx.x - 1) + lane; const bool has_row = blockIdx.x > 0 && row < ld;
	double *Arow = (double *)__builtin_assume_aligned(S.A + (size_t)(has_row ? row : k0) * ld + k0, 16);
	double x[CB];
	if (has_row) { // issued first: in flight while the diagonal block is factored
#pragma unroll
		for (int c = 0; c < CB; c++) x[c] = Arow[c];
	}
	double a[CB], rinv[CB];
	{ const double *src = (const double *)__builtin_assume_aligned(S.A + (size_t)(k0 + (lane & (CB - 1))) * ld + k0, 16);
	  const double *rh = (const double *)__builtin_assume_aligned(S.rhs + k0, 16);
#pragma unroll
	  for (int c = 0; c < CB; c++) { const double v = src[c], b = rh[c]; a[c] = lane < CB ? (c <= lane ? v : 0.0) : (lane == CB ? b : 0.0); } }
	if (!chol_block_regs(a, rinv, lane)) { if (blockIdx.x == 0 && lane == 0) *S.flag = 1; return; }
	if (blockIdx.x == 0) {
		if (lane < CB) {
			double *dst = (double *)__builtin_assume_aligned(S.Ldiag + (size_t)(k0 + lane) * CB, 16);
#pragma unroll
			for (int c = 0; c < CB; c++) dst[c] = a[c];
		} else if (lane == CB) {
#pragma unroll
			for (int c = 0; c < CB; c++) S.y[k0 + c] = a[c];
		}
		return;
	}
	if (lane <= CB) {
#pragma unroll
		for (int c = 0; c < CB; c++) Ls[lane * (CB + 1) + c] = a[c];
	}
	if (lane == 0) {
#pragma unroll
		for (int c = 0; c < CB; c++) ri[c] = rinv[c];
	}
	__syncthreads();
	if (!has_row) return;
	double acc = 0;
#pragma unroll
	for (int j = 0; j < CB; j++) { // x_j = (a_j - sum_{m<j} x_m L[j][m]) / L[j][j]
		double s = x[j];
#pragma unroll
		for (int m = 0; m < j; m++) s -= x[m] * Ls[j * (CB + 1) + m];
		x[j] = s * ri[j];
		acc += x[j] * Ls[CB * (CB + 1) + j];
	}
#pragma unroll
	for (int c = 0; c < CB; c++) Arow[c] = x[c];
	S.rhs[row] -= acc;
}

// Trailing update after panel k0: tile (ti, tj), tj <= ti, of the matrix below/right of the panel: C -= X_i X_j^t, X = A[:, k0 .. k0+CB). C is loaded straight
// into the MFMA accumulators (D = (-X_i) X_j^t + C) while the operands travel through LDS: two dependent memory phases instead of three.
__global__ void __launch_bounds__(256) k_chol_update(const Gang G, int k0) {
	__shared__ double Xi[CT * (CB + 1)], Xj[CT * (CB + 1)];
	BIG_ENTER(); (void)p; const BigSys S = gang_sys(G, gw);
	{ const int below = S.ld - k0 - CB, nt = below > 0 ? (below + CT - 1) / CT : 0; if ((int)blockIdx.x >= nt * (nt + 1) / 2) return; }
	if (*S.flag) return;
	int t = blockIdx.x, ti = 0; while ((ti + 1) * (ti + 2) / 2 <= t) ti++; const int tj = t - ti * (ti + 1) / 2; // linear tile index -> (ti, tj) of the lower triangle
	const int base = k0 + CB, i0 = base + CT * ti, j0 = base + CT * tj, ld = S.ld, tid = threadIdx.x;
	const int w = tid >> 6, lane = tid & 63, wr = w >> 1, wc = w & 1;
	const bool active = !(ti == tj && wc > wr); // the strictly upper quarter of a diagonal tile is never read
	f64x4 acc[2][2];
#pragma unroll
	for (int a = 0; a < 2; a++)
#pragma unroll
		for (int b = 0; b < 2; b++)
#pragma unroll
			for (int r = 0; r < 4; r++) { // D: col = lane & 15, row = (lane >> 4) + 4 r
				const int gi = i0 + wr * 32 + a * 16 + (lane >> 4) + 4 * r, gj = j0 + wc * 32 + b * 16 + (lane & 15);
				acc[a][b][r] = (active && gi < ld && gj < ld) ? S.A[(size_t)gi * ld + gj] : 0.0;
			}
	for (int e = tid; e < CT * CB; e += 256) {
		const int r = e / CB, c = e % CB;
		Xi[r * (CB + 1) + c] = (i0 + r < ld) ? -S.A[(size_t)(i0 + r) * ld + k0 + c] : 0.0;
		Xj[r * (CB + 1) + c] = (j0 + r < ld) ? S.A[(size_t)(j0 + r) * ld + k0 + c] : 0.0;
	}
	__syncthreads();
	if (!active) return;
#pragma unroll
	for (int kk = 0; kk < CB / 4; kk++) {
		double fa[2], fb[2];
#pragma unroll
		for (int a = 0; a < 2; a++) fa[a] = Xi[(wr * 32 + a * 16 + (lane & 15)) * (CB + 1) + 4 * kk + (lane >> 4)];   // A[i][k]
#pragma unroll
		for (int b = 0; b < 2; b++) fb[b] = Xj[(wc * 32 + b * 16 + (lane & 15)) * (CB + 1) + 4 * kk + (lane >> 4)];   // B[k][j] = X_j[j][k]
#pragma unroll
		for (int a = 0; a < 2; a++)
#pragma unroll
			for (int b = 0; b < 2; b++) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[a], fb[b], acc[a][b], 0, 0, 0);
	}
#pragma unroll
	for (int a = 0; a < 2; a++)
#pragma unroll
		for (int b = 0; b < 2; b++)
#pragma unroll
			for (int r = 0; r < 4; r++) {
				const int gi = i0 + wr * 32 + a * 16 + (lane >> 4) + 4 * r, gj = j0 + wc * 32 + b * 16 + (lane & 15);
				if (gi < ld && gj < ld) S.A[(size_t)gi * ld + gj] = acc[a][b][r];
			}
}

// ---- panel step k AND the trailing update of step k-1 in ONE launch (round 4): launch k0 holds
//   * the panel workgroups of step k0 (256 threads): they first apply the update of step k0-CB to what they are about to read -- their 64 rows of the panel columns and the
//     32 x 32 diagonal block, C -= X_prev X_prev^t with the same MFMA sequence as k_chol_update (bit-identical numbers), four wavefronts, results handed over in LDS --
//     then the first wavefront factors the diagonal block and solves the rows in ONE loop (chol_block_solve_regs: the row solve x_k -= x_j L_kj shares the v_readlane of
//     L_kj with the factorisation's own update; same operations in the same order as k_chol_panel's separate substitution, no LDS copy of L, no barrier);
//   * the tile workgroups of step k0-CB's update for the columns right of the panel (region [k0+CB, ld)^2): they only read X_prev, so they run beside the panel chain.
// One launch per 32 columns instead of two, and the update's time disappears behind the panel's dependent chain.
__device__ __forceinline__ bool chol_block_solve_regs(double (&a)[CB], double (&x)[CB], double &acc, int lane) {
	bool ok = true; // (branch-free, as in chol_block_regs)
#pragma unroll
	for (int j = 0; j < CB; j++) {
		const double d = lane_bcast(a[j], j);
		ok &= (d > 0.0);
		const double r = rsqrt_nr(d);
		const double l = (lane == j) ? d * r : ((lane > j) ? a[j] * r : 0.0);
		a[j] = l;
		const double xj = x[j] * r; x[j] = xj;
#pragma unroll
		for (int k = j + 1; k < CB; k++) { const double lk = lane_bcast(l, k); a[k] -= l * lk; x[k] -= xj * lk; asm volatile("" : "+v"(a[k]), "+v"(x[k]));
			/* pins the pair where the broadcast is: instruction selection otherwise emits the whole factorisation first and the substitution after it,
			every L_kj kept (spilled) for the second pass */ }
		__builtin_amdgcn_sched_barrier(0); // (as in chol_block_regs: no SGPR spills)
	}
#pragma unroll
	for (int j = 0; j < CB; j++) acc += x[j] * lane_bcast(a[j], CB); // lane CB carried the right-hand side: its row is y_k now
	return ok;
}
// The same chain with ONE array (round 5, late): the rows a panel wavefront solves are augmented rows of the block exactly like the right-hand side -- lane >= CB carries a row
// r^t and comes out as (L^-1 r)^t --, so the solve needs no second array: lanes 0 .. CB-1 the diagonal block, lane CB the right-hand side, lanes CB+1 .. 63 THIRTY-ONE panel
// rows, one multiply-add per remaining column instead of two. The 64 rows of a panel workgroup go to three wavefronts (31 + 31 + 2) that run the chain side by side, each with
// its own copy of the diagonal block. Same operations on the same numbers as chol_block_solve_regs: bit-identical factors. acc (lanes > CB): the row's share of y.
#ifndef SRBA_CHAIN_U
#define SRBA_CHAIN_U 2
#endif
__device__ __forceinline__ bool chol_block_rows_regs(double (&v)[CB], double &acc, int lane) {
	bool ok = true;
#pragma unroll
	for (int j = 0; j < CB; j++) {
		const double d = lane_bcast(v[j], j);
		ok &= (d > 0.0);
		const double r = rsqrt_nr(d);
		const double l = (lane == j) ? d * r : ((lane > j) ? v[j] * r : 0.0);
		v[j] = l;
#pragma unroll
		for (int k = j + 1; k < CB; k += SRBA_CHAIN_U) { // SRBA_CHAIN_U columns per pin: the later v_readlane pairs fill the wait states the first multiply-add would spend in an s_nop
			double lk[SRBA_CHAIN_U];
#pragma unroll
			for (int u = 0; u < SRBA_CHAIN_U; u++) lk[u] = (k + u < CB) ? lane_bcast(l, k + u) : 0.0;
#pragma unroll
			for (int u = 0; u < SRBA_CHAIN_U; u++) if (k + u < CB) v[k + u] -= l * lk[u];
#pragma unroll
			for (int u = 0; u < SRBA_CHAIN_U; u++) if (k + u < CB) asm volatile("" : "+v"(v[k + u])); }
		__builtin_amdgcn_sched_barrier(0);
	}
#pragma unroll
	for (int j = 0; j < CB; j++) acc += v[j] * lane_bcast(v[j], CB);
	return ok;
}
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) k_chol_step(const Gang G, int k0) { // two workgroups per CU: left alone the kernel takes 232 + 32 registers -- eight
	// over the budget that lets a second workgroup in -- and the many tile workgroups of the early steps queue behind one another
	__shared__ double sh[2 * CT * (CB + 1)];
	BIG_ENTER(); (void)p; const BigSys S = gang_sys(G, gw);
	const int ld = S.ld, tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
	if (k0 >= ld) return; // this window is smaller than the largest of the gang
	const int below = ld - k0 - CB, npan = 1 + (below + 63) / 64, kp = k0 - CB;
	if (*S.flag) return; // an earlier panel met a non-positive pivot
	if ((int)blockIdx.x >= npan) { // ---- trailing update of step kp, columns right of this panel (k_chol_update's tile body with X = A[:, kp .. kp+CB))
		if (k0 == 0) return;
		const int nt = below > 0 ? (below + CT - 1) / CT : 0; int t = (int)blockIdx.x - npan; if (t >= nt * (nt + 1) / 2) return;
		double *Xi = sh, *Xj = sh + CT * (CB + 1);
		int ti = 0; while ((ti + 1) * (ti + 2) / 2 <= t) ti++; const int tj = t - ti * (ti + 1) / 2;
		const int base = k0 + CB, i0 = base + CT * ti, j0 = base + CT * tj, wr = w >> 1, wc = w & 1;
		const bool active = !(ti == tj && wc > wr);
		f64x4 acc[2][2];
#pragma unroll
		for (int a = 0; a < 2; a++)
#pragma unroll
			for (int b = 0; b < 2; b++)
#pragma unroll
				for (int r = 0; r < 4; r++) {
					const int gi = i0 + wr * 32 + a * 16 + (lane >> 4) + 4 * r, gj = j0 + wc * 32 + b * 16 + (lane & 15);
					acc[a][b][r] = (active && gi < ld && gj < ld) ? S.A[(size_t)gi * ld + gj] : 0.0;
				}
		for (int e = tid; e < CT * CB; e += 256) {
			const int r = e / CB, c = e % CB;
			Xi[r * (CB + 1) + c] = (i0 + r < ld) ? -S.A[(size_t)(i0 + r) * ld + kp + c] : 0.0;
			Xj[r * (CB + 1) + c] = (j0 + r < ld) ? S.A[(size_t)(j0 + r) * ld + kp + c] : 0.0;
		}
		__syncthreads();
		if (!active) return;
#pragma unroll
		for (int kk = 0; kk < CB / 4; kk++) {
			double fa[2], fb[2];
#pragma unroll
			for (int a = 0; a < 2; a++) fa[a] = Xi[(wr * 32 + a * 16 + (lane & 15)) * (CB + 1) + 4 * kk + (lane >> 4)];
#pragma unroll
			for (int b = 0; b < 2; b++) fb[b] = Xj[(wc * 32 + b * 16 + (lane & 15)) * (CB + 1) + 4 * kk + (lane >> 4)];
#pragma unroll
			for (int a = 0; a < 2; a++)
#pragma unroll
				for (int b = 0; b < 2; b++) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[a], fb[b], acc[a][b], 0, 0, 0);
		}
#pragma unroll
		for (int a = 0; a < 2; a++)
#pragma unroll
			for (int b = 0; b < 2; b++)
#pragma unroll
				for (int r = 0; r < 4; r++) {
					const int gi = i0 + wr * 32 + a * 16 + (lane >> 4) + 4 * r, gj = j0 + wc * 32 + b * 16 + (lane & 15);
					if (gi < ld && gj < ld) S.A[(size_t)gi * ld + gj] = acc[a][b][r];
				}
		return;
	}
	// ---- panel workgroup b: rows k0+CB+64(b-1) .. +63 (b >= 1); b = 0 publishes L_kk and y_k
	const int b = blockIdx.x, row0 = k0 + CB + 64 * (b - 1);
	// the chain runs on wavefronts 0 .. 2 (workgroup 0: wavefront 0 alone): lanes 0 .. CB-1 the diagonal block, lane CB the right-hand side, lane CB+1+i row 31 w + i of the 64
	const int xr = 31 * w + lane - (CB + 1), row = row0 + xr; const bool has_row = b > 0 && w < 3 && lane > CB && xr < 64 && row < ld;
	double a[CB];
	if (k0 == 0) {
		if (w > 2 || (b == 0 && w != 0)) return;
		const double *Arow = (const double *)__builtin_assume_aligned(S.A + (size_t)(has_row ? row : k0) * ld + k0, 16);
		const double *src = (const double *)__builtin_assume_aligned(S.A + (size_t)(k0 + (lane & (CB - 1))) * ld + k0, 16), *rh = (const double *)__builtin_assume_aligned(S.rhs + k0, 16);
#pragma unroll
		for (int c = 0; c < CB; c++) { const double v = src[c], bb = rh[c], xv = Arow[c]; a[c] = lane < CB ? (c <= lane ? v : 0.0) : (lane == CB ? bb : (has_row ? xv : 0.0)); }
	} else {
		double *Xo = sh, *Xk = sh + CT * (CB + 1); // -X_prev of the own rows | X_prev of rows k0 .. k0+CB-1
		// C of this wavefront's tiles straight into the accumulators: own rows 16 w .. 16 w + 15, both column halves; wavefronts 0..2 also take the tiles (0,0) (1,0) (1,1) of the diagonal block
		f64x4 co[2], cd; const int dr = (w == 0) ? 0 : 16, dc = (w == 2) ? 16 : 0;
#pragma unroll
		for (int h = 0; h < 2; h++)
#pragma unroll
			for (int r = 0; r < 4; r++) { const int gi = row0 + 16 * w + (lane >> 4) + 4 * r, gj = k0 + 16 * h + (lane & 15); co[h][r] = (b > 0 && gi < ld) ? S.A[(size_t)gi * ld + gj] : 0.0; }
#pragma unroll
		for (int r = 0; r < 4; r++) { const int gi = k0 + dr + (lane >> 4) + 4 * r, gj = k0 + dc + (lane & 15); cd[r] = (w < 3) ? S.A[(size_t)gi * ld + gj] : 0.0; }
		for (int e = tid; e < CT * CB; e += 256) { const int r = e / CB, c = e % CB; Xo[r * (CB + 1) + c] = (b > 0 && row0 + r < ld) ? -S.A[(size_t)(row0 + r) * ld + kp + c] : 0.0; }
		for (int e = tid; e < CB * CB; e += 256) { const int r = e / CB, c = e % CB; Xk[r * (CB + 1) + c] = S.A[(size_t)(k0 + r) * ld + kp + c]; }
		__syncthreads();
#pragma unroll
		for (int kk = 0; kk < CB / 4; kk++) {
			const double fa = Xo[(16 * w + (lane & 15)) * (CB + 1) + 4 * kk + (lane >> 4)];
			const double fd = -Xk[(dr + (lane & 15)) * (CB + 1) + 4 * kk + (lane >> 4)];
			double fb[2];
#pragma unroll
			for (int h = 0; h < 2; h++) fb[h] = Xk[(16 * h + (lane & 15)) * (CB + 1) + 4 * kk + (lane >> 4)];
#pragma unroll
			for (int h = 0; h < 2; h++) co[h] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa, fb[h], co[h], 0, 0, 0);
			if (w < 3) cd = __builtin_amdgcn_mfma_f64_16x16x4f64(fd, fb[dc >> 4], cd, 0, 0, 0);
		}
		__syncthreads(); // every wavefront has read its operands: the updated blocks take their place (own rows | diagonal block)
#pragma unroll
		for (int h = 0; h < 2; h++)
#pragma unroll
			for (int r = 0; r < 4; r++) Xo[(16 * w + (lane >> 4) + 4 * r) * (CB + 1) + 16 * h + (lane & 15)] = co[h][r];
		if (w < 3) {
#pragma unroll
			for (int r = 0; r < 4; r++) Xk[(dr + (lane >> 4) + 4 * r) * (CB + 1) + dc + (lane & 15)] = cd[r];
		}
		__syncthreads();
		if (w > 2 || (b == 0 && w != 0)) return;
		const double *rh = (const double *)__builtin_assume_aligned(S.rhs + k0, 16); const int xo = has_row ? xr : 0;
#pragma unroll
		for (int c = 0; c < CB; c++) { const double xv = Xo[xo * (CB + 1) + c], v = Xk[(lane & (CB - 1)) * (CB + 1) + c], bb = rh[c];
			a[c] = lane < CB ? (c <= lane ? v : 0.0) : (lane == CB ? bb : (has_row ? xv : 0.0)); }
	}
	double acc = 0;
	if (!chol_block_rows_regs(a, acc, lane)) { if (b == 0 && lane == 0) *S.flag = 1; return; }
	if (b == 0) {
		if (lane < CB) {
			double *dst = (double *)__builtin_assume_aligned(S.Ldiag + (size_t)(k0 + lane) * CB, 16);
#pragma unroll
			for (int c = 0; c < CB; c++) dst[c] = a[c];
		} else if (lane == CB) {
#pragma unroll
			for (int c = 0; c < CB; c++) S.y[k0 + c] = a[c];
		}
		return;
	}
	if (!has_row) return;
	double *Arow = (double *)__builtin_assume_aligned(S.A + (size_t)row * ld + k0, 16);
#pragma unroll
	for (int c = 0; c < CB; c++) Arow[c] = a[c];
	S.rhs[row] -= acc;
}

// ---- the whole factorisation in ONE launch ("a device-resident factorisation does not exist"): the panel steps and trailing updates above as phases of a
// persistent kernel, separated by grid-wide barriers (a counter in HBM: every workgroup adds one and waits for all `gridDim.x` of the phase; release / acquire fences at
// agent scope carry the matrix across the eight L2s). The grid is sized so that every workgroup is resident (at most 120 workgroups of 256 threads and 34 KB of LDS per
// factorisation, a few factorisations side by side on 256 CUs); a wait that lasts more than ~0.5 s gives up and raises flag 2 (reported as an error by the host) instead
// of hanging the device. Same arithmetic in the same order as k_chol_panel / k_chol_update: bit-identical factors.
__device__ __forceinline__ bool grid_barrier(unsigned *count, unsigned target, int *flag) {
	__syncthreads();
	if (threadIdx.x == 0) {
		__threadfence();
		atomicAdd(count, 1u);
		long long spins = 0;
		while ((int)(__hip_atomic_load(count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
			__builtin_amdgcn_s_sleep(4);
			if (++spins > (1ll << 23)) { atomicExch(flag, 2); break; }
		}
		__threadfence();
	}
	__syncthreads();
	return __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0;
}
__global__ void __launch_bounds__(256) k_chol_persistent(const BigSys S, unsigned *bar) {
	__shared__ double sh[2 * CT * (CB + 1)]; // trailing update: X_i | X_j ; panel step: L_kk with the right-hand side row (33 x 33) | reciprocal diagonal
	const int G = gridDim.x, b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6, ld = S.ld;
	unsigned target = 0; bool ok = true;
	for (int k0 = 0; k0 < ld && ok; k0 += CB) {
		const int below = ld - k0 - CB, nchunk = 1 + (below + 63) / 64; // chunk 0 publishes L_kk and y_k, chunk c >= 1 solves rows k0 + CB + 64 (c - 1) .. + 63 of the panel
		// ---- panel step (first wavefront of every workgroup: the diagonal block is factored redundantly, as in k_chol_panel, so that every workgroup sees a bad pivot itself)
		if (w == 0) {
			double *Ls = sh, *ri = sh + (CB + 1) * (CB + 1);
			for (int c = b; c < nchunk || c == b; c += G) { // (every workgroup factors at least once per step)
				const int row = k0 + CB + 64 * (c - 1) + lane; const bool has_row = c >= 1 && c < nchunk && row < ld;
				double *Arow = (double *)__builtin_assume_aligned(S.A + (size_t)(has_row ? row : k0) * ld + k0, 16);
				double x[CB];
				if (has_row) {
#pragma unroll
					for (int q = 0; q < CB; q++) x[q] = Arow[q];
				}
				double a[CB], rinv[CB];
				{ const double *src = (const double *)__builtin_assume_aligned(S.A + (size_t)(k0 + (lane & (CB - 1))) * ld + k0, 16);
				  const double *rh = (const double *)__builtin_assume_aligned(S.rhs + k0, 16);
#pragma unroll
				  for (int q = 0; q < CB; q++) { const double v = src[q], bb = rh[q]; a[q] = lane < CB ? (q <= lane ? v : 0.0) : (lane == CB ? bb : 0.0); } }
				if (!chol_block_regs(a, rinv, lane)) { if (b == 0 && lane == 0) atomicExch(S.flag, 1); ok = false; break; }
				if (c == 0) {
					if (lane < CB) {
						double *dst = (double *)__builtin_assume_aligned(S.Ldiag + (size_t)(k0 + lane) * CB, 16);
#pragma unroll
						for (int q = 0; q < CB; q++) dst[q] = a[q];
					} else if (lane == CB) {
#pragma unroll
						for (int q = 0; q < CB; q++) S.y[k0 + q] = a[q];
					}
				} else if (c < nchunk) {
					if (lane <= CB) {
#pragma unroll
						for (int q = 0; q < CB; q++) Ls[lane * (CB + 1) + q] = a[q];
					}
					if (lane == 0) {
#pragma unroll
						for (int q = 0; q < CB; q++) ri[q] = rinv[q];
					}
					__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
					if (has_row) {
						double acc = 0;
#pragma unroll
						for (int j = 0; j < CB; j++) {
							double sm = x[j];
#pragma unroll
							for (int m = 0; m < j; m++) sm -= x[m] * Ls[j * (CB + 1) + m];
							x[j] = sm * ri[j];
							acc += x[j] * Ls[CB * (CB + 1) + j];
						}
#pragma unroll
						for (int q = 0; q < CB; q++) Arow[q] = x[q];
						S.rhs[row] -= acc;
					}
					__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
				}
				if (c >= nchunk) break;
			}
		}
		// a bad pivot is seen by the first wavefront of EVERY workgroup (same data, same arithmetic): all of them leave the loop after this barrier
		{ __shared__ int bad; if (tid == 0) bad = 0; __syncthreads(); if (w == 0 && lane == 0 && !ok) bad = 1; __syncthreads(); ok = bad == 0; }
		target += G; if (!grid_barrier(bar, target, S.flag)) break;
		if (!ok || below <= 0) break;
		// ---- trailing update C -= X_i X_j^t, one 64 x 64 tile of the lower triangle per workgroup and pass
		double *Xi = sh, *Xj = sh + CT * (CB + 1);
		const int nt = (below + CT - 1) / CT, ntile = nt * (nt + 1) / 2, base = k0 + CB;
		for (int t = b; t < ntile; t += G) {
			int ti = 0; while ((ti + 1) * (ti + 2) / 2 <= t) ti++; const int tj = t - ti * (ti + 1) / 2;
			const int i0 = base + CT * ti, j0 = base + CT * tj, wr = w >> 1, wc = w & 1;
			const bool active = !(ti == tj && wc > wr);
			f64x4 acc[2][2];
#pragma unroll
			for (int a2 = 0; a2 < 2; a2++)
#pragma unroll
				for (int b2 = 0; b2 < 2; b2++)
#pragma unroll
					for (int r = 0; r < 4; r++) {
						const int gi = i0 + wr * 32 + a2 * 16 + (lane >> 4) + 4 * r, gj = j0 + wc * 32 + b2 * 16 + (lane & 15);
						acc[a2][b2][r] = (active && gi < ld && gj < ld) ? S.A[(size_t)gi * ld + gj] : 0.0;
					}
			for (int e = tid; e < CT * CB; e += 256) {
				const int r = e / CB, cc = e % CB;
				Xi[r * (CB + 1) + cc] = (i0 + r < ld) ? -S.A[(size_t)(i0 + r) * ld + k0 + cc] : 0.0;
				Xj[r * (CB + 1) + cc] = (j0 + r < ld) ? S.A[(size_t)(j0 + r) * ld + k0 + cc] : 0.0;
			}
			__syncthreads();
			if (active) {
#pragma unroll
				for (int kk = 0; kk < CB / 4; kk++) {
					double fa[2], fb[2];
#pragma unroll
					for (int a2 = 0; a2 < 2; a2++) fa[a2] = Xi[(wr * 32 + a2 * 16 + (lane & 15)) * (CB + 1) + 4 * kk + (lane >> 4)];
#pragma unroll
					for (int b2 = 0; b2 < 2; b2++) fb[b2] = Xj[(wc * 32 + b2 * 16 + (lane & 15)) * (CB + 1) + 4 * kk + (lane >> 4)];
#pragma unroll
					for (int a2 = 0; a2 < 2; a2++)
#pragma unroll
						for (int b2 = 0; b2 < 2; b2++) acc[a2][b2] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[a2], fb[b2], acc[a2][b2], 0, 0, 0);
				}
#pragma unroll
				for (int a2 = 0; a2 < 2; a2++)
#pragma unroll
					for (int b2 = 0; b2 < 2; b2++)
#pragma unroll
						for (int r = 0; r < 4; r++) {
							const int gi = i0 + wr * 32 + a2 * 16 + (lane >> 4) + 4 * r, gj = j0 + wc * 32 + b2 * 16 + (lane & 15);
							if (gi < ld && gj < ld) S.A[(size_t)gi * ld + gj] = acc[a2][b2][r];
						}
			}
			__syncthreads(); // the operands in LDS are rewritten by the next tile
		}
		target += G; if (!grid_barrier(bar, target, S.flag)) break;
	}
}

// L^t x = y in place in S.y (one workgroup per window): block rows from the last to the first. The CB x CB triangular solve runs in the first wavefront with column
// `lane` of L_kk in registers (lane c forms x_c, one v_readlane broadcast per step); all four wavefronts then eliminate x_k from the rows above, all CB loads of a row in
// flight. The registers of the NEXT diagonal block are requested right after the solve (its own are dead then), so that load rides under the elimination.
// (A pipelined form -- y in LDS, the first wavefront alone on the chain with the coupling block in registers, the other three eliminating one step behind -- was built and is
// slower, 628 against 221 us per launch: a step then lasts as long as the elimination by 192 threads, which is what bounds the step here too; profiles/r04_cfg4_timeline.txt.)
__global__ void __launch_bounds__(256) k_chol_bsub(const Gang G) {
	__shared__ double xs[CB];
	BIG_ENTER(); (void)p; const BigSys S = gang_sys(G, gw);
	const int tid = threadIdx.x, lane = tid & 63, ld = S.ld, nblk = ld / CB;
	if (*S.flag) return;
	double a[CB]; // a[m] = L[m][lane] of the current diagonal block (first wavefront)
	int cbv = CB; asm volatile("" : "+v"(cbv)); // (row stride of the diagonal factors as a vector value, as for ldv below)
	if (tid < 64) { const double *row = S.Ldiag + (size_t)(nblk - 1) * CB * CB + (lane < CB ? lane : 0);
#pragma unroll
		for (int m = 0; m < CB; m++) { const double v = *row; a[m] = (lane < CB) ? v : 0.0; row += cbv; }
	}
	for (int kb = nblk - 1; kb >= 0; kb--) {
		const int k0 = kb * CB;
		if (tid < 64) {
			double acc = lane < CB ? S.y[k0 + lane] : 0.0, dsel = 1.0;
#pragma unroll
			for (int c = 0; c < CB; c++) dsel = (lane == c) ? a[c] : dsel; // lane c picks its diagonal element (selects: the array stays in registers) ...
			const double dinv = 1.0 / dsel;                                  // ... and all the reciprocals of the diagonal are ONE division
#pragma unroll
			for (int c = CB - 1; c >= 0; c--) { // x_c = (y_c - sum_{m>c} L[m][c] x_m) / L[c][c]
				const double xc = lane_bcast(acc * dinv, c);
				acc = (lane == c) ? xc : ((lane < c) ? acc - a[c] * xc : acc); // (branch-free, the update pinned to its broadcast, one scheduling region per step: left to itself the compiler
				asm volatile("" : "+v"(acc)); __builtin_amdgcn_sched_barrier(0); //  issued the v_readlane pairs of many steps first and spilled them -- 154 SGPR spills, cf. chol_block_solve_regs)
			}
			if (lane < CB) { xs[lane] = acc; S.y[k0 + lane] = acc; }
			if (kb > 0) { const double *row = S.Ldiag + (size_t)(k0 - CB) * CB + (lane < CB ? lane : 0);
#pragma unroll
				for (int m = 0; m < CB; m++) { const double v = *row; a[m] = (lane < CB) ? v : 0.0; row += cbv; }
			}
		}
		__syncthreads();
		int ldv = ld; asm volatile("" : "+v"(ldv)); // (the row stride as a VECTOR value: with a scalar one the 32 row addresses below are 32 scalar register pairs -- 154 SGPR spills)
		for (int i = tid; i < k0; i += 256) { // y_i -= sum_c L[k0+c][i] x_c
			double v[CB]; const double *col = S.A + (size_t)k0 * ld + i;
#pragma unroll
			for (int c = 0; c < CB; c++) { v[c] = *col; col += ldv; }
			double s = 0;
#pragma unroll
			for (int c = 0; c < CB; c++) s += v[c] * xs[c];
			S.y[i] -= s;
		}
		__syncthreads();
	}
}

// ------------------------------------------------------------------------------------------------ grid-wide phases over ONE capsule (index p)
#define BIG_GID() (blockIdx.x * blockDim.x + threadIdx.x)
#define BIG_STRIDE() (gridDim.x * blockDim.x)

template <int FAM> __global__ void __launch_bounds__(256) kb_spantree(const Batch B, const DevParams prm, const Gang G, int only_needed, int use_skip) {
	BIG_ENTER();
	typedef Worker<FAM> W; typedef typename W::PO PO; typedef typename W::pose_t pose_t; constexpr int PD = W::PD;
	if (use_skip && BIG_FLAG()) return;
	const ProbDesc &d = B.desc[p]; const int cnt = only_needed ? d.n_need : d.n_pairs;
	for (int q = BIG_GID(); q < cnt; q += BIG_STRIDE()) {
		const int pr = only_needed ? B.need_idx[d.o_pair + q] : q;
		pose_t acc = PO::ident();
		for (int k = B.pair_path_off[d.o_ppoff + pr]; k < B.pair_path_off[d.o_ppoff + pr + 1]; k++) {
			const int pe = B.path_edge[d.o_path + k]; const pose_t ed = PO::ld(B.edge + (d.o_edge + (pe >> 1)) * PD);
			acc = (pe & 1) ? comp(acc, inv(ed)) : comp(acc, ed);
		}
		PO::st(B.pose + (d.o_pair + pr) * 2 * PD, acc); PO::st(B.pose + ((d.o_pair + pr) * 2 + 1) * PD, inv(acc));
	}
}
template <int FAM> __global__ void __launch_bounds__(256) kb_jac_init(const Batch B, const DevParams prm, const Gang G) {
	BIG_ENTER();
	const ProbDesc &d = B.desc[p];
	for (int i = BIG_GID(); i < d.n_valid; i += BIG_STRIDE()) { B.valid[d.o_valid + i] = 1; B.first_fail[d.o_valid + i] = 0x7fffffff; }
}
template <int FAM> __global__ void __launch_bounds__(128) kb_jac(const Batch B, const DevParams prm, const Gang G) {
	BIG_ENTER();
	Worker<FAM> Wk(B, B.desc[p], prm); const ProbDesc &d = B.desc[p];
	for (int b = BIG_GID(); b < d.n_bp + d.n_bf; b += BIG_STRIDE()) { if (b < d.n_bp) Wk.jac_dh_dp(b); else Wk.jac_dh_df(b - d.n_bp); }
}
template <int FAM> __global__ void __launch_bounds__(256) kb_jac_post(const Batch B, const DevParams prm, const Gang G) {
	BIG_ENTER(); // invalid-row semantics of Worker::phase_jacobians
	typedef Worker<FAM> W; constexpr int P = W::P, L = W::L, O = W::O; const ProbDesc &d = B.desc[p];
	for (int b = BIG_GID(); b < d.n_bp + d.n_bf; b += BIG_STRIDE()) {
		if (b < d.n_bp) {
			const int vs = B.obs_valid[d.o_obs + B.bp_res[d.o_bp + b]], ff = B.first_fail[d.o_valid + vs]; B.bp_ok[d.o_bp + b] = (ff == 0x7fffffff);
			if (ff != 0x7fffffff) B.valid[d.o_valid + vs] = 0;
			if (ff == b) { double *J = B.Jp + (long long)(d.o_bp + b) * O * P; for (int k = 0; k < O * P; k++) J[k] = 0; }
		} else {
			const int bb = b - d.n_bp, vs = B.obs_valid[d.o_obs + B.bf_res[d.o_bf + bb]], ff = B.first_fail[d.o_valid + vs]; B.bf_ok[d.o_bf + bb] = (ff == 0x7fffffff);
			if (ff != 0x7fffffff) B.valid[d.o_valid + vs] = 0;
			if (ff == b) { double *J = B.Jf + (long long)(d.o_bf + bb) * O * L; for (int k = 0; k < O * L; k++) J[k] = 0; }
		}
	}
}
// sum of N per-thread values over a 256-thread workgroup in a fixed order: wavefront sums (DPP tree), then the four wavefronts; the result is valid in thread k < N
template <int N> __device__ __forceinline__ double wg_sum(const double (&acc)[N], double *sh /* 4 N */) {
	const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
	for (int k = 0; k < N; k++) { const double v = wave_sum(acc[k]); if (lane == 0) sh[w * N + k] = v; }
	__syncthreads();
	const int k = threadIdx.x < N ? threadIdx.x : 0;
	return (sh[k] + sh[N + k]) + (sh[2 * N + k] + sh[3 * N + k]);
}
constexpr int BIG_HEAVY = 48;   // Hessian / Schur blocks with more terms than this are summed by a whole workgroup

// Hessian blocks (K6): one thread per block; the U_Ap blocks with many terms (an edge near the root of a deep window collects thousands of observations) are left
// to kb_hessian_heavy.
template <int FAM> __global__ void __launch_bounds__(128) kb_hessian(const Batch B, const DevParams prm, const Gang G) {
	BIG_ENTER(); int *ninv_out = G.iscal + gw * 8;
	typedef Worker<FAM> W; constexpr int P = W::P, L = W::L, O = W::O; Worker<FAM> Wk(B, B.desc[p], prm); const ProbDesc &d = B.desc[p];
	const double *Jp = B.Jp + d.o_bp * O * P, *Jf = B.Jf + d.o_bf * O * L; const unsigned char *rp = B.bp_ok + d.o_bp, *rf = B.bf_ok + d.o_bf;
	const bool latch = prm.solver != SRBA_SOLVER_NO_SCHUR_SPARSE_CHOL; int ninv = 0;
	const int total = d.n_hap + (W::T::REL ? 0 : d.n_hf + d.n_hapf);
	for (int b = BIG_GID(); b < total; b += BIG_STRIDE()) {
		if (b < d.n_hap) {
			const int tb = B.hap_term_off[d.o_hapoff + b], te = B.hap_term_off[d.o_hapoff + b + 1]; if (te - tb > BIG_HEAVY) continue;
			const long long g = d.o_hap + b; ninv += Wk.template hess_block<P, P>(B.HAp + g * P * P, latch ? B.HAp0 + g * P * P : nullptr, B.hap_t1 + d.o_hapt, B.hap_t2 + d.o_hapt, tb, te, Jp, Jp,
				rp, rp); }
		else if constexpr (!W::T::REL) {
			if (b < d.n_hap + d.n_hf) { const int q = b - d.n_hap; ninv += Wk.template hess_block<L, L>(B.Hf + (d.o_hf + q) * L * L, nullptr, B.hf_t1 + d.o_hft, B.hf_t2 + d.o_hft,
				B.hf_term_off[d.o_hfoff + q], B.hf_term_off[d.o_hfoff + q + 1], Jf, Jf, rf, rf); }
			else { const int q = b - d.n_hap - d.n_hf; ninv += Wk.template hess_block<P, L>(B.HApf + (d.o_hapf + q) * P * L, nullptr, B.hapf_t1 + d.o_hapft, B.hapf_t2 + d.o_hapft,
				B.hapf_term_off[d.o_hapfoff + q], B.hapf_term_off[d.o_hapfoff + q + 1], Jp, Jf, rp, rf); }
		}
	}
	if (ninv) atomicAdd(ninv_out, ninv);
}
// one workgroup per U_Ap block (grid = n_hap; the light ones return at once): terms strided over the threads, fixed-order reduction
template <int FAM> __global__ void __launch_bounds__(256) kb_hessian_heavy(const Batch B, const DevParams prm, const Gang G) {
	BIG_ENTER(); int *ninv_out = G.iscal + gw * 8;
	typedef Worker<FAM> W; constexpr int P = W::P, O = W::O; Worker<FAM> Wk(B, B.desc[p], prm); const ProbDesc &d = B.desc[p];
	__shared__ double sh[4 * P * P];
	if ((int)blockIdx.x >= d.n_hap) return;
	const int b = blockIdx.x, tb = B.hap_term_off[d.o_hapoff + b], te = B.hap_term_off[d.o_hapoff + b + 1];
	if (te - tb <= BIG_HEAVY) return;
	const double *Jp = B.Jp + d.o_bp * O * P; const unsigned char *rp = B.bp_ok + d.o_bp; const int *t1 = B.hap_t1 + d.o_hapt, *t2 = B.hap_t2 + d.o_hapt;
	double H[P * P]; int ninv = 0;
#pragma unroll
	for (int k = 0; k < P * P; k++) H[k] = 0;
	for (int t = tb + threadIdx.x; t < te; t += 256) {
		const int b1 = t1[t], b2 = t2[t];
		if (rp[b1] && rp[b2]) Wk.template hess_term<P, P>(H, Jp + (long long)b1 * O * P, Jp + (long long)b2 * O * P); else ninv++;
	}
	const double v = wg_sum<P * P>(H, sh) * ((prm.noise == SRBA_NOISE_IDENTITY) ? prm.inv_sigma : 1.0);
	if (threadIdx.x < P * P) { const long long g = d.o_hap + b; B.HAp[g * P * P + threadIdx.x] = v; if (prm.solver != SRBA_SOLVER_NO_SCHUR_SPARSE_CHOL) B.HAp0[g * P * P + threadIdx.x] = v; }
	if (ninv) atomicAdd(ninv_out, ninv);
}
// per-workgroup partial sums in a fixed order; kb_reduce adds them sequentially (deterministic)
template <int FAM> __global__ void __launch_bounds__(256) kb_residuals(const Batch B, const DevParams prm, const Gang G, int to_trial_copy, int use_skip) {
	BIG_ENTER();
	if (use_skip && BIG_FLAG()) return;
	Worker<FAM> Wk(B, B.desc[p], prm); constexpr int O = Worker<FAM>::O; const ProbDesc &d = B.desc[p];
	const int ngx = big_grid_dev(d.n_obs, 256); if ((int)blockIdx.x >= ngx) return; // the partition of the sum is the window's own
	double *out = to_trial_copy ? B.resid2 : B.resid, *partial = G.part + (size_t)gw * 3 * kBigPart;
	double acc = 0;
	for (int i = BIG_GID(); i < d.n_obs; i += ngx * 256) { double r[O]; acc += Wk.residual_row(i, r); for (int k = 0; k < O; k++) out[(long long)(d.o_obs + i) * O + k] = r[k]; }
	__shared__ double sh[4]; const double v = wave_sum(acc);
	if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
	__syncthreads();
	if (threadIdx.x == 0) partial[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
// kind: 0 = one partial per workgroup of kb_residuals (n_obs items), 1 = kb_maxdiag (nK + nF), 2 = kb_dot (n_scal)
__global__ void __launch_bounds__(256) kb_reduce(const Batch B, const Gang G, int which, int kind, int slot, int is_max) { // one workgroup per window, fixed order (n <= 4096)
	__shared__ double sh[4];
	BIG_ENTER(); const ProbDesc &d = B.desc[p];
	const int n = big_grid_dev(kind == 0 ? d.n_obs : (kind == 1 ? d.nK + d.nF : d.n_scal), 256);
	const double *partial = G.part + ((size_t)gw * 3 + which) * kBigPart; double *out = G.scal + gw * 16 + slot;
	double v = 0; for (int i = threadIdx.x; i < n; i += 256) v = is_max ? fmax(v, partial[i]) : v + partial[i];
	v = is_max ? wave_max(v) : wave_sum(v);
	if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
	__syncthreads();
	if (threadIdx.x == 0) *out = is_max ? fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3])) : (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
__global__ void kb_set_lambda(const Gang G, const GangLambda lam) { const int w = threadIdx.x; if (w < kGang && ((G.mask >> w) & 1u)) G.scal[w * 16 + BS_LAMBDA] = lam.v[w]; }
// what the host path did with device-to-device copies, for all windows of a gang at once: 0 = grad0 -> grad, 1 = grad -> grad0 (extension: the Schur kernels reduce grad in place),
	// 2 = residuals of the accepted trial -> current
__global__ void __launch_bounds__(256) kb_copy_vec(const Batch B, const Gang G, int kind, int O) {
	BIG_ENTER(); const ProbDesc &d = B.desc[p];
	if (kind == 2) { const double *s = B.resid2 + (long long)d.o_obs * O; double *t = B.resid + (long long)d.o_obs * O; for (long long k = BIG_GID(); k < (long long)d.n_obs * O;
		k += BIG_STRIDE()) t[k] = s[k]; }
	else { const double *s = (kind == 0 ? B.grad0 : B.grad) + d.o_scal; double *t = (kind == 0 ? B.grad : B.grad0) + d.o_scal; for (int k = BIG_GID(); k < d.n_scal; k += BIG_STRIDE()) t[k] = s[k]; }
}
template <int O, int M> __device__ __forceinline__ void grad_term(double (&acc)[M], const double *A, const double *r, const DevParams &prm) {
	double lr[O], a[O * M]; ldn<O>(lr, r); ldn<O * M>(a, A);
	if (prm.noise == SRBA_NOISE_CONSTANT_MATRIX) { double t[O]; for (int k = 0; k < O; k++) { double q = 0; for (int j = 0; j < O; j++) q += prm.lambda[k * O + j] * lr[j]; t[k] = q; } for (int k = 0;
		k < O; k++) lr[k] = t[k]; }
#pragma unroll
	for (int q = 0; q < M; q++) { double sm = 0;
#pragma unroll
		for (int k = 0; k < O; k++) sm += a[k * M + q] * lr[k];
		acc[q] += sm; }
}
// Gradient (K5). Workgroups 0 .. nK-1: one per unknown edge, its dh_dAp blocks strided over the 256 threads (an edge of a deep window has 10^3..10^4 of them),
// fixed-order reduction; the following workgroups: one thread per unknown landmark (tens of dh_df blocks each).
template <int FAM> __global__ void __launch_bounds__(256) kb_gradient(const Batch B, const DevParams prm, const Gang G) {
	BIG_ENTER(); const double *resid = B.resid;
	typedef Worker<FAM> W; constexpr int P = W::P, L = W::L, O = W::O; const ProbDesc &d = B.desc[p];
	const double sc = (prm.noise == SRBA_NOISE_IDENTITY) ? prm.inv_sigma : 1.0; double *g = B.grad + d.o_scal;
	__shared__ double sh[4 * P];
	if ((int)blockIdx.x >= d.nK + (d.nF + 255) / 256) return;
	if ((int)blockIdx.x < d.nK) {
		const int ci = blockIdx.x, bb = B.colp_off[d.o_colp + ci], be = B.colp_off[d.o_colp + ci + 1];
		double acc[P];
#pragma unroll
		for (int q = 0; q < P; q++) acc[q] = 0;
		for (int b = bb + threadIdx.x; b < be; b += 256) grad_term<O, P>(acc, B.Jp + (long long)(d.o_bp + b) * O * P, resid + (long long)(d.o_obs + B.bp_res[d.o_bp + b]) * O, prm);
		const double v = wg_sum<P>(acc, sh);
		if (threadIdx.x < P) g[ci * P + threadIdx.x] = v * sc;
		return;
	}
	if constexpr (!W::T::REL) {
		const int ci = ((int)blockIdx.x - d.nK) * 256 + threadIdx.x; if (ci >= d.nF) return;
		double acc[L];
#pragma unroll
		for (int q = 0; q < L; q++) acc[q] = 0;
		for (int b = B.colf_off[d.o_colf + ci]; b < B.colf_off[d.o_colf + ci + 1]; b++) grad_term<O, L>(acc, B.Jf + (long long)(d.o_bf + b) * O * L,
			resid + (long long)(d.o_obs + B.bf_res[d.o_bf + b]) * O, prm);
		for (int q = 0; q < L; q++) g[d.nK * P + ci * L + q] = acc[q] * sc;
	}
}
template <int FAM> __global__ void __launch_bounds__(256) kb_maxdiag(const Batch B, const DevParams prm, const Gang G) {
	BIG_ENTER();
	typedef Worker<FAM> W; constexpr int P = W::P, L = W::L; const ProbDesc &d = B.desc[p]; double mx = 0;
	const int ngx = big_grid_dev(d.nK + d.nF, 256); if ((int)blockIdx.x >= ngx) return;
	double *partial = G.part + ((size_t)gw * 3 + 1) * kBigPart;
	for (int i = BIG_GID(); i < d.nK + d.nF; i += ngx * 256) {
		if (i < d.nK) { const double *H = B.HAp + (d.o_hap + B.hap_diag[d.o_unk + i]) * P * P; for (int k = 0; k < P; k++) mx = fmax(mx, H[k * P + k]); }
		else { const int l = i - d.nK; const double *H = B.Hf + (d.o_hf + B.hf_diag[d.o_ulm + l]) * L * L; for (int k = 0; k < L; k++) mx = fmax(mx, H[k * L + k]); }
	}
	__shared__ double sh[4]; const double v = wave_max(mx);
	if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
	__syncthreads();
	if (threadIdx.x == 0) partial[blockIdx.x] = fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3]));
}
// rho denominator sum dl (lambda dl + g) and |g|_inf
template <int FAM> __global__ void __launch_bounds__(256) kb_dot(const Batch B, const DevParams prm, const Gang G, int use_skip) {
	BIG_ENTER();
	if (use_skip && BIG_FLAG()) return;
	const double lambda = G.scal[gw * 16 + BS_LAMBDA]; const ProbDesc &d = B.desc[p]; const double *dl = B.delta + d.o_scal, *g = B.grad + d.o_scal; double den = 0, ninf = 0;
	const int ngx = big_grid_dev(d.n_scal, 256); if ((int)blockIdx.x >= ngx) return;
	double *partial_den = G.part + ((size_t)gw * 3 + 1) * kBigPart, *partial_ninf = G.part + ((size_t)gw * 3 + 2) * kBigPart;
	for (int k = BIG_GID(); k < d.n_scal; k += ngx * 256) { den += dl[k] * (lambda * dl[k] + g[k]); ninf = fmax(ninf, fabs(g[k])); }
	__shared__ double sh[8]; const double v = wave_sum(den), m = wave_max(ninf);
	if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6] = v; sh[4 + (threadIdx.x >> 6)] = m; }
	__syncthreads();
	if (threadIdx.x == 0) { partial_den[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]); partial_ninf[blockIdx.x] = fmax(fmax(sh[4], sh[5]), fmax(sh[6], sh[7])); }
}
// ---- Schur complement (schur.h:180-311), grid-wide
template <int FAM> __global__ void __launch_bounds__(128) kb_schur_inv(const Batch B, const DevParams prm, const Gang G) {
	BIG_ENTER();
	typedef Worker<FAM> W; constexpr int P = W::P, L = W::L; const ProbDesc &d = B.desc[p]; const double lambda = G.scal[gw * 16 + BS_LAMBDA];
	if constexpr (!W::T::REL) {
		for (int l = BIG_GID(); l < d.nF; l += BIG_STRIDE()) {
			double M[L * L], Mi[L * L]; const double *src = B.Hf + (d.o_hf + B.hf_diag[d.o_ulm + l]) * L * L;
			for (int k = 0; k < L * L; k++) M[k] = src[k];
			for (int k = 0; k < L; k++) M[k * L + k] += lambda;
			const bool ok = fullpiv_inverse<L>(M, Mi); B.hf_ok[d.o_ulm + l] = ok ? 1 : 0;
			if (ok) for (int k = 0; k < L * L; k++) B.Hfinv[(d.o_ulm + l) * L * L + k] = Mi[k];
		}
		for (int k = BIG_GID(); k < d.n_hap * P * P; k += BIG_STRIDE()) B.HAp[d.o_hap * P * P + k] = B.HAp0[d.o_hap * P * P + k]; // restore from the snapshot (schur.h:188)
	}
}
// H_Ap(i,j) -= sum_l W_il Hf_l^-1 W_jl^t (schur.h:213-260). One workgroup per U_Ap block: the terms (landmarks seen through both edges, up to
// all of them for a diagonal block) are strided over the 256 threads, each term writes its own Y = W Hf^-1 where the gradient / back-substitution need it.
template <int FAM> __global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) kb_schur_reduce(const Batch B, const DevParams prm, const Gang G) {
	// (172 registers left to itself: two wavefronts per SIMD; the launch waits for its gathers)
	BIG_ENTER();
	typedef Worker<FAM> W; constexpr int P = W::P, L = W::L; const ProbDesc &d = B.desc[p];
	if constexpr (!W::T::REL) {
		__shared__ double sh[4 * P * P];
		if ((int)blockIdx.x >= d.n_hap) return;
		const int b = blockIdx.x, tb = B.sch_term_off[d.o_hapoff + b], te = B.sch_term_off[d.o_hapoff + b + 1];
		if (tb == te) return;
		double Hl[P * P];
#pragma unroll
		for (int k = 0; k < P * P; k++) Hl[k] = 0;
		for (int t = tb + threadIdx.x; t < te; t += 256) {
			const int l = B.sch_lm[d.o_sch + t]; if (!B.hf_ok[d.o_ulm + l]) continue;
			const double *W1 = B.HApf + (d.o_hapf + B.sch_b1[d.o_sch + t]) * P * L, *W2 = B.HApf + (d.o_hapf + B.sch_b2[d.o_sch + t]) * P * L, *Hi = B.Hfinv + (d.o_ulm + l) * L * L;
			double Y[P * L], w1[P * L], w2[P * L], hi[L * L];
			ldn<P * L>(w1, W1); ldn<P * L>(w2, W2); ldn<L * L>(hi, Hi); // 16-byte requests at 8-byte alignment: the launch is bound by the gathers' address traffic (every lane its own blocks),
				// not by flops, L2 locality or the reductions (profiles/r04_cfg4_schur_reduce_variants.txt)
#pragma unroll
			for (int i = 0; i < P; i++)
#pragma unroll
				for (int j = 0; j < L; j++) { double s = 0;
#pragma unroll
					for (int k = 0; k < L; k++) s += w1[i * L + k] * hi[k * L + j];
					Y[i * L + j] = s; }
#pragma unroll
			for (int i = 0; i < P; i++)
#pragma unroll
				for (int j = 0; j < P; j++) { double s = 0;
#pragma unroll
					for (int k = 0; k < L; k++) s += Y[i * L + k] * w2[j * L + k];
					Hl[i * P + j] += s; }
			const int yw = B.sch_yw[d.o_sch + t];
			if (yw >= 0) stn<P * L>(B.YW + (d.o_yw + yw) * P * L, Y);
		}
		const double v = wg_sum<P * P>(Hl, sh);
		if (threadIdx.x < P * P) B.HAp[(d.o_hap + b) * P * P + threadIdx.x] -= v;
	}
}
// The same reduction with ONE WAVEFRONT per U_Ap block (round 5). A deep window of cfg4 has ~2 300 U_Ap blocks with 16 ... 440 terms (mean 156: tools/diag_cfg4_schur_hist.py). The
// workgroup-per-block form above spends a block's life waiting: four dependent round trips (term range -> indices -> hf_ok -> operands), a 36-value reduction through LDS with two
// barriers, three workgroups per CU in flight -- 6 % of the FP64 rate, 530 us per launch for sixteen windows. Here a wavefront walks its block 64 terms at a time (packed 16-byte
// term records, the next pass's records requested with this pass's operands, hf_ok read beside them), keeps the 36 sums in registers across passes and reduces them once with DPP;
// no LDS, no barrier, four independent blocks per workgroup = twelve blocks per CU in flight, the longest blocks first (records sorted at upload: ProbDesc::n_vb). One wavefront
// sums a block in a fixed order: reproducible run to run. (Measured and dropped on the way: a LANE per light block, <= 64 terms -- 125 us for 5 % of the terms.)
#ifndef SRBA_SCHUR_WAVES
#define SRBA_SCHUR_WAVES 2 /* wavefronts per SIMD: 175 registers, nothing spilled; at three (168, 16 spilled in the term loop) the launch is 4 % of a cfg4 step slower, at four 17 % */
#endif
template <int FAM> __global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SRBA_SCHUR_WAVES, SRBA_SCHUR_WAVES))) kb_schur_reduce_wave(const Batch B, const DevParams prm, const Gang G, int xcd_rows) {
	// xcd_rows (gridDim.y a multiple of 8): the workgroups of window w all run on XCD w % 8 -- workgroups go to the XCDs round-robin in dispatch order (x fastest), so the linear
	// id is re-read as (XCD, position on that XCD) and the position as (window, workgroup of the window). A window's W blocks (8 MB, each read ~ 13 times) then stay in ONE L2
	// instead of passing through all eight.
	int gw = blockIdx.y, bx = blockIdx.x;
	if (xcd_rows) { const unsigned id = blockIdx.x + blockIdx.y * gridDim.x, xcd = id & 7u, j = id >> 3; gw = (int)(xcd + 8u * (j / gridDim.x)); bx = (int)(j % gridDim.x); }
	if (!((G.mask >> gw) & 1u)) return; const int p = G.p[gw];
	typedef Worker<FAM> W; constexpr int P = W::P, L = W::L; const ProbDesc &d = B.desc[p];
	if constexpr (!W::T::REL) {
		const int v = bx * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63; if (v >= d.n_vb) return;
		const int *vr = B.sch_vb + 4 * (d.o_vb + v);
		const int t0 = __builtin_amdgcn_readfirstlane(vr[0]), t1 = __builtin_amdgcn_readfirstlane(vr[1]), b = __builtin_amdgcn_readfirstlane(vr[2]);
		const int4 *tr = (const int4 *)(B.sch_rec + 4 * d.o_sch); const double *HApf = B.HApf + d.o_hapf * P * L, *Hfinv = B.Hfinv + d.o_ulm * L * L; const int *ok = B.hf_ok + d.o_ulm;
		double Hl[P * P];
#pragma unroll
		for (int k = 0; k < P * P; k++) Hl[k] = 0;
		int4 nx = tr[min(t0 + lane, t1 - 1)];
		for (int t = t0; t < t1; t += 64) {
			const int4 cur = nx; const bool act = t + lane < t1;
			if (t + 64 < t1) nx = tr[min(t + 64 + lane, t1 - 1)]; // (clamped, unconditional: the records of the next pass travel with the operands of this one)
			double w1[P * L], w2[P * L], hi[L * L];
			ldn<P * L>(w1, HApf + (long long)cur.y * P * L); ldn<P * L>(w2, HApf + (long long)cur.z * P * L); ldn<L * L>(hi, Hfinv + (long long)cur.x * L * L);
			const bool use = act && ok[cur.x] != 0;
			double *Yout = (use && cur.w >= 0) ? B.YW + (d.o_yw + cur.w) * P * L : nullptr;
#pragma unroll
			for (int i = 0; i < P; i++) { double y[L];
#pragma unroll
				for (int j = 0; j < L; j++) { double s = 0;
#pragma unroll
					for (int k = 0; k < L; k++) s += w1[i * L + k] * hi[k * L + j];
					y[j] = s; }
#pragma unroll
				for (int j = 0; j < P; j++) { double s = 0;
#pragma unroll
					for (int k = 0; k < L; k++) s += y[k] * w2[j * L + k];
					Hl[i * P + j] += use ? s : 0.0; }
				if (Yout) {
#pragma unroll
					for (int j = 0; j < L; j++) Yout[i * L + j] = y[j];
				}
			}
		}
		double mine = 0;
#pragma unroll
		for (int k = 0; k < P * P; k++) { const double tot = wave_sum(Hl[k]); mine = (lane == k) ? tot : mine; }
		if (lane < P * P) B.HAp[(d.o_hap + b) * P * P + lane] -= mine;
	}
}
// g_Ap(i) -= sum_l Y_il g_f(l) (schur.h:262-283): one workgroup per unknown edge
template <int FAM> __global__ void __launch_bounds__(256) kb_schur_grad(const Batch B, const DevParams prm, const Gang G) {
	BIG_ENTER();
	typedef Worker<FAM> W; constexpr int P = W::P, L = W::L; const ProbDesc &d = B.desc[p];
	if constexpr (!W::T::REL) {
		__shared__ double sh[4 * P];
		double *g = B.grad + d.o_scal; const double *gf = g + d.nK * P;
		if ((int)blockIdx.x >= d.nK) return;
		const int i = blockIdx.x, b = B.hap_diag[d.o_unk + i];
		double acc[P];
#pragma unroll
		for (int r = 0; r < P; r++) acc[r] = 0;
		for (int t = B.sch_term_off[d.o_hapoff + b] + threadIdx.x; t < B.sch_term_off[d.o_hapoff + b + 1]; t += 256) {
			const int l = B.sch_lm[d.o_sch + t]; if (!B.hf_ok[d.o_ulm + l]) continue;
			double Y[P * L], gl[L]; ldn<P * L>(Y, B.YW + (d.o_yw + B.sch_yw[d.o_sch + t]) * P * L); ldn<L>(gl, gf + l * L);
#pragma unroll
			for (int r = 0; r < P; r++) { double s = 0;
#pragma unroll
				for (int k = 0; k < L; k++) s += Y[r * L + k] * gl[k];
				acc[r] += s; }
		}
		const double v = wg_sum<P>(acc, sh);
		if (threadIdx.x < P) g[i * P + threadIdx.x] -= v;
	}
}
template <int FAM> __global__ void __launch_bounds__(128) kb_schur_features(const Batch B, const DevParams prm, const Gang G, int use_skip) {
	BIG_ENTER();
	typedef Worker<FAM> W; constexpr int P = W::P, L = W::L; const ProbDesc &d = B.desc[p];
	if (use_skip && BIG_FLAG()) return;
	if constexpr (!W::T::REL) {
		double *g = B.grad + d.o_scal, *dl = B.delta + d.o_scal;
		for (int l = BIG_GID(); l < d.nF; l += BIG_STRIDE()) {
			if (!B.hf_ok[d.o_ulm + l]) continue;
			double gl[L]; for (int k = 0; k < L; k++) gl[k] = g[d.nK * P + l * L + k];
			for (int q = B.lm_hapf_off[d.o_lmoff + l]; q < B.lm_hapf_off[d.o_lmoff + l + 1]; q++) {
				const int hb = B.lm_hapf_idx[d.o_hapf + q], i = B.hapf_i[d.o_hapf + hb]; double Wm[P * L], di[P]; ldn<P * L>(Wm, B.HApf + (d.o_hapf + hb) * P * L); ldn<P>(di, dl + i * P);
				for (int k = 0; k < L; k++) { double s = 0; for (int r = 0; r < P; r++) s += Wm[r * L + k] * di[r]; gl[k] -= s; }
			}
			const double *Hi = B.Hfinv + (d.o_ulm + l) * L * L;
			for (int k = 0; k < L; k++) g[d.nK * P + l * L + k] = gl[k];
			for (int r = 0; r < L; r++) { double s = 0; for (int k = 0; k < L; k++) s += Hi[r * L + k] * gl[k]; dl[d.nK * P + l * L + r] = s; }
		}
	}
}
// (H + lambda I) into the dense lower triangle + right-hand side; identity padding up to ld
__global__ void kb_dense_clear(const Gang G) { BIG_ENTER(); (void)p; const BigSys S = gang_sys(G, gw); for (size_t k = BIG_GID(); k < (size_t)S.ld * S.ld; k += BIG_STRIDE()) {
	const int r = (int)(k / S.ld), c = (int)(k % S.ld); S.A[k] = (r == c && r >= S.n) ? 1.0 : 0.0; } if (BIG_GID() == 0) *S.flag = 0; }
template <int FAM> __global__ void __launch_bounds__(128) kb_dense_assemble(const Batch B, const DevParams prm, const Gang G, int full_system) {
	BIG_ENTER(); const BigSys S = gang_sys(G, gw);
	typedef Worker<FAM> W; constexpr int P = W::P, L = W::L; const ProbDesc &d = B.desc[p]; const double lambda = G.scal[gw * 16 + BS_LAMBDA];
	const int total = d.n_hap + (full_system ? d.n_hapf + d.n_hf : 0);
	for (int b = BIG_GID(); b < total; b += BIG_STRIDE()) {
		if (b < d.n_hap) { // upper block (i <= j) -> lower triangle: A[Pj+q][Pi+r] = H[r][q]
			const int i = B.hap_i[d.o_hap + b], j = B.hap_j[d.o_hap + b]; const double *H = B.HAp + (d.o_hap + b) * P * P;
			for (int r = 0; r < P; r++) for (int q = 0; q < P; q++) { if (i == j && q < r) continue; S.A[(size_t)(P * j + q) * S.ld + P * i + r] = H[r * P + q] + ((i == j && r == q) ? lambda : 0.0); }
		} else if (b < d.n_hap + d.n_hapf) { // (edge i, landmark j): row of the landmark, column of the edge
			const int q0 = b - d.n_hap, i = B.hapf_i[d.o_hapf + q0], j = B.hapf_j[d.o_hapf + q0]; const double *H = B.HApf + (d.o_hapf + q0) * P * L;
			for (int r = 0; r < P; r++) for (int q = 0; q < L; q++) S.A[(size_t)(P * d.nK + L * j + q) * S.ld + P * i + r] = H[r * L + q];
		} else {
			const int q0 = b - d.n_hap - d.n_hapf, i = B.hf_i[d.o_hf + q0], j = B.hf_j[d.o_hf + q0]; const double *H = B.Hf + (d.o_hf + q0) * L * L;
			for (int r = 0; r < L; r++) for (int q = 0; q < L; q++) { if (i == j && q < r) continue; S.A[(size_t)(P * d.nK + L * j + q) * S.ld + P * d.nK + L * i + r] = H[r * L + q] + ((i == j && r
				== q) ? lambda : 0.0); }
		}
	}
	const double *g = B.grad + d.o_scal;
	for (int k = BIG_GID(); k < S.ld; k += BIG_STRIDE()) S.rhs[k] = k < S.n ? g[k] : 0.0;
}
__global__ void kb_take_delta(const Batch B, const Gang G) { BIG_ENTER(); const BigSys S = gang_sys(G, gw); if (*S.flag) return; const ProbDesc &d = B.desc[p]; double *dl = B.delta + d.o_scal;
	for (int k = BIG_GID(); k < d.n_scal; k += BIG_STRIDE()) if (k < S.n) dl[k] = S.y[k]; else if (S.n == d.n_scal) dl[k] = 0; }
// K12 backup + K11 apply / restore
template <int FAM> __global__ void __launch_bounds__(128) kb_apply(const Batch B, const DevParams prm, const Gang G, int use_skip) {
	BIG_ENTER();
	typedef Worker<FAM> W; typedef typename W::PO PO; constexpr int P = W::P, L = W::L, PD = W::PD; const ProbDesc &d = B.desc[p]; const double *dl = B.delta + d.o_scal;
	if (use_skip && BIG_FLAG()) return;
	for (int i = BIG_GID(); i < d.nK + d.nF * L + d.n_req; i += BIG_STRIDE()) {
		if (i < d.nK) { double *e = B.edge + (d.o_edge + i) * PD, *o = B.old_edge + (d.o_unk + i) * PD; for (int k = 0; k < PD; k++) o[k] = e[k]; PO::st(e, comp(PO::expm(dl + i * P), PO::ld(e))); }
		else if (i < d.nK + d.nF * L) { const int k = i - d.nK; B.old_ulm[d.o_ulm * L + k] = B.ulm[d.o_ulm * L + k]; B.ulm[d.o_ulm * L + k] += dl[d.nK * P + k]; }
		else { const int r = i - d.nK - d.nF * L; const double *s = B.pose + (d.o_pair * 2 + B.req_idx[d.o_req + r]) * PD; double *o = B.old_pose + (d.o_req + r) * PD; for (int k = 0; k < PD;
			k++) o[k] = s[k]; }
	}
}
template <int FAM> __global__ void __launch_bounds__(128) kb_restore(const Batch B, const DevParams prm, const Gang G) {
	BIG_ENTER();
	typedef Worker<FAM> W; constexpr int L = W::L, PD = W::PD; const ProbDesc &d = B.desc[p];
	for (int i = BIG_GID(); i < d.nK + d.nF * L + d.n_req; i += BIG_STRIDE()) {
		if (i < d.nK) { for (int k = 0; k < PD; k++) B.edge[(d.o_edge + i) * PD + k] = B.old_edge[(d.o_unk + i) * PD + k]; }
		else if (i < d.nK + d.nF * L) { const int k = i - d.nK; B.ulm[d.o_ulm * L + k] = B.old_ulm[d.o_ulm * L + k]; }
		else { const int r = i - d.nK - d.nF * L; double *s = B.pose + (d.o_pair * 2 + B.req_idx[d.o_req + r]) * PD; const double *o = B.old_pose + (d.o_req + r) * PD; for (int k = 0; k < PD;
			k++) s[k] = o[k]; }
	}
}
template <int FAM> __global__ void __launch_bounds__(128) kb_cov_recovery(const Batch B, const DevParams prm, const Gang G, int schur_active) {
	BIG_ENTER();
	typedef Worker<FAM> W; constexpr int L = W::L; const ProbDesc &d = B.desc[p];
	if constexpr (!W::T::REL) for (int l = BIG_GID(); l < d.nF; l += BIG_STRIDE()) {
		const bool ok = prm.cov_recovery == 1 && (schur_active ? (B.hf_ok[d.o_ulm + l] != 0) : true); B.ulm_inf_valid[d.o_ulm + l] = ok ? 1 : 0;
		if (ok) for (int k = 0; k < L * L; k++) B.ulm_inf[(d.o_ulm + l) * L * L + k] = B.Hf[(d.o_hf + B.hf_diag[d.o_ulm + l]) * L * L + k];
	}
}

#endif // SRBA_BIG_DECLS_ONLY
} // namespace srbadev
