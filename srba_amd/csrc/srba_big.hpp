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

typedef double f64x4 __attribute__((ext_vector_type(4)));

// In-LDS Cholesky of a CB x CB block (row-major, leading dimension CB+1) by one wavefront: lane = row. Uniform result.
__device__ __forceinline__ bool chol_block_lds(double *S, int lane) {
	bool ok = true;
	for (int j = 0; j < CB; j++) {
		const double d = S[j * (CB + 1) + j];
		if (!(d > 0.0)) { ok = false; break; }
		const double r = 1.0 / sqrt(d);
		double lij = 0;
		if (lane > j && lane < CB) { lij = S[lane * (CB + 1) + j] * r; S[lane * (CB + 1) + j] = lij; }
		if (lane == j) S[j * (CB + 1) + j] = d * r;
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		if (lane > j && lane < CB) for (int k = j + 1; k <= lane; k++) S[lane * (CB + 1) + k] -= lij * S[k * (CB + 1) + j];
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
	}
	return ok;
}

// Panel step k0: workgroup 0 owns the diagonal block (writes L_kk to Ldiag and y_k), workgroup b >= 1 owns rows k0+CB+64(b-1) .. +63 of the panel.
__global__ void __launch_bounds__(64) k_chol_panel(const BigSys S, int k0) {
	__shared__ double Ls[CB * (CB + 1)]; __shared__ double ys[CB];
	const int lane = threadIdx.x, ld = S.ld;
	for (int e = lane; e < CB * CB; e += 64) { const int r = e / CB, c = e % CB; Ls[r * (CB + 1) + c] = (c <= r) ? S.A[(size_t)(k0 + r) * ld + k0 + c] : 0.0; }
	__syncthreads();
	if (!chol_block_lds(Ls, lane)) { if (blockIdx.x == 0 && lane == 0) *S.flag = 1; return; }
	// y_k = L_kk^-1 rhs_k (every workgroup needs it; only workgroup 0 publishes it)
	if (lane < CB) ys[lane] = S.rhs[k0 + lane];
	__syncthreads();
	for (int j = 0; j < CB; j++) {
		if (lane == j) ys[j] = ys[j] / Ls[j * (CB + 1) + j];
		__syncthreads();
		if (lane > j && lane < CB) ys[lane] -= Ls[lane * (CB + 1) + j] * ys[j];
		__syncthreads();
	}
	if (blockIdx.x == 0) {
		for (int e = lane; e < CB * CB; e += 64) { const int r = e / CB, c = e % CB; S.Ldiag[(size_t)(k0 + r) * CB + c] = Ls[r * (CB + 1) + c]; }
		if (lane < CB) S.y[k0 + lane] = ys[lane];
		return;
	}
	const int row = k0 + CB + 64 * (blockIdx.x - 1) + lane;
	if (row >= S.ld) return;
	double *Arow = S.A + (size_t)row * ld + k0;
	double x[CB];
#pragma unroll
	for (int c = 0; c < CB; c++) x[c] = Arow[c];
	double acc = 0;
#pragma unroll
	for (int j = 0; j < CB; j++) { // x_j = (a_j - sum_{m<j} x_m L[j][m]) / L[j][j]
		double s = x[j];
#pragma unroll
		for (int m = 0; m < j; m++) s -= x[m] * Ls[j * (CB + 1) + m];
		x[j] = s / Ls[j * (CB + 1) + j];
		acc += x[j] * ys[j];
	}
#pragma unroll
	for (int c = 0; c < CB; c++) Arow[c] = x[c];
	S.rhs[row] -= acc;
}

// Trailing update after panel k0: tile (ti, tj), tj <= ti, of the matrix below/right of the panel: C -= X_i X_j^t, X = A[:, k0 .. k0+CB)
__global__ void __launch_bounds__(256) k_chol_update(const BigSys S, int k0, int ntile) {
	__shared__ double Xi[CT * (CB + 1)], Xj[CT * (CB + 1)];
	// linear tile index -> (ti, tj) of the lower triangle
	int t = blockIdx.x, ti = 0; while ((ti + 1) * (ti + 2) / 2 <= t) ti++; const int tj = t - ti * (ti + 1) / 2;
	(void)ntile;
	const int base = k0 + CB, i0 = base + CT * ti, j0 = base + CT * tj, ld = S.ld, tid = threadIdx.x;
	for (int e = tid; e < CT * CB; e += 256) {
		const int r = e / CB, c = e % CB;
		Xi[r * (CB + 1) + c] = (i0 + r < ld) ? S.A[(size_t)(i0 + r) * ld + k0 + c] : 0.0;
		Xj[r * (CB + 1) + c] = (j0 + r < ld) ? S.A[(size_t)(j0 + r) * ld + k0 + c] : 0.0;
	}
	__syncthreads();
	const int w = tid >> 6, lane = tid & 63, wr = w >> 1, wc = w & 1;
	if (ti == tj && wc > wr) return; // strictly upper part of a diagonal tile
	f64x4 acc[2][2];
#pragma unroll
	for (int a = 0; a < 2; a++)
#pragma unroll
		for (int b = 0; b < 2; b++) acc[a][b] = (f64x4){0, 0, 0, 0};
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
			for (int r = 0; r < 4; r++) { // D: col = lane & 15, row = (lane >> 4) + 4 r
				const int gi = i0 + wr * 32 + a * 16 + (lane >> 4) + 4 * r, gj = j0 + wc * 32 + b * 16 + (lane & 15);
				if (gi < ld && gj < ld) S.A[(size_t)gi * ld + gj] -= acc[a][b][r];
			}
}

// L^t x = y in place in S.y (one workgroup): block rows from the last to the first
__global__ void __launch_bounds__(256) k_chol_bsub(const BigSys S) {
	__shared__ double Ls[CB * (CB + 1)], xs[CB];
	const int tid = threadIdx.x, ld = S.ld, nblk = ld / CB;
	for (int kb = nblk - 1; kb >= 0; kb--) {
		const int k0 = kb * CB;
		for (int e = tid; e < CB * CB; e += 256) { const int r = e / CB, c = e % CB; Ls[r * (CB + 1) + c] = S.Ldiag[(size_t)(k0 + r) * CB + c]; }
		if (tid < CB) xs[tid] = S.y[k0 + tid];
		__syncthreads();
		for (int c = CB - 1; c >= 0; c--) { // x_c = (y_c - sum_{m>c} L[m][c] x_m) / L[c][c]
			if (tid == c) xs[c] = xs[c] / Ls[c * (CB + 1) + c];
			__syncthreads();
			if (tid < c) xs[tid] -= Ls[c * (CB + 1) + tid] * xs[c];
			__syncthreads();
		}
		if (tid < CB) S.y[k0 + tid] = xs[tid];
		for (int i = tid; i < k0; i += 256) { // y_i -= sum_c L[k0+c][i] x_c
			double s = 0;
#pragma unroll 8
			for (int c = 0; c < CB; c++) s += S.A[(size_t)(k0 + c) * ld + i] * xs[c];
			S.y[i] -= s;
		}
		__syncthreads();
	}
}

// ------------------------------------------------------------------------------------------------ grid-wide phases over ONE capsule (index p)
#define BIG_GID() (blockIdx.x * blockDim.x + threadIdx.x)
#define BIG_STRIDE() (gridDim.x * blockDim.x)

template <int FAM> __global__ void __launch_bounds__(256) kb_spantree(const Batch B, const DevParams prm, int p, int only_needed) {
	typedef Worker<FAM> W; typedef typename W::PO PO; typedef typename W::pose_t pose_t; constexpr int PD = W::PD;
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
template <int FAM> __global__ void __launch_bounds__(256) kb_jac_init(const Batch B, const DevParams prm, int p) {
	const ProbDesc &d = B.desc[p];
	for (int i = BIG_GID(); i < d.n_valid; i += BIG_STRIDE()) { B.valid[d.o_valid + i] = 1; B.first_fail[d.o_valid + i] = 0x7fffffff; }
}
template <int FAM> __global__ void __launch_bounds__(128) kb_jac(const Batch B, const DevParams prm, int p) {
	Worker<FAM> Wk(B, B.desc[p], prm); const ProbDesc &d = B.desc[p];
	for (int b = BIG_GID(); b < d.n_bp + d.n_bf; b += BIG_STRIDE()) { if (b < d.n_bp) Wk.jac_dh_dp(b); else Wk.jac_dh_df(b - d.n_bp); }
}
template <int FAM> __global__ void __launch_bounds__(256) kb_jac_post(const Batch B, const DevParams prm, int p) { // invalid-row semantics of Worker::phase_jacobians
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
template <int FAM> __global__ void __launch_bounds__(128) kb_hessian(const Batch B, const DevParams prm, int p, int *ninv_out) {
	typedef Worker<FAM> W; constexpr int P = W::P, L = W::L, O = W::O; Worker<FAM> Wk(B, B.desc[p], prm); const ProbDesc &d = B.desc[p];
	const double *Jp = B.Jp + d.o_bp * O * P, *Jf = B.Jf + d.o_bf * O * L; const unsigned char *rp = B.bp_ok + d.o_bp, *rf = B.bf_ok + d.o_bf;
	const bool latch = prm.solver != SRBA_SOLVER_NO_SCHUR_SPARSE_CHOL; int ninv = 0;
	const int total = d.n_hap + (W::T::REL ? 0 : d.n_hf + d.n_hapf);
	for (int b = BIG_GID(); b < total; b += BIG_STRIDE()) {
		if (b < d.n_hap) { const long long g = d.o_hap + b; ninv += Wk.template hess_block<P, P>(B.HAp + g * P * P, latch ? B.HAp0 + g * P * P : nullptr, B.hap_t1 + d.o_hapt, B.hap_t2 + d.o_hapt, B.hap_term_off[d.o_hapoff + b], B.hap_term_off[d.o_hapoff + b + 1], Jp, Jp, rp, rp); }
		else if constexpr (!W::T::REL) {
			if (b < d.n_hap + d.n_hf) { const int q = b - d.n_hap; ninv += Wk.template hess_block<L, L>(B.Hf + (d.o_hf + q) * L * L, nullptr, B.hf_t1 + d.o_hft, B.hf_t2 + d.o_hft, B.hf_term_off[d.o_hfoff + q], B.hf_term_off[d.o_hfoff + q + 1], Jf, Jf, rf, rf); }
			else { const int q = b - d.n_hap - d.n_hf; ninv += Wk.template hess_block<P, L>(B.HApf + (d.o_hapf + q) * P * L, nullptr, B.hapf_t1 + d.o_hapft, B.hapf_t2 + d.o_hapft, B.hapf_term_off[d.o_hapfoff + q], B.hapf_term_off[d.o_hapfoff + q + 1], Jp, Jf, rp, rf); }
		}
	}
	if (ninv) atomicAdd(ninv_out, ninv);
}
// per-workgroup partial sums in a fixed order; kb_reduce adds them sequentially (deterministic)
template <int FAM> __global__ void __launch_bounds__(256) kb_residuals(const Batch B, const DevParams prm, int p, double *out, double *partial) {
	Worker<FAM> Wk(B, B.desc[p], prm); constexpr int O = Worker<FAM>::O; const ProbDesc &d = B.desc[p];
	double acc = 0;
	for (int i = BIG_GID(); i < d.n_obs; i += BIG_STRIDE()) { double r[O]; acc += Wk.residual_row(i, r); for (int k = 0; k < O; k++) out[(long long)(d.o_obs + i) * O + k] = r[k]; }
	__shared__ double sh[4]; const double v = wave_sum(acc);
	if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
	__syncthreads();
	if (threadIdx.x == 0) partial[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
__global__ void kb_reduce(const double *partial, int n, double *out, int is_max) { // one thread: n is at most a few thousand
	if (threadIdx.x || blockIdx.x) return;
	double s = 0; for (int i = 0; i < n; i++) s = is_max ? fmax(s, partial[i]) : s + partial[i];
	*out = s;
}
template <int FAM> __global__ void __launch_bounds__(128) kb_gradient(const Batch B, const DevParams prm, int p, const double *resid) { // one thread per unknown column, blocks in ascending order
	typedef Worker<FAM> W; constexpr int P = W::P, L = W::L, O = W::O; const ProbDesc &d = B.desc[p];
	const double sc = (prm.noise == SRBA_NOISE_IDENTITY) ? prm.inv_sigma : 1.0; double *g = B.grad + d.o_scal;
	for (int col = BIG_GID(); col < d.nK + d.nF; col += BIG_STRIDE()) {
		const bool isp = col < d.nK; const int M = isp ? P : L, ci = isp ? col : col - d.nK;
		const int bb = isp ? B.colp_off[d.o_colp + ci] : B.colf_off[d.o_colf + ci], be = isp ? B.colp_off[d.o_colp + ci + 1] : B.colf_off[d.o_colf + ci + 1];
		double acc[6] = {0, 0, 0, 0, 0, 0};
		for (int b = bb; b < be; b++) {
			const double *A = isp ? B.Jp + (long long)(d.o_bp + b) * O * P : B.Jf + (long long)(d.o_bf + b) * O * L;
			const double *r = resid + (long long)(d.o_obs + (isp ? B.bp_res[d.o_bp + b] : B.bf_res[d.o_bf + b])) * O;
			double lr[O]; for (int k = 0; k < O; k++) lr[k] = r[k];
			if (prm.noise == SRBA_NOISE_CONSTANT_MATRIX) { double t[O]; for (int k = 0; k < O; k++) { double q = 0; for (int j = 0; j < O; j++) q += prm.lambda[k * O + j] * lr[j]; t[k] = q; } for (int k = 0; k < O; k++) lr[k] = t[k]; }
			for (int q = 0; q < M; q++) { double sm = 0; for (int k = 0; k < O; k++) sm += A[k * M + q] * lr[k]; acc[q] += sm; }
		}
		double *go = isp ? g + ci * P : g + d.nK * P + ci * L;
		for (int q = 0; q < M; q++) go[q] = acc[q] * sc;
	}
}
template <int FAM> __global__ void __launch_bounds__(256) kb_maxdiag(const Batch B, const DevParams prm, int p, double *partial) {
	typedef Worker<FAM> W; constexpr int P = W::P, L = W::L; const ProbDesc &d = B.desc[p]; double mx = 0;
	for (int i = BIG_GID(); i < d.nK + d.nF; i += BIG_STRIDE()) {
		if (i < d.nK) { const double *H = B.HAp + (d.o_hap + B.hap_diag[d.o_unk + i]) * P * P; for (int k = 0; k < P; k++) mx = fmax(mx, H[k * P + k]); }
		else { const int l = i - d.nK; const double *H = B.Hf + (d.o_hf + B.hf_diag[d.o_ulm + l]) * L * L; for (int k = 0; k < L; k++) mx = fmax(mx, H[k * L + k]); }
	}
	__shared__ double sh[4]; const double v = wave_max(mx);
	if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
	__syncthreads();
	if (threadIdx.x == 0) partial[blockIdx.x] = fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3]));
}
// rho denominator sum dl (lambda dl + g) and |g|_inf
template <int FAM> __global__ void __launch_bounds__(256) kb_dot(const Batch B, const DevParams prm, int p, double lambda, double *partial_den, double *partial_ninf) {
	const ProbDesc &d = B.desc[p]; const double *dl = B.delta + d.o_scal, *g = B.grad + d.o_scal; double den = 0, ninf = 0;
	for (int k = BIG_GID(); k < d.n_scal; k += BIG_STRIDE()) { den += dl[k] * (lambda * dl[k] + g[k]); ninf = fmax(ninf, fabs(g[k])); }
	__shared__ double sh[8]; const double v = wave_sum(den), m = wave_max(ninf);
	if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6] = v; sh[4 + (threadIdx.x >> 6)] = m; }
	__syncthreads();
	if (threadIdx.x == 0) { partial_den[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]); partial_ninf[blockIdx.x] = fmax(fmax(sh[4], sh[5]), fmax(sh[6], sh[7])); }
}
// ---- Schur complement (schur.h:180-311), grid-wide
template <int FAM> __global__ void __launch_bounds__(128) kb_schur_inv(const Batch B, const DevParams prm, int p, double lambda) {
	typedef Worker<FAM> W; constexpr int P = W::P, L = W::L; const ProbDesc &d = B.desc[p];
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
template <int FAM> __global__ void __launch_bounds__(128) kb_schur_reduce(const Batch B, const DevParams prm, int p) {
	typedef Worker<FAM> W; constexpr int P = W::P, L = W::L; const ProbDesc &d = B.desc[p];
	if constexpr (!W::T::REL) {
		for (int b = BIG_GID(); b < d.n_hap; b += BIG_STRIDE()) {
			double *H = B.HAp + (d.o_hap + b) * P * P; const int tb = B.sch_term_off[d.o_hapoff + b], te = B.sch_term_off[d.o_hapoff + b + 1];
			if (tb == te) continue;
			double Hl[P * P]; for (int k = 0; k < P * P; k++) Hl[k] = H[k];
			for (int t = tb; t < te; t++) {
				const int l = B.sch_lm[d.o_sch + t]; if (!B.hf_ok[d.o_ulm + l]) continue;
				const double *W1 = B.HApf + (d.o_hapf + B.sch_b1[d.o_sch + t]) * P * L, *W2 = B.HApf + (d.o_hapf + B.sch_b2[d.o_sch + t]) * P * L, *Hi = B.Hfinv + (d.o_ulm + l) * L * L;
				double Y[P * L];
				for (int i = 0; i < P; i++) for (int j = 0; j < L; j++) { double s = 0; for (int k = 0; k < L; k++) s += W1[i * L + k] * Hi[k * L + j]; Y[i * L + j] = s; }
				for (int i = 0; i < P; i++) for (int j = 0; j < P; j++) { double s = 0; for (int k = 0; k < L; k++) s += Y[i * L + k] * W2[j * L + k]; Hl[i * P + j] -= s; }
				const int yw = B.sch_yw[d.o_sch + t]; if (yw >= 0) for (int k = 0; k < P * L; k++) B.YW[(d.o_yw + yw) * P * L + k] = Y[k];
			}
			for (int k = 0; k < P * P; k++) H[k] = Hl[k];
		}
	}
}
template <int FAM> __global__ void __launch_bounds__(128) kb_schur_grad(const Batch B, const DevParams prm, int p) {
	typedef Worker<FAM> W; constexpr int P = W::P, L = W::L; const ProbDesc &d = B.desc[p];
	if constexpr (!W::T::REL) {
		double *g = B.grad + d.o_scal; const double *gf = g + d.nK * P;
		for (int i = BIG_GID(); i < d.nK; i += BIG_STRIDE()) {
			const int b = B.hap_diag[d.o_unk + i]; double acc[P]; for (int r = 0; r < P; r++) acc[r] = g[i * P + r];
			for (int t = B.sch_term_off[d.o_hapoff + b]; t < B.sch_term_off[d.o_hapoff + b + 1]; t++) {
				const int l = B.sch_lm[d.o_sch + t]; if (!B.hf_ok[d.o_ulm + l]) continue;
				const double *Y = B.YW + (d.o_yw + B.sch_yw[d.o_sch + t]) * P * L;
				for (int r = 0; r < P; r++) { double s = 0; for (int k = 0; k < L; k++) s += Y[r * L + k] * gf[l * L + k]; acc[r] -= s; }
			}
			for (int r = 0; r < P; r++) g[i * P + r] = acc[r];
		}
	}
}
template <int FAM> __global__ void __launch_bounds__(128) kb_schur_features(const Batch B, const DevParams prm, int p) {
	typedef Worker<FAM> W; constexpr int P = W::P, L = W::L; const ProbDesc &d = B.desc[p];
	if constexpr (!W::T::REL) {
		double *g = B.grad + d.o_scal, *dl = B.delta + d.o_scal;
		for (int l = BIG_GID(); l < d.nF; l += BIG_STRIDE()) {
			if (!B.hf_ok[d.o_ulm + l]) continue;
			double gl[L]; for (int k = 0; k < L; k++) gl[k] = g[d.nK * P + l * L + k];
			for (int q = B.lm_hapf_off[d.o_lmoff + l]; q < B.lm_hapf_off[d.o_lmoff + l + 1]; q++) {
				const int hb = B.lm_hapf_idx[d.o_hapf + q], i = B.hapf_i[d.o_hapf + hb]; const double *Wm = B.HApf + (d.o_hapf + hb) * P * L;
				for (int k = 0; k < L; k++) { double s = 0; for (int r = 0; r < P; r++) s += Wm[r * L + k] * dl[i * P + r]; gl[k] -= s; }
			}
			const double *Hi = B.Hfinv + (d.o_ulm + l) * L * L;
			for (int k = 0; k < L; k++) g[d.nK * P + l * L + k] = gl[k];
			for (int r = 0; r < L; r++) { double s = 0; for (int k = 0; k < L; k++) s += Hi[r * L + k] * gl[k]; dl[d.nK * P + l * L + r] = s; }
		}
	}
}
// (H + lambda I) into the dense lower triangle + right-hand side; identity padding up to ld
__global__ void kb_dense_clear(const BigSys S) { for (size_t k = BIG_GID(); k < (size_t)S.ld * S.ld; k += BIG_STRIDE()) { const int r = (int)(k / S.ld), c = (int)(k % S.ld); S.A[k] = (r == c && r >= S.n) ? 1.0 : 0.0; } if (BIG_GID() == 0) *S.flag = 0; }
template <int FAM> __global__ void __launch_bounds__(128) kb_dense_assemble(const Batch B, const DevParams prm, int p, const BigSys S, double lambda, int full_system) {
	typedef Worker<FAM> W; constexpr int P = W::P, L = W::L; const ProbDesc &d = B.desc[p];
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
			for (int r = 0; r < L; r++) for (int q = 0; q < L; q++) { if (i == j && q < r) continue; S.A[(size_t)(P * d.nK + L * j + q) * S.ld + P * d.nK + L * i + r] = H[r * L + q] + ((i == j && r == q) ? lambda : 0.0); }
		}
	}
	const double *g = B.grad + d.o_scal;
	for (int k = BIG_GID(); k < S.ld; k += BIG_STRIDE()) S.rhs[k] = k < S.n ? g[k] : 0.0;
}
__global__ void kb_take_delta(const Batch B, int p, const BigSys S) { const ProbDesc &d = B.desc[p]; double *dl = B.delta + d.o_scal; for (int k = BIG_GID(); k < d.n_scal; k += BIG_STRIDE()) if (k < S.n) dl[k] = S.y[k]; else if (S.n == d.n_scal) dl[k] = 0; }
// K12 backup + K11 apply / restore
template <int FAM> __global__ void __launch_bounds__(128) kb_apply(const Batch B, const DevParams prm, int p) {
	typedef Worker<FAM> W; typedef typename W::PO PO; constexpr int P = W::P, L = W::L, PD = W::PD; const ProbDesc &d = B.desc[p]; const double *dl = B.delta + d.o_scal;
	for (int i = BIG_GID(); i < d.nK + d.nF * L + d.n_req; i += BIG_STRIDE()) {
		if (i < d.nK) { double *e = B.edge + (d.o_edge + i) * PD, *o = B.old_edge + (d.o_unk + i) * PD; for (int k = 0; k < PD; k++) o[k] = e[k]; PO::st(e, comp(PO::expm(dl + i * P), PO::ld(e))); }
		else if (i < d.nK + d.nF * L) { const int k = i - d.nK; B.old_ulm[d.o_ulm * L + k] = B.ulm[d.o_ulm * L + k]; B.ulm[d.o_ulm * L + k] += dl[d.nK * P + k]; }
		else { const int r = i - d.nK - d.nF * L; const double *s = B.pose + (d.o_pair * 2 + B.req_idx[d.o_req + r]) * PD; double *o = B.old_pose + (d.o_req + r) * PD; for (int k = 0; k < PD; k++) o[k] = s[k]; }
	}
}
template <int FAM> __global__ void __launch_bounds__(128) kb_restore(const Batch B, const DevParams prm, int p) {
	typedef Worker<FAM> W; constexpr int L = W::L, PD = W::PD; const ProbDesc &d = B.desc[p];
	for (int i = BIG_GID(); i < d.nK + d.nF * L + d.n_req; i += BIG_STRIDE()) {
		if (i < d.nK) { for (int k = 0; k < PD; k++) B.edge[(d.o_edge + i) * PD + k] = B.old_edge[(d.o_unk + i) * PD + k]; }
		else if (i < d.nK + d.nF * L) { const int k = i - d.nK; B.ulm[d.o_ulm * L + k] = B.old_ulm[d.o_ulm * L + k]; }
		else { const int r = i - d.nK - d.nF * L; double *s = B.pose + (d.o_pair * 2 + B.req_idx[d.o_req + r]) * PD; const double *o = B.old_pose + (d.o_req + r) * PD; for (int k = 0; k < PD; k++) s[k] = o[k]; }
	}
}
template <int FAM> __global__ void __launch_bounds__(128) kb_cov_recovery(const Batch B, const DevParams prm, int p, int schur_active) {
	typedef Worker<FAM> W; constexpr int L = W::L; const ProbDesc &d = B.desc[p];
	if constexpr (!W::T::REL) for (int l = BIG_GID(); l < d.nF; l += BIG_STRIDE()) {
		const bool ok = prm.cov_recovery == 1 && (schur_active ? (B.hf_ok[d.o_ulm + l] != 0) : true); B.ulm_inf_valid[d.o_ulm + l] = ok ? 1 : 0;
		if (ok) for (int k = 0; k < L * L; k++) B.ulm_inf[(d.o_ulm + l) * L * L + k] = B.Hf[(d.o_hf + B.hf_diag[d.o_ulm + l]) * L * L + k];
	}
}

} // namespace srbadev
