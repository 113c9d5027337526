/*
 * srba_assemble.hip -- the fused normal-equations kernel (description and tables: srba_assemble.hpp); its own translation unit, so that it builds in seconds.
 */
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../include/srba_hip.h"
#include "srba_device.hpp"
extern __shared__ double srba_lds[]; // per capsule of the bin: Hessian blocks | gradient | poses of the unknown edges
#include "srba_assemble.hpp"

namespace srbadev {

__device__ __forceinline__ void asm_sync() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
#ifndef SRBA_ASM_KO
#define SRBA_ASM_KO 0 /* knock-out experiments (wrong results): 1 no Hessian stores, 2 no residual gather, 4 no pose gather, 8 no LDS adds, 16 no gradient stores, 32 plain (volatile) LDS stores instead of adds */
#endif
__device__ __forceinline__ void lds_add(double *p, double v) { // ds_add_f64, nothing returned
	if (SRBA_ASM_KO & 32) *(volatile double *)p = v; else __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// A block of this family is J = sg * K, K = [ c s k2 ; -s c k3 ; 0 0 1 ] with k2 = x s - y c, k3 = x c + y s: FOUR numbers {c, s, k2, k3} and a sign.
// LAMBDA: 0 = identity (scaled afterwards), 1 = diagonal matrix (the usual information matrix of a relative-pose observation: a third of the products), 2 = full matrix
// With the sign inside: b = sg * {c, s, k2, k3}, J = [ b0 b1 b2 ; -b1 b0 b3 ; 0 0 sg ].  M = Lambda * J(b);  H = J(a)^t * M
template <int LAMBDA> __device__ __forceinline__ void asm_lambda_sk(double (&M)[9], const double (&b)[4], const double sg, const double *l) {
	if constexpr (LAMBDA == 1) { M[0] = l[0] * b[0]; M[1] = l[0] * b[1]; M[2] = l[0] * b[2]; M[3] = -(l[4] * b[1]); M[4] = l[4] * b[0]; M[5] = l[4] * b[3]; M[6] = 0; M[7] = 0; M[8] = l[8] * sg; }
	else if constexpr (LAMBDA == 2) {
#pragma unroll
		for (int k = 0; k < 3; k++) { M[3 * k] = l[3 * k] * b[0] - l[3 * k + 1] * b[1]; M[3 * k + 1] = l[3 * k] * b[1] + l[3 * k + 1] * b[0];
			M[3 * k + 2] = l[3 * k] * b[2] + l[3 * k + 1] * b[3] + l[3 * k + 2] * sg; }
	} else { M[0] = b[0]; M[1] = b[1]; M[2] = b[2]; M[3] = -b[1]; M[4] = b[0]; M[5] = b[3]; M[6] = 0; M[7] = 0; M[8] = sg; }
}
__device__ __forceinline__ void asm_skt_m(double *H, const double (&a)[4], const double sg, const double (&M)[9]) {
#pragma unroll
	for (int j = 0; j < 3; j++) { H[j] = a[0] * M[j] - a[1] * M[3 + j]; H[3 + j] = a[1] * M[j] + a[0] * M[3 + j]; H[6 + j] = a[2] * M[j] + a[3] * M[3 + j] + sg * M[6 + j]; }
}

#ifndef SRBA_ASM_WAVES
#define SRBA_ASM_WAVES 3 /* wavefronts per SIMD the register budget is cut for */
#endif

#ifndef SRBA_ASM_CH
#define SRBA_ASM_CH 2 /* passes whose gathers are in flight together */
#endif
struct AsmGather { double D[3][4]; double r[3]; };   // what a row gathers: {x, y, cos, sin} of the pose D of each block, its residual
// fields of a row record (srba_assemble.hpp)
__device__ __forceinline__ int asm_m(const uint4 &q) { return (int)((q.x >> 28) & 3u); }
__device__ __forceinline__ unsigned asm_iD(const uint4 &q, int a) { return a == 0 ? q.x & 0x3fffu : a == 1 ? (q.x >> 14) & 0x3fffu : q.y & 0x3fffu; }
__device__ __forceinline__ unsigned asm_col(const uint4 &q, int a) { return a == 0 ? q.y >> 25 : a == 1 ? q.z & 0x7fu : (q.z >> 7) & 0x7fu; }
__device__ __forceinline__ unsigned asm_xb(const uint4 &q, int s) { return s == 0 ? (q.z >> 14) & 0x7ffu : s == 1 ? q.w & 0x7ffu : (q.w >> 11) & 0x7ffu; }
__device__ __forceinline__ unsigned asm_flags(const uint4 &q) { return q.z >> 25; }

__device__ __forceinline__ void asm_gather(AsmGather &G, const uint4 &q, const double *pose0, const double *res0) {
	const int m = asm_m(q);
	// a pose is [x y phi cos sin]: the blocks need x, y, cos, sin -- two 16-byte requests per lane instead of three
#pragma unroll
	for (int a = 0; a < 3; a++) if (a < m) { const unsigned iD = asm_iD(q, a);
		const double *pd = pose0 + (iD ? iD - 1 : 0u) * 5u;
		if (!(SRBA_ASM_KO & 4)) { ldn<2>(G.D[a], pd); ldn<2>(G.D[a] + 2, pd + 3); } else { G.D[a][0] = (double)iD; G.D[a][1] = 1; G.D[a][2] = 0.6; G.D[a][3] = 0.8; }
	}
	if (m > 0) { const unsigned row = (q.y >> 14) & 0x7ffu; if (!(SRBA_ASM_KO & 2)) ldn<3>(G.r, res0 + row * 3u); else { G.r[0] = (double)row; G.r[1] = 1; G.r[2] = 2; } }
}

// the sums of one row: gradient entries and diagonal Hessian block of each of its unknowns, the cross terms of its pairs of blocks
template <int LAMBDA> __device__ __forceinline__ void asm_row_sums(const uint4 &q, const AsmGather &G, const double *lam, double *Hb, double *gb, const double *eb, const int *dgt) {
	const int m = asm_m(q); if (m == 0) return;
	const unsigned fl = asm_flags(q);
	double t[3];
	if constexpr (LAMBDA == 2) { for (int k = 0; k < 3; k++) t[k] = lam[k * 3] * G.r[0] + lam[k * 3 + 1] * G.r[1] + lam[k * 3 + 2] * G.r[2]; }
	else if constexpr (LAMBDA == 1) { for (int k = 0; k < 3; k++) t[k] = lam[k * 4] * G.r[k]; }
	else { for (int k = 0; k < 3; k++) t[k] = G.r[k]; }
	double K[3][4], M[3][9];
#pragma unroll
	for (int a = 0; a < 3; a++) if (a < m) {
		double x = G.D[a][0], y = G.D[a][1], c = G.D[a][2], s = G.D[a][3]; const unsigned col = asm_col(q, a);
		if (asm_iD(q, a) == 0) { x = 0; y = 0; c = 1; s = 0; }
		const bool inverse = ((fl >> a) & 1u) != 0;
		if (inverse) { // D' = p (+) D (jacobians.h:684-711), p = the edge's own pose (staged in LDS)
			const double *pp = eb + col * 5u; const double px = pp[0], py = pp[1], pc = pp[3], ps = pp[4];
			const double nx = px + x * pc - y * ps, ny = py + x * ps + y * pc, nc = pc * c - ps * s, ns = ps * c + pc * s; x = nx; y = ny; c = nc; s = ns;
		}
		// J = sg K: the sign goes into the four numbers (an exact operation), J^t Lambda r and every product of two blocks then carry theirs by themselves
		const double sg = inverse ? -1.0 : 1.0, k2 = x * s - y * c, k3 = x * c + y * s;
		K[a][0] = sg * c; K[a][1] = sg * s; K[a][2] = sg * k2; K[a][3] = sg * k3;
		asm_lambda_sk<LAMBDA>(M[a], K[a], sg, lam);
		if (!(SRBA_ASM_KO & 8)) {
			double *go = gb + 3 * col; // J^t Lambda r
			lds_add(go, K[a][0] * t[0] - K[a][1] * t[1]); lds_add(go + 1, K[a][1] * t[0] + K[a][0] * t[1]); lds_add(go + 2, K[a][2] * t[0] + K[a][3] * t[1] + sg * t[2]);
			double Hd[9]; asm_skt_m(Hd, K[a], sg, M[a]); // J^t Lambda J: symmetric, the upper triangle is summed (mirrored afterwards)
			double *ho = Hb + 9 * dgt[col]; lds_add(ho, Hd[0]); lds_add(ho + 1, Hd[1]); lds_add(ho + 2, Hd[2]); lds_add(ho + 4, Hd[4]); lds_add(ho + 5, Hd[5]); lds_add(ho + 8, Hd[8]);
		}
	}
	// cross terms (a, b), a < b: J_a^t Lambda J_b = sg_a sg_b K_a^t Lambda K_b into the block (unknown of a, unknown of b)
#pragma unroll
	for (int sidx = 0; sidx < 3; sidx++) { const int a = sidx == 2 ? 1 : 0, b = sidx == 0 ? 1 : 2; const unsigned xb = asm_xb(q, sidx);
		if (b < m && xb != 0x7ffu && !(SRBA_ASM_KO & 8)) {
			double v[9]; asm_skt_m(v, K[a], ((fl >> a) & 1u) ? -1.0 : 1.0, M[b]); double *ho = Hb + 9 * xb;
#pragma unroll
			for (int k = 0; k < 9; k++) lds_add(ho + k, v[k]);
		}
	}
}

template <int LAMBDA, int WPW>
__global__ void __launch_bounds__(64 * WPW) __attribute__((amdgpu_waves_per_eu(SRBA_ASM_WAVES))) k_assemble_se2rel(const Batch B, const DevParams prm, const AsmTables T) {
	constexpr int CH = SRBA_ASM_CH;
	// a workgroup is a bin of capsules whose LDS images share its allocation (packed at upload); its wavefronts work independently, one capsule each
	const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	const AsmDesc &d = T.desc[blockIdx.x * WPW + wave]; // descriptors in bin order; wave-uniform: scalar loads
	if (d.pidx < 0) return;
	long long *tick = B.phase_cycles ? B.phase_cycles + 16 * (long long)d.pidx : nullptr; // SRBA_HIP_PHASE_TIMING=1: slots 0..3 = start, image ready, sums done, end (100 MHz ticks)
	const int tid = threadIdx.x & 63, n_rec = d.n_rec, rounds = (n_rec + 63) >> 6, n_hap = d.n_hap, nK = d.nK;
	if (tick && tid == 0) { tick[0] = wall_clock64(); // slot 4: where it ran (HW_ID: wave [3:0] SIMD [5:4] CU [11:8] SH [12] SE [15:13]; XCC_ID) -- tools/diag_assemble.py builds the per-CU timeline from it
		tick[4] = (long long)(unsigned)__builtin_amdgcn_s_getreg(4 | (31 << 11)) | (long long)(__builtin_amdgcn_s_getreg(20 | (31 << 11)) & 0xf) << 32; }
	double *Hb = srba_lds + (d.lds_off >> 3), *gb = Hb + 9 * n_hap, *eb = gb + 3 * nK + ((n_hap + nK) & 1); // (eb on 16 bytes)
	int *dgt = (int *)(eb + ((5 * nK + 1) & ~1)); // diagonal Hessian block of every unknown
	const double *lam = prm.lambda; // wave-uniform: stays in scalar registers
	const double sc = (prm.noise == SRBA_NOISE_IDENTITY) ? prm.inv_sigma : 1.0; const bool latch = prm.solver != SRBA_SOLVER_NO_SCHUR_SPARSE_CHOL,
		keep = (prm.ext & SRBA_EXT_SCHUR_KEEPS_GRADIENT) != 0;
	const double *pose0 = B.pose + d.o_pose, *edge0 = B.edge + d.o_edge, *res0 = B.resid + d.o_res;
	const uint4 *rp = (const uint4 *)(T.rec + d.o_rec) + tid;
	// requests that depend on the descriptor only leave together: the records of the first CH passes, the poses of the unknown edges, the diagonal block of every unknown
	uint4 R[CH];
#pragma unroll
	for (int u = 0; u < CH; u++) { R[u] = make_uint4(0, 0, 0, 0); if (64 * u + tid < n_rec) R[u] = rp[64 * u]; }
	{ int v[2];
#pragma unroll
		for (int u = 0; u < 2; u++) v[u] = B.hap_diag[d.o_unk + min(64 * u + tid, nK - 1)];
#pragma unroll
		for (int u = 0; u < 2; u++) if (64 * u + tid < nK) dgt[64 * u + tid] = v[u]; }
	{ const int ne = 5 * nK; // the poses of the unknown edges, 16 bytes per request (eb is on 16 bytes and has room for an odd tail); a window has at most 127 unknowns: five requests per lane
		f64x2u v[5];
#pragma unroll
		for (int u = 0; u < 5; u++) { const int k = 128 * u + 2 * tid; v[u] = *(const f64x2u *)(edge0 + min(k, ne - 2 + (ne & 1))); } // (clamped: an odd tail reads one double of what follows)
#pragma unroll
		for (int u = 0; u < 5; u++) { const int k = 128 * u + 2 * tid; if (k < ne) *(f64x2u *)(eb + k) = v[u]; } }
	{ const int nz = 9 * n_hap + 3 * nK; f64x2u z; z.x = 0; z.y = 0; for (int k = 2 * tid; k < nz; k += 128) *(f64x2u *)(Hb + k) = z; } // (the image has room for an odd tail)
	AsmGather G[CH];
#pragma unroll
	for (int u = 0; u < CH; u++) asm_gather(G[u], R[u], pose0, res0);
	asm_sync();
	if (tick && tid == 0) tick[1] = wall_clock64();
	for (int c0 = 0; c0 < rounds; c0 += CH) { // CH passes at a time: their gathers travel together, the records of the next CH under their arithmetic
		uint4 N[CH]; const bool more = c0 + CH < rounds;
#pragma unroll
		for (int u = 0; u < CH; u++) { N[u] = make_uint4(0, 0, 0, 0); if (more && 64 * (c0 + CH + u) + tid < n_rec) N[u] = rp[64 * (c0 + CH + u)]; }
#pragma unroll
		for (int u = 0; u < CH; u++) if (u == 0 || c0 + u < rounds) asm_row_sums<LAMBDA>(R[u], G[u], lam, Hb, gb, eb, dgt);
		if (more) {
#pragma unroll
			for (int u = 0; u < CH; u++) { R[u] = N[u]; asm_gather(G[u], R[u], pose0, res0); }
		}
	}
	asm_sync();
	if (tick && tid == 0) tick[2] = wall_clock64();
	// the lower triangle of the diagonal blocks, lambda guess
	double dmax = 0;
	for (int k = tid; k < nK; k += 64) { double *h = Hb + 9 * dgt[k]; const double h1 = h[1], h2 = h[2], h5 = h[5]; h[3] = h1; h[6] = h2; h[7] = h5; dmax = fmax(dmax, fmax(h[0], fmax(h[4], h[8]))); }
	asm_sync();
	// the Hessian blocks and the gradient leave as contiguous spans (16 bytes per lane and request)
	if (!(SRBA_ASM_KO & 1)) {
		double *Hg = B.HAp + d.o_hap * 9, *H0 = B.HAp0 + d.o_hap * 9; const int nh = 9 * n_hap; typedef double f64x2a __attribute__((ext_vector_type(2), aligned(16)));
		for (int k0 = 0; k0 < nh; k0 += 512) { f64x2a v[4]; // four 16-byte reads of the image in flight, then the stores
#pragma unroll
			for (int u = 0; u < 4; u++) { const int k = k0 + 128 * u + 2 * tid; if (k < nh) v[u] = *(const f64x2a *)(Hb + k); } // (an odd tail reads the pad / first gradient entry)
#pragma unroll
			for (int u = 0; u < 4; u++) { const int k = k0 + 128 * u + 2 * tid;
				if (k + 1 < nh) { f64x2u w; w.x = v[u].x * sc; w.y = v[u].y * sc; *(f64x2u *)(Hg + k) = w; if (latch) *(f64x2u *)(H0 + k) = w; }
				else if (k < nh) { const double w = v[u].x * sc; Hg[k] = w; if (latch) H0[k] = w; } }
		}
	}
	if (!(SRBA_ASM_KO & 16)) { double *go = B.grad + d.o_scal, *g0 = B.grad0 + d.o_scal; const int ng = 3 * nK;
		for (int k = 2 * tid; k < ng; k += 128) {
			if (k + 1 < ng) { f64x2u w; w.x = gb[k] * sc; w.y = gb[k + 1] * sc; *(f64x2u *)(go + k) = w; if (keep) *(f64x2u *)(g0 + k) = w; }
			else { const double w = gb[k] * sc; go[k] = w; if (keep) g0[k] = w; } } }
	const double l0 = 1e-3 * (wave_max(dmax) * sc);
	if (tid == 0) { B.lambda_io[d.pidx] = l0; B.results[d.pidx].num_invalid_jacobs = 0; if (tick) tick[3] = wall_clock64(); }
}

int asm_launch(int lambda_mode, int wpw, int n_bins, size_t lds_bytes, hipStream_t stream, const Batch &B, const DevParams &prm, const AsmTables &T) {
	static bool attr_done[3][3] = {}; // the bins may be larger than the 64 KB a launch may ask for by default
	auto go = [&](auto kernel, int which, int threads) -> int { const int wi = wpw == 4 ? 2 : wpw == 2 ? 1 : 0;
		if (lds_bytes > 64 * 1024 && !attr_done[which][wi]) { hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
			if (e != hipSuccess) return (int)e; attr_done[which][wi] = true; }
		hipLaunchKernelGGL(kernel, dim3(n_bins), dim3(threads), lds_bytes, stream, B, prm, T);
		return (int)hipGetLastError();
	};
#define SRBA_ASM_GO(L) (wpw == 4 ? go(k_assemble_se2rel<L, 4>, L, 256) : wpw == 2 ? go(k_assemble_se2rel<L, 2>, L, 128) : go(k_assemble_se2rel<L, 1>, L, 64))
	return lambda_mode == 2 ? SRBA_ASM_GO(2) : lambda_mode == 1 ? SRBA_ASM_GO(1) : SRBA_ASM_GO(0);
#undef SRBA_ASM_GO
}

unsigned long long asm_layout_signature() { return sizeof(Batch) * 10007ull + sizeof(DevParams); }

void asm_config(int &wpw, int &bin_bytes) {
	wpw = ASM_MAX_WPW; int kb = ASM_DEFAULT_BIN_KB;
	if (const char *e = getenv("SRBA_HIP_ASM_WPW")) { const int v = atoi(e); if (v == 1 || v == 2 || v == 4) wpw = v; }
	if (const char *e = getenv("SRBA_HIP_ASM_BIN_KB")) { const int v = atoi(e); if (v >= 4 && v <= 160) kb = v; }
	bin_bytes = kb * 1024;
}

// Bins: the largest remaining image opens a bin, takes the next largest ones that fit beside it (similar windows end at similar times) and then the smallest ones that fill what is
// left. Bins come out largest first = dispatch order.
int asm_plan(int n, const AsmDesc *dsc, const int *rounds, size_t cap, int wpw, int bin_bytes, AsmDesc *out, int32_t *rest, int &n_rest) {
	std::vector<size_t> need(n); std::vector<int> fit; fit.reserve(n); n_rest = 0; cap = std::min(cap, (size_t)bin_bytes);
	for (int p = 0; p < n; p++) { need[p] = asm_image_bytes(dsc[p].n_hap, dsc[p].nK); if (rounds[p] > 0 && need[p] <= cap) fit.push_back(p); else rest[n_rest++] = p; }
	std::stable_sort(fit.begin(), fit.end(), [&](int x, int y) { return need[x] != need[y] ? need[x] < need[y] : rounds[x] < rounds[y]; });
	int nb = 0; AsmDesc none; std::memset(&none, 0, sizeof(none)); none.pidx = -1;
	for (size_t lo = 0, hi = fit.size(); lo < hi;) { AsmDesc *e = out + (size_t)wpw * nb; size_t used = 0; int w = 0;
		while (w < wpw && lo < hi && used + need[fit[hi - 1]] <= (size_t)bin_bytes) { hi--; e[w] = dsc[fit[hi]]; e[w].lds_off = (int)used; used += need[fit[hi]]; w++; }
		while (w < wpw && lo < hi && used + need[fit[lo]] <= (size_t)bin_bytes) { e[w] = dsc[fit[lo]]; e[w].lds_off = (int)used; used += need[fit[lo]]; w++; lo++; }
		for (; w < wpw; w++) e[w] = none;
		nb++; }
	// dispatch order: the bins of the largest windows hold few capsules for their LDS; spread over the first part of the launch, with bins of small windows between them, they
	// keep more wavefronts per CU in flight than one after the other. SRBA_HIP_ASM_MIX="F,S": the F % largest bins are spread evenly over the first S % of the sequence
	static int mixF = -1, mixS = 0; if (mixF < 0) { mixF = ASM_MIX_F; mixS = ASM_MIX_S; if (const char *e = getenv("SRBA_HIP_ASM_MIX")) { if (sscanf(e, "%d,%d", &mixF, &mixS) != 2) { mixF = 0; mixS = 0; } } }
	if (mixF > 0 && mixS > 0 && nb > 8) { std::vector<AsmDesc> t(out, out + (size_t)wpw * nb); const int nF = std::max(1, (int)((long long)nb * mixF / 100)), span = std::max(nF, (int)((long long)nb * mixS / 100));
		std::vector<int> src(nb, -1); for (int j = 0; j < nF; j++) src[(int)((long long)j * span / nF)] = j;
		for (int q = 0, rest_i = nF; q < nb; q++) if (src[q] < 0) src[q] = rest_i++;
		for (int q = 0; q < nb; q++) std::memcpy(out + (size_t)wpw * q, t.data() + (size_t)wpw * src[q], sizeof(AsmDesc) * wpw); }
	return nb;
}

// ---- host: the records of one capsule (srba_assemble.hpp)
int asm_pack(const srba_problem_capsule &k, AsmRec *dst) {
	const int nK = k.n_unk_edges, n_obs = k.n_obs, n_bp = k.n_bp;
	if (n_bp < 1 || 2 * (long long)k.n_pairs > ASM_MAX_POSE || n_obs > ASM_MAX_ROW + 1 || nK > ASM_MAX_NK || nK < 1 || k.n_hap > ASM_MAX_HAP) return 0;
	// the blocks of every row, in block order (= ascending unknown: the blocks are listed unknown by unknown)
	std::vector<unsigned char> cnt(n_obs, 0); std::vector<int> blk(3 * (size_t)n_obs, -1), loc(n_bp, 0);
	for (int b = 0; b < n_bp; b++) { const int r = k.bp_res[b]; if (r < 0 || r >= n_obs || cnt[r] == 3 || k.bp_D[b] < -1 || k.bp_D[b] >= 2 * k.n_pairs || k.bp_col[b] < 0 || k.bp_col[b] >= nK) return 0;
		if (cnt[r] > 0 && k.bp_col[blk[3 * (size_t)r + cnt[r] - 1]] >= k.bp_col[b]) return 0; loc[b] = cnt[r]; blk[3 * (size_t)r + cnt[r]++] = b; }
	// the diagonal Hessian block of unknown i sums exactly the self products of the blocks of its column
	for (int i = 0; i < nK; i++) { const int bb = k.colp_off[i], be = k.colp_off[i + 1], hd = k.hap_diag[i]; if (be < bb || hd < 0 || hd >= k.n_hap || k.hap_term_off[hd + 1] - k.hap_term_off[hd] != be - bb) return 0;
		for (int b = bb; b < be; b++) { const int t = k.hap_term_off[hd] + (b - bb); if (k.bp_col[b] != i || k.hap_t1[t] != b || k.hap_t2[t] != b) return 0; } }
	if (k.colp_off[nK] != n_bp) return 0;
	// every off-diagonal term pairs two blocks of one row, the lower block first
	std::vector<uint16_t> xb(3 * (size_t)n_obs, 0x7ff); long long n_off = 0;
	for (int h = 0; h < k.n_hap; h++) if (k.hap_i[h] != k.hap_j[h]) for (int t = k.hap_term_off[h]; t < k.hap_term_off[h + 1]; t++) {
		const int t1 = k.hap_t1[t], t2 = k.hap_t2[t]; if (t1 < 0 || t2 < 0 || t1 >= n_bp || t2 >= n_bp) return 0;
		const int r = k.bp_res[t1], a = loc[t1], b = loc[t2]; if (k.bp_res[t2] != r || a >= b || k.bp_col[t1] != k.hap_i[h] || k.bp_col[t2] != k.hap_j[h]) return 0;
		uint16_t &s = xb[3 * (size_t)r + (a + b - 1)]; if (s != 0x7ff) return 0; s = (uint16_t)h; n_off++; }
	if (n_off + n_bp != k.n_hap_terms) return 0;
	// Rows with blocks, by number of blocks (three, two, one), each class sorted by its unknowns and dealt round-robin to 16-lane groups of its own: an LDS instruction costs per group of 16
	// lanes that has a lane in it, times the lanes of the group that hit one address (tools/probes/lds_atomic_rate.hip). Rows that add to the same gradient entries and Hessian blocks are
	// neighbours in the sorted list and land in different groups; the 27 instructions of a third block find their rows (8 % of all) in two groups instead of one or two lanes of every group.
	auto key = [&](int r) { unsigned long long v = 0; for (int a = 0; a < 3; a++) v = v << 16 | (unsigned long long)(a < cnt[r] ? k.bp_col[blk[3 * (size_t)r + a]] + 1 : 0); return v; };
	static const int by_m = getenv("SRBA_HIP_ASM_BY_M") ? atoi(getenv("SRBA_HIP_ASM_BY_M")) : 1; // 0: one class (the deal of the first version of this kernel)
	int g0 = 0;
	for (int mc = 3; mc >= 1; mc--) { std::vector<int> act; for (int r = 0; r < n_obs; r++) if (by_m ? cnt[r] == mc : (mc == 3 && cnt[r] > 0)) act.push_back(r);
		if (act.empty()) continue;
		std::stable_sort(act.begin(), act.end(), [&](int x, int y) { return key(x) < key(y); });
		const int n_act = (int)act.size(), ng = (n_act + 15) / 16;
		for (int i = 0; i < n_act; i++) { const int r = act[i], m = cnt[r]; AsmRec &R = dst[(size_t)(g0 + i % ng) * 16 + i / ng]; uint32_t D[3] = {0, 0, 0}, col[3] = {0, 0, 0}, fl = 0;
			for (int a = 0; a < m; a++) { const int b = blk[3 * (size_t)r + a]; D[a] = (uint32_t)(k.bp_D[b] + 1); col[a] = (uint32_t)k.bp_col[b]; if (!k.bp_normal[b]) fl |= 1u << a; }
			for (int s = 0; s < 3; s++) { const int a = s == 2 ? 1 : 0, b = s == 0 ? 1 : 2; if (b < m && (((fl >> a) ^ (fl >> b)) & 1u)) fl |= 1u << (3 + s); }
			const uint16_t *x = &xb[3 * (size_t)r];
			R.w[0] = D[0] | D[1] << 14 | (uint32_t)m << 28; R.w[1] = D[2] | (uint32_t)r << 14 | col[0] << 25; R.w[2] = col[1] | col[2] << 7 | (uint32_t)x[0] << 14 | fl << 25; R.w[3] = (uint32_t)x[1] | (uint32_t)x[2] << 11; }
		g0 += ng; }
	return 16 * g0;
}
} // namespace srbadev
