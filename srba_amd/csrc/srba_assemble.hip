/*
 * srba_assemble.hip -- the fused normal-equations kernel (description and tables: srba_assemble.hpp); its own translation unit, so that it builds in seconds.
 */
#include <hip/hip_runtime.h>
#include "../../include/srba_hip.h"
#include "srba_device.hpp"
extern __shared__ double srba_lds[]; // five numbers per Jacobian block, then the gradient of the capsule
#include "srba_assemble.hpp"

namespace srbadev {

__device__ __forceinline__ void asm_sync() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }

// O[lane] <- sum of O over the lanes (start of the run of equal keys that `lane` belongs to) .. lane; runs are contiguous. The shuffles of a step are all issued
// before the (exec-masked) additions.
template <int N> __device__ __forceinline__ void asm_scan_up(double (&O)[N], int key, int lane) {
#pragma unroll
	for (int off = 1; off < 64; off <<= 1) {
		const int ko = __shfl_up(key, off); double o[N];
#pragma unroll
		for (int k = 0; k < N; k++) o[k] = __shfl_up(O[k], off);
		if (lane >= off && ko == key) {
#pragma unroll
			for (int k = 0; k < N; k++) O[k] += o[k];
		}
	}
}

// A block of this family is J = sg * K, K = [ c s k2 ; -s c k3 ; 0 0 1 ] with k2 = x s - y c, k3 = x c + y s: FOUR numbers {c, s, k2, k3} and a sign.
// M = Lambda * K(b)   (identity Lambda: M = K)
// LAMBDA: 0 = identity (scaled afterwards), 1 = diagonal matrix (the usual information matrix of a relative-pose observation: a third of the products), 2 = full matrix
template <int LAMBDA> __device__ __forceinline__ void asm_lambda_k(double (&M)[9], const double (&b)[4], const double *l) {
	if constexpr (LAMBDA == 1) { M[0] = l[0] * b[0]; M[1] = l[0] * b[1]; M[2] = l[0] * b[2]; M[3] = -(l[4] * b[1]); M[4] = l[4] * b[0]; M[5] = l[4] * b[3]; M[6] = 0; M[7] = 0; M[8] = l[8]; }
	else if constexpr (LAMBDA == 2) {
#pragma unroll
		for (int k = 0; k < 3; k++) { M[3 * k] = l[3 * k] * b[0] - l[3 * k + 1] * b[1]; M[3 * k + 1] = l[3 * k] * b[1] + l[3 * k + 1] * b[0];
			M[3 * k + 2] = l[3 * k] * b[2] + l[3 * k + 1] * b[3] + l[3 * k + 2]; }
	} else { M[0] = b[0]; M[1] = b[1]; M[2] = b[2]; M[3] = -b[1]; M[4] = b[0]; M[5] = b[3]; M[6] = 0; M[7] = 0; M[8] = 1; }
}
// H = K(a)^t * M (row-major 3 x 3)
__device__ __forceinline__ void asm_kt_m(double *H, const double (&a)[4], const double (&M)[9]) {
#pragma unroll
	for (int j = 0; j < 3; j++) { H[j] = a[0] * M[j] - a[1] * M[3 + j]; H[3 + j] = a[1] * M[j] + a[0] * M[3 + j]; H[6 + j] = a[2] * M[j] + a[3] * M[3 + j] + M[6 + j]; }
}

// Sums over runs of consecutive items when every lane owns C consecutive items: a run inside one lane is summed serially and emitted on the spot; the piece of a run
// that a lane leaves open to its right goes through ONE prefix scan over the lanes, and the lane that holds the last item of such a run adds what its left neighbours
// summed. N values per item.
template <int N> struct AsmRuns {
	double acc[N], lead[N]; unsigned lead_lo, lead_hi, cur_lo, cur_hi; bool open, started, have_lead;
	__device__ __forceinline__ void init() { open = false; started = false; have_lead = false; lead_lo = lead_hi = cur_lo = cur_hi = 0;
#pragma unroll
		for (int k = 0; k < N; k++) { acc[k] = 0; lead[k] = 0; } }
	// item `idx_in_lane` of this lane: value v, record words lo / hi (bit 30 of `flags`: first of its run, bit 31: last); emit(lo, hi, total) writes a finished run
	template <class Emit> __device__ __forceinline__ void item(int idx_in_lane, const double (&v)[N], unsigned lo, unsigned hi, unsigned flags, Emit emit) {
		const bool first = (flags & 0x40000000u) != 0, fresh = first || idx_in_lane == 0; // (idx 0 without the flag: the run began in an earlier lane)
		if (fresh) started = first;
#pragma unroll
		for (int k = 0; k < N; k++) acc[k] = fresh ? v[k] : acc[k] + v[k];
		cur_lo = lo; cur_hi = hi;
		if (flags & 0x80000000u) {
			open = false;
			if (started) emit(lo, hi, acc);
			else { have_lead = true; lead_lo = lo; lead_hi = hi;
#pragma unroll
				for (int k = 0; k < N; k++) lead[k] = acc[k];
			}
		} else open = true;
	}
	// after the last item: key(lo, hi) identifies the run
	template <class Key, class Emit> __device__ __forceinline__ void finish(int lane, Key key, Emit emit) {
		double O[N];
#pragma unroll
		for (int k = 0; k < N; k++) O[k] = open ? acc[k] : 0.0;
		const int okey = open ? key(cur_lo, cur_hi) : -1 - lane;
		asm_scan_up<N>(O, okey, lane);
		const int kprev = __shfl_up(okey, 1); double P[N];
#pragma unroll
		for (int k = 0; k < N; k++) P[k] = __shfl_up(O[k], 1);
		if (have_lead) {
			if (lane > 0 && kprev == key(lead_lo, lead_hi)) {
#pragma unroll
				for (int k = 0; k < N; k++) lead[k] += P[k];
			}
			emit(lead_lo, lead_hi, lead);
		}
	}
};

#ifndef SRBA_ASM_WAVES
#define SRBA_ASM_WAVES 2 /* wavefronts per SIMD the register budget is cut for */
#endif
#ifndef SRBA_ASM_KO
#define SRBA_ASM_KO 0 /* knock-out experiments (wrong results): 1 no Hessian stores, 2 no residual gather, 4 no pose gather, 16 no gradient stores */
#endif
#ifndef SRBA_ASM_U
#define SRBA_ASM_U 4   /* blocks / terms in flight per lane */
#endif
template <int LAMBDA>
__global__ void __launch_bounds__(64 * ASM_WAVES_PER_WG) __attribute__((amdgpu_waves_per_eu(SRBA_ASM_WAVES))) k_assemble_se2rel(const Batch B, const DevParams prm, const AsmTables T) {
	constexpr int PD = 5, U = SRBA_ASM_U;
	// a workgroup is a bin of up to four capsules whose LDS images share its allocation (packed at upload); its wavefronts work independently, one capsule each
	const AsmDesc &d = T.desc[blockIdx.x * ASM_WAVES_PER_WG + (threadIdx.x >> 6)]; // descriptors in bin order: one dependent load less than bin -> capsule -> descriptor
	if (d.pidx < 0) return;
	long long *tick = B.phase_cycles ? B.phase_cycles + 16 * (long long)d.pidx : nullptr; // SRBA_HIP_PHASE_TIMING=1: slots 0..3 = start, end of A, end of B, end (100 MHz ticks)
	if (tick && (threadIdx.x & 63) == 0) tick[0] = wall_clock64();
	const bool STAGE = d.stage != 0; // the Hessian blocks go through LDS and leave as one contiguous span; 0 (large windows): every block is stored by the lane that summed it, half the image
	const int tid = threadIdx.x & 63, n_bp = d.n_bp, n_terms = d.n_terms, cb = d.cb, ct = d.ct, n_hap = d.n_hap, nK = d.nK;
	// LDS image of the capsule: four numbers per block slot | the Hessian blocks | the gradient | the unknown edges' own poses
	const int nslot = 64 * cb;
	double *K4 = srba_lds + (d.lds_off >> 3), *Hb = K4 + 4 * nslot, *gb = Hb + (STAGE ? 9 * n_hap : 0), *eb = gb + 3 * nK;
	double *Hglob = B.HAp + d.o_hap * 9;
	const double *lam = prm.lambda; // wave-uniform: stays in scalar registers
	const double sc = (prm.noise == SRBA_NOISE_IDENTITY) ? prm.inv_sigma : 1.0; const bool latch = prm.solver != SRBA_SOLVER_NO_SCHUR_SPARSE_CHOL,
		keep = (prm.ext & SRBA_EXT_SCHUR_KEEPS_GRADIENT) != 0;
	double dmax = 0;
	const double *pose0 = B.pose + d.o_pose, *edge0 = B.edge + d.o_edge, *res0 = B.resid + d.o_res;
	const unsigned long long *br = T.blk + d.o_bp, *tr = T.term + d.o_hapt;
	// requests that do not depend on one another leave together: the first block records, the first term records, the poses of the unknown edges (one contiguous span)
	unsigned long long mn[U], rec[U];
#pragma unroll
	for (int u = 0; u < U; u++) if (u < cb) mn[u] = br[min(tid * cb + u, n_bp - 1)]; // (wave-uniform conditions) clamped, unconditional loads
#pragma unroll
	for (int u = 0; u < U; u++) rec[u] = (n_terms > 0 && u < ct) ? tr[min(tid * ct + u, n_terms - 1)] : 0ull;
	for (int k = tid; k < PD * nK; k += 64) eb[k] = edge0[k];
	asm_sync();
	// ---- A: blocks (cb consecutive blocks per lane; the blocks are sorted by unknown): four numbers per block to LDS, gradient and diagonal Hessian block per unknown
	auto emitA = [&](unsigned lo, unsigned hi, const double (&tot)[9]) { // tot: gradient (3), upper triangle of the diagonal block (00 01 02 11 12 22; Lambda is symmetric)
		const int col = (lo >> 16) & 0x1fff, diag = hi >> 16;
		double *go = gb + 3 * col; go[0] = tot[0] * sc; go[1] = tot[1] * sc; go[2] = tot[2] * sc;
		const double H[9] = {tot[3] * sc, tot[4] * sc, tot[5] * sc, tot[4] * sc, tot[6] * sc, tot[7] * sc, tot[5] * sc, tot[7] * sc, tot[8] * sc};
		if (STAGE) { double *ho = Hb + 9 * diag;
#pragma unroll
			for (int k = 0; k < 9; k++) ho[k] = H[k];
		} else { stn<9>(Hglob + 9 * (long long)diag, H); if (latch) stn<9>(B.HAp0 + (d.o_hap + diag) * 9, H); }
		dmax = fmax(dmax, fmax(H[0], fmax(H[4], H[8])));
	};
#ifdef SRBA_ASM_TICKS
#define ASM_TICK(slot, cond) do { if (tick && (cond)) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); if (tid == 0) tick[slot] = wall_clock64(); } } while (0)
#else
#define ASM_TICK(slot, cond) do { } while (0)
#endif
	AsmRuns<9> RA; RA.init();
	ASM_TICK(4, true);
	for (int s0 = 0; s0 < cb; s0 += U) {
		unsigned long long m[U];
#pragma unroll
		for (int u = 0; u < U; u++) m[u] = mn[u];
#pragma unroll
		for (int u = 0; u < U; u++) if (s0 + U + u < cb) mn[u] = br[min(tid * cb + s0 + U + u, n_bp - 1)]; // the records of the next four blocks travel with this group's gathers
		// a pose is [x y phi cos sin]: the blocks need x, y, cos, sin -- two 16-byte requests per lane instead of three
		double D[U][4], r[U][3];
#pragma unroll
		for (int u = 0; u < U; u++) if (s0 + u < cb) {
			const unsigned lo = (unsigned)m[u], hi = (unsigned)(m[u] >> 32); const int iD = (int)(lo & 0xffff) - 1;
			const double *pd = pose0 + (unsigned)max(iD, 0) * PD; if (!(SRBA_ASM_KO & 4)) { ldn<2>(D[u], pd); ldn<2>(D[u] + 2, pd + 3); } else { D[u][0] = (double)lo; D[u][1] = 1; D[u][2] = 0.6;
				D[u][3] = 0.8; }
			if (!(SRBA_ASM_KO & 2)) ldn<3>(r[u], res0 + (hi & 0xffff) * 3); else { r[u][0] = (double)hi; r[u][1] = 1; r[u][2] = 2; }
		}
		ASM_TICK(5, s0 == 0);
#pragma unroll
		for (int u = 0; u < U; u++) if (s0 + u < cb) {
			const unsigned lo = (unsigned)m[u], hi = (unsigned)(m[u] >> 32); const int iD = (int)(lo & 0xffff) - 1; const bool inverse = (lo & 0x20000000u) != 0;
			if (tid * cb + s0 + u < n_bp) {
				double x = D[u][0], y = D[u][1], c = D[u][2], s = D[u][3];
				if (iD < 0) { x = 0; y = 0; c = 1; s = 0; }
				if (inverse) { // D' = p (+) D (jacobians.h:684-711), p = the edge's own pose (staged in LDS)
					const double *pp = eb + ((lo >> 16) & 0x1fff) * PD; const double px = pp[0], py = pp[1], pc = pp[3], ps = pp[4];
					const double nx = px + x * pc - y * ps, ny = py + x * ps + y * pc, nc = pc * c - ps * s, ns = ps * c + pc * s; x = nx; y = ny; c = nc; s = ns;
				}
				const double kk[4] = {c, s, x * s - y * c, x * c + y * s};
				double *dst = K4 + (s0 + u) * 64 + tid; // four planes of 64 cb slots: consecutive lanes write consecutive doubles (no bank conflict)
#pragma unroll
				for (int k = 0; k < 4; k++) dst[k * nslot] = kk[k];
				double v[9], t[3], M[9], Hd[9];
				if constexpr (LAMBDA == 2) { for (int k = 0; k < 3; k++) t[k] = lam[k * 3] * r[u][0] + lam[k * 3 + 1] * r[u][1] + lam[k * 3 + 2] * r[u][2]; }
				else if constexpr (LAMBDA == 1) { for (int k = 0; k < 3; k++) t[k] = lam[k * 4] * r[u][k]; }
				else { for (int k = 0; k < 3; k++) t[k] = r[u][k]; }
				const double sg = inverse ? -1.0 : 1.0;
				v[0] = sg * (c * t[0] - s * t[1]); v[1] = sg * (s * t[0] + c * t[1]); v[2] = sg * (kk[2] * t[0] + kk[3] * t[1] + t[2]); // J^t Lambda r, J = sg K
				asm_lambda_k<LAMBDA>(M, kk, lam); asm_kt_m(Hd, kk, M);                                                               // J^t Lambda J = K^t Lambda K (symmetric)
				v[3] = Hd[0]; v[4] = Hd[1]; v[5] = Hd[2]; v[6] = Hd[4]; v[7] = Hd[5]; v[8] = Hd[8];
				RA.item(s0 + u, v, lo, hi, lo, emitA);
			}
		}
		ASM_TICK(6, s0 == 0);
	}
	ASM_TICK(7, true);
	RA.finish(tid, [](unsigned lo, unsigned) { return (int)((lo >> 16) & 0x1fff); }, emitA);
	asm_sync();
	if (tick && tid == 0) tick[1] = wall_clock64();
	// ---- B: off-diagonal Hessian blocks: ct consecutive terms per lane (the term list is sorted by Hessian block)
	if (n_terms > 0) {
		auto emitB = [&](unsigned, unsigned hi, const double (&tot)[9]) {
			const long long blk = hi & 0x3fffffff;
			if (STAGE) { double *ho = Hb + 9 * blk;
#pragma unroll
				for (int k = 0; k < 9; k++) ho[k] = tot[k] * sc;
			} else { double H[9];
#pragma unroll
				for (int k = 0; k < 9; k++) H[k] = tot[k] * sc;
				stn<9>(Hglob + 9 * blk, H); if (latch) stn<9>(B.HAp0 + (d.o_hap + blk) * 9, H); }
		};
		AsmRuns<9> RB; RB.init();
		for (int s0 = 0; s0 < ct; s0 += U) {
			unsigned long long cur[U];
#pragma unroll
			for (int u = 0; u < U; u++) cur[u] = rec[u];
#pragma unroll
			for (int u = 0; u < U; u++) if (s0 + U + u < ct) rec[u] = tr[min(tid * ct + s0 + U + u, n_terms - 1)]; // the next group
#pragma unroll
			for (int u = 0; u < U; u++) if (s0 + u < ct) {
				if (tid * ct + s0 + u < n_terms) {
					const unsigned lo = (unsigned)cur[u], hi = (unsigned)(cur[u] >> 32);
					double A[4], Bm[4], M[9], v[9];
					const double *pa = K4 + (lo & 0x7fff), *pb = K4 + (lo >> 16);
#pragma unroll
					for (int k = 0; k < 4; k++) { A[k] = pa[k * nslot]; Bm[k] = pb[k * nslot]; }
					asm_lambda_k<LAMBDA>(M, Bm, lam); asm_kt_m(v, A, M);
					if (lo & 0x8000u) { // the two blocks have opposite directions: J1^t Lambda J2 = - K1^t Lambda K2
#pragma unroll
						for (int k = 0; k < 9; k++) v[k] = -v[k];
					}
					RB.item(s0 + u, v, lo, hi, hi, emitB);
				}
			}
		}
		RB.finish(tid, [](unsigned, unsigned hi) { return (int)(hi & 0x3fffffff); }, emitB);
	}
	asm_sync();
	if (tick && tid == 0) tick[2] = wall_clock64();
	// ---- C: the Hessian blocks and the gradient leave as contiguous spans (16 bytes per lane and request), lambda guess
	if (STAGE && !(SRBA_ASM_KO & 1)) {
		double *Hg = B.HAp + d.o_hap * 9, *H0 = B.HAp0 + d.o_hap * 9; const int nh = 9 * n_hap;
		for (int k = 2 * tid; k < nh; k += 128) {
			if (k + 1 < nh) { f64x2u v; v.x = Hb[k]; v.y = Hb[k + 1]; *(f64x2u *)(Hg + k) = v; if (latch) *(f64x2u *)(H0 + k) = v; }
			else { Hg[k] = Hb[k]; if (latch) H0[k] = Hb[k]; }
		}
	}
	if (!(SRBA_ASM_KO & 16)) { double *go = B.grad + d.o_scal; for (int k = tid; k < 3 * nK; k += 64) { const double v = gb[k]; go[k] = v; if (keep) B.grad0[d.o_scal + k] = v; } }
	const double l0 = 1e-3 * wave_max(dmax);
	if (tid == 0) { B.lambda_io[d.pidx] = l0; B.results[d.pidx].num_invalid_jacobs = 0; if (tick) tick[3] = wall_clock64(); }
}

int asm_launch(int lambda_mode, int n_bins, size_t lds_bytes, hipStream_t stream, const Batch &B, const DevParams &prm, const AsmTables &T) {
	static bool attr_done[3] = {false, false, false}; // the bins are larger than the 64 KB a launch may ask for by default
	auto go = [&](auto kernel, int which) -> int {
		if (lds_bytes > 64 * 1024 && !attr_done[which]) { hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, ASM_BIN_BYTES);
			if (e != hipSuccess) return (int)e; attr_done[which] = true; }
		hipLaunchKernelGGL(kernel, dim3(n_bins), dim3(64 * ASM_WAVES_PER_WG), lds_bytes, stream, B, prm, T);
		return (int)hipGetLastError();
	};
	return lambda_mode == 2 ? go(k_assemble_se2rel<2>, 2) : lambda_mode == 1 ? go(k_assemble_se2rel<1>, 1) : go(k_assemble_se2rel<0>, 0);
}
} // namespace srbadev
