/*
 * srba_hip.hip -- kernels + C ABI (include/srba_hip.h) of the MI355X back-end.  Build: hipcc --offload-arch=gfx950.
 *
 * Kernels (all templated on the model family):
 *   k_lm_run        : the whole Levenberg-Marquardt loop of optimize_edges() (impl/optimize_edges.h:256-751), one wavefront per capsule,
 *                     persistent workgroups pulling capsules of one LDS size class from a counter
 *   k_spantree, k_residuals, k_linearize, k_solve, k_apply, k_rollback : the same phases as separate launches (stepwise API);
 *                     srba_flat.hpp: the batch-flat form of K1
 *   srba_big.hpp    : grid-wide phases + dense blocked Cholesky for ONE large capsule, LM control on the host (big_lm_run)
 * HBM layout: every capsule array is concatenated batch-wide into one device arena (SoA, 256-byte aligned sub-arrays); a second arena
 * holds state + workspaces.  A per-capsule descriptor (ProbDesc) carries sizes and element offsets.
 */
#include "srba_device.hpp"
#include "srba_wg.hpp"
#include <algorithm>
#include <iterator>
#include <map>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <atomic>
#include <exception>
#include <memory>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

using namespace srbadev;

// =================================================================================================== device: phases that need the dense system
namespace srbadev {

extern __shared__ double srba_lds[]; // block-sparse system of the capsule (diag | off | rhs) when it fits
// (LDS operands are addressed from this symbol where it matters: a pointer that went through SparseSys -- whose numbers may live in HBM -- is a generic pointer, and
//  its loads become flat_load instead of ds_read)

template <int FAM, bool LEAN = false, int G = 64>
struct Solver : public Worker<FAM, LEAN, G> {
	typedef Worker<FAM, LEAN, G> W; using W::B; using W::d; using W::prm; using W::tid; using W::E; using W::U; using W::Pz; using W::Eo; using W::Uo;
	static constexpr int P = W::P, L = W::L, O = W::O, PD = W::PD;
	double *red2w = nullptr; // G = 128: two doubles of LDS behind the image (group reductions, the solver's verdict)
	__device__ Solver(const Batch &B_, const ProbDesc &d_, const DevParams &p_, double *red_ = nullptr, int cp_ = 0) : W(B_, d_, p_, cp_), red2w(red_) {}
	__device__ __forceinline__ bool schur_active() const { return prm.solver != SRBA_SOLVER_NO_SCHUR_SPARSE_CHOL && d.nF > 0 && d.nK > 0; }

	// K7 + K8 (schur.h:180-268). Mutates HAp and minus_grad in place like the reference.
	__device__ __forceinline__ void schur_reduce(double lambda, long long *pc = nullptr) { this->fresh(); long long tq = pc ? wall_clock64() : 0;
		if constexpr (!W::T::REL) {
			for (int l = tid; l < d.nF; l += G) {
				double M[L * L], Mi[L * L]; const double *src = B.Hf + (d.o_hf + B.hf_diag[d.o_ulm + l]) * L * L;
				for (int k = 0; k < L * L; k++) M[k] = src[k];
				for (int k = 0; k < L; k++) M[k * L + k] += lambda;
				const bool ok = fullpiv_inverse<L>(M, Mi);
				B.hf_ok[d.o_ulm + l] = ok ? 1 : 0;
				if (ok) for (int k = 0; k < L * L; k++) B.Hfinv[(d.o_ulm + l) * L * L + k] = Mi[k];
			}
			for (int k = tid; k < d.n_hap * P * P; k += G) B.HAp[d.o_hap * P * P + k] = B.HAp0[d.o_hap * P * P + k];
			__syncthreads();
			if (pc) { if (tid == 0) pc[14] += wall_clock64() - tq; tq = wall_clock64(); }
			// Balanced over the lanes: the flat term list of the capsule (sorted by U_Ap block) is cut into 64 equal runs, one per lane. A lane keeps the running block in
			// registers and adds it to HBM when its run moves on to the next block (a block cut by a run boundary receives two or three such additions: atomics), so the
			// pass is as long as 1/64 of the terms, not as the diagonal block with the longest list in every group of 64 blocks. The terms of the diagonal blocks also
			// carry the gradient correction g_i -= Y_t g_l (K8), which used to be a third sweep over Y stored in HBM by this one.
			double *g = B.grad + d.o_scal; const double *gf = g + d.nK * P;
			{
				const int T = B.sch_term_off[d.o_hapoff + d.n_hap], per = (T + G - 1) / G, tb = tid * per, te = min(T, tb + per);
				const int *s_lm = B.sch_lm + d.o_sch, *s_b1 = B.sch_b1 + d.o_sch, *s_b2 = B.sch_b2 + d.o_sch, *s_yw = B.sch_yw + d.o_sch, *s_blk = B.sch_tblk + d.o_sch;
				int cur = -1; bool curdiag = false; double Hl[P * P], ga[P];
				auto flush = [&]() __attribute__((always_inline)) {
					if (cur < 0) return;
					const long long cw = W::wide(cur); // (an index proved non-negative that enters 64-bit address arithmetic: see Worker::wide)
					double *H = B.HAp + (d.o_hap + cw) * P * P;
#pragma unroll
					for (int k = 0; k < P * P; k++) unsafeAtomicAdd(H + k, Hl[k]); // global_atomic_add_f64 (the workspace is ordinary device memory)
					if (curdiag) { double *gi = g + B.hap_i[d.o_hap + cw] * P;
#pragma unroll
						for (int r = 0; r < P; r++) unsafeAtomicAdd(gi + r, ga[r]); }
				};
				constexpr int NT = 2; // terms per pass: indices and records of both are requested before either is used
				for (int t = tb; t < te; t += NT) {
					int lq[NT], aq[NT], cq[NT], ywq[NT], kq[NT]; bool okq[NT];
#pragma unroll
					for (int u = 0; u < NT; u++) { const bool live = t + u < te; const int tu = live ? t + u : t; lq[u] = s_lm[tu]; aq[u] = s_b1[tu]; cq[u] = s_b2[tu]; ywq[u] = s_yw[tu];
						kq[u] = s_blk[tu]; okq[u] = live; }
#pragma unroll
					for (int u = 0; u < NT; u++) okq[u] = okq[u] && B.hf_ok[d.o_ulm + lq[u]] != 0;
					double W1[NT][P * L], W2[NT][P * L], Hi[NT][L * L], gl[NT][L];
#pragma unroll
					for (int u = 0; u < NT; u++) { ldn<P * L>(W1[u], B.HApf + (d.o_hapf + aq[u]) * P * L); ldn<P * L>(W2[u], B.HApf + (d.o_hapf + cq[u]) * P * L); ldn<L * L>(Hi[u],
						B.Hfinv + (d.o_ulm + lq[u]) * L * L); ldn<L>(gl[u], gf + lq[u] * L); }
#pragma unroll
					for (int u = 0; u < NT; u++) {
						if (!okq[u]) continue;
						const int kb = kq[u];
						if (kb != cur) { flush(); cur = kb; curdiag = ywq[u] >= 0;
#pragma unroll
							for (int k = 0; k < P * P; k++) Hl[k] = 0;
#pragma unroll
							for (int r = 0; r < P; r++) ga[r] = 0; }
						double Y[P * L];
#pragma unroll
						for (int i = 0; i < P; i++)
#pragma unroll
							for (int j = 0; j < L; j++) { double sm = 0;
#pragma unroll
								for (int k = 0; k < L; k++) sm += W1[u][i * L + k] * Hi[u][k * L + j];
								Y[i * L + j] = sm; }
#pragma unroll
						for (int i = 0; i < P; i++)
#pragma unroll
							for (int j = 0; j < P; j++) { double sm = 0;
#pragma unroll
								for (int k = 0; k < L; k++) sm += Y[i * L + k] * W2[u][j * L + k];
								Hl[i * P + j] -= sm; }
						if (curdiag) {
#pragma unroll
							for (int r = 0; r < P; r++) { double sm = 0;
#pragma unroll
								for (int k = 0; k < L; k++) sm += Y[r * L + k] * gl[u][k];
								ga[r] -= sm; } }
					}
				}
				flush();
			}
			__syncthreads();
			if (pc) { if (tid == 0) pc[15] += wall_clock64() - tq; }
		}
	}
	// ---- Workgroup path with the U_Ap blocks in LDS (ProbDesc::hs_lds; round 5). The landmark kernels run at 2 - 3 TB/s of HBM traffic: the one-wavefront forms above read both Jacobian
	// blocks of every K6 term and both U_Apf blocks of every Schur term from memory (terms sorted by OUTPUT block, the sum kept in registers), i.e. every input block 3 .. 9 times per pass.
	// Here the OUTPUT lives on chip -- n_hap x P x P doubles of LDS, summed with ds_add_f64 -- and the terms are sorted by INPUT (K6: by observation, K7/K8: by landmark), a lane per
	// term: the lanes of a wavefront touch the same few input blocks at the same time, which then come from the vector L1; HBM sees every input block about once.
	// (block stride P P + 1 doubles: with 36 the same element of 64 different blocks falls on 8 of the 32 bank pairs of the LDS -- an 8-way conflict on every ds_add_f64 of the term loops)
	static constexpr int HS = P * P + 1;
	__device__ __forceinline__ double *hs() const { return srba_lds + WG_HS; }
	// A window whose U_Ap blocks do not all fit the LDS of a CU (more than 515: a third of the windows of the cfg3 room, half of its work) is swept in PANELS: block ranges that do fit, each
	// with its own share of the two term lists (the host sorts the lists by panel first: every term still runs once). ProbDesc::n_panel, Batch::ptab.
	__device__ __forceinline__ const int *ptab() const { return B.ptab + d.o_ptab; }
	// the LDS blocks [b0, b1) -> U_Ap (and its latch, schur.h:38) in memory, 16 bytes per lane and request
	__device__ __forceinline__ void store_hs(bool latch_too, int b0, int b1) { this->fresh();
		const double *H = hs(); double *Hg = B.HAp + (d.o_hap + b0) * P * P, *H0 = B.HAp0 + (d.o_hap + b0) * P * P; const int n_acc = (b1 - b0) * P * P; static_assert((P * P) % 2 == 0,
			"pairs of doubles inside a block");
		for (int k = 2 * tid; k < n_acc; k += 2 * G) { const int b = k / (P * P), e = k - b * (P * P); f64x2u v; v.x = H[b * HS + e]; v.y = H[b * HS + e + 1]; *(f64x2u *)(Hg + k) = v;
			if (latch_too) *(f64x2u *)(H0 + k) = v; }
	}
	__device__ __forceinline__ void store_hs(bool latch_too) { store_hs(latch_too, 0, d.n_hap); } // (single-panel windows: the reduced blocks go to memory once, when the run ends)
	// K6 (sparse_hessian_update_numeric.h:22-60): U_Ap summed in LDS from the term list sorted by observation; U_f and U_Apf as before (their lists are a landmark's observations: short)
	__device__ __forceinline__ int phase_hessian_lds() { this->fresh();
		double *H = hs(); const int *pt = ptab(); const int np = d.n_panel;
		const double *Jp = B.Jp + d.o_bp * O * P; const unsigned char *rp = B.bp_ok + d.o_bp; const int *rec = B.hapo + d.o_hapo * 3;
		const double sc = (prm.noise == SRBA_NOISE_IDENTITY) ? prm.inv_sigma : 1.0;
		int ninv = 0;
		for (int q = 0; q < np; q++) {
			const int pb0 = pt[q], pb1 = pt[q + 1], t0 = pt[np + 1 + q], t1 = pt[np + 2 + q];
			for (int k = tid; k < (pb1 - pb0) * HS; k += G) H[k] = 0;
			__syncthreads();
			// (one term in flight per lane, the next record requested ahead. Two terms in flight -- 96 more registers -- were measured and changed nothing, tools/r5_session6.sh:
			//  with the blocks of a wavefront's terms shared through the vector L1 the loop is bound by its 36 ds_add_f64 per term, not by the loads)
			int b1 = 0, b2 = 0, blk = 0;
			if (t0 + tid < t1) { b1 = rec[3 * (t0 + tid)]; b2 = rec[3 * (t0 + tid) + 1]; blk = rec[3 * (t0 + tid) + 2]; }
			for (int t = t0 + tid; t < t1; t += G) {
				const int tn = t + G; int n1 = 0, n2 = 0, nb = 0; if (tn < t1) { n1 = rec[3 * tn]; n2 = rec[3 * tn + 1]; nb = rec[3 * tn + 2]; }
				double A[O * P], Bm[O * P]; ldn<O * P>(A, Jp + (long long)b1 * O * P); ldn<O * P>(Bm, Jp + (long long)b2 * O * P);
				if (rp[b1] && rp[b2]) {
					double *dst = H + (blk - pb0) * HS;
#pragma unroll
					for (int i = 0; i < P; i++) { double row[P]; W::template hess_row<P, P>(row, A, Bm, i);
#pragma unroll
						for (int j = 0; j < P; j++) atomicAdd(dst + i * P + j, row[j] * sc); }
				} else ninv++;
				b1 = n1; b2 = n2; blk = nb;
			}
			if (q == 0) ninv += this->phase_hessian_landmark_blocks();
			__syncthreads();
			store_hs(prm.solver != SRBA_SOLVER_NO_SCHUR_SPARSE_CHOL, pb0, pb1);
			if (q + 1 < np) __syncthreads();
		}
		return ninv;
	}
	// K7 + K8 (schur.h:180-268) AND the assembly of the reduced system into the tile layout (assemble_tiles): the reduced U_Ap blocks start from the latch in LDS, a lane per Schur term (sorted
	// by landmark) adds -Y W^t with Y = W Hf^-1 formed per term from cached inputs, and the finished blocks go from LDS straight into the lower triangle of 16 x 16 tiles -- panel by panel.
	__device__ __forceinline__ void schur_assemble_lds(const SparseSys &S, double lambda, long long *pc = nullptr) { this->fresh(); long long tq = pc ? wall_clock64() : 0;
		if constexpr (!W::T::REL) {
			double *H = hs(), *gacc = srba_lds + WG_GACC; const int *pt = ptab(); const int np = d.n_panel;
			const int n = d.n_sys, ntl = S.nt; double *T = S.tiles;
			for (int l = tid; l < d.nF; l += G) {
				double M[L * L], Mi[L * L]; const double *src = B.Hf + (d.o_hf + B.hf_diag[d.o_ulm + l]) * L * L;
				for (int k = 0; k < L * L; k++) M[k] = src[k];
				for (int k = 0; k < L; k++) M[k * L + k] += lambda;
				const bool ok = fullpiv_inverse<L>(M, Mi);
				B.hf_ok[d.o_ulm + l] = ok ? 1 : 0;
				if (ok) for (int k = 0; k < L * L; k++) B.Hfinv[(d.o_ulm + l) * L * L + k] = Mi[k];
			}
			{ const long long n2 = 128LL * (ntl + 1) * (ntl + 2) / 2; f64x2u z; z.x = 0; z.y = 0; for (long long k = tid; k < n2; k += G) *(f64x2u *)(T + 2 * k) = z; } // the tile area is cleared ...
			for (int k = tid; k < d.nK * P; k += G) gacc[k] = 0;
			auto at = [&](int r, int c) __attribute__((always_inline)) -> double * { return T + 256 * (long long)wg_tile(r >> 4, c >> 4) + wg_frag_off(r & 15, c & 15); }; // element (r, c), r >= c
			const int *rec = B.schl + d.o_schl * 4; const double *gf = B.grad + d.o_scal + d.nK * P;
			for (int q = 0; q < np; q++) {
				const int pb0 = pt[q], pb1 = pt[q + 1], t0 = pt[2 * np + 2 + q], t1 = pt[2 * np + 3 + q], n_acc = (pb1 - pb0) * P * P;
				{ const double *H0 = B.HAp0 + (d.o_hap + pb0) * P * P;
				  for (int k = 2 * tid; k < n_acc; k += 2 * G) { const int b = k / (P * P), e = k - b * (P * P); const f64x2u v = *(const f64x2u *)(H0 + k); H[b * HS + e] = v.x;
				  	H[b * HS + e + 1] = v.y; } }
				__syncthreads();
				if (pc && q == 0) { if (tid == 0) pc[14] += wall_clock64() - tq; tq = wall_clock64(); }
				int l = 0, b1 = 0, b2 = 0, w = 0;
				if (t0 + tid < t1) { const int t = t0 + tid; l = rec[4 * t]; b1 = rec[4 * t + 1]; b2 = rec[4 * t + 2]; w = rec[4 * t + 3]; }
				for (int t = t0 + tid; t < t1; t += G) {
					const int tn = t + G; int nl = 0, n1 = 0, n2 = 0, nw = 0; if (tn < t1) { nl = rec[4 * tn]; n1 = rec[4 * tn + 1]; n2 = rec[4 * tn + 2]; nw = rec[4 * tn + 3]; }
					double W1[P * L], W2[P * L], Hi[L * L], gl[L];
					ldn<P * L>(W1, B.HApf + (d.o_hapf + b1) * P * L); ldn<P * L>(W2, B.HApf + (d.o_hapf + b2) * P * L); ldn<L * L>(Hi, B.Hfinv + (d.o_ulm + l) * L * L); ldn<L>(gl, gf + l * L);
					if (B.hf_ok[d.o_ulm + l] != 0) {
						const int blk = w & 0xffff, e = (w >> 16) & 0x7fff; const bool diag = w < 0;
						double *dst = H + (blk - pb0) * HS;
#pragma unroll
						for (int i = 0; i < P; i++) {
							double y[L];
#pragma unroll
							for (int j = 0; j < L; j++) { double sm = 0;
#pragma unroll
								for (int k = 0; k < L; k++) sm += W1[i * L + k] * Hi[k * L + j];
								y[j] = sm; }
#pragma unroll
							for (int j = 0; j < P; j++) { double sm = 0;
#pragma unroll
								for (int k = 0; k < L; k++) sm += y[k] * W2[j * L + k];
								atomicAdd(dst + i * P + j, -sm); }
							if (diag) { double sm = 0;
#pragma unroll
								for (int k = 0; k < L; k++) sm += y[k] * gl[k];
								atomicAdd(gacc + e * P + i, -sm); }
						}
					}
					l = nl; b1 = n1; b2 = n2; w = nw;
				}
				__syncthreads();
				if (pc && q + 1 == np) { if (tid == 0) pc[15] += wall_clock64() - tq; tq = wall_clock64(); }
				// ... and the panel's reduced blocks leave LDS for the tiles: a lane per block ROW, every upper-triangle block (i <= j) transposed into the lower triangle
				for (int e0 = tid; e0 < (pb1 - pb0) * P; e0 += G) {
					const int b = e0 / P, r = e0 - b * P; const int i = B.hap_i[d.o_hap + pb0 + b], j = B.hap_j[d.o_hap + pb0 + b]; const double *src = H + b * HS + r * P;
#pragma unroll
					for (int c = 0; c < P; c++) {
						if (i != j) *at(P * j + c, P * i + r) = src[c];
						else if (r >= c) *at(P * i + r, P * i + c) = src[c] + (r == c ? lambda : 0.0); // (a diagonal block holds both triangles)
					}
				}
				if (np > 1) store_hs(false, pb0, pb1); // (what the reference's in-place Schur complement leaves in HAp; a single panel is written once, when the run ends)
				if (q + 1 < np) __syncthreads();
			}
			{ double *g = B.grad + d.o_scal; for (int k = tid; k < d.nK * P; k += G) g[k] += gacc[k]; }
			__syncthreads();
			const double *g = B.grad + d.o_scal; // the (corrected) gradient is tile row nt; rows beyond n_sys get an identity diagonal
			for (int k = tid; k < 16 * ntl; k += G) { T[256 * (long long)wg_tile(ntl, k >> 4) + wg_frag_off(0, k & 15)] = (k < n) ? g[k] : 0.0; if (k >= n) *at(k, k) = 1.0; }
			__syncthreads();
			if (pc) { if (tid == 0) pc[10] += wall_clock64() - tq; }
		}
	}
	// K10 (schur.h:271-311)
	__device__ __forceinline__ void schur_features() { this->fresh();
		if constexpr (!W::T::REL && LEAN) { // workgroup kernels: a landmark's U_Apf blocks were a serial chain of (index, block, increment) loads per lane, with as many lanes busy as the window has
			// landmarks. Two passes instead: a lane per U_Apf block forms W^t delta_i (three numbers, parked in the block's slot of Yh), then a lane per landmark sums its blocks' three
			// numbers (independent loads), finishes g_l and multiplies by Hf^-1. Same sums in the same order per landmark.
			double *g = B.grad + d.o_scal, *dl = B.delta + d.o_scal;
			for (int hb = tid; hb < d.n_hapf; hb += G) {
				const int i = B.hapf_i[d.o_hapf + hb]; double Wm[P * L], di[P]; ldn<P * L>(Wm, B.HApf + (d.o_hapf + hb) * P * L); ldn<P>(di, dl + i * P);
				double pk[L];
#pragma unroll
				for (int k = 0; k < L; k++) { double sm = 0;
#pragma unroll
					for (int r = 0; r < P; r++) sm += Wm[r * L + k] * di[r];
					pk[k] = sm; }
				double *o = B.Yh + (d.o_hapf + hb) * P * L;
#pragma unroll
				for (int k = 0; k < L; k++) o[k] = pk[k];
			}
			__syncthreads();
			for (int l = tid; l < d.nF; l += G) {
				if (!B.hf_ok[d.o_ulm + l]) continue;
				double gl[L]; ldn<L>(gl, g + d.nK * P + l * L);
				const int qb = B.lm_hapf_off[d.o_lmoff + l], qe = B.lm_hapf_off[d.o_lmoff + l + 1];
				for (int q = qb; q < qe; q += 4) { // four blocks' numbers in flight
					double pk[4][L];
#pragma unroll
					for (int u = 0; u < 4; u++) if (q + u < qe) { const int hb = B.lm_hapf_idx[d.o_hapf + q + u]; const double *o = B.Yh + (d.o_hapf + hb) * P * L;
#pragma unroll
						for (int k = 0; k < L; k++) pk[u][k] = o[k]; }
#pragma unroll
					for (int u = 0; u < 4; u++) if (q + u < qe) {
#pragma unroll
						for (int k = 0; k < L; k++) gl[k] -= pk[u][k]; }
				}
				double Hi[L * L]; ldn<L * L>(Hi, B.Hfinv + (d.o_ulm + l) * L * L);
#pragma unroll
				for (int k = 0; k < L; k++) g[d.nK * P + l * L + k] = gl[k];
#pragma unroll
				for (int r = 0; r < L; r++) { double sm = 0;
#pragma unroll
					for (int k = 0; k < L; k++) sm += Hi[r * L + k] * gl[k];
					dl[d.nK * P + l * L + r] = sm; }
			}
			__syncthreads();
		} else
		if constexpr (!W::T::REL) {
			double *g = B.grad + d.o_scal, *dl = B.delta + d.o_scal;
			for (int l = tid; l < d.nF; l += G) {
				if (!B.hf_ok[d.o_ulm + l]) continue;
				double gl[L]; for (int k = 0; k < L; k++) gl[k] = g[d.nK * P + l * L + k];
				for (int q = B.lm_hapf_off[d.o_lmoff + l]; q < B.lm_hapf_off[d.o_lmoff + l + 1]; q++) {
					const int hb = B.lm_hapf_idx[d.o_hapf + q]; const int i = B.hapf_i[d.o_hapf + hb];
					const double *Wm = B.HApf + (d.o_hapf + hb) * P * L;
					for (int k = 0; k < L; k++) { double s = 0; for (int r = 0; r < P; r++) s += Wm[r * L + k] * dl[i * P + r]; gl[k] -= s; }
				}
				const double *Hi = B.Hfinv + (d.o_ulm + l) * L * L;
				for (int k = 0; k < L; k++) g[d.nK * P + l * L + k] = gl[k];
				for (int r = 0; r < L; r++) { double s = 0; for (int k = 0; k < L; k++) s += Hi[r * L + k] * gl[k]; dl[d.nK * P + l * L + r] = s; }
			}
			__syncthreads();
		}
	}
	// (H + lambda I) scattered into the block-sparse storage + right-hand side (lev-marq_solvers.h:88-150 / :303-325 / :492-519)
	__device__ __forceinline__ void put(const SparseSys &S, int row, int col, double v) const { // symmetric entry (row,col) of the original numbering
		const int pr = 3 * S.perm[row / 3] + row % 3, pc = 3 * S.perm[col / 3] + col % 3;
		double *p = pr >= pc ? sp_elem(S, pr, pc) : sp_elem(S, pc, pr); if (p) *p = v;
	}
	// store element (r,q) of an upper-triangle sub-block whose destination block is `dst` (>=0 off-diagonal, stored transposed; <0 diagonal)
	// dst >= 0: off-diagonal block (dst>>1), bit0 = stored transposed ; dst < 0: diagonal block -(1+dst) (lower triangle kept) ; 0x80000000: duplicate, skipped
	__device__ __forceinline__ void put_dst(const SparseSys &S, int dst, int r3, int q3, double v) const {
		if (dst >= 0) S.off[9 * (dst >> 1) + ((dst & 1) ? q3 * 3 + r3 : r3 * 3 + q3)] = v;
		else if (dst != (int)0x80000000 && r3 <= q3) S.diag[9 * (-1 - dst) + q3 * 3 + r3] = v;
	}
	// copy one 3x3 block (row stride LD) to its destination; diagonal destinations keep the lower triangle and get +lambda on the diagonal
	template <int LD> __device__ __forceinline__ void put_block(const SparseSys &S, int dst, const double *H, double lambda) const {
		double v[9];
#pragma unroll
		for (int r = 0; r < 3; r++)
#pragma unroll
			for (int q = 0; q < 3; q++) v[r * 3 + q] = H[r * LD + q];
		if (dst >= 0) {
			double *o = S.off + 9 * (dst >> 1);
			if (dst & 1) {
#pragma unroll
				for (int r = 0; r < 3; r++)
#pragma unroll
					for (int q = 0; q < 3; q++) o[q * 3 + r] = v[r * 3 + q];
			} else {
#pragma unroll
				for (int k = 0; k < 9; k++) o[k] = v[k];
			}
		} else {
			double *o = S.diag + 9 * (-1 - dst);
#pragma unroll
			for (int r = 0; r < 3; r++)
#pragma unroll
				for (int q = r; q < 3; q++) o[q * 3 + r] = v[r * 3 + q] + (r == q ? lambda : 0.0);
		}
	}
	__device__ __forceinline__ void assemble(const SparseSys &S, double lambda) { this->fresh();
		const int n = d.n_sys, nb = d.nb;
		if (d.aligned) { // every block is overwritten whole by put_block below, except the fill-in blocks (listed by the host): zero only those
			for (int i = tid; i < d.n_fill; i += G) { double *o = S.diag + 9 * B.sp_fill[d.o_spfill + i];
#pragma unroll
				for (int q = 0; q < 9; q++) o[q] = 0; }
		} else {
			for (int k = tid; k < 9 * nb; k += G) S.diag[k] = 0;
			for (int k = tid; k < 9 * S.nnzoff; k += G) S.off[k] = 0;
		}
		const double *g = B.grad + d.o_scal;
		for (int k = tid; k < 3 * nb; k += G) S.rhs[3 * S.perm[k / 3] + k % 3] = (k < n) ? g[k] : 0.0;
		if (!d.aligned) __syncthreads(); // (aligned: the fill-in blocks zeroed above, the right-hand side and the blocks written below are disjoint pieces of the image -- no order between them,
			// and the loads of the gradient and of the Hessian blocks are in flight together)
		constexpr int PB = P / 3;
		// one lane per aligned 3x3 sub-block: its 9 loads are in flight together (one memory round trip per pass instead of one per element)
		for (int sb = tid; sb < d.n_hap * PB * PB; sb += G) {
			const int dst = B.hap_dst[d.o_hap * PB * PB + sb]; if (dst == (int)0x80000000) continue;
			const int b = sb / (PB * PB), si = (sb / PB) % PB, sj = sb % PB;
			put_block<P>(S, dst, B.HAp + (d.o_hap + b) * P * P + si * 3 * P + sj * 3, lambda);
		}
		if constexpr (!W::T::REL) if (prm.solver == SRBA_SOLVER_NO_SCHUR_SPARSE_CHOL) {
			const int base = P * d.nK;
			if (d.aligned) {
				if constexpr (L == 3) {
					for (int sb = tid; sb < d.n_hapf * PB; sb += G) {
						const int dst = B.hapf_dst[d.o_hapf * PB + sb]; if (dst == (int)0x80000000) continue;
						put_block<L>(S, dst, B.HApf + (d.o_hapf + sb / PB) * P * L + (sb % PB) * 3 * L, 0.0);
					}
					for (int b = tid; b < d.n_hf; b += G) {
						const int dst = B.hf_dst[d.o_hf + b]; if (dst == (int)0x80000000) continue;
						put_block<L>(S, dst, B.Hf + (d.o_hf + b) * L * L, lambda);
					}
				}
			} else { // landmark blocks straddle 3x3 block boundaries (L == 2): generic per-element placement
				for (int e = tid; e < d.n_hapf * P * L; e += G) {
					const int b = e / (P * L), r = (e / L) % P, q = e % L;
					put(S, P * B.hapf_i[d.o_hapf + b] + r, base + L * B.hapf_j[d.o_hapf + b] + q, B.HApf[(d.o_hapf + b) * P * L + r * L + q]);
				}
				for (int e = tid; e < d.n_hf * L * L; e += G) {
					const int b = e / (L * L), r = (e / L) % L, q = e % L; const int i = B.hf_i[d.o_hf + b], j = B.hf_j[d.o_hf + b];
					if (i == j && r > q) continue;
					put(S, base + L * i + r, base + L * j + q, B.Hf[(d.o_hf + b) * L * L + r * L + q] + ((i == j && r == q) ? lambda : 0.0));
				}
			}
		}
		for (int k = n + tid; k < 3 * nb; k += G) S.diag[9 * S.perm[k / 3] + 4 * (k % 3)] = 1.0; // identity padding
		__syncthreads();
	}
	// solve(lambda): returns false if not positive definite (uniform across the wavefront)
	__device__ __forceinline__ bool solve(const SparseSys &S, double lambda, long long *pc = nullptr) {
		long long t0 = 0;
#define STIC() do { if (pc) { __syncthreads(); t0 = wall_clock64(); } } while (0)
#define STOC(slot) do { if (pc) { __syncthreads(); if (tid == 0) pc[slot] += wall_clock64() - t0; } } while (0)
		if (schur_active() && (prm.ext & SRBA_EXT_SCHUR_KEEPS_GRADIENT)) { // (extension, default off) every solve starts from the gradient K5 produced: same lane -> same elements as keep_gradient()
			double *g = B.grad + d.o_scal; const double *g0 = B.grad0 + d.o_scal; for (int k = tid; k < d.n_scal; k += G) g[k] = g0[k]; __syncthreads(); }
		if constexpr (G > 64 && !W::T::REL) { // landmark window on a workgroup (the host sends Schur windows only): U_Ap in LDS, dense LL^t on the matrix cores (srba_wg.hpp)
			STIC(); schur_assemble_lds(S, lambda, pc); STOC(9);
			STIC(); const bool okw = wg_chol_solve<G / 64>(S.tiles, S.linv, S.nt, (lds_f64 *)srba_lds); STOC(11);
			if (!okw) return false;
			double *dlw = B.delta + d.o_scal;
			for (int k = tid; k < d.n_scal; k += G) dlw[k] = (k < d.n_sys) ? S.rhs[k] : 0.0;
			__syncthreads();
			STIC(); if (schur_active()) schur_features(); STOC(13);
			return true;
		}
		STIC(); if (schur_active()) schur_reduce(lambda, pc); STOC(9);
		STIC(); assemble(S, lambda); STOC(10);
		// (the dense block layouts are never chosen for the relative-pose families -- srba_hip_upload_problems -- whose kernels therefore carry the sparse solver only: the
		//  headline kernel sits 22 VGPRs below the two-wavefronts-per-SIMD limit)
		STIC(); bool ok;
		if constexpr (G > 64 && !W::T::REL) { ok = false; /* (unreachable: the workgroup branch above returned) */ } else if constexpr (G > 64) {
			// two wavefronts per capsule (sparse layout only): the first one factors and substitutes, the verdict travels through the reduction scratch behind the image
			int *flag = (int *)red2w; if (threadIdx.x < 64) { const bool k1 = sp_factor_fsub_rows(S); if (k1) sp_bsub_rows(S); if (threadIdx.x == 0) *flag = k1 ? 1 : 0; }
			__syncthreads(); ok = *flag != 0; __syncthreads();
		} else if constexpr (W::T::REL || !W::T::SE3) ok = sp_factor_fsub_rows(S); else ok = d.dense_blocks == 2 ? (S.row_lds ? sp_factor_fsub_dense_left(S,
			(lds_f64 *)srba_lds + ((S.nb + 1) / 2 + 16), (lds_f64 *)srba_lds + ((S.nb + 1) / 2 + 16 + 18 * S.nb)) : sp_factor_fsub_dense<true>(S)) : (S.dense ? sp_factor_fsub_dense<false>(S) :
			sp_factor_fsub_rows(S)); STOC(11);
		if (!ok) return false;
		STIC(); if constexpr (G > 64) { /* done above */ } else if constexpr (W::T::REL || !W::T::SE3) sp_bsub_rows(S); else { if (d.dense_blocks == 2) { if (S.row_lds) sp_bsub_dense_left(S,
			(lds_f64 *)srba_lds + ((S.nb + 1) / 2 + 16 + 18 * S.nb)); else sp_bsub_dense<true>(S); } else if (S.dense) sp_bsub_dense<false>(S); else sp_bsub_rows(S); }
		double *dl = B.delta + d.o_scal;
		for (int k = tid; k < d.n_scal; k += G) dl[k] = (k < d.n_sys) ? S.sol(k) : 0.0;
		if (schur_active()) __syncthreads(); /* (K10 reads the increments of the edges back from memory; without landmarks to solve for,
			the loop takes them from the LDS image: no reader waits for these stores) */ STOC(12);
		STIC(); if (schur_active()) schur_features(); STOC(13);
		return true;
#undef STIC
#undef STOC
	}
	__device__ __forceinline__ void keep_gradient() { // call after phase_gradient + barrier
		if (schur_active() && (prm.ext & SRBA_EXT_SCHUR_KEEPS_GRADIENT)) { const double *g = B.grad + d.o_scal; double *g0 = B.grad0 + d.o_scal; for (int k = tid; k < d.n_scal; k += G) g0[k] = g[k]; }
	}
	__device__ __forceinline__ SparseSys make_sys(double *lds) const {
		SparseSys S; S.tiles = S.linv = nullptr; S.nt = 0;
		if constexpr (G > 64 && !W::T::REL) { // a landmark window on a workgroup (srba_wg.hpp): tile system in the capsule's HBM workspace, x in LDS; no symbolic structure, no permutation
			S.nb = d.nb; S.nnzoff = 0; S.dense = 3; S.row_lds = nullptr; S.col_off = S.row = S.item = S.rptr = S.rent = S.perm = nullptr; S.diag = S.off = nullptr;
			S.nt = (d.n_sys + WT - 1) / WT; S.tiles = B.dense + d.o_dense; S.linv = S.tiles + 256 * (long long)((S.nt + 1) * (S.nt + 2) / 2); S.rhs = lds + 768;
			return S;
		}
		S.nb = d.nb; S.nnzoff = d.nnzoff; S.dense = (W::T::REL || !W::T::SE3) ? 0 : d.dense_blocks;
		S.row_lds = (S.dense == 2 && B.dense_left && d.nb <= 168) ? lds + (d.nb + 1) / 2 + 16 : nullptr; // HBM-resident layout, left-looking sweeps: 21 nb doubles of LDS after the permutation (two
			// rows of the factor | y)
		S.col_off = B.sp_col_off + d.o_spcol; S.row = B.sp_row + d.o_sprow; S.item = B.sp_tgt + d.o_spitem; S.perm = B.sp_perm + d.o_spperm;
		S.rptr = B.sp_rptr + d.o_spcol; S.rent = B.sp_rcol + d.o_sprow;
		double *base = (!W::T::REL && W::T::SE3 && d.dense_blocks == 2) ? B.dense + d.o_dense : lds; // 2: the numbers live in an HBM workspace, LDS holds the permutation only
		S.diag = base; S.off = base + 9 * d.nb; S.rhs = S.off + 9 * d.nnzoff;
		if (S.dense) { // numbers only: every index of the dense block layout is arithmetic; the block permutation is the one table kept
			int *p0 = d.dense_blocks == 2 ? (int *)lds : (int *)(S.rhs + 3 * d.nb);
			for (int k = tid; k < d.nb; k += G) p0[k] = S.perm[k];
			S.perm = p0; S.col_off = S.row = S.item = S.rptr = S.rent = nullptr;
			__syncthreads();
		} else { // symbolic structure next to the numbers (packed): the factorisation's dependent index loads hit LDS, not L2
			int *ip = (int *)(S.rhs + 3 * d.nb); int *c0 = ip, *rp0 = c0 + d.nb + 1, *p0 = rp0 + d.nb + 1, *r0 = p0 + d.nb, *re0 = r0 + d.nnzoff, *t0 = re0 + d.nnzoff;
			for (int k = tid; k <= d.nb; k += G) { c0[k] = S.col_off[k]; rp0[k] = S.rptr[k]; }
			for (int k = tid; k < d.nb; k += G) p0[k] = S.perm[k];
			for (int k = tid; k < d.nnzoff; k += G) { r0[k] = S.row[k]; re0[k] = S.rent[k]; } // (the item / row-view words were packed at upload)
			for (int k = tid; k < d.n_items; k += G) t0[k] = S.item[k];
			S.col_off = c0; S.rptr = rp0; S.perm = p0; S.row = r0; S.rent = re0; S.item = t0;
			__syncthreads();
		}
		return S;
	}

	// K12 backup + K11 apply (optimize_edges.h:491-557)
	__device__ __forceinline__ void apply_update() { this->fresh();
		typedef typename W::PO PO;
		const double *dl = B.delta + d.o_scal;
		for (int i = tid; i < d.nK; i += G) {
			double *e = E() + (d.o_edge + i) * PD, *o = B.old_edge + (d.o_unk + i) * PD;
			for (int k = 0; k < PD; k++) o[k] = e[k];
			const typename W::pose_t np = comp(PO::expm(dl + i * P), PO::ld(e));
			PO::st(e, np);
		}
		for (int k = tid; k < d.nF * L; k += G) { B.old_ulm[d.o_ulm * L + k] = U()[d.o_ulm * L + k]; U()[d.o_ulm * L + k] += dl[d.nK * P + k]; }
		for (int r = tid; r < d.n_req; r += 2 * G) { // two poses per lane and pass: both loads before the stores
			const int r2 = r + G; const bool two = r2 < d.n_req;
			const double *s = Pz() + (d.o_pair * 2 + B.req_idx[d.o_req + r]) * PD, *s2 = two ? Pz() + (d.o_pair * 2 + B.req_idx[d.o_req + r2]) * PD : s;
			double v[PD], v2[PD];
#pragma unroll
			for (int k = 0; k < PD; k++) { v[k] = s[k]; v2[k] = s2[k]; }
			double *o = B.old_pose + (d.o_req + r) * PD, *o2 = B.old_pose + (d.o_req + r2) * PD;
#pragma unroll
			for (int k = 0; k < PD; k++) o[k] = v[k];
			if (two) {
#pragma unroll
				for (int k = 0; k < PD; k++) o2[k] = v2[k];
			}
		}
		__syncthreads();
	}
	// K12 + K11 with the increment of the edges taken straight from the solved right-hand side in LDS (no round trip through B.delta) and, when it fits, a copy of
	// ALL edge poses of the capsule left in the (now idle) off-diagonal area of the LDS image for the spanning-tree refresh that follows. Returns that copy or nullptr.
	__device__ __forceinline__ const double *apply_update_lds(const SparseSys &S) { this->fresh();
		typedef typename W::PO PO; typedef typename W::pose_t pose_t;
		const double *dl = B.delta + d.o_scal;
		const bool stage = d.dense_in_lds && d.n_edges * PD <= 9 * S.nnzoff; double *el = S.off;
		for (int i = tid; i < (stage ? d.n_edges : d.nK); i += G) {
			double *e = E() + (d.o_edge + i) * PD; pose_t cur = PO::ld(e);
			if (i < d.nK) {
				double inc[P];
#pragma unroll
				for (int k = 0; k < P; k++) { const int q = i * P + k; inc[k] = S.sol(q); }
				PO::st(B.old_edge + (d.o_unk + i) * PD, cur);
				cur = comp(PO::expm(inc), cur);
				PO::st(e, cur);
			}
			if (stage) { double t[PD]; PO::to(t, cur);
#pragma unroll
				for (int k = 0; k < PD; k++) el[i * PD + k] = t[k]; }
		}
		for (int k = tid; k < d.nF * L; k += G) { B.old_ulm[d.o_ulm * L + k] = U()[d.o_ulm * L + k]; U()[d.o_ulm * L + k] += dl[d.nK * P + k]; }
		for (int r = tid; r < d.n_req; r += 2 * G) { // two poses per lane and pass: both loads before the stores
			const int r2 = r + G; const bool two = r2 < d.n_req;
			const double *s = Pz() + (d.o_pair * 2 + B.req_idx[d.o_req + r]) * PD, *s2 = two ? Pz() + (d.o_pair * 2 + B.req_idx[d.o_req + r2]) * PD : s;
			double v[PD], v2[PD]; ldn<PD>(v, s); ldn<PD>(v2, s2);
			stn<PD>(B.old_pose + (d.o_req + r) * PD, v);
			if (two) stn<PD>(B.old_pose + (d.o_req + r2) * PD, v2);
		}
		// with the staged copy the refresh that follows reads LDS only, and nothing reads the arrays written here before the next full barrier (end of that refresh):
		// the hand-off is an LDS one (no wait for the global stores to be acknowledged)
		if (stage) grp_lds_sync<G>(); else __syncthreads();
		return stage ? el : nullptr;
	}
	// K11 for the double-buffered loop: the trial unknowns exp(delta) (+) edge, lm + delta go to the OTHER copy (Bt), nothing is backed up (a rejected trial simply leaves the
	// accepted copy as it is); like apply_update_lds the increment comes from the solved right-hand side in LDS and all edge poses of the trial are staged in the idle part of the
	// LDS image for the spanning-tree refresh that follows. Returns that copy or nullptr.
	__device__ __forceinline__ const double *apply_trial(const SparseSys &S) { this->fresh();
		typedef typename W::PO PO; typedef typename W::pose_t pose_t;
		const double *dl = B.delta + d.o_scal;
		const bool stage = d.dense_in_lds && d.n_edges * PD <= 9 * S.nnzoff; double *el = S.off;
		for (int i = tid; i < (stage ? d.n_edges : d.nK); i += G) {
			pose_t cur = PO::ld(E() + (d.o_edge + i) * PD);
			if (i < d.nK) {
				double inc[P];
#pragma unroll
				for (int k = 0; k < P; k++) { const int q = i * P + k; inc[k] = S.sol(q); }
				cur = comp(PO::expm(inc), cur);
				PO::st(Eo() + (d.o_edge + i) * PD, cur);
			}
			if (stage) { double t[PD]; PO::to(t, cur);
#pragma unroll
				for (int k = 0; k < PD; k++) el[i * PD + k] = t[k]; }
		}
		{ double *uo = Uo(); const double *ua = U(); for (int k = tid; k < d.nF * L; k += G) uo[d.o_ulm * L + k] = ua[d.o_ulm * L + k] + dl[d.nK * P + k]; }
		if (stage) grp_lds_sync<G>(); else __syncthreads();
		return stage ? el : nullptr;
	}
	__device__ __forceinline__ void restore() { this->fresh(); // optimize_edges.h:664-680
		for (int i = tid; i < d.nK * PD; i += G) E()[d.o_edge * PD + i] = B.old_edge[d.o_unk * PD + i];
		for (int k = tid; k < d.nF * L; k += G) U()[d.o_ulm * L + k] = B.old_ulm[d.o_ulm * L + k];
		for (int r = tid; r < d.n_req; r += G) { double *s = Pz() + (d.o_pair * 2 + B.req_idx[d.o_req + r]) * PD; const double *o = B.old_pose + (d.o_req + r) * PD; for (int k = 0; k < PD;
			k++) s[k] = o[k]; }
		__syncthreads();
	}
};


#ifdef SRBA_WAVES_PER_EU
#define SRBA_OCC __attribute__((amdgpu_waves_per_eu(SRBA_WAVES_PER_EU, SRBA_WAVES_PER_EU)))
#else
#define SRBA_OCC
#endif
#ifndef SRBA_LM_DB
#define SRBA_LM_DB 1   /* the fused loop keeps two copies of the unknowns and of the spanning-tree poses (trial -> the other copy, accept = flip) instead of backup / restore */
#endif
#ifndef SRBA_FUSE_K4
#define SRBA_FUSE_K4 1 /* a trial's residuals compose the poses of the refreshed pairs themselves (Worker::phase_residuals_fused); 0: refresh the pose table, then gather from it (rounds 1-3) */
#endif
template <int FAM, bool DB = (SRBA_LM_DB != 0), bool LEAN = false, int G = 64, bool FUSE_K4 = (SRBA_FUSE_K4 != 0) && (LEAN || G > 64) /* k_lm_run itself (every family,
	253 registers for the headline one) keeps the table path: the fused form costs it 13 registers and with them its second wavefront per SIMD */,
          bool LND = false /* B0 lies in global memory (a kernel that takes `const Batch *`): every phase works through its own laundered reference to it and to the descriptor (srba_device.hpp lnd)
          	*/>
__device__ __forceinline__ void lm_one(const Batch &B0, const DevParams &prm, const int pidx, double *red = nullptr /* G = 128: LDS scratch of the group reductions */) {
	const ProbDesc &d = B0.desc[pidx];
	// the batch record for this function's own few accesses: a reference of its own at every use (LND), like the phases (srba_device.hpp lnd)
	auto LB = [&]() __attribute__((always_inline)) -> const Batch & { if constexpr (LND) return lnd(B0); else return B0; };
	int cur = 0, last_rej = 0; bool lazy_rej = false; // DB: which copy holds the accepted state; the last evaluated trial was rejected (lazy_rej: ... and the pose table of the trial copy was not
		// refreshed for it)
	typedef Solver<FAM, LEAN, G> Sv;
	// a worker on copy `cp` of the state for ONE phase call (LND: through references nothing else shares -- the pointers a phase uses are loaded when it starts and die when it ends)
	auto Wk = [&](int cp) __attribute__((always_inline)) -> Sv { if constexpr (LND) return Sv(lnd(B0), lnd(d), lnd(prm), red, cp); else return Sv(B0, d, prm, red, cp); };
	Solver<FAM, LEAN, G> S(B0, d, prm, red); // (for the questions that touch no array: schur_active)
	constexpr int P = Solver<FAM, LEAN, G>::P, L = Solver<FAM, LEAN, G>::L, O = Solver<FAM, LEAN, G>::O;
	const SparseSys A = Wk(0).make_sys(srba_lds);
	const int tid = threadIdx.x; srba_lm_result *out = LB().results + pidx;
	const int nObs = d.n_obs, n = d.n_scal;
	double *resid, *resid2; { const Batch &Bl = LB(); resid = Bl.resid; resid2 = Bl.resid2; }

	long long *pc; { const Batch &Bl = LB(); pc = Bl.phase_cycles ? Bl.phase_cycles + (long long)pidx * 16 : nullptr; } long long tc0 = 0;
#define TIC() do { if (pc) { __syncthreads(); tc0 = wall_clock64(); } } while (0)
#define TOC(slot) do { if (pc) { __syncthreads(); if (tid == 0) pc[slot] += wall_clock64() - tc0; } } while (0)
	// K6: the LDS image of the system is idle while the capsule is linearised (it is assembled per trial): the U_Ap accumulators of the term-parallel form live there
	const bool hess_terms = G <= 128 && LB().hess_terms && d.dense_in_lds && d.n_hap * P * P <= 9 * (d.nb + d.nnzoff); // (the term-parallel form cuts the list between TWO wavefronts;
		// workgroup windows keep no system in LDS anyway)
	bool hs_reduced = false; // workgroup path, U_Ap in LDS: the LDS blocks hold the Schur-reduced system of the last solve (what the reference leaves in HAp), not yet written back
	auto hessian = [&](int cp) __attribute__((always_inline)) -> int { Sv X = Wk(cp); if constexpr (G > 64 && !Tr<FAM>::REL) { hs_reduced = false; return X.phase_hessian_lds(); }
		else return hess_terms ? X.phase_hessian_terms(srba_lds) /* == A.diag: the LDS image, idle while the capsule is linearised */ + X.phase_hessian_landmark_blocks() : X.phase_hessian(); };
	double lambda, nu = 2.0, total_err, RMSE;
	int iter = 0, trials = 0, n_notpd = 0, n_acc = 0, n_relin = 0, stopmask = 0; bool stop = false;
	TIC(); Wk(0).phase_spantree(false, nullptr, DB ? (double *)LB().pose1 : nullptr); // S5 (DB: both copies of the poses)
	if constexpr (DB) { constexpr int PD = Solver<FAM, LEAN, G>::PD; const Batch &Bl = LB(); // the second copy of the unknowns (the fixed edges of the paths stay equal in both for good)
		for (int k = tid; k < d.n_edges * PD; k += G) Bl.edge1[d.o_edge * PD + k] = Bl.edge[d.o_edge * PD + k];
		for (int k = tid; k < d.nF * L; k += G) Bl.ulm1[d.o_ulm * L + k] = Bl.ulm[d.o_ulm * L + k]; }
	__syncthreads(); TOC(0);
	TIC(); Wk(0).phase_jacobians(); TOC(1); // S6,S7
	TIC(); const int ninv = (int)grp_sum<G>((double)hessian(0), red); // S10
	__syncthreads(); TOC(2);
	if (tid == 0) {
		out->status = 0; out->num_iters = 0; out->num_trials = 0; out->num_not_pd = 0; out->num_accepted = 0; out->num_relinearized = 0; out->stop_reason = 0;
		out->num_invalid_jacobs = ninv; out->num_observations = nObs; out->num_jacobians = d.n_bp + d.n_bf; out->num_span_tree_numeric_updates = d.n_pairs;
		for (int k = 0; k < SRBA_TRACE_LEN; k++) { out->trace_chi2[k] = NAN; out->trace_lambda[k] = NAN; out->trace_rho[k] = NAN; }
		out->lambda_last_trial = NAN;
	}
	if ((long long)O * nObs < (long long)n) { if (tid == 0) out->status = 1; return; } // S11
	lambda = Wk(0).lambda_guess(red); // S12
	TIC(); total_err = Wk(0).phase_residuals(resid, red); TOC(3); // S13
	RMSE = sqrt(total_err / nObs);
	if (tid == 0) { out->lambda_init = lambda; out->total_sqr_error_init = total_err; }
	__syncthreads();
	TIC(); Wk(0).phase_gradient(resid); // S14
	__syncthreads(); Wk(0).keep_gradient(); TOC(4);
	for (; iter < prm.max_iters && !stop; iter++) {
		double rho = 0;
		if (lambda >= prm.max_lambda) { stop = true; stopmask |= 1 << SRBA_STOP_LAMBDA; }
		if (RMSE < prm.max_err) { stop = true; stopmask |= 1 << SRBA_STOP_RMSE; }
		while (rho <= 0 && !stop) {
			const int tr = trials++;
			if (tid == 0) { if (tr < SRBA_TRACE_LEN) out->trace_lambda[tr] = lambda; out->lambda_last_trial = lambda; }
			const int ca = DB ? cur : 0, ct = DB ? (cur ^ 1) : 0; // accepted / trial copy of the state (the same one without DB)
			TIC(); const bool solved = Wk(ca).solve(A, lambda, pc); TOC(5); hs_reduced = true;
			if (!solved) {
				n_notpd++; lambda *= nu; nu *= 2.0; stop = (lambda > prm.max_lambda); if (stop) stopmask |= 1 << SRBA_STOP_LAMBDA;
				__syncthreads();
				continue;
			}
			TIC(); const double *edge_lds = DB ? Wk(ca).apply_trial(A) : Wk(ca).apply_update_lds(A); TOC(6);
			// (DB, edges staged, every refreshed path of four edges or fewer: the residuals compose their poses themselves and the refresh of the pose table waits for an accepted trial)
			const bool lazy = DB && FUSE_K4 && edge_lds != nullptr && d.need_flat != 0;
			double new_err;
			if (lazy) { TIC(); new_err = Wk(ct).phase_residuals_fused(resid2, red, edge_lds); TOC(3); }
			else { TIC(); Wk(ct).phase_spantree(true, edge_lds);
				__syncthreads(); TOC(7);
				TIC(); new_err = Wk(ct).phase_residuals(resid2, red); TOC(3); }
			const double new_RMSE = sqrt(new_err / nObs);
			const double err_red = total_err > 0 ? (total_err - new_err) / total_err : 0;
			double den = 0; { const Batch &Bl = LB(); const double *dl = Bl.delta + d.o_scal, *g = Bl.grad + d.o_scal;
				if (S.schur_active() || !d.dense_in_lds) { for (int k = tid; k < n; k += G) den += dl[k] * (lambda * dl[k] + g[k]); }
				else for (int k = tid; k < n; k += G) { const double dk = A.sol(k); den += dk * (lambda * dk + g[k]); } } // (the solved right-hand side is still in the LDS image: the same numbers,
					// no round trip through memory)
			den = grp_sum<G>(den, red);
			rho = (total_err - new_err) / den;
			if (tid == 0 && tr < SRBA_TRACE_LEN) { out->trace_chi2[tr] = new_err; out->trace_rho[tr] = rho; }
			if (rho > 0) {
				n_acc++;
				const bool relin = (err_red < 0 || err_red > prm.min_relin);
				{ double *t = resid; resid = resid2; resid2 = t; }
				total_err = new_err; RMSE = new_RMSE;
				if constexpr (DB) { cur ^= 1; last_rej = 0; } // the trial copy is the accepted one from here on
				__syncthreads();
				if (lazy) { TIC(); Wk(ct).phase_spantree(true, edge_lds); __syncthreads(); TOC(7); } // the poses of the accepted trial, for the Jacobians and for the output
				if (relin) { n_relin++; TIC(); Wk(ct).phase_jacobians(); TOC(1); TIC(); hessian(ct); __syncthreads(); TOC(2); }
				TIC(); Wk(ct).phase_gradient(resid);
				__syncthreads(); Wk(ct).keep_gradient(); TOC(4);
				double ninf = 0; { const double *g = LB().grad + d.o_scal; for (int k = tid; k < n; k += G) ninf = fmax(ninf, fabs(g[k])); }
				ninf = grp_max<G>(ninf, red);
				if (ninf <= 1e-15) { stop = true; stopmask |= 1 << SRBA_STOP_GRADIENT; }
				if (RMSE < prm.max_err) { stop = true; stopmask |= 1 << SRBA_STOP_RMSE; }
				if (rho > prm.max_rho) { stop = true; stopmask |= 1 << SRBA_STOP_RHO; }
				lambda *= 1.0 / 3.0; nu = 2.0;
			} else {
				if constexpr (DB) { last_rej = 1; lazy_rej = lazy; } else { TIC(); Wk(ca).restore(); TOC(8); }
				lambda *= nu; nu *= 2.0; stop = (lambda > prm.max_lambda); if (stop) stopmask |= 1 << SRBA_STOP_LAMBDA;
			}
		}
	}
	if (!stop) stopmask |= 1 << SRBA_STOP_MAX_ITERS;
	if constexpr (G > 64 && !Tr<FAM>::REL) { if (hs_reduced && d.n_panel == 1 && S.schur_active()) { __syncthreads(); Wk(0).store_hs(false); } }
		// (the reference's Schur complement works on HAp in place: its reduced blocks are what a caller reads after the run)
	// S17: crpLandmarksApprox
	if constexpr (!Solver<FAM, LEAN, G>::W::T::REL) {
		const Batch &Bl = LB();
		for (int l = tid; l < d.nF; l += G) {
			const bool ok = prm.cov_recovery == 1 && (S.schur_active() ? (Bl.hf_ok[d.o_ulm + l] != 0) : true);
			Bl.ulm_inf_valid[d.o_ulm + l] = ok ? 1 : 0;
			if (ok) for (int k = 0; k < L * L; k++) Bl.ulm_inf[(d.o_ulm + l) * L * L + k] = Bl.Hf[(d.o_hf + Bl.hf_diag[d.o_ulm + l]) * L * L + k];
		}
	}
	if (tid == 0) {
		out->num_iters = iter; out->num_trials = trials; out->num_not_pd = n_notpd; out->num_accepted = n_acc; out->num_relinearized = n_relin; out->stop_reason = stopmask;
		out->total_sqr_error_final = total_err; out->obs_rmse = RMSE; out->lambda_final = lambda;
	}
	if constexpr (DB) { // the reference's partial restore, where it becomes visible: a rejected trial refreshed BOTH poses of every pair in use and only the ones Jacobian blocks read went back
		// (optimize_edges.h:664-670)
		constexpr int PD = Solver<FAM, LEAN, G>::PD; const Batch &Bl = LB(); double *pose_a = cur ? Bl.pose1 : Bl.pose, *pose_t = cur ? Bl.pose : Bl.pose1; // accepted / trial copy of the pose table
		__syncthreads();
		if (last_rej && lazy_rej) { Wk(cur ^ 1).phase_spantree(true, nullptr); __syncthreads(); } // (the last rejected trial's poses, from its edges in the trial copy)
		if (last_rej) for (int q = tid; q < 2 * d.n_need; q += G) {
			const long long ps = 2LL * Bl.need_idx[d.o_pair + (q >> 1)] + (q & 1);
			if (!Bl.pose_req[d.o_pair * 2 + ps]) { double v[PD]; ldn<PD>(v, pose_t + (d.o_pair * 2 + ps) * PD); stn<PD>(pose_a + (d.o_pair * 2 + ps) * PD, v); }
		}
	}
	if constexpr (DB) { // the accepted state goes back to the primary arrays
		constexpr int PD = Solver<FAM, LEAN, G>::PD; const Batch &Bl = LB();
		__syncthreads();
		if (cur) {
			for (int k = tid; k < d.nK * PD; k += G) Bl.edge[d.o_edge * PD + k] = Bl.edge1[d.o_edge * PD + k];
			for (int k = tid; k < d.nF * L; k += G) Bl.ulm[d.o_ulm * L + k] = Bl.ulm1[d.o_ulm * L + k];
			for (int q = tid; q < 2 * d.n_need; q += G) { const long long ps = 2LL * Bl.need_idx[d.o_pair + (q >> 1)] + (q & 1); double v[PD]; ldn<PD>(v, Bl.pose1 + (d.o_pair * 2 + ps) * PD);
				stn<PD>(Bl.pose + (d.o_pair * 2 + ps) * PD, v); }
		}
	}
	(void)P;
}
// The LM loop of ONE capsule as one of sc->W replicas that speculate on the lambda ladder (SpecCtl, srba_device.hpp; k_lm_spec below): lm_one<FAM, true, true, 128> with the trial
// evaluation replaced by "take the outcome of step sp_j of the current round". The control flow (stop rules, counters, traces, the partial-restore quirk) is lm_one's, statement by
// statement; kept as a separate function so that the register allocation of the batch kernels does not depend on it. Relative-pose families (no landmark unknowns, no Schur complement).
template <int FAM>
__device__ __forceinline__ void lm_spec(const Batch &B0, const DevParams &prm, const int pidx, double *red, const SpecCtl *sc) {
	constexpr bool LEAN = true, FUSE_K4 = (SRBA_FUSE_K4 != 0); constexpr int G = 2 * SRBA_WG;
	static_assert(Tr<FAM>::REL, "speculation: no landmark unknowns");
	const ProbDesc &d = B0.desc[pidx];
	const Batch &B = B0;
	int sp_j = sc->W, sp_round = sc->round0, rej_owner = 0, rej_round = 0; const double *own_el = nullptr; // SPEC: next outcome of the round to take (W: none left), rounds so far,
		// who evaluated the last rejected trial and in which round, this replica's staged trial edges
	int cur = 0, last_rej = 0; bool lazy_rej = false; // DB: which copy holds the accepted state; the last evaluated trial was rejected (lazy_rej: ... and the pose table of the trial copy was not
		// refreshed for it)
	Solver<FAM, LEAN, G> S(B, d, prm, red);
	constexpr int P = Solver<FAM, LEAN, G>::P, L = Solver<FAM, LEAN, G>::L, O = Solver<FAM, LEAN, G>::O;
	const SparseSys A = S.make_sys(srba_lds);
	const int tid = threadIdx.x; srba_lm_result *out = B.results + pidx;
	const int nObs = d.n_obs, n = d.n_scal;
	double *resid = B.resid, *resid2 = B.resid2;

	long long *pc = B.phase_cycles ? B.phase_cycles + (long long)pidx * 16 : nullptr; long long tc0 = 0;
#define TIC() do { if (pc) { __syncthreads(); tc0 = wall_clock64(); } } while (0)
#define TOC(slot) do { if (pc) { __syncthreads(); if (tid == 0) pc[slot] += wall_clock64() - tc0; } } while (0)
	// K6: the LDS image of the system is idle while the capsule is linearised (it is assembled per trial): the U_Ap accumulators of the term-parallel form live there
	const bool hess_terms = B.hess_terms && d.dense_in_lds && d.n_hap * P * P <= 9 * (d.nb + d.nnzoff);
	auto hessian = [&](Solver<FAM, LEAN, G> &X) -> int { return hess_terms ? X.phase_hessian_terms(srba_lds) /* == A.diag: the LDS image,
		idle while the capsule is linearised */ + X.phase_hessian_landmark_blocks() : X.phase_hessian(); };
	double lambda, nu = 2.0, total_err, RMSE;
	int iter = 0, trials = 0, n_notpd = 0, n_acc = 0, n_relin = 0, stopmask = 0; bool stop = false;
	TIC(); S.phase_spantree(false, nullptr, B0.pose1); // S5 (both copies of the poses)
	{ constexpr int PD = Solver<FAM, LEAN, G>::PD; // the second copy of the unknowns (the fixed edges of the paths stay equal in both for good)
		for (int k = tid; k < d.n_edges * PD; k += G) B0.edge1[d.o_edge * PD + k] = B0.edge[d.o_edge * PD + k]; }
	__syncthreads(); TOC(0);
	TIC(); S.phase_jacobians(); TOC(1); // S6,S7
	TIC(); const int ninv = (int)grp_sum<G>((double)hessian(S), red); // S10
	__syncthreads(); TOC(2);
	if (tid == 0) {
		out->status = 0; out->num_iters = 0; out->num_trials = 0; out->num_not_pd = 0; out->num_accepted = 0; out->num_relinearized = 0; out->stop_reason = 0;
		out->num_invalid_jacobs = ninv; out->num_observations = nObs; out->num_jacobians = d.n_bp + d.n_bf; out->num_span_tree_numeric_updates = d.n_pairs;
		for (int k = 0; k < SRBA_TRACE_LEN; k++) { out->trace_chi2[k] = NAN; out->trace_lambda[k] = NAN; out->trace_rho[k] = NAN; }
		out->lambda_last_trial = NAN;
	}
	if ((long long)O * nObs < (long long)n) { if (tid == 0) out->status = 1; return; } // S11
	lambda = S.lambda_guess(red); // S12
	TIC(); total_err = S.phase_residuals(resid, red); TOC(3); // S13
	RMSE = sqrt(total_err / nObs);
	if (tid == 0) { out->lambda_init = lambda; out->total_sqr_error_init = total_err; }
	__syncthreads();
	TIC(); S.phase_gradient(resid); // S14
	__syncthreads(); S.keep_gradient(); TOC(4);
	for (; iter < prm.max_iters && !stop; iter++) {
		double rho = 0;
		if (lambda >= prm.max_lambda) { stop = true; stopmask |= 1 << SRBA_STOP_LAMBDA; }
		if (RMSE < prm.max_err) { stop = true; stopmask |= 1 << SRBA_STOP_RMSE; }
		while (rho <= 0 && !stop) {
			const int tr = trials++;
			if (tid == 0) { if (tr < SRBA_TRACE_LEN) out->trace_lambda[tr] = lambda; out->lambda_last_trial = lambda; }
			const Batch Bt = copy_view(B0, cur ^ 1); Solver<FAM, LEAN, G> Sa(B0, d, prm, red, cur), St(B0, d, prm, red, cur ^ 1); // accepted / trial copy of the state
			bool lazy; const double *edge_lds = nullptr; double new_err, new_RMSE, err_red;
			{
				lazy = FUSE_K4 && d.need_flat != 0 && d.n_edges * Solver<FAM, LEAN, G>::PD <= 9 * d.nnzoff; // (dense_in_lds is a condition of the launch)
				if (sp_j == sc->W) { // no outcome left of the last round (or the state has changed since): all replicas evaluate their step of the ladder that starts at (lambda, nu)
					if (last_rej && rej_round == sp_round) { const double *src = sc->xdelta + ((rej_round & 1) * sc->W + rej_owner) * sc->xstride; double *keep = B.old_edge + d.o_unk * Solver<FAM,
						LEAN, G>::PD; // (that round's increments are overwritten two rounds from now: keep the last rejected one; old_edge is idle in the double-buffered loop)
						for (int k = tid; k < n; k += G) keep[k] = spec_ld(src + k); }
					sp_round++;
					double lam_w = lambda, nu_w = nu; for (int q = 0; q < sc->w; q++) { lam_w *= nu_w; nu_w *= 2.0; }
					TIC(); const bool ok_w = Sa.solve(A, lam_w, pc); TOC(5);
					double rho_w = NAN, err_w = NAN;
					if (ok_w) {
						double *xd = sc->xdelta + ((sp_round & 1) * sc->W + sc->w) * sc->xstride;
						for (int k = tid; k < n; k += G) xd[k] = A.sol(k);
						TIC(); own_el = Sa.apply_trial(A); TOC(6);
						TIC(); if (lazy) err_w = St.phase_residuals_fused(resid2, red, own_el); else { St.phase_spantree(true, own_el); __syncthreads(); err_w = St.phase_residuals(resid2, red); }
							TOC(3);
						double den = 0; { const double *g = B.grad + d.o_scal; for (int k = tid; k < n; k += G) { const double dk = A.sol(k); den += dk * (lam_w * dk + g[k]); } }
						den = grp_sum<G>(den, red);
						rho_w = (total_err - err_w) / den;
					}
					TIC(); const bool all_there = spec_exchange(*sc, sp_round, ok_w ? 2 : 1, rho_w, err_w, lam_w); TOC(14);
					if (!all_there) { if (tid == 0) out->status = 2; stop = true; break; } // a replica is not resident: the host re-runs the capsule on the sequential path (spec_fallback)
					sp_j = 0;
				}
				const double *b = sc->box + ((sp_round & 1) * sc->W + sp_j) * 4; const int code = (int)spec_ld(b);
				if ((code != 1 && code != 2) || spec_ld(b + 3) != lambda) { if (tid == 0) out->status = 3; stop = true; break; } // every replica answered (spec_exchange said so) and yet the
					// box holds no outcome, or one for another lambda: the protocol itself lost step -- status 3, which the host reports as an error (status 2, a replica that was not
					// resident in time, is the only case it repairs by re-running the capsule on the sequential path)
				if (code != 2) { // that step of the ladder was not positive definite
					n_notpd++; lambda *= nu; nu *= 2.0; stop = (lambda > prm.max_lambda); if (stop) stopmask |= 1 << SRBA_STOP_LAMBDA;
					sp_j++;
					__syncthreads();
					continue;
				}
				rho = spec_ld(b + 1); new_err = spec_ld(b + 2);
				new_RMSE = sqrt(new_err / nObs);
				err_red = total_err > 0 ? (total_err - new_err) / total_err : 0;
			}
			if (tid == 0 && tr < SRBA_TRACE_LEN) { out->trace_chi2[tr] = new_err; out->trace_rho[tr] = rho; }
			if (rho > 0) {
				n_acc++;
				{ // the accepted trial is replica sp_j's: the others re-apply its increment to their copy of the accepted state (same numbers, same arithmetic)
					if (sp_j != sc->w && lazy) { // ... or, where a trial leaves nothing but its edges and its residuals (lazy), copy those from the winner's arena (written before its exchange,
						// untouched until two accepted trials from now)
						constexpr int PD = Solver<FAM, LEAN, G>::PD; typedef typename Solver<FAM, LEAN, G>::PO PO; const long long off = sc->stride * (sp_j - sc->w);
						const double *we = (const double *)((const char *)Bt.edge + off) + d.o_edge * PD, *wr = (const double *)((const char *)resid2 + off) + (long long)d.o_obs * O;
							double *el = A.off;
						TIC(); __syncthreads();
						for (int i = tid; i < d.n_edges; i += G) { const typename Solver<FAM, LEAN, G>::pose_t e = PO::ld(we + i * PD); if (i < d.nK) PO::st(Bt.edge + (d.o_edge + i) * PD, e);
							double t[PD]; PO::to(t, e);
#pragma unroll
							for (int k = 0; k < PD; k++) el[i * PD + k] = t[k]; }
						{ double *mr = resid2 + (long long)d.o_obs * O; for (int k = tid; k < nObs * O; k += G) mr[k] = wr[k]; }
						__syncthreads(); edge_lds = el; TOC(6);
					} else if (sp_j != sc->w) { const double *src = sc->xdelta + ((sp_round & 1) * sc->W + sp_j) * sc->xstride; double *dl = B.delta + d.o_scal;
						__syncthreads();
						for (int k = tid; k < n; k += G) { const double v = spec_ld(src + k); A.rhs[3 * A.perm[k / 3] + k % 3] = v; dl[k] = v; }
						__syncthreads();
						TIC(); edge_lds = Sa.apply_trial(A); TOC(6);
						TIC(); St.phase_spantree(true, edge_lds); __syncthreads(); (void)St.phase_residuals(resid2, red); TOC(3);
					} else edge_lds = own_el;
					sp_j = sc->W; // the state changes: the other outcomes of the round are void
				}
				const bool relin = (err_red < 0 || err_red > prm.min_relin);
				{ double *t = resid; resid = resid2; resid2 = t; }
				total_err = new_err; RMSE = new_RMSE;
				cur ^= 1; last_rej = 0; // the trial copy is the accepted one from here on
				__syncthreads();
				if (lazy) { TIC(); St.phase_spantree(true, edge_lds); __syncthreads(); TOC(7); } // the poses of the accepted trial, for the Jacobians and for the output
				if (relin) { n_relin++; TIC(); St.phase_jacobians(); TOC(1); TIC(); hessian(St); __syncthreads(); TOC(2); }
				TIC(); St.phase_gradient(resid);
				__syncthreads(); St.keep_gradient(); TOC(4);
				double ninf = 0; { const double *g = B.grad + d.o_scal; for (int k = tid; k < n; k += G) ninf = fmax(ninf, fabs(g[k])); }
				ninf = grp_max<G>(ninf, red);
				if (ninf <= 1e-15) { stop = true; stopmask |= 1 << SRBA_STOP_GRADIENT; }
				if (RMSE < prm.max_err) { stop = true; stopmask |= 1 << SRBA_STOP_RMSE; }
				if (rho > prm.max_rho) { stop = true; stopmask |= 1 << SRBA_STOP_RHO; }
				lambda *= 1.0 / 3.0; nu = 2.0;
			} else {
				last_rej = 1; lazy_rej = lazy; rej_owner = sp_j; rej_round = sp_round; sp_j++;
				lambda *= nu; nu *= 2.0; stop = (lambda > prm.max_lambda); if (stop) stopmask |= 1 << SRBA_STOP_LAMBDA;
			}
		}
	}
	if (!stop) stopmask |= 1 << SRBA_STOP_MAX_ITERS;
	if (tid == 0) {
		out->num_iters = iter; out->num_trials = trials; out->num_not_pd = n_notpd; out->num_accepted = n_acc; out->num_relinearized = n_relin; out->stop_reason = stopmask;
		out->total_sqr_error_final = total_err; out->obs_rmse = RMSE; out->lambda_final = lambda;
	}
	{ // the reference's partial restore, where it becomes visible: a rejected trial refreshed BOTH poses of every pair in use and only the ones Jacobian blocks read went back
		// (optimize_edges.h:664-670)
		constexpr int PD = Solver<FAM, LEAN, G>::PD; const Batch Ba = copy_view(B0, cur), Bt = copy_view(B0, cur ^ 1);
		__syncthreads();
		if (last_rej && !(rej_round == sp_round && rej_owner == sc->w)) { // the trial copy of this replica holds ITS last trial: make it the sequentially last evaluated one (the quirk below shows it)
			const double *src = rej_round == sp_round ? sc->xdelta + ((rej_round & 1) * sc->W + rej_owner) * sc->xstride : B.old_edge + d.o_unk * PD;
			for (int k = tid; k < n; k += G) A.rhs[3 * A.perm[k / 3] + k % 3] = rej_round == sp_round ? spec_ld(src + k) : src[k];
			__syncthreads();
			Solver<FAM, LEAN, G> Sa(B0, d, prm, red, cur), Sl(B0, d, prm, red, cur ^ 1); const double *el = Sa.apply_trial(A);
			if (!lazy_rej) Sl.phase_spantree(true, el);
			__syncthreads();
		}
		if (last_rej && lazy_rej) { Solver<FAM, LEAN, G> Sl(B0, d, prm, red, cur ^ 1); Sl.phase_spantree(true, nullptr); __syncthreads(); } // (the last rejected trial's poses,
			// from its edges in the trial copy)
		if (last_rej) for (int q = tid; q < 2 * d.n_need; q += G) {
			const long long ps = 2LL * B0.need_idx[d.o_pair + (q >> 1)] + (q & 1);
			if (!B0.pose_req[d.o_pair * 2 + ps]) { double v[PD]; ldn<PD>(v, Bt.pose + (d.o_pair * 2 + ps) * PD); stn<PD>(Ba.pose + (d.o_pair * 2 + ps) * PD, v); }
		}
	}
	{ // the accepted state goes back to the primary arrays
		constexpr int PD = Solver<FAM, LEAN, G>::PD;
		__syncthreads();
		if (cur) {
			for (int k = tid; k < d.nK * PD; k += G) B0.edge[d.o_edge * PD + k] = B0.edge1[d.o_edge * PD + k];
			for (int q = tid; q < 2 * d.n_need; q += G) { const long long ps = 2LL * B0.need_idx[d.o_pair + (q >> 1)] + (q & 1); double v[PD]; ldn<PD>(v, B0.pose1 + (d.o_pair * 2 + ps) * PD);
				stn<PD>(B0.pose + (d.o_pair * 2 + ps) * PD, v); }
		}
	}
	(void)P; (void)L;
}
// Persistent workgroups: a launch covers one LDS size class with a grid of at most the number of wavefronts the chip can hold for that class; every
// wavefront pulls capsules (sorted longest-first inside the class) from a shared counter until the class is exhausted. A launch therefore has ONE
// tail (its last capsules) instead of one per chunk, and the chip stays full while big (LDS-bound) and small (wave-slot-bound) classes drain side by side.
#ifndef SRBA_LM_BY_POINTER
// the fused LM kernels take `const Batch *` (a device copy of the batch record, uploaded with the input arena) and every phase of lm_one works through its own laundered reference
// (srba_device.hpp lnd): the batch's ~100 array pointers are scalar loads at the start of the phase that uses them instead of 220 - 530 spilled scalar registers. 0: the record as a kernel
// argument (rounds 1 - 4)
#define SRBA_LM_BY_POINTER 1
#endif
#if SRBA_LM_BY_POINTER
#define SRBA_LM_BATCH_ARG const Batch *__restrict__ Bptr, const DevParams *__restrict__ Pptr
#define SRBA_LM_BATCH_REF const Batch &B = *Bptr; const DevParams &prm = *Pptr
#define SRBA_LM_BATCH_VAL(c) (const srbadev::Batch *)(c)->d_batch, (const srbadev::DevParams *)((c)->d_batch + 1)
#define SRBA_LM_LND true
#else
#define SRBA_LM_BATCH_ARG const Batch B, const DevParams prm
#define SRBA_LM_BATCH_REF
#define SRBA_LM_BATCH_VAL(c) (c)->B, (c)->dp
#define SRBA_LM_LND false
#endif
template <int FAM>
__global__ void __launch_bounds__(SRBA_WG) SRBA_OCC k_lm_run(SRBA_LM_BATCH_ARG, int first, int count, int *next) { SRBA_LM_BATCH_REF;
	for (;;) {
		int i = 0; if (threadIdx.x == 0) { i = atomicAdd(next, 1); if (i == 0) *(long long *)(next + 2) = wall_clock64(); /* when the class started (srba_hip_launch_order): record = {counter, pad,
			stamp} */ }
		i = __builtin_amdgcn_readfirstlane(i);
		if (i >= count) break;
		lm_one<FAM, (SRBA_LM_DB != 0), false, 64, false, SRBA_LM_LND>(B, prm, B.order[first + i]);
		__syncthreads(); // the LDS image and the symbolic copy are rebuilt by the next capsule
	}
}

// holds a class stream back for a while before its persistent launch (staggered start of the class launches, plan_launches)
__global__ void k_delay(int us) { const long long t0 = wall_clock64(); while (wall_clock64() - t0 < 100LL * us) __builtin_amdgcn_s_sleep(20); }
// The same loop for the small size classes of a big batch: three wavefronts per SIMD (at most 168 registers; Worker<FAM, LEAN = true> keeps fewer loads in flight per lane). On the 23 460 windows
// with at most 31 unknown edges of the benchmark batch: 16.6 ms against 19.0 ms for k_lm_run, whose two wavefronts per SIMD leave 40 % of the LDS of a CU unused while those classes run;
// the big (LDS-bound) classes are 3 % slower with it and keep k_lm_run. Only instantiated where the plan uses it (relative-pose SE2: plan_launches).
#ifndef SRBA_LEAN_WAVES
#define SRBA_LEAN_WAVES 3 /* wavefronts per SIMD k_lm_run_lean is compiled for (4: 128 registers) */
#endif
template <int FAM>
__global__ void __launch_bounds__(SRBA_WG) __attribute__((amdgpu_waves_per_eu(SRBA_LEAN_WAVES, SRBA_LEAN_WAVES))) k_lm_run_lean(SRBA_LM_BATCH_ARG, int first, int count, int *next) { SRBA_LM_BATCH_REF;
	for (;;) {
		int i = 0; if (threadIdx.x == 0) { i = atomicAdd(next, 1); if (i == 0) *(long long *)(next + 2) = wall_clock64(); }
		i = __builtin_amdgcn_readfirstlane(i);
		if (i >= count) break;
		lm_one<FAM, (SRBA_LM_DB != 0), true, 64, (SRBA_FUSE_K4 != 0), SRBA_LM_LND>(B, prm, B.order[first + i]);
		__syncthreads();
	}
}

// Two wavefronts per capsule for the big, LDS-bound windows of a big batch (round 4). Those windows hold 20 .. 43 KB of LDS each, so 4 .. 7 of them fill a CU and their wavefronts sit (nearly) alone
// on a SIMD; the launch then waits for the latency of their trials (tools/diag_residency.py: dead LDS that removes a resident window costs its share of the throughput), nearly half of which
// is spent in the lane-parallel phases (spanning-tree refresh, residuals, Jacobians, Hessian terms, gradient, assembly of the LDS image: 71 of 154 us per trial for the 67-edge windows) at one
// record gather per lane and pass. With a second wavefront on the same LDS image those phases run twice as wide; the block-sparse factorisation and substitution stay on the first wavefront.
// Worker / Solver<FAM, LEAN, 128>: strides of 128, workgroup barriers, group reductions through two doubles of LDS behind the image (fixed order: deterministic), the U_Ap terms cut between two
// Hessian blocks. Three wavefronts per SIMD (the LEAN register diet): six such workgroups per CU.
template <int FAM>
__global__ void __launch_bounds__(2 * SRBA_WG) __attribute__((amdgpu_waves_per_eu(3, 3))) k_lm_run2(SRBA_LM_BATCH_ARG, int first, int count, int *next, int lds_doubles) { SRBA_LM_BATCH_REF;
	double *red = srba_lds + lds_doubles; int *slot = (int *)(red + 2);
	for (;;) {
		if (threadIdx.x == 0) { const int i0 = atomicAdd(next, 1); *slot = i0; if (i0 == 0) *(long long *)(next + 2) = wall_clock64(); }
		__syncthreads(); const int i = *slot; __syncthreads();
		if (i >= count) break;
		lm_one<FAM, (SRBA_LM_DB != 0), true, 2 * SRBA_WG, (SRBA_FUSE_K4 != 0), SRBA_LM_LND>(B, prm, B.order[first + i], red);
		__syncthreads();
	}
}

#ifndef SRBA_WG_WAVES
#define SRBA_WG_WAVES 2 /* wavefronts per SIMD the workgroup kernels are compiled for (256 registers) */
#endif
#define SRBA_WG_TOP (SRBA_WG_WAVES >= 3 ? 768 : 512) /* threads of the one-per-CU workgroup class: every wave slot of the CU */
#ifndef SRBA_WG_BY_POINTER
// the workgroup kernels read the ~100 array pointers of the batch from a device copy of `Batch` (scalar loads where a pointer is used) instead of keeping them all in scalar registers from
// the kernel arguments: their phases last tens of microseconds, a scalar load is nothing there, and 450 - 530 spilled scalars (v_readlane restores: a quarter of the instructions of the
// Hessian and Schur loops) are
#define SRBA_WG_BY_POINTER 1
#endif
#if SRBA_WG_BY_POINTER
#define SRBA_WG_BATCH_ARG const Batch *__restrict__ Bptr, const DevParams *__restrict__ Pptr
#define SRBA_WG_BATCH_REF const Batch &B = *Bptr; const DevParams &prm = *Pptr
#define SRBA_WG_BATCH_VAL(c) (const srbadev::Batch *)(c)->d_batch, (const srbadev::DevParams *)((c)->d_batch + 1)
#else
#define SRBA_WG_BATCH_ARG const Batch B, const DevParams prm
#define SRBA_WG_BATCH_REF
#define SRBA_WG_BATCH_VAL(c) (c)->B, (c)->dp
#endif
// One WORKGROUP of G = 128 or 256 threads per capsule for the SE3 landmark families (round 5; srba_wg.hpp): every lane-parallel phase runs G wide, the Schur-reduced system is a
// lower triangle of 16 x 16 tiles in the capsule's HBM workspace and is factored on the matrix cores by the G / 64 wavefronts. Replaces, for the windows the plan sends here, the
// one-wavefront k_lm_run<3..6> (504 - 512 registers, one wavefront per SIMD, the 3x3-block sweeps of a 40 - 59-edge stereo window 6.5 of the 15.5 ms of a trial). Registers: the
// LEAN diet (one Schur term, one Hessian term, one spanning-tree pair in flight) under a cap of 256 -- two wavefronts per SIMD, i.e. two 256-thread workgroups per CU.
// LDS: WG_LDS_DOUBLES (the solver's staging, x, reduction scratch, flags, the work counter's slot).
template <int FAM, int G>
__global__ void __launch_bounds__(G) __attribute__((amdgpu_waves_per_eu(SRBA_WG_WAVES, SRBA_WG_WAVES))) k_lm_wg(SRBA_WG_BATCH_ARG, int first, int count, int *next) {
	SRBA_WG_BATCH_REF;
	double *red = srba_lds + WG_RED; int *slot = (int *)(red + 13); // (red[0 .. 11]: group reductions, red + 12: the solver's flag)
	for (;;) {
		if (threadIdx.x == 0) { const int i0 = atomicAdd(next, 1); *slot = i0; if (i0 == 0) *(long long *)(next + 2) = wall_clock64(); }
		__syncthreads(); const int i = *slot; __syncthreads();
		if (i >= count) break;
		lm_one<FAM, (SRBA_LM_DB != 0), true, G, false, (SRBA_WG_BY_POINTER != 0)>(B, prm, B.order[first + i], red);
		__syncthreads();
	}
}
template <int FAM, int G> __global__ void __launch_bounds__(G) __attribute__((amdgpu_waves_per_eu(SRBA_WG_WAVES, SRBA_WG_WAVES))) k_solve_wg(const Batch B, const DevParams prm, int first) {
	// (G = 768 only with SRBA_WG_WAVES = 3)
	const int pidx = B.order[first + blockIdx.x]; const ProbDesc &d = B.desc[pidx]; Solver<FAM, true, G> S(B, d, prm, srba_lds + WG_RED);
	const SparseSys A = S.make_sys(srba_lds);
	const bool ok = S.solve(A, B.lambda_io[pidx]);
	if (d.n_panel == 1 && S.schur_active()) { __syncthreads(); S.store_hs(false); }
	if (threadIdx.x == 0) B.notpd[pidx] = ok ? 0 : 1;
}

// A batch of ONE capsule (the per-key-frame use of the engine): W replicas of it, one workgroup of two wavefronts each, speculate on the lambda ladder (SpecCtl, srba_device.hpp).
// Replica w works in its own copy of the work arena (`stride` bytes apart, zeroed at upload); replica 0's is the one the host reads back.
template <int FAM>
__global__ void __launch_bounds__(2 * SRBA_WG) __attribute__((amdgpu_waves_per_eu(3, 3))) k_lm_spec(const Batch B, const DevParams prm, int lds_doubles, long long stride, SpecCtl sc) {
	double *red = srba_lds + lds_doubles; sc.w = blockIdx.x; sc.stride = stride;
	const Batch Bw = shift_work(B, stride * sc.w); const ProbDesc &d = B.desc[0];
	if (!sc.w) { constexpr int PD = Tr<FAM>::PD; for (int k = threadIdx.x; k < d.nK * PD; k += 2 * SRBA_WG) sc.edge_backup[k] = B.edge[d.o_edge * PD + k]; }
		// (the host's way back if the replicas lose step: SpecCtl::edge_backup)
	if (sc.w) { constexpr int PD = Tr<FAM>::PD; // the accepted state the run starts from (replica 0's: srba_hip_reset_state, or what an earlier run left);
		// replica 0 writes there after the first exchange only
		for (int k = threadIdx.x; k < d.n_edges * PD; k += 2 * SRBA_WG) Bw.edge[d.o_edge * PD + k] = B.edge[d.o_edge * PD + k];
		__syncthreads(); }
	lm_spec<FAM>(Bw, prm, 0, red, &sc);
}

// ---- stepwise kernels
template <int FAM> __global__ void __launch_bounds__(SRBA_WG) k_spantree(const Batch B, const DevParams prm, int only_needed) { Solver<FAM> S(B, B.desc[blockIdx.x], prm);
	S.phase_spantree(only_needed != 0); }
template <int FAM> __global__ void __launch_bounds__(SRBA_WG) k_residuals(const Batch B, const DevParams prm) {
	Solver<FAM> S(B, B.desc[blockIdx.x], prm); const double e = S.phase_residuals(B.resid, srba_lds, true); if (threadIdx.x == 0) B.chi2[blockIdx.x] = e;
}
template <int FAM> __global__ void __launch_bounds__(SRBA_WG) k_linearize(const Batch B, const DevParams prm, int lds_doubles, const int *list /* capsule of every workgroup,
	or NULL: the whole batch */) {
	constexpr int P = Worker<FAM>::P;
	const int pidx = list ? list[blockIdx.x] : blockIdx.x;
	Solver<FAM> S(B, B.desc[pidx], prm);
	const ProbDesc &d = B.desc[pidx];
	S.phase_jacobians();
	// U_Ap blocks term-parallel with LDS accumulators when they fit the launch's LDS (lds_doubles; the first 16 doubles are the reduction scratch), else one lane per block
	int nv = (d.n_hap * P * P <= lds_doubles - 16) ? S.phase_hessian_terms(srba_lds + 16) + S.phase_hessian_landmark_blocks() : S.phase_hessian();
	const int ninv = (int)block_sum((double)nv, srba_lds); __syncthreads();
	S.phase_gradient(B.resid); __syncthreads(); S.keep_gradient();
	const double l0 = S.lambda_guess(srba_lds);
	if (threadIdx.x == 0) { B.lambda_io[pidx] = l0; B.results[pidx].num_invalid_jacobs = ninv; }
}
// K2 (+ K3) alone: the Jacobian blocks in HBM, for srba_hip_debug_read after a fused srba_hip_linearize
template <int FAM> __global__ void __launch_bounds__(SRBA_WG) k_jacobians_only(const Batch B, const DevParams prm) { Solver<FAM> S(B, B.desc[blockIdx.x], prm); S.phase_jacobians(); }
// K6 alone on whatever Jacobian blocks are in device memory (every row taken as valid): the algebra of the reference's SchurTests starts from given blocks
template <int FAM> __global__ void __launch_bounds__(SRBA_WG) k_hessian_only(const Batch B, const DevParams prm) {
	Solver<FAM> S(B, B.desc[blockIdx.x], prm); const ProbDesc &d = B.desc[blockIdx.x];
	for (int b = threadIdx.x; b < d.n_bp; b += SRBA_WG) B.bp_ok[d.o_bp + b] = 1;
	for (int b = threadIdx.x; b < d.n_bf; b += SRBA_WG) B.bf_ok[d.o_bf + b] = 1;
	__syncthreads();
	S.phase_hessian();
}
template <int FAM> __global__ void __launch_bounds__(SRBA_WG) k_solve(const Batch B, const DevParams prm, int first) {
	const int pidx = B.order[first + blockIdx.x]; const ProbDesc &d = B.desc[pidx]; Solver<FAM> S(B, d, prm);
	const SparseSys A = S.make_sys(srba_lds);
	const bool ok = S.solve(A, B.lambda_io[pidx]);
	if (threadIdx.x == 0) B.notpd[pidx] = ok ? 0 : 1;
}
template <int FAM> __global__ void __launch_bounds__(SRBA_WG) k_apply(const Batch B, const DevParams prm) { Solver<FAM> S(B, B.desc[blockIdx.x], prm); S.apply_update(); }
template <int FAM> __global__ void __launch_bounds__(SRBA_WG) k_rollback(const Batch B, const DevParams prm) { Solver<FAM> S(B, B.desc[blockIdx.x], prm); S.restore(); }

#ifdef SRBA_PROBE_KERNELS /* diagnostic build only (tools/probe_phases.sh): every phase of the workgroup kernel as a kernel of its own, to read its register need from the code object */
#define SRBA_PROBE(NAME, BODY) template <int FAM> __global__ void __launch_bounds__(256) NAME(const Batch B, const DevParams prm) { Solver<FAM, true, 256> S(B, B.desc[blockIdx.x], prm, \
	srba_lds + WG_RED); const SparseSys A = S.make_sys(srba_lds); (void)A; BODY; }
SRBA_PROBE(kp_spantree, S.phase_spantree(false, nullptr, B.pose1))
SRBA_PROBE(kp_spantree_need, S.phase_spantree(true, nullptr))
SRBA_PROBE(kp_jacobians, S.phase_jacobians())
SRBA_PROBE(kp_hessian, B.notpd[blockIdx.x] = S.phase_hessian())
SRBA_PROBE(kp_hessian_lds, B.notpd[blockIdx.x] = S.phase_hessian_lds())
SRBA_PROBE(kp_schur_lds, S.schur_assemble_lds(A, B.lambda_io[blockIdx.x]))
SRBA_PROBE(kp_gradient, S.phase_gradient(B.resid))
SRBA_PROBE(kp_residuals, B.chi2[blockIdx.x] = S.phase_residuals(B.resid, srba_lds + WG_RED))
SRBA_PROBE(kp_schur, S.schur_reduce(B.lambda_io[blockIdx.x]))
SRBA_PROBE(kp_chol, B.notpd[blockIdx.x] = wg_chol_solve<4>(A.tiles, A.linv, A.nt, (lds_f64 *)srba_lds))
SRBA_PROBE(kp_features, S.schur_features())
SRBA_PROBE(kp_apply, S.apply_trial(A))
template <int FAM> void probe_instantiate() { Batch B; DevParams p; hipLaunchKernelGGL(kp_spantree<FAM>, 1, 256, 0, 0, B, p); hipLaunchKernelGGL(kp_spantree_need<FAM>, 1, 256, 0, 0, B, p);
	hipLaunchKernelGGL(kp_jacobians<FAM>, 1, 256, 0, 0, B, p); hipLaunchKernelGGL(kp_hessian<FAM>, 1, 256, 0, 0, B, p);
	hipLaunchKernelGGL(kp_gradient<FAM>, 1, 256, 0, 0, B, p); hipLaunchKernelGGL(kp_hessian_lds<FAM>, 1, 256, 0, 0, B, p); hipLaunchKernelGGL(kp_schur_lds<FAM>, 1, 256, 0, 0, B, p);
		hipLaunchKernelGGL(kp_residuals<FAM>, 1, 256, 0, 0, B, p); hipLaunchKernelGGL(kp_schur<FAM>, 1, 256, 0, 0, B, p);
	hipLaunchKernelGGL(kp_chol<FAM>, 1, 256, 0, 0, B, p); hipLaunchKernelGGL(kp_features<FAM>, 1, 256, 0, 0, B, p); hipLaunchKernelGGL(kp_apply<FAM>, 1, 256, 0, 0, B, p); }
template void probe_instantiate<SRBA_PROBE_KERNELS>();
#endif

} // namespace srbadev
#define SRBA_BIG_DECLS_ONLY /* the kernels of srba_big.hpp are compiled in srba_big.hip: here only its constants and records */
#include "srba_big.hpp"
#include "srba_flat.hpp"
#include "srba_assemble.hpp"
namespace srbadev {
// ---- whole-map squared error (eval_overall_error.h:15-137): a plain streaming pair of kernels over ONE problem (desc[0]), grid-stride
// K1 over all (observer, base) pairs: compose the breadth-first path from the root of the pair (spantree_create_complete.h:96-124)
template <int FAM> __global__ void __launch_bounds__(256) k_overall_pairs(const Batch B, const DevParams prm) {
	typedef Worker<FAM> W; typedef typename W::PO PO; typedef typename W::pose_t pose_t; constexpr int PD = W::PD;
	const ProbDesc &d = B.desc[0];
	for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < d.n_pairs; p += gridDim.x * blockDim.x) {
		pose_t acc = PO::ident();
		for (int k = B.pair_path_off[p]; k < B.pair_path_off[p + 1]; k++) {
			const int pe = B.path_edge[k]; const pose_t ed = PO::ld(B.edge + (long long)(pe >> 1) * PD);
			acc = (pe & 1) ? comp(acc, inv(ed)) : comp(acc, ed);
		}
		PO::st(B.pose + (long long)p * 2 * PD, acc); PO::st(B.pose + ((long long)p * 2 + 1) * PD, inv(acc));
	}
}
// K4 over all observations; one partial sum per workgroup (fixed order -> deterministic), summed by the host
template <int FAM> __global__ void __launch_bounds__(256) k_overall_residuals(const Batch B, const DevParams prm, double *partial) {
	Worker<FAM> Wk(B, B.desc[0], prm); constexpr int O = Worker<FAM>::O;
	const ProbDesc &d = B.desc[0];
	double acc = 0;
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < d.n_obs; i += gridDim.x * blockDim.x) { double r[O]; acc += Wk.residual_row(i, r); }
	__shared__ double sh[4];
	const double v = wave_sum(acc);
	if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
	__syncthreads();
	if (threadIdx.x == 0) partial[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

} // namespace srbadev

// =================================================================================================== host side
#include "srba_ctx.hpp"
thread_local std::string g_last_error;

// Symbolic block factorisation of one capsule's system (natural block order): the numeric kernel never discovers structure.
struct Symbolic { std::vector<int32_t> fill; std::vector<int32_t> col_off, row, item_off, tgt, ab, rptr, rcol, rblk, perm, hap_dst, hapf_dst, hf_dst; int max_cn = 0; bool aligned = true; };
static void symbolic_factor(const srba_problem_capsule &k, const ProbDesc &d, int P, int L, bool full_system, Symbolic &out) {
	const int nb = d.nb;
	// 1) block-level adjacency of the system (original numbering)
	std::vector<std::vector<int> > adj(nb);
	auto add_scalar_block = [&](int r0, int c0, int nr, int nc) {
		for (int i = 0; i < nr; i++) for (int j = 0; j < nc; j++) { const int a = (r0 + i) / 3, b = (c0 + j) / 3; if (a != b) { adj[a].push_back(b); adj[b].push_back(a); } }
	};
	for (int b = 0; b < k.n_hap; b++) add_scalar_block(P * k.hap_i[b], P * k.hap_j[b], P, P);
	if (full_system) {
		for (int b = 0; b < k.n_hapf; b++) add_scalar_block(P * k.hapf_i[b], P * d.nK + L * k.hapf_j[b], P, L);
		for (int b = 0; b < k.n_hf; b++) add_scalar_block(P * d.nK + L * k.hf_i[b], P * d.nK + L * k.hf_j[b], L, L);
	}
	for (int a = 0; a < nb; a++) { std::sort(adj[a].begin(), adj[a].end()); adj[a].erase(std::unique(adj[a].begin(), adj[a].end()), adj[a].end()); }
	// 2) fill-reducing order: exact minimum degree on the elimination graph (ties: lowest index). Stands where CSparse's cs_amd stands in the
	//    reference (cs_schol(order=1)); any permutation gives the same solution up to rounding.
	out.perm.assign(nb, 0);
	{
		std::vector<std::vector<int> > g(adj); std::vector<char> gone(nb, 0); std::vector<int> tmp;
		for (int step = 0; step < nb; step++) {
			int best = -1; size_t bestdeg = ~size_t(0);
			for (int v = 0; v < nb; v++) if (!gone[v] && g[v].size() < bestdeg) { bestdeg = g[v].size(); best = v; }
			out.perm[best] = step; gone[best] = 1;
			const std::vector<int> nbrs = g[best];
			for (size_t x = 0; x < nbrs.size(); x++) { // connect the neighbours into a clique, drop the eliminated node
				std::vector<int> &gx = g[nbrs[x]]; tmp.clear();
				std::set_union(gx.begin(), gx.end(), nbrs.begin(), nbrs.end(), std::back_inserter(tmp));
				gx.clear(); for (size_t q = 0; q < tmp.size(); q++) if (tmp[q] != best && tmp[q] != nbrs[x]) gx.push_back(tmp[q]);
			}
			g[best].clear();
		}
	}
	// 3) symbolic factorisation in the permuted numbering: cols[c] = block rows r > c of column c of L
	std::vector<std::vector<int> > cols(nb);
	for (int a = 0; a < nb; a++) for (size_t q = 0; q < adj[a].size(); q++) { const int pa = out.perm[a], pb = out.perm[adj[a][q]]; if (pa < pb) cols[pa].push_back(pb); }
	for (int c = 0; c < nb; c++) {
		std::vector<int> &v = cols[c]; std::sort(v.begin(), v.end()); v.erase(std::unique(v.begin(), v.end()), v.end());
		if (!v.empty()) { std::vector<int> &par = cols[v[0]]; par.insert(par.end(), v.begin() + 1, v.end()); }
	}
	out.col_off.assign(nb + 1, 0); out.row.clear(); out.item_off.assign(nb + 1, 0); out.tgt.clear(); out.ab.clear(); out.max_cn = 0;
	for (int c = 0; c < nb; c++) { out.row.insert(out.row.end(), cols[c].begin(), cols[c].end()); out.col_off[c + 1] = (int32_t)out.row.size(); }
	for (int c = 0; c < nb; c++) {
		const std::vector<int> &v = cols[c]; const int cn = (int)v.size(); out.max_cn = std::max(out.max_cn, cn);
		for (int a = 0; a < cn; a++) for (int b = 0; b <= a; b++) { // item t = a(a+1)/2 + b : target block (v[a], v[b])
			out.ab.push_back((a << 16) | b);
			if (a == b) { out.tgt.push_back(-1 - v[a]); continue; }
			const std::vector<int> &cb = cols[v[b]]; const int pos = (int)(std::lower_bound(cb.begin(), cb.end(), v[a]) - cb.begin());
			out.tgt.push_back(out.col_off[v[b]] + pos);
		}
		out.item_off[c + 1] = (int32_t)out.tgt.size();
	}
	// row view of the strictly-lower blocks (backward substitution pushes along block-rows)
	out.rptr.assign(nb + 1, 0); out.rcol.assign(out.row.size(), 0); out.rblk.assign(out.row.size(), 0);
	for (size_t i = 0; i < out.row.size(); i++) out.rptr[out.row[i] + 1]++;
	for (int a = 0; a < nb; a++) out.rptr[a + 1] += out.rptr[a];
	{ std::vector<int32_t> fill(out.rptr.begin(), out.rptr.end() - 1);
	  for (int c = 0; c < nb; c++) for (int i = out.col_off[c]; i < out.col_off[c + 1]; i++) { const int q = fill[out.row[i]]++; out.rcol[q] = c; out.rblk[q] = i; } }
	// 4) destination of every aligned 3x3 sub-block (a,b) of the UPPER-triangle Hessian blocks, original numbering a <= b
	auto dst_of = [&](int a, int b) -> int32_t {
		if (a == b) return -1 - out.perm[a];
		if (a > b) return (int32_t)0x80000000; // lower half of a symmetric diagonal block: duplicate, skipped
		const int pa = out.perm[a], pb = out.perm[b]; const int lo = std::min(pa, pb), hi = std::max(pa, pb);
		const std::vector<int> &cb = cols[lo]; const int pos = (int)(std::lower_bound(cb.begin(), cb.end(), hi) - cb.begin());
		return ((out.col_off[lo] + pos) << 1) | (pa < pb ? 1 : 0); // pa<pb: the factor stores block (pb,pa) = transpose of the given one
	};
	const int PB = P / 3;
	out.hap_dst.clear(); out.hapf_dst.clear(); out.hf_dst.clear();
	for (int b = 0; b < k.n_hap; b++) for (int si = 0; si < PB; si++) for (int sj = 0; sj < PB; sj++) out.hap_dst.push_back(dst_of(k.hap_i[b] * PB + si, k.hap_j[b] * PB + sj));
	out.aligned = !(full_system && L != 3 && k.n_unk_lms > 0);
	if (full_system && L == 3) {
		const int lb = PB * d.nK;
		for (int b = 0; b < k.n_hapf; b++) for (int si = 0; si < PB; si++) out.hapf_dst.push_back(dst_of(k.hapf_i[b] * PB + si, lb + k.hapf_j[b]));
		for (int b = 0; b < k.n_hf; b++) out.hf_dst.push_back(dst_of(lb + k.hf_i[b], lb + k.hf_j[b]));
	} else { out.hapf_dst.assign((size_t)k.n_hapf * PB, 0); out.hf_dst.assign(k.n_hf, 0); }
	// 5) blocks without a source (fill-in; unified index: diag k -> k, off-diagonal i -> nb+i)
	{ std::vector<char> covered(nb + out.row.size(), 0);
	  auto mark = [&](const std::vector<int32_t> &v) { for (size_t i = 0; i < v.size(); i++) { const int32_t dsti = v[i]; if (dsti == (int32_t)0x80000000) continue;
	  	covered[dsti >= 0 ? nb + (dsti >> 1) : -1 - dsti] = 1; } };
	  mark(out.hap_dst); if (full_system && L == 3) { mark(out.hapf_dst); mark(out.hf_dst); }
	  out.fill.clear(); for (size_t u = 0; u < covered.size(); u++) if (!covered[u]) out.fill.push_back((int32_t)u); }
}

// The same outputs for the DENSE block layout of a mid-size, nearly full system (ProbDesc::dense_blocks): identity order, every block (r > c) present at
// c (nb-1) - c (c-1)/2 + (r-c-1), no index arrays for the device (the solver computes them); only the destinations of the Hessian sub-blocks and the list of
// blocks without a source (to be zeroed at assembly) are needed.
static void symbolic_dense(const srba_problem_capsule &k, const ProbDesc &d, int P, int L, bool full_system, Symbolic &out) {
	const int nb = d.nb; out = Symbolic();
	out.perm.resize(nb); for (int q = 0; q < nb; q++) out.perm[q] = q;
	out.col_off.assign(nb + 1, 0); out.item_off.assign(nb + 1, 0); out.rptr.assign(nb + 1, 0); out.max_cn = nb - 1;
	auto cstart = [&](int c) { return c * (nb - 1) - c * (c - 1) / 2; };
	auto dst_of = [&](int a, int b) -> int32_t {
		if (a == b) return -1 - a;
		if (a > b) return (int32_t)0x80000000; // lower half of a symmetric diagonal block: duplicate, skipped
		return ((cstart(a) + (b - a - 1)) << 1) | 1; // the factor stores block (b,a) = transpose of the given upper block
	};
	const int PB = P / 3;
	for (int b = 0; b < k.n_hap; b++) for (int si = 0; si < PB; si++) for (int sj = 0; sj < PB; sj++) out.hap_dst.push_back(dst_of(k.hap_i[b] * PB + si, k.hap_j[b] * PB + sj));
	out.aligned = !(full_system && L != 3 && k.n_unk_lms > 0);
	if (full_system && L == 3) {
		const int lb = PB * d.nK;
		for (int b = 0; b < k.n_hapf; b++) for (int si = 0; si < PB; si++) out.hapf_dst.push_back(dst_of(k.hapf_i[b] * PB + si, lb + k.hapf_j[b]));
		for (int b = 0; b < k.n_hf; b++) out.hf_dst.push_back(dst_of(lb + k.hf_i[b], lb + k.hf_j[b]));
	} else { out.hapf_dst.assign((size_t)k.n_hapf * PB, 0); out.hf_dst.assign(k.n_hf, 0); }
	const int nnz = nb * (nb - 1) / 2; std::vector<char> covered(nb + nnz, 0);
	auto mark = [&](const std::vector<int32_t> &v) { for (size_t i = 0; i < v.size(); i++) { const int32_t dsti = v[i]; if (dsti == (int32_t)0x80000000) continue;
		covered[dsti >= 0 ? nb + (dsti >> 1) : -1 - dsti] = 1; } };
	mark(out.hap_dst); if (full_system && L == 3) { mark(out.hapf_dst); mark(out.hf_dst); }
	for (size_t u = 0; u < covered.size(); u++) if (!covered[u]) out.fill.push_back((int32_t)u);
}


// Index-range check of one capsule (every index the kernels dereference): a wrong capsule is reported at upload instead of reading out of bounds on the device
static const char *validate_capsule(const srba_problem_capsule &k) {
	auto in = [](int v, int lo, int hi) { return v >= lo && v < hi; };
	if (k.n_edges < 0 || k.n_unk_edges < 0 || k.n_unk_lms < 0 || k.n_known_lms < 0 || k.n_pairs < 0 || k.n_path < 0 || k.n_obs < 0 || k.n_valid < 0 || k.n_bp < 0 || k.n_bf < 0 || k.n_hap < 0 ||
		k.n_hf < 0 || k.n_hapf < 0 || k.n_sch_terms < 0) return "negative size";
	if (k.n_unk_edges > k.n_edges || k.n_unk_edges + k.n_unk_lms == 0) return "no unknowns / more unknown edges than edges";
	const int np2 = 2 * k.n_pairs;
	auto lmref = [&](int v) { return v >= 0 ? v < k.n_unk_lms : (-1 - v) < k.n_known_lms; };
	// every array is checked for presence before any loop below reads it (validation runs on upload worker threads: a crash here would take the process down off the calling thread)
	if (!k.edge_pose || (k.n_unk_lms && !k.ulm_pos) || (k.n_known_lms && !k.klm_pos) || (k.n_obs && (!k.obs_pose || !k.obs_lm || !k.obs_valid || !k.obs_z)) || (k.n_path && !k.path_edge)) return
		"null data array";
	if (k.n_path > 0 && k.n_pairs == 0) return "path entries without pairs";
	if (k.n_pairs && (!k.pair_path_off || !k.pair_needed || !k.pose_required || k.pair_path_off[0] != 0 || k.pair_path_off[k.n_pairs] != k.n_path)) return "pair_path_off";
	for (int i = 0; i < k.n_pairs; i++) if (k.pair_path_off[i + 1] < k.pair_path_off[i]) return "pair_path_off not monotone";
	for (int i = 0; i < k.n_path; i++) if (k.path_edge[i] < 0 || (k.path_edge[i] >> 1) >= k.n_edges) return "path_edge";
	for (int i = 0; i < k.n_obs; i++) if (!in(k.obs_pose[i], -1, np2) || !lmref(k.obs_lm[i]) || !in(k.obs_valid[i], 0, std::max(k.n_valid, 1))) return "observation table";
	if (k.n_bp && (!k.bp_col || !k.bp_res || !k.bp_A || !k.bp_D || !k.bp_lm || !k.bp_normal)) return "null dh_dAp table";
	if (k.n_bf && (!k.bf_col || !k.bf_res || !k.bf_pose)) return "null dh_df table";
	if (k.n_unk_edges && (!k.colp_off || k.colp_off[0] != 0 || k.colp_off[k.n_unk_edges] != k.n_bp)) return "colp_off";
	for (int i = 0; i < k.n_unk_edges; i++) if (k.colp_off[i + 1] < k.colp_off[i]) return "colp_off not monotone";
	for (int i = 0; i < k.n_bp; i++) if (!in(k.bp_col[i], 0, k.n_unk_edges) || !in(k.bp_res[i], 0, k.n_obs) || !in(k.bp_A[i], -1, np2) || !in(k.bp_D[i], -1,
		np2) || !lmref(k.bp_lm[i])) return "dh_dAp block table";
	if (k.n_unk_lms && (!k.colf_off || k.colf_off[0] != 0 || k.colf_off[k.n_unk_lms] != k.n_bf)) return "colf_off";
	for (int i = 0; i < k.n_unk_lms; i++) if (k.colf_off[i + 1] < k.colf_off[i]) return "colf_off not monotone";
	for (int i = 0; i < k.n_bf; i++) if (!in(k.bf_col[i], 0, k.n_unk_lms) || !in(k.bf_res[i], 0, k.n_obs) || !in(k.bf_pose[i], -1, np2)) return "dh_df block table";
	if (k.n_hap && (!k.hap_i || !k.hap_j || !k.hap_term_off || (k.n_hap_terms && (!k.hap_t1 || !k.hap_t2)))) return "null HAp table";
	if (k.n_unk_edges && !k.hap_diag) return "null hap_diag";
	if (k.n_hap && (k.hap_term_off[0] != 0 || k.hap_term_off[k.n_hap] != k.n_hap_terms)) return "hap_term_off";
	for (int i = 0; i < k.n_hap; i++) if (!in(k.hap_i[i], 0, k.n_unk_edges) || !in(k.hap_j[i], 0, k.n_unk_edges) || k.hap_term_off[i + 1] < k.hap_term_off[i]) return "HAp block table";
	for (int i = 0; i < k.n_hap_terms; i++) if (!in(k.hap_t1[i], 0, k.n_bp) || !in(k.hap_t2[i], 0, k.n_bp)) return "HAp terms";
	for (int i = 0; i < k.n_unk_edges; i++) if (!in(k.hap_diag[i], 0, k.n_hap)) return "hap_diag";
	if (k.n_hf && (!k.hf_i || !k.hf_j || !k.hf_term_off || !k.hf_diag || (k.n_hf_terms && (!k.hf_t1 || !k.hf_t2)))) return "null Hf table";
	if (k.n_hf && (k.hf_term_off[0] != 0 || k.hf_term_off[k.n_hf] != k.n_hf_terms)) return "hf_term_off";
	for (int i = 0; i < k.n_hf; i++) if (k.hf_term_off[i + 1] < k.hf_term_off[i]) return "hf_term_off not monotone";
	for (int i = 0; i < k.n_hf; i++) if (!in(k.hf_i[i], 0, k.n_unk_lms) || !in(k.hf_j[i], 0, k.n_unk_lms)) return "Hf block table";
	for (int i = 0; i < k.n_hf_terms; i++) if (!in(k.hf_t1[i], 0, k.n_bf) || !in(k.hf_t2[i], 0, k.n_bf)) return "Hf terms";
	for (int i = 0; i < k.n_unk_lms && k.n_hf; i++) if (!in(k.hf_diag[i], 0, k.n_hf)) return "hf_diag";
	if (k.n_hapf && (!k.hapf_i || !k.hapf_j || !k.hapf_term_off || !k.hapf_t1 || !k.hapf_t2 || !k.lm_hapf_off || !k.lm_hapf_idx)) return "null HApf table";
	if (k.n_hapf && (k.hapf_term_off[0] != 0 || k.hapf_term_off[k.n_hapf] != k.n_hapf_terms)) return "hapf_term_off";
	for (int i = 0; i < k.n_hapf; i++) if (k.hapf_term_off[i + 1] < k.hapf_term_off[i]) return "hapf_term_off not monotone";
	if (k.n_hapf) { if (k.lm_hapf_off[0] != 0 || k.lm_hapf_off[k.n_unk_lms] != k.n_hapf) return "lm_hapf_off"; for (int i = 0; i < k.n_unk_lms;
		i++) if (k.lm_hapf_off[i + 1] < k.lm_hapf_off[i]) return "lm_hapf_off not monotone";
		for (int i = 0; i < k.n_hapf; i++) if (!in(k.lm_hapf_idx[i], 0, k.n_hapf)) return "lm_hapf_idx"; }
	for (int i = 0; i < k.n_hapf; i++) if (!in(k.hapf_i[i], 0, k.n_unk_edges) || !in(k.hapf_j[i], 0, k.n_unk_lms)) return "HApf block table";
	for (int i = 0; i < k.n_hapf_terms; i++) if (!in(k.hapf_t1[i], 0, k.n_bp) || !in(k.hapf_t2[i], 0, k.n_bf)) return "HApf terms";
	if (k.n_sch_terms) { if (!k.sch_term_off || !k.sch_b1 || !k.sch_b2 || !k.sch_lm || k.sch_term_off[0] != 0 || k.sch_term_off[k.n_hap] != k.n_sch_terms) return "sch_term_off";
		for (int i = 0; i < k.n_hap; i++) if (k.sch_term_off[i + 1] < k.sch_term_off[i]) return "sch_term_off not monotone";
		for (int i = 0; i < k.n_sch_terms; i++) if (!in(k.sch_b1[i], 0, k.n_hapf) || !in(k.sch_b2[i], 0, k.n_hapf) || !in(k.sch_lm[i], 0, k.n_unk_lms)) return "Schur terms"; }
	return nullptr;
}

// first element of slice q when cnt items are dealt round-robin to nq slices
static inline int slice_begin(int cnt, int q, int nq) { return q * (cnt / nq) + std::min(q, cnt % nq); }

// Launch plan of the fused LM kernel. Every size class is one or more launches (a launch has ONE dynamic-LDS size); the HIP runtime
// multiplexes streams onto 4 hardware queues and kernels of one queue run in order, so the plan uses n_queues streams and decides what
// shares the chip at any time. Small capsules are wave-slot bound (VGPRs), big ones LDS bound: running them side by side fills both.
//   sched 0: every class cut into n_queues interleaved slices, one per stream (all queues walk the classes in step, biggest first)
//   sched 1: classes cut into chunks, chunks dealt to the least-loaded queue (cost ~ sum of system sizes); even queues run their chunks
//            biggest-LDS first, odd queues smallest first -> big and small capsules overlap for the whole run
//   sched 2: one stream per class, biggest first
static void plan_launches(srba_hip_ctx *c, const int32_t *ord) {
	c->plan.clear(); const int nq = c->n_queues;
	struct fin { srba_hip_ctx *c; ~fin() { // grid of every job: persistent launches hold as many wavefronts as the chip can keep resident for that LDS size, the rest one per capsule
		int batch_total = 0; for (int k = 0; k < SRBA_NCLS; k++) batch_total += c->cls_count[k];
		for (size_t j = 0; j < c->plan.size(); j++) { LaunchJob &J = c->plan[j]; J.grid = J.count;
			if (c->sched == 3) { const size_t lds = c->cls_lds[J.cls] + c->lds_pad; const int fit = lds ? (int)std::max<size_t>(1, (size_t)c->lds_per_cu / lds) : c->waves_per_cu; J.grid = std::max(1,
				std::min(J.count, c->n_cu * std::min(c->waves_per_cu, fit)));
				if (J.cls >= SRBA_NLDS) J.grid = std::max(1, std::min(J.count, c->n_cu * (J.cls == SRBA_CLS_WG512 ? 1 : (J.cls == SRBA_CLS_WG256 ? SRBA_WG_WAVES : 2 * SRBA_WG_WAVES))));
					// workgroup classes: 256 registers -> two wavefronts per SIMD = two 256-thread / four 128-thread workgroups per CU
				J.lean = (c->lean_on && c->params.family == SRBA_SE2_RELPOSE2D && J.cls < SRBA_NCLS - 1 && fit >= 9 && J.count >= c->lean_min_count) ? 1 : 0;
				if (J.lean) J.grid = std::max(1, std::min(J.count, c->n_cu * std::min(4 * SRBA_LEAN_WAVES, fit)));
				// (also for a batch of a few capsules -- the per-key-frame use of the engine is a batch of ONE: its latency is the whole cost, 1.55 -> 1.41 ms per key-frame of the sequential run)
				J.two = (c->two_on && c->params.family == SRBA_SE2_RELPOSE2D && J.cls < SRBA_NCLS - 1 && lds > 0 && ((lds >= (size_t)c->two_from_kb * 1024 && J.count >= c->two_min_count) ||
					batch_total <= 4)) ? 1 : 0; if (J.two) J.lean = 0;
				if (J.two) J.grid = std::max(1, std::min(J.count, c->n_cu * std::min(6, fit))); }
		}
		// Staggered start (round 4): the persistent launches of all classes are enqueued at once on their own streams, and which workgroups the dispatcher places first was a race --
		// when the small (wave-slot bound) classes won it they filled every wave slot, the big (LDS bound, longest running) capsules trickled in late and the launch ended in their tail:
		// 42-43 ms instead of 38 ms on the benchmark batch, from one launch to the next (tools/diag_launch_order.py, profiles/r04_launch_order.txt). Largest-footprint-first is now enforced:
		// the stream of job j is held back by a one-thread delay kernel for stagger_ns x (workgroups of all the jobs before it) -- the time the dispatcher needs to place those.
		if (c->sched == 3 && c->plan.size() > 1) { long long ahead = 0; for (size_t j = 0; j < c->plan.size(); j++) { LaunchJob &J = c->plan[j];
			J.delay_us = (int)std::min<long long>(c->stagger_max_us, ahead * c->stagger_ns / 1000); if (j < c->delay_us.size()) J.delay_us = c->delay_us[j]; ahead += J.grid; } }
		if (getenv("SRBA_HIP_PLAN_DEBUG")) for (size_t j = 0; j < c->plan.size(); j++) { const LaunchJob &J = c->plan[j]; std::fprintf(stderr,
			"[plan] job %zu: stream %d class %d lds %zu B capsules %d grid %d delay %d us%s\n", j, J.queue, J.cls, c->cls_lds[J.cls], J.count, J.grid, J.delay_us,
			J.lean ? " (lean: three wavefronts per SIMD)" : (J.two ? " (two wavefronts per capsule)" : "")); } } } finish = {c};
	if (c->sched == 3) { // one persistent launch per size class, every class on its own stream, biggest LDS footprint first (the HBM class is the biggest)
		int q = 0; const int qmax = std::max(1, c->class_streams);
		for (int k = SRBA_NCLS - 2; k >= 0; k--) if (c->cls_count[k]) { c->plan.push_back({q % qmax, k, c->cls_first[k], c->cls_count[k], 0.0, 0}); q++; }
		c->n_streams_used = std::max(std::min(q, qmax), 1); return;
	}
	// cost of a chunk ~ sum over its capsules of (system size) x (LDS footprint): a trial takes time ~ nb, and how many capsules run at once
	// is set by the LDS they hold (measured on the benchmark: 43 us per loop-closure window vs 5.5 us per typical window, chip-wide)
	auto cost_of = [&](int cls, int first, int count) { const double w = cls == SRBA_NCLS - 1 ? 24.0 : std::max(1.0, (double)c->cls_lds[cls] / 8192.0); double s = 0; for (int i = 0; i < count;
		i++) s += (c->desc[ord[first + i]].nb + 4) * w; return s; };
	if (c->sched == 2) {
		int q = 0; const int qmax = std::max(1, c->class_streams);
		for (int k = SRBA_NCLS - 2; k >= 0; k--) if (c->cls_count[k]) { c->plan.push_back({q % qmax, k, c->cls_first[k], c->cls_count[k], 0.0, 0}); q++; }
		c->n_streams_used = std::max(std::min(q, qmax), 1); return;
	}
	c->n_streams_used = nq;
	if (c->sched == 0) {
		int rr = 0;
		for (int k = SRBA_NCLS - 2; k >= 0; k--) if (c->cls_count[k]) {
			const int cnt = c->cls_count[k], parts = cnt >= 16 * nq ? nq : 1;
			for (int q = 0; q < parts; q++) { const int a = slice_begin(cnt, q, parts), b = slice_begin(cnt, q + 1, parts); if (b > a) c->plan.push_back({parts == 1 ? (rr++ % nq) : q, k,
				c->cls_first[k] + a, b - a, 0.0, 0}); }
		}
		return;
	}
	std::vector<LaunchJob> jobs;
	for (int k = SRBA_NCLS - 2; k >= 0; k--) if (c->cls_count[k]) {
		const int cnt = c->cls_count[k], parts = std::max(1, std::min(c->max_parts_per_queue * nq, cnt / c->min_chunk));
		for (int q = 0; q < parts; q++) { const int a = slice_begin(cnt, q, parts), b = slice_begin(cnt, q + 1, parts); if (b > a) jobs.push_back({0, k, c->cls_first[k] + a, b - a, cost_of(k,
			c->cls_first[k] + a, b - a), 0}); }
	}
	std::vector<size_t> by_cost(jobs.size()); for (size_t i = 0; i < jobs.size(); i++) by_cost[i] = i;
	std::stable_sort(by_cost.begin(), by_cost.end(), [&](size_t a, size_t b) { return jobs[a].cost > jobs[b].cost; });
	std::vector<double> load(nq, 0.0); std::vector<std::vector<LaunchJob> > perq(nq);
	for (size_t i = 0; i < by_cost.size(); i++) { int q = 0; for (int x = 1; x < nq; x++) if (load[x] < load[q]) q = x; LaunchJob J = jobs[by_cost[i]]; J.queue = q; load[q] += J.cost;
		perq[q].push_back(J); }
	for (int q = 0; q < nq; q++) { // class index grows with the LDS footprint (the HBM class last = biggest)
		std::stable_sort(perq[q].begin(), perq[q].end(), [&](const LaunchJob &a, const LaunchJob &b) { return (q & 1) ? a.cls < b.cls : a.cls > b.cls; });
	}
	for (size_t i = 0;; i++) { bool any = false; for (int q = 0; q < nq; q++) if (i < perq[q].size()) { c->plan.push_back(perq[q][i]); any = true; } if (!any) break; }
}


static void make_dev_params(const srba_hip_params &p, DevParams &dp, const FamDims &dm) {
	std::memset(&dp, 0, sizeof(dp));
	dp.solver = p.solver; dp.noise = p.noise; dp.sensor_pose = p.sensor_pose; dp.max_iters = p.max_iters; dp.use_robust_kernel = p.use_robust_kernel; dp.cov_recovery = p.cov_recovery;
		dp.ext = p.extensions;
	dp.inv_sigma = 1.0 / p.std_noise_observations;
	for (int i = 0; i < dm.O * dm.O; i++) dp.lambda[i] = p.lambda[i];
	dp.kernel_param = p.kernel_param; dp.max_err = p.max_error_per_obs_to_stop; dp.max_rho = p.max_rho; dp.max_lambda = p.max_lambda; dp.min_relin = p.min_error_reduction_ratio_to_relinearize;
	for (int i = 0; i < 3; i++) dp.SPt[i] = p.sensor_pose_se3[i];
	for (int i = 0; i < 9; i++) dp.SPR[i] = p.sensor_pose_se3[3 + i];
	for (int i = 0; i < 4; i++) { dp.camL[i] = p.cam_left[i]; dp.camR[i] = p.cam_right[i]; }
	// R2L = (-)rightCameraPose (models/sensors.h:193): quaternion -> R, then invert
	const double r = p.right_cam_pose[3], x = p.right_cam_pose[4], y = p.right_cam_pose[5], z = p.right_cam_pose[6];
	const double R[9] = {r * r + x * x - y * y - z * z, 2 * (x * y - r * z), 2 * (z * x + r * y), 2 * (x * y + r * z), r * r - x * x + y * y - z * z, 2 * (y * z - r * x), 2 * (z * x - r * y),
		2 * (y * z + r * x), r * r - x * x - y * y + z * z};
	for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) dp.R2LR[3 * i + j] = R[3 * j + i];
	for (int i = 0; i < 3; i++) dp.R2Lt[i] = -(R[i] * p.right_cam_pose[0] + R[3 + i] * p.right_cam_pose[1] + R[6 + i] * p.right_cam_pose[2]);
}

static int check_params(const srba_hip_params *p) {
	if (!p || p->family < 0 || p->family >= SRBA_NUM_FAMILIES) { g_last_error = "bad family"; return -1; }
	if (p->solver < 0 || p->solver > 2) { g_last_error = "bad solver"; return -1; }
	if (p->noise == SRBA_NOISE_IDENTITY && !(p->std_noise_observations > 0)) { g_last_error = "std_noise_observations must be > 0"; return -1; }
	if (p->sensor_pose == SRBA_SENSOR_POSE_SE3 && kDims[p->family].PD != 12 && p->family != SRBA_SE2_STEREO) {
		g_last_error = "sensor_pose_on_robot_se3 is supported with SE3 keyframe poses and with <SE2, Euclidean3D, StereoCamera>"; return -1; }
	if (p->family == SRBA_SE3_RELPOSE3D && p->sensor_pose != SRBA_SENSOR_POSE_NONE) { g_last_error = "relative-pose observations take no sensor pose"; return -1; }
	return 0;
}

// Host side of an upload is per-capsule work on disjoint outputs (validation, symbolic factorisation, packing into the staging arena): spread over threads.
template <class F> static void parallel_ranges(int n, int threads, F fn) { // fn(begin, end, thread); an exception of a worker (std::bad_alloc) is rethrown in the caller
	if (threads <= 1 || n < 24) { fn(0, n, 0); return; }
	if (n < 512) threads = std::min(threads, std::max(2, n / 3)); // a round of a map sweep is a batch of tens to hundreds of capsules (~30 us of host work each): a few threads pay, thirty-two do not
	std::atomic<int> next(0); const int chunk = n < 512 ? std::max(1, n / (threads * 2)) : std::max(16, n / (threads * 8)); std::exception_ptr err; std::atomic<bool> failed(false);
	auto work = [&](int t) {
		try { for (;;) { const int b = next.fetch_add(chunk); if (b >= n || failed.load()) break; fn(b, std::min(n, b + chunk), t); } }
		catch (...) { if (!failed.exchange(true)) err = std::current_exception(); }
	};
	std::vector<std::thread> th;
	try { for (int t = 1; t < threads; t++) th.emplace_back(work, t); } catch (...) { /* fewer threads than asked for: the ones that started share the work */ }
	work(0); for (auto &x : th) x.join();
	if (failed.load() && err) std::rethrow_exception(err);
}

extern "C" {

int srba_family_dims(int family, int *P, int *L, int *O, int *PD) {
	if (family < 0 || family >= SRBA_NUM_FAMILIES) return -1;
	if (P) *P = kDims[family].P; if (L) *L = kDims[family].L; if (O) *O = kDims[family].O; if (PD) *PD = kDims[family].PD; return 0;
}

void srba_hip_params_default(srba_hip_params *p, int family) {
	std::memset(p, 0, sizeof(*p));
	p->family = family; p->solver = SRBA_SOLVER_SCHUR_DENSE_CHOL; p->noise = SRBA_NOISE_IDENTITY; p->sensor_pose = SRBA_SENSOR_POSE_NONE;
	p->std_noise_observations = 1.0; for (int i = 0; i < 6; i++) p->lambda[i * 6 + i] = 1.0; // overwritten by the caller for O x O
	for (int i = 0; i < 9; i++) p->sensor_pose_se3[3 + i] = (i % 4 == 0) ? 1.0 : 0.0;
	p->cam_left[0] = p->cam_left[1] = p->cam_right[0] = p->cam_right[1] = 1.0; p->right_cam_pose[3] = 1.0;
	p->max_iters = 20; p->use_robust_kernel = 0; p->kernel_param = 3.0; p->max_error_per_obs_to_stop = 1e-6; p->max_rho = 10.0; p->max_lambda = 1e20;
	p->min_error_reduction_ratio_to_relinearize = 0.01; p->cov_recovery = 1;
}

const char *srba_hip_last_error(const srba_hip_ctx *ctx) { return ctx ? ctx->error.c_str() : g_last_error.c_str(); }

srba_hip_ctx *srba_hip_create(int device, const srba_hip_params *params) {
	if (check_params(params) != 0) return nullptr;
	int ndev = 0; hipError_t e = hipGetDeviceCount(&ndev);
	if (e != hipSuccess || ndev <= 0) { g_last_error = std::string("no HIP device available: ") + hipGetErrorString(e); return nullptr; }
	if (device < 0) { if (hipGetDevice(&device) != hipSuccess) device = 0; }
	if (device >= ndev) { g_last_error = "HIP device index out of range"; return nullptr; }
	if ((e = hipSetDevice(device)) != hipSuccess) { g_last_error = std::string("hipSetDevice: ") + hipGetErrorString(e); return nullptr; }
	if (!with_family(params->family, [](auto) {})) { g_last_error = "srba_hip_create: this build of the library holds no kernels for the requested family (SRBA_ONLY_FAMILY build)"; return nullptr; }
	if (big_layout_signature() != layout_signature() || srbadev::asm_layout_signature() != sizeof(srbadev::Batch) * 10007ull + sizeof(srbadev::DevParams)) { g_last_error =
		"srba_hip_create: the translation units of libsrba_hip.so were compiled with different SRBA_* settings (their shared records differ)"; return nullptr; }
	srba_hip_ctx *c = new srba_hip_ctx();
	c->device = device; c->params = *params; c->dm = kDims[params->family]; make_dev_params(*params, c->dp, c->dm);
	std::memset(&c->B, 0, sizeof(c->B)); std::memset(&c->stats, 0, sizeof(c->stats));
	{ const char *e = getenv("SRBA_HIP_PHASE_TIMING"); c->phase_timing = (e && e[0] == '1'); }
	{ const char *e = getenv("SRBA_HIP_MAX_LDS_KB"); c->max_lds_kb = e ? atoi(e) : 1 << 20; } // test knob: systems above this many KB are factored in the HBM workspace (0 = all of them)
	{ const char *e = getenv("SRBA_HIP_LDS_PAD"); c->lds_pad = e ? (size_t)atol(e) : 0; } // diagnostics: extra LDS bytes per workgroup (lowers residency)
	{ const char *e = getenv("SRBA_HIP_CHUNK"); if (e && atoi(e) > 0) c->min_chunk = atoi(e); e = getenv("SRBA_HIP_PARTS"); if (e && atoi(e) > 0) c->max_parts_per_queue = atoi(e); }
		// tuning knobs of the launch plan
	{ const char *e = getenv("SRBA_HIP_SCHED"); if (e) c->sched = atoi(e); } // tuning knob: launch plan (see plan_launches)
	{ hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, device) == hipSuccess && pr.multiProcessorCount > 0) c->n_cu = pr.multiProcessorCount; }
	{ const char *e = getenv("SRBA_HIP_CLASS_STREAMS"); if (e && atoi(e) > 0) c->class_streams = atoi(e); }
	{ const char *e = getenv("SRBA_HIP_WAVES_PER_CU"); if (e && atoi(e) > 0) c->waves_per_cu = atoi(e); e = getenv("SRBA_HIP_LDS_PER_CU_KB"); if (e && atoi(e) > 0) c->lds_per_cu = atoi(e) * 1024; }
		// tuning knobs: resident wavefronts / LDS per CU assumed by the persistent plan
	{ const char *e = getenv("SRBA_HIP_QUEUES"); if (e && atoi(e) >= 1 && atoi(e) < SRBA_NCLS) c->n_queues = atoi(e); } // tuning knob: concurrent launch streams
	if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess || false) { g_last_error = "cannot create HIP stream/events"; delete c; return nullptr; }
	bool ok = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) == hipSuccess;
	{ const char *e = getenv("SRBA_HIP_CLASS_PRIO"); if (e) c->class_prio = atoi(e); e = getenv("SRBA_HIP_STAGGER_NS"); if (e && atoi(e) >= 0) c->stagger_ns = atoi(e); e = getenv("SRBA_HIP_TWO");
		if (e) c->two_on = atoi(e) != 0; e = getenv("SRBA_HIP_SPEC"); if (e) { c->spec_on = atoi(e) != 0; if (atoi(e) >= 2) c->spec_w = std::min((int)srba_hip_ctx::kSpecMaxW, atoi(e)); }
		e = getenv("SRBA_HIP_TWO_FROM_KB"); if (e && atoi(e) > 0) c->two_from_kb = atoi(e); e = getenv("SRBA_HIP_TWO_MIN_COUNT"); if (e && atoi(e) >= 1) c->two_min_count = atoi(e);
		e = getenv("SRBA_HIP_LEAN"); if (e) c->lean_on = atoi(e) != 0; e = getenv("SRBA_HIP_LEAN_MIN_COUNT"); if (e && atoi(e) >= 1) c->lean_min_count = atoi(e); e = getenv("SRBA_HIP_DELAY_US");
		if (e) { std::string t(e); size_t p0 = 0; while (p0 <= t.size()) { size_t p1 = t.find(',', p0); if (p1 == std::string::npos) p1 = t.size(); c->delay_us.push_back(atoi(t.substr(p0,
		p1 - p0).c_str())); p0 = p1 + 1; } } }
	int pr_least = 0, pr_greatest = 0; hipDeviceGetStreamPriorityRange(&pr_least, &pr_greatest);
	for (int k = 1; k < SRBA_NCLS && ok; k++) { // (class_prio 1: the streams of the biggest classes -- low stream index, see plan_launches -- get the highest priority, 2: the lowest)
		const int split = 4, pri = c->class_prio == 0 ? 0 : ((c->class_prio == 1) == (k < split) ? pr_greatest : pr_least);
		ok = (c->class_prio ? hipStreamCreateWithPriority(&c->cls_stream[k], hipStreamNonBlocking, pri) : hipStreamCreateWithFlags(&c->cls_stream[k],
			hipStreamNonBlocking)) == hipSuccess && hipEventCreateWithFlags(&c->cls_done[k], hipEventDisableTiming) == hipSuccess; }
	if (!ok || hipMalloc((void **)&c->d_next, sizeof(int) * 4 * kMaxJobs) != hipSuccess || hipMalloc((void **)&c->d_part, 8 * 3 * kBigPart * srbadev::kGang) != hipSuccess || hipMalloc((void
		**)&c->d_scal, (8 * 16 + 4 * 8) * srbadev::kGang) != hipSuccess) { g_last_error = "cannot create HIP stream/events"; delete c; return nullptr; }
	c->spec_test_drop = getenv("SRBA_HIP_SPEC_TEST_DROP") != nullptr;
	{ const char *e = getenv("SRBA_HIP_WG"); if (e) c->wg_on = atoi(e) != 0; e = getenv("SRBA_HIP_WG_FROM"); if (e && atoi(e) >= 1) c->wg_from_sys = atoi(e); e = getenv("SRBA_HIP_WG256_FROM");
		if (e && atoi(e) >= 1) c->wg256_from_sys = atoi(e); e = getenv("SRBA_HIP_WG_HS"); if (e) c->wg_hs = atoi(e) != 0; }
	{ const char *e = getenv("SRBA_HIP_BIG_MIN_SYS"); if (e && atoi(e) >= 0) c->big_min_sys = atoi(e); } // tuning / test knob
	{ const char *e = getenv("SRBA_HIP_DENSE_BLOCKS"); if (e) c->dense_blocks_ok = atoi(e) != 0; }
	{ const unsigned hc = std::thread::hardware_concurrency(); c->upload_threads = (int)std::min(32u, std::max(1u, hc)); const char *e = getenv("SRBA_HIP_UPLOAD_THREADS");
		if (e) c->upload_threads = std::max(1, atoi(e)); } // host threads of srba_hip_upload_problems
	{ const char *e = getenv("SRBA_HIP_HBM_FROM_KB"); if (e) c->hbm_from_kb = atoi(e); } // (default 48; 0 = off) in batches of 1024+ capsules,
		// landmark windows whose LDS image needs this many KB or more keep their system in HBM instead: above 40 KB the LDS, not the registers, limits the wavefronts resident per CU
	{ const char *e = getenv("SRBA_HIP_DENSE_LEFT"); if (e) c->dense_left = atoi(e) != 0; } // 0: right-looking sweeps on the HBM-resident dense layout (round-2 first version)
	{ const char *e = getenv("SRBA_HIP_LIN_TERMS"); if (e) c->lin_terms = atoi(e) != 0; }
	{ const char *e = getenv("SRBA_HIP_LM_TERMS"); if (e) c->lm_terms = atoi(e) != 0; }
	{ const char *e = getenv("SRBA_HIP_ASSEMBLE_MAX_KB"); if (e) c->asm_max_kb = atoi(e); }            // capsules whose LDS image exceeds this take the unfused kernel (tests)
	{ const char *e = getenv("SRBA_HIP_DISTINCT_OBS"); if (e) c->od_on = atoi(e) != 0; }               // 0 = K4 evaluates every residual row, copies included (rounds 1-5)
	{ const char *e = getenv("SRBA_HIP_ASSEMBLE"); if (e) c->asm_on = atoi(e) != 0; }                   // 0 = srba_hip_linearize always runs the unfused kernel (Jacobian blocks through HBM)
	{ const char *e = getenv("SRBA_HIP_FLAT"); if (e) c->use_flat = atoi(e) != 0; }                     // 0 = one wavefront per capsule for the stepwise spanning-tree launch
	{ const char *e = getenv("SRBA_HIP_BIG_TIME_EVERY"); if (e && atoi(e) >= 1) c->big_time_every = atoi(e); }
	{ const char *e = getenv("SRBA_HIP_BIG_FRESH"); if (e) c->big_fresh = atoi(e) != 0; }
	{ const char *e = getenv("SRBA_HIP_SCHUR_XCD"); if (e) c->sch_xcd = atoi(e) != 0; } { const char *e = getenv("SRBA_HIP_SCHUR_SORT"); if (e) c->sch_sort = atoi(e) != 0; }
	{ const char *e = getenv("SRBA_HIP_SCHUR_WAVE"); if (e) c->sch_wave = atoi(e) != 0; }
	{ const char *e = getenv("SRBA_HIP_BIG_GANGS"); if (e && atoi(e) >= 1) c->big_gangs = std::min(atoi(e), kBigLanes); }
	{ const char *e = getenv("SRBA_HIP_BIG_LANES"); if (e && atoi(e) >= 1) { c->big_lanes_max = std::min(atoi(e), kBigLanes); c->big_gang_slots = std::min(atoi(e), srbadev::kGang); } }
	{ const char *e = getenv("SRBA_HIP_GANG_FROM_NB"); if (e) c->gang_from_nb = atoi(e); } // large capsules of one batch in flight at once
	{ const char *e = getenv("SRBA_HIP_BIG_PERSISTENT"); if (e) c->big_persistent = atoi(e) != 0; }   // 1 = the blocked Cholesky of the big path as ONE persistent launch with grid barriers
		// (k_chol_persistent) instead of one launch per panel step and per trailing update (~60 launches); measured slower, DESIGN 4c
	{ const char *e = getenv("SRBA_HIP_BIG_FUSED_STEP"); if (e) c->big_fused_step = atoi(e) != 0; }    // 0 = panel step and trailing update as two launches per 32 columns (k_chol_panel,
		// k_chol_update)
	{ const char *e = getenv("SRBA_HIP_BIG_GANG"); if (e) c->big_gang = atoi(e) != 0; }                // 0 = one host thread + stream per large window instead of the lock-step gang on one stream
		// 
	return c;
}

int srba_hip_set_params(srba_hip_ctx *c, const srba_hip_params *params) {
	if (!c) return -1;
	if (check_params(params) != 0) { c->fail(g_last_error); return -1; }
	if (params->family != c->params.family) { c->fail("srba_hip_set_params: the family of a context cannot change"); return -1; }
	if (c->n_prob && (params->solver != c->params.solver || params->noise != c->params.noise || params->extensions != c->params.extensions)) c->n_prob = 0;
		// the uploaded batch was laid out for the old solver / noise policy: upload again
	c->params = *params; make_dev_params(*params, c->dp, c->dm); c->batch_copied = false; /* (the kernels read the parameters from the device copy) */ return 0;
}

int srba_hip_destroy(srba_hip_ctx *c) {
	if (!c) return 0;
	hipSetDevice(c->device);
	for (int i = 0; i < kBigLanes; i++) { BigLane &l = c->lanes[i]; if (l.h_fetch) hipHostFree(l.h_fetch); if (l.e0) hipEventDestroy(l.e0); if (l.e1) hipEventDestroy(l.e1); if (i > 0) {
		if (l.d_part) hipFree(l.d_part); if (l.d_scal) hipFree(l.d_scal); if (l.stream) hipStreamDestroy(l.stream); } }
	if (c->d_in) hipFree(c->d_in); if (c->d_wk) hipFree(c->d_wk); if (c->d_next) hipFree(c->d_next); if (c->d_batch) hipFree(c->d_batch); if (c->d_spec) hipFree(c->d_spec); c->h_in.release();
		if (c->h_out) hipHostFree(c->h_out); if (c->ev_h2d) hipEventDestroy(c->ev_h2d); if (c->d_part) hipFree(c->d_part); if (c->d_scal) hipFree(c->d_scal);
	for (int i = 0; i < srba_hip_ctx::kRing; i++) { if (c->ring0[i]) hipEventDestroy(c->ring0[i]); if (c->ring1[i]) hipEventDestroy(c->ring1[i]); }
	if (c->ev_fork) hipEventDestroy(c->ev_fork);
	for (int k = 1; k < SRBA_NCLS; k++) { if (c->cls_done[k]) hipEventDestroy(c->cls_done[k]); if (c->cls_stream[k]) hipStreamDestroy(c->cls_stream[k]); }
	if (c->stream) hipStreamDestroy(c->stream);
	delete c; return 0;
}

void *srba_hip_stream(srba_hip_ctx *c) { return c ? (void *)c->stream : nullptr; }
double srba_hip_last_kernel_ms(srba_hip_ctx *c) { return c ? c->last_ms : 0.0; }
int srba_hip_kernel_ms_history(srba_hip_ctx *c, double *out_ms, int n) {
	if (!c || !out_ms || n < 0) return -1;
	if (hipStreamSynchronize(c->stream) != hipSuccess) return -1;
	const int have = (int)std::min<long long>(std::min<long long>(n, c->n_launches), srba_hip_ctx::kRing);
	for (int i = 0; i < have; i++) { // most recent first
		const int slot = (int)((c->n_launches - 1 - i) % srba_hip_ctx::kRing); float ms = 0;
		if (hipEventElapsedTime(&ms, c->ring0[slot], c->ring1[slot]) != hipSuccess) return i;
		out_ms[i] = ms;
	}
	return have;
}

// Class of a capsule on the workgroup path of the SE3 landmark families (k_lm_wg, srba_wg.hpp), or -1: Schur solvers, a reduced system of wg_from_sys .. 16 WG_NT_MAX scalars, and U_Ap blocks
// that fit the LDS of the workgroup shape (128 threads: four per CU, 40 KB; 256: two, 80 KB; 512: one, 159 KB). *lds_bytes: what the window needs. Windows with more blocks than a CU's
// LDS holds keep the one-wavefront kernel (or the multi-workgroup path).
static int wg_class_of(const srba_hip_ctx *c, const srba_problem_capsule &k, bool schur_solver, size_t *lds_bytes, int *panels = nullptr) {
	if (panels) *panels = 1;
	const int P = c->dm.P, L = c->dm.L; const int n_sys = P * k.n_unk_edges;
	if (!(c->wg_on && c->wg_hs && c->max_lds_kb > 0 /* (0: the test knob that sends every window to the multi-workgroup path) */ && c->dm.PD == 12 && L == 3 && schur_solver && k.n_unk_lms > 0 &&
		k.n_unk_edges > 0 && c->gang_from_nb <= 0)) return -1;
	if (n_sys < c->wg_from_sys || n_sys > 16 * srbadev::WG_NT_MAX || k.n_hap >= 65536 || k.n_unk_edges >= 32768) return -1;
	const size_t need = 8 * ((size_t)srbadev::WG_HS + (size_t)k.n_hap * (P * P + 1)); if (lds_bytes) *lds_bytes = need;
	if (need <= (size_t)160 * 1024 / (2 * SRBA_WG_WAVES) && n_sys < c->wg256_from_sys) return SRBA_CLS_WG128; // (two wavefronts: 2 x SRBA_WG_WAVES workgroups share the CU's LDS)
	if (need <= (size_t)160 * 1024 / SRBA_WG_WAVES) return SRBA_CLS_WG256;
	if (need <= (size_t)159 * 1024) return SRBA_CLS_WG512; // (nearly the whole LDS of a CU: 515 blocks)
	// more blocks than a CU's LDS holds: the window is swept in panels of equal size (ProbDesc::n_panel)
	const int cap = (int)(((size_t)159 * 1024 / 8 - (size_t)srbadev::WG_HS) / (size_t)(P * P + 1)), np = (k.n_hap + cap - 1) / cap, psize = (k.n_hap + np - 1) / np;
	if (np > 16) return -1;
	if (panels) *panels = np; if (lds_bytes) *lds_bytes = 8 * ((size_t)srbadev::WG_HS + (size_t)psize * (P * P + 1));
	return SRBA_CLS_WG512;
}
static int upload_problems_impl(srba_hip_ctx *c, const srba_problem_capsule *caps, int n);
int srba_hip_upload_problems(srba_hip_ctx *c, const srba_problem_capsule *caps, int n) { // no C++ exception crosses the C ABI
	try { return upload_problems_impl(c, caps, n); }
	catch (const std::exception &e) { if (c) { c->n_prob = 0; c->fail(std::string("upload: ") + e.what()); } return -1; }
	catch (...) { if (c) { c->n_prob = 0; c->fail("upload: unknown exception"); } return -1; }
}
// every observation row of the relative-pose SE2 family is valid (its Jacobian blocks have no failure case): the flags the unfused kernel rewrites at every call are set once per upload for the fused
	// one
static int set_asm_flags(srba_hip_ctx *c) {
	if (c->asm_flags_set) return 0;
	if (c->n_valid_total) HIPCHK(c, hipMemsetD32Async((hipDeviceptr_t)(c->d_wk + c->off_valid), 1, (size_t)c->n_valid_total, c->stream));
	if (c->n_bp_total) HIPCHK(c, hipMemsetD8Async((hipDeviceptr_t)(c->d_wk + c->off_bp_ok), 1, (size_t)c->n_bp_total, c->stream));
	c->asm_flags_set = true; return 0;
}
static int upload_problems_impl(srba_hip_ctx *c, const srba_problem_capsule *caps, int n) {
	if (!c || !caps || n <= 0) { if (c) c->fail("upload: bad arguments"); return -1; }
	HIPCHK(c, hipSetDevice(c->device));
	c->big_chol_ms = c->big_chol_flops = 0; c->big_chol_count = c->big_chol_seqs = 0; c->big_chol_nmax = 0; c->big_t_flops = 0; c->big_t_seqs = 0;
	c->n_prob = 0; // whatever was uploaded before stops being launchable / readable now: a failed upload leaves the context empty, not half-updated
	static const bool host_timing = getenv("SRBA_HIP_HOST_TIMING") != nullptr; static double acc[4] = {0, 0, 0, 0}; static long long calls = 0; auto now = []() { return std::chrono::duration<double,
		std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }; const double ht0 = host_timing ? now() : 0; double ht1 = 0, ht2 = 0;
	const int P = c->dm.P, L = c->dm.L, O = c->dm.O, PD = c->dm.PD, PDX = c->dm.PDX();
	const bool schur_solver = c->params.solver != SRBA_SOLVER_NO_SCHUR_SPARSE_CHOL;
	c->desc.assign(n, ProbDesc()); srba_batch_stats &st = c->stats; std::memset(&st, 0, sizeof(st)); st.n_problems = n;
	// ---- pass 1: descriptors and totals
	long long t_ptab = 0, t_hapo = 0, t_schl = 0, t_hrec = 0, t_edge = 0, t_unk = 0, t_ulm = 0, t_klm = 0, t_pair = 0, t_path = 0, t_obs = 0, t_valid = 0, t_bp = 0, t_bf = 0, t_hap = 0, t_hapt = 0,
		t_hf = 0, t_hft = 0, t_hapf = 0, t_hapft = 0, t_sch = 0, t_req = 0, t_scal = 0, t_yw = 0, t_dense = 0, t_vb = 0; bool any_vb = false;
	std::vector<int> cls(n, 0), big_lds(n, 0); int cls_nbmax[SRBA_NCLS] = {0}; size_t wg_lds[3] = {0, 0, 0}; std::vector<Symbolic> sym(n); long long t_spcol = 0, t_sprow = 0, t_spitem = 0,
		t_spfill = 0;
	// capsules whose system cannot fit one wavefront's LDS even as bare numbers (more than 63 block rows) but is no deep-window system either: a handful go to the
	// multi-workgroup path; when the batch holds many, they keep one wavefront each with the system in HBM (see below)
	bool many_mid = false; { int cnt = 0; for (int p = 0; p < n; p++) { const int nsys = (schur_solver && caps[p].n_unk_lms > 0 && caps[p].n_unk_edges > 0) ? P * caps[p].n_unk_edges : P *
		caps[p].n_unk_edges + L * caps[p].n_unk_lms; if ((nsys + 2) / 3 > 63 && nsys <= c->big_min_sys) cnt++; } many_mid = cnt > 32; }
	std::vector<const char *> why(n, nullptr);
	parallel_ranges(n, c->upload_threads, [&](int b, int e, int) { // validation and the block-sparse symbolic factorisation (the expensive part of this pass) of every capsule
		for (int p = b; p < e; p++) {
			const srba_problem_capsule &k = caps[p];
			if ((why[p] = validate_capsule(k)) != nullptr) continue;
			ProbDesc t; t.nK = k.n_unk_edges; t.nF = k.n_unk_lms; t.n_scal = P * t.nK + L * t.nF; t.n_sys = (schur_solver && t.nF > 0 && t.nK > 0) ? P * t.nK : t.n_scal; t.nb = (t.n_sys + 2) / 3;
			if (schur_solver && t.nK == 0) continue;
			const bool wg = wg_class_of(c, k, schur_solver, nullptr) >= 0; // (the workgroup path needs no symbolic analysis)
			if (t.n_sys <= c->big_min_sys && !wg) symbolic_factor(k, t, P, L, !schur_solver, sym[p]);
		}
	});
	for (int p = 0; p < n; p++) {
		const srba_problem_capsule &k = caps[p]; ProbDesc &d = c->desc[p];
		if (why[p]) { c->fail(std::string("upload: malformed capsule (") + why[p] + ")"); return -1; }
		d.n_edges = k.n_edges; d.nK = k.n_unk_edges; d.nF = k.n_unk_lms; d.n_klm = k.n_known_lms; d.n_pairs = k.n_pairs; d.n_obs = k.n_obs; d.n_valid = k.n_valid; d.n_bp = k.n_bp; d.n_bf = k.n_bf;
		d.n_hap = k.n_hap; d.n_hf = k.n_hf; d.n_hapf = k.n_hapf; d.n_sch = k.n_sch_terms; d.n_hapt = k.n_hap_terms;
		{ int sp = k.n_hap_terms; for (int b = 0; b <= k.n_hap; b++) if (2 * (long long)k.hap_term_off[b] >= k.n_hap_terms) { sp = k.hap_term_off[b]; break; } d.hapt_split = sp; }
			// k_lm_run2: where the second wavefront's share of the U_Ap terms begins (a block boundary)
		d.n_scal = P * d.nK + L * d.nF; d.n_sys = (schur_solver && d.nF > 0 && d.nK > 0) ? P * d.nK : d.n_scal;
		if (schur_solver && d.nK == 0) { c->fail("upload: Schur solvers need at least one unknown kf2kf edge (the reference has the same restriction, schur.h:34)"); return -1; }
		int nreq = 0; for (int i = 0; i < 2 * k.n_pairs; i++) nreq += k.pose_required[i] ? 1 : 0; d.n_req = nreq;
		d.o_edge = t_edge; d.o_unk = t_unk; d.o_ulm = t_ulm; d.o_klm = t_klm; d.o_pair = t_pair; d.o_ppoff = t_pair + p; d.o_path = t_path; d.o_obs = t_obs; d.o_valid = t_valid;
		d.o_bp = t_bp; d.o_colp = t_unk + p; d.o_bf = t_bf; d.o_colf = t_ulm + p; d.o_hap = t_hap; d.o_hapoff = t_hap + p; d.o_hapt = t_hapt; d.o_hf = t_hf; d.o_hfoff = t_hf + p; d.o_hft = t_hft;
		d.o_hapf = t_hapf; d.o_hapfoff = t_hapf + p; d.o_hapft = t_hapft; d.o_sch = t_sch; d.o_lmoff = t_ulm + p; d.o_req = t_req; d.o_scal = t_scal; d.o_yw = t_yw; d.o_dense = t_dense;
		d.nb = (d.n_sys + 2) / 3;
		size_t wg_need = 0; int wg_panels = 1; const int wg_cls = wg_class_of(c, k, schur_solver, &wg_need, &wg_panels); const bool to_wg = wg_cls >= 0; // one workgroup, tile system in HBM,
			// U_Ap blocks in LDS, matrix cores (srba_wg.hpp)
		const bool surely_big = d.n_sys > c->big_min_sys || to_wg; // far beyond what one wavefront's LDS holds (or a workgroup window): dense system on the multi-workgroup path,
			// no block-sparse symbolic analysis
		if (!surely_big) { /* sym[p]: computed above */ }
		else { Symbolic &y = sym[p]; y.col_off.assign(d.nb + 1, 0); y.item_off.assign(d.nb + 1, 0); y.rptr.assign(d.nb + 1, 0); y.perm.resize(d.nb); for (int q = 0; q < d.nb; q++) y.perm[q] = q;
			y.hap_dst.assign((size_t)k.n_hap * (P / 3) * (P / 3), 0); y.hapf_dst.assign((size_t)k.n_hapf * (P / 3), 0); y.hf_dst.assign(k.n_hf, 0); y.aligned = true; }
		d.nnzoff = (int)sym[p].row.size(); d.n_items = (int)sym[p].tgt.size(); d.aligned = sym[p].aligned ? 1 : 0; d.dense_blocks = 0;
		size_t n_ints = 2 * ((size_t)d.nb + 1) + 2 * (size_t)d.nnzoff + (size_t)d.n_items + (size_t)d.nb;
		bool packable = d.nb + d.nnzoff < 16384 && d.nb < 16384 && sym[p].max_cn < 512; // item / row-entry words of the LDS copy
		size_t tri_n = 9 * (size_t)d.nb + 9 * (size_t)d.nnzoff + 3 * (size_t)d.nb + (n_ints + 1) / 2; // diag | off | rhs | symbolic ints
		const bool rel_family = c->params.family == SRBA_SE2_RELPOSE2D || c->params.family == SRBA_SE3_RELPOSE3D || c->dm.P == 3;
		const bool to_gang = c->gang_from_nb > 0 && !rel_family && schur_solver && caps[p].n_unk_lms > 0 && caps[p].n_unk_edges > 0 && d.nb >= c->gang_from_nb;
			// big enough for the lock-step multi-workgroup path to beat one wavefront // relative-pose and SE2 families: their kernels carry the sparse solver only (one 3x3 block per edge: the
			// sparse image fits)
		if (!surely_big && c->dense_blocks_ok && !rel_family) { // nearly full factor: the dense block layout (numbers + the permutation only) is smaller than the sparse one with its item list
			const size_t nnz_d = (size_t)d.nb * (d.nb - 1) / 2, tri_d = 9 * (size_t)d.nb + 9 * nnz_d + 3 * (size_t)d.nb + ((size_t)d.nb + 1) / 2;
			if (tri_d < tri_n && tri_d * 8 <= 152 * 1024) { symbolic_dense(k, d, P, L, !schur_solver, sym[p]); d.dense_blocks = 1; d.nnzoff = (int)nnz_d; d.n_items = 0;
				d.aligned = sym[p].aligned ? 1 : 0; n_ints = d.nb; packable = true; tri_n = tri_d; }
		}
		// LDS footprint x residency time is what bounds the batch: capsules are grouped in fine size classes so that each launch
		// reserves little more LDS per wavefront than its capsules need
		size_t bytes = tri_n * 8;
		static const int kClsKB[SRBA_NLDS] = {6, 8, 10, 12, 14, 16, 20, 24, 28, 32, 40, 48, 56, 64, 80, 96, 112, 128, 152};
		cls[p] = SRBA_NCLS - 1; for (int q = 0; q < SRBA_NLDS && packable && !surely_big && !to_gang && kClsKB[q] <= c->max_lds_kb; q++) if (bytes <= (size_t)kClsKB[q] * 1024) { cls[p] = q; break; }
		long long wave_ws = 0; // doubles of HBM workspace of a capsule that keeps one wavefront but holds its (dense block) system in HBM
		d.hap_chunked = 0; d.n_hrec = k.n_hap; d.o_hrec = t_hrec; t_hrec += k.n_hap; // K6 work records: one per block
		d.hs_lds = 0; d.o_hapo = t_hapo; d.o_schl = t_schl; d.n_panel = 1; d.o_ptab = t_ptab;
		if (to_wg) { d.n_panel = wg_panels; t_ptab += 3 * (wg_panels + 1); const int nt = (d.n_sys + srbadev::WT - 1) / srbadev::WT; d.dense_blocks = 3; d.nnzoff = 0; d.n_items = 0;
			wave_ws = srbadev::wg_ws_doubles(nt);
			cls[p] = wg_cls; d.hs_lds = 1; wg_lds[wg_cls - SRBA_NLDS] = std::max(wg_lds[wg_cls - SRBA_NLDS], wg_need); t_hapo += k.n_hap_terms; t_schl += k.n_sch_terms; }
		if ((cls[p] == SRBA_NCLS - 1 ? many_mid : (c->hbm_from_kb > 0 && n >= 1024 && kClsKB[cls[p]] >= c->hbm_from_kb && d.nb <= 168)) && !surely_big && !to_gang && !to_wg && c->dense_blocks_ok &&
			!rel_family) {
			// Does not fit any LDS class and the batch has many like it: the multi-workgroup path would run them a few at a time from the host. They stay on the
			// one-wavefront kernel with the dense block system in an HBM workspace (slow per capsule, but thousands run side by side).
			const size_t nnz_d = (size_t)d.nb * (d.nb - 1) / 2;
			symbolic_dense(k, d, P, L, !schur_solver, sym[p]); d.dense_blocks = 2; d.nnzoff = (int)nnz_d; d.n_items = 0; d.aligned = sym[p].aligned ? 1 : 0;
			tri_n = ((size_t)d.nb + 1) / 2 + 16 + (c->dense_left ? 21 * (size_t)d.nb : 0); bytes = tri_n * 8; /* LDS: the permutation (+ two rows of the factor and the right-hand side: left-looking
				sweeps) */
			cls[p] = 0; while (cls[p] < SRBA_NLDS - 1 && bytes > (size_t)kClsKB[cls[p]] * 1024) cls[p]++;
			wave_ws = 12 * (long long)d.nb + 9 * (long long)nnz_d;
		}
		const int n_row_store = (int)sym[p].row.size(); // entries of the row / row-view arrays in the input arena (none in the dense layout)
		d.o_spcol = t_spcol; d.o_sprow = t_sprow; d.o_spitem = t_spitem; d.o_spperm = t_spcol - p;
		t_spcol += d.nb + 1; t_sprow += n_row_store; t_spitem += (long long)sym[p].tgt.size();
		d.n_fill = (int)sym[p].fill.size(); d.o_spfill = t_spfill; t_spfill += d.n_fill;
		d.dense_in_lds = (cls[p] < SRBA_NLDS && d.dense_blocks < 2) ? 1 : 0; if (cls[p] < SRBA_NLDS) cls_nbmax[cls[p]] = std::max(cls_nbmax[cls[p]], (int)tri_n);
		const long long big_ld = cls[p] < SRBA_NCLS - 1 ? 0 : ((d.n_sys + srbadev::CB - 1) / srbadev::CB) * srbadev::CB; // dense path: ld x ld matrix + ld x CB diagonal factors + rhs + y
		int nyw = 0; if (k.n_sch_terms > 0) for (int b = 0; b < k.n_hap; b++) if (k.hap_i[b] == k.hap_j[b]) nyw += k.sch_term_off[b + 1] - k.sch_term_off[b];
		d.n_vb = 0; d.o_vb = t_vb; // Schur reduction, a wavefront per U_Ap block (multi-workgroup class): one work record per block that has terms
		if (c->sch_wave && cls[p] == SRBA_NCLS - 1 && k.n_sch_terms > 0) { any_vb = true; for (int b = 0; b < k.n_hap; b++) if (k.sch_term_off[b + 1] > k.sch_term_off[b]) d.n_vb++;
			t_vb += d.n_vb; }
		long long npath_needed = 0; for (int q = 0; q < k.n_pairs; q++) if (k.pair_needed[q]) { st.n_pairs_needed++; npath_needed += k.pair_path_off[q + 1] - k.pair_path_off[q]; }
		st.n_path_needed += npath_needed;
		t_edge += k.n_edges; t_unk += d.nK; t_ulm += d.nF; t_klm += d.n_klm; t_pair += k.n_pairs; t_path += k.n_path; t_obs += k.n_obs; t_valid += k.n_valid; t_bp += k.n_bp; t_bf += k.n_bf;
		t_hap += k.n_hap; t_hapt += k.n_hap_terms; t_hf += k.n_hf; t_hft += k.n_hf_terms; t_hapf += k.n_hapf; t_hapft += k.n_hapf_terms; t_sch += k.n_sch_terms; t_req += nreq; t_scal += d.n_scal;
			t_yw += nyw; t_dense += (big_ld * big_ld + big_ld * srbadev::CB + 2 * big_ld + wave_ws + 1) & ~1LL; /* (even: the tile systems of the workgroup path move 16 bytes at a time) */ big_lds[p]
			= (int)big_ld;
	}
	st.n_edges = t_edge; st.n_unk_edges = t_unk; st.n_unk_lms = t_ulm; st.n_pairs = t_pair; st.n_path = t_path; st.n_obs = t_obs; st.n_bp = t_bp; st.n_bf = t_bf; st.n_hap = t_hap;
		st.n_hap_terms = t_hapt;
	st.n_hf_terms = t_hft; st.n_hapf_terms = t_hapft; st.n_sch_terms = t_sch; st.n_scalars = t_scal;
	c->tot_edge = t_edge; c->tot_ulm = t_ulm;
	// ---- input arena layout
	Arena in; struct { size_t desc, edge0, ulm0, klm, obs_z, pair_path_off, path_edge, obs_pose, obs_lm, obs_valid, bp_col, bp_res, bp_A, bp_D, bp_lm, colp_off, bf_col, bf_res, bf_pose, colf_off,
		hap_i, hap_j, hap_term_off, hap_t1, hap_t2, hap_tblk, hf_i, hf_j, hf_term_off, hf_t1, hf_t2, hapf_i, hapf_j, hapf_term_off, hapf_t1, hapf_t2, hap_diag, hf_diag, sch_term_off, sch_b1, sch_b2,
			sch_lm, sch_yw, sch_tblk, sch_vb, sch_rec,
		lm_hapf_off, lm_hapf_idx, req_idx, need_idx, need_rec, obs_rec, od_tab, pair_needed, pose_req, bp_normal, order, sp_col_off, sp_row, sp_item_off, sp_tgt, sp_rptr, sp_rcol, sp_perm, sp_fill, hap_rec,
			hapo, schl, ptab, hap_dst, hapf_dst, hf_dst, asm_rec, asm_desc, asm_list; } o;
	o.desc = in.add(sizeof(ProbDesc) * n);
	o.edge0 = in.add(8 * t_edge * PDX); o.ulm0 = in.add(8 * t_ulm * L); o.klm = in.add(8 * t_klm * L); o.obs_z = in.add(8 * t_obs * O);
	o.pair_path_off = in.add(4 * (t_pair + n)); o.path_edge = in.add(4 * t_path); o.obs_pose = in.add(4 * t_obs); o.obs_lm = in.add(4 * t_obs); o.obs_valid = in.add(4 * t_obs);
	o.bp_col = in.add(4 * t_bp); o.bp_res = in.add(4 * t_bp); o.bp_A = in.add(4 * t_bp); o.bp_D = in.add(4 * t_bp); o.bp_lm = in.add(4 * t_bp); o.colp_off = in.add(4 * (t_unk + n));
	o.bf_col = in.add(4 * t_bf); o.bf_res = in.add(4 * t_bf); o.bf_pose = in.add(4 * t_bf); o.colf_off = in.add(4 * (t_ulm + n));
	o.hap_i = in.add(4 * t_hap); o.hap_j = in.add(4 * t_hap); o.hap_term_off = in.add(4 * (t_hap + n)); o.hap_t1 = in.add(4 * t_hapt); o.hap_t2 = in.add(4 * t_hapt); o.hap_tblk = in.add(4 * t_hapt);
	o.hf_i = in.add(4 * t_hf); o.hf_j = in.add(4 * t_hf); o.hf_term_off = in.add(4 * (t_hf + n)); o.hf_t1 = in.add(4 * t_hft); o.hf_t2 = in.add(4 * t_hft);
	o.hapf_i = in.add(4 * t_hapf); o.hapf_j = in.add(4 * t_hapf); o.hapf_term_off = in.add(4 * (t_hapf + n)); o.hapf_t1 = in.add(4 * t_hapft); o.hapf_t2 = in.add(4 * t_hapft);
	o.hap_diag = in.add(4 * t_unk); o.hf_diag = in.add(4 * t_ulm);
	o.sch_term_off = in.add(4 * (t_hap + n)); o.sch_b1 = in.add(4 * t_sch); o.sch_b2 = in.add(4 * t_sch); o.sch_lm = in.add(4 * t_sch); o.sch_yw = in.add(4 * t_sch); o.sch_tblk = in.add(4 * t_sch);
	o.sch_vb = in.add(16 * (size_t)t_vb); o.sch_rec = in.add(any_vb ? 16 * (size_t)t_sch : 0);
	o.lm_hapf_off = in.add(4 * (t_ulm + n)); o.lm_hapf_idx = in.add(4 * t_hapf); o.req_idx = in.add(4 * t_req); o.need_idx = in.add(4 * t_pair);
		o.need_rec = in.add(4 * 5 * std::max<long long>(t_pair, 1)); o.obs_rec = in.add(4 * 5 * std::max<long long>(t_obs, 1)); o.od_tab = in.add(4 * 12 * std::max<long long>(t_valid, 1)); o.pair_needed = in.add(t_pair); o.pose_req = in.add(2 * t_pair);
		o.bp_normal = in.add(t_bp); o.order = in.add(4 * (size_t)n);
	o.sp_col_off = in.add(4 * t_spcol); o.sp_row = in.add(4 * t_sprow); o.sp_item_off = in.add(4 * t_spcol); o.sp_tgt = in.add(4 * t_spitem); o.sp_rptr = in.add(4 * t_spcol);
		o.sp_rcol = in.add(4 * t_sprow); o.sp_perm = in.add(4 * t_spcol); o.hap_rec = in.add(4 * 3 * std::max<long long>(t_hrec, 1)); o.hapo = in.add(4 * 3 * t_hapo); o.schl = in.add(4 * 4 * t_schl);
		o.ptab = in.add(4 * std::max<long long>(t_ptab, 1)); o.sp_fill = in.add(4 * std::max<long long>(t_spfill, 1));
	o.hap_dst = in.add(4 * t_hap * (P / 3) * (P / 3)); o.hapf_dst = in.add(4 * t_hapf * (P / 3)); o.hf_dst = in.add(4 * t_hf);
	bool asm_fam = c->asm_on && c->params.family == SRBA_SE2_RELPOSE2D;
	if (asm_fam && c->dp.noise == SRBA_NOISE_CONSTANT_MATRIX) for (int i = 0; i < 3; i++) for (int j = 0; j < i; j++) if (c->dp.lambda[3 * i + j] != c->dp.lambda[3 * j + i]) asm_fam = false;
		// the fused kernel sums the upper triangle of J^t Lambda J only
	// fused normal-equations kernel (srba_assemble.hpp): 32-byte row records, room per capsule known from its sizes
	std::vector<long long> asm_ro(asm_fam ? n + 1 : 1, 0); if (asm_fam) for (int p = 0; p < n; p++) asm_ro[p + 1] = asm_ro[p] + srbadev::asm_rec_room(caps[p].n_obs, caps[p].n_bp);
	o.asm_rec = in.add(asm_fam ? sizeof(srbadev::AsmRec) * (size_t)std::max<long long>(asm_ro[n], 1) : 0);
		o.asm_desc = in.add(asm_fam ? sizeof(srbadev::AsmDesc) * (size_t)srbadev::ASM_MAX_WPW * (size_t)n : 0); /* (at most one bin per capsule) */ o.asm_list = in.add(asm_fam ? 4 * (size_t)n :
		0);
	std::vector<int> asm_rounds(asm_fam ? n : 0, 0); // per capsule: passes of the fused kernel (0: it does not fit its packed records)
	in.add(0);
	if (c->h2d_pending) { HIPCHK(c, hipEventSynchronize(c->ev_h2d)); c->h2d_pending = false; } // (an upload nobody waited for may still be reading the staging buffer)
	if (c->h_in_cap < in.size + 256) { c->h_in.release(); c->h_in_cap = 0; const size_t want = in.size + 256 <= srba_hip_ctx::kPinnedMax / 2 ? 2 * (in.size + 256) : in.size + 256;
		// uninitialised: cleared below, in parallel
		if (want <= srba_hip_ctx::kPinnedMax && hipHostMalloc((void **)&c->h_in.p, want, hipHostMallocDefault) == hipSuccess) c->h_in.pinned = true; else { (void)hipGetLastError();
			c->h_in.p = new char[want]; c->h_in.pinned = false; }
		c->h_in_cap = want; }
	char *h = c->h_in.get();
	{ const size_t tot = in.size + 256, slab = (size_t)4 << 20; const int nslab = (int)((tot + slab - 1) / slab);
	  parallel_ranges(std::max(nslab, 512), nslab > 1 ? c->upload_threads : 1, [&](int b, int e, int) { for (int q = b; q < e && q < nslab; q++) std::memset(h + (size_t)q * slab, 0, std::min(slab,
	  	tot - (size_t)q * slab)); }); }
	c->in_off_edge0 = o.edge0; c->in_off_ulm0 = o.ulm0; c->h_off_order = o.order;
	if (host_timing) ht1 = now();
	// ---- pass 2: pack
#define CPY(dstoff, elem_off, src, count, T) do { if ((count) > 0) std::memcpy(h + (dstoff) + sizeof(T) * (size_t)(elem_off), (src), sizeof(T) * (size_t)(count)); } while (0)
	std::vector<int64_t> acc_blocks(std::max(1, c->upload_threads), 0), acc_items(std::max(1, c->upload_threads), 0);
	parallel_ranges(n, c->upload_threads, [&](int p_begin, int p_end, int thread) {
	for (int p = p_begin; p < p_end; p++) {
		const srba_problem_capsule &k = caps[p]; const ProbDesc &d = c->desc[p];
		if (PDX == PD) { CPY(o.edge0, d.o_edge * PD, k.edge_pose, (size_t)k.n_edges * PD, double); }
		else { double *e = (double *)(h + o.edge0) + d.o_edge * PDX; for (int q = 0; q < k.n_edges; q++) { const double *s3 = k.edge_pose + 3 * (size_t)q; e[5 * q] = s3[0]; e[5 * q + 1] = s3[1];
			e[5 * q + 2] = s3[2]; e[5 * q + 3] = std::cos(s3[2]); e[5 * q + 4] = std::sin(s3[2]); } }
		CPY(o.ulm0, d.o_ulm * L, k.ulm_pos, (size_t)d.nF * L, double); CPY(o.klm, d.o_klm * L, k.klm_pos, (size_t)d.n_klm * L, double); CPY(o.obs_z, d.o_obs * O, k.obs_z, (size_t)k.n_obs * O, double);
		CPY(o.pair_path_off, d.o_ppoff, k.pair_path_off, k.n_pairs + 1, int32_t); CPY(o.path_edge, d.o_path, k.path_edge, k.n_path, int32_t);
		CPY(o.obs_pose, d.o_obs, k.obs_pose, k.n_obs, int32_t); CPY(o.obs_lm, d.o_obs, k.obs_lm, k.n_obs, int32_t); CPY(o.obs_valid, d.o_obs, k.obs_valid, k.n_obs, int32_t);
		CPY(o.bp_col, d.o_bp, k.bp_col, k.n_bp, int32_t); CPY(o.bp_res, d.o_bp, k.bp_res, k.n_bp, int32_t); CPY(o.bp_A, d.o_bp, k.bp_A, k.n_bp, int32_t); CPY(o.bp_D, d.o_bp, k.bp_D, k.n_bp, int32_t);
			CPY(o.bp_lm, d.o_bp, k.bp_lm, k.n_bp, int32_t);
		CPY(o.colp_off, d.o_colp, k.colp_off, d.nK + 1, int32_t);
		CPY(o.bf_col, d.o_bf, k.bf_col, k.n_bf, int32_t); CPY(o.bf_res, d.o_bf, k.bf_res, k.n_bf, int32_t); CPY(o.bf_pose, d.o_bf, k.bf_pose, k.n_bf, int32_t);
		if (k.colf_off) CPY(o.colf_off, d.o_colf, k.colf_off, d.nF + 1, int32_t);
		CPY(o.hap_i, d.o_hap, k.hap_i, k.n_hap, int32_t); CPY(o.hap_j, d.o_hap, k.hap_j, k.n_hap, int32_t); CPY(o.hap_term_off, d.o_hapoff, k.hap_term_off, k.n_hap + 1, int32_t); CPY(o.hap_t1,
			d.o_hapt, k.hap_t1, k.n_hap_terms, int32_t); CPY(o.hap_t2, d.o_hapt, k.hap_t2, k.n_hap_terms, int32_t);
		{ int32_t *tb = (int32_t *)(h + o.hap_tblk) + d.o_hapt; for (int b = 0; b < k.n_hap; b++) for (int t = k.hap_term_off[b]; t < k.hap_term_off[b + 1]; t++) tb[t] = b; }
		CPY(o.hf_i, d.o_hf, k.hf_i, k.n_hf, int32_t); CPY(o.hf_j, d.o_hf, k.hf_j, k.n_hf, int32_t); if (k.hf_term_off) CPY(o.hf_term_off, d.o_hfoff, k.hf_term_off, k.n_hf + 1, int32_t); CPY(o.hf_t1,
			d.o_hft, k.hf_t1, k.n_hf_terms, int32_t); CPY(o.hf_t2, d.o_hft, k.hf_t2, k.n_hf_terms, int32_t);
		CPY(o.hapf_i, d.o_hapf, k.hapf_i, k.n_hapf, int32_t); CPY(o.hapf_j, d.o_hapf, k.hapf_j, k.n_hapf, int32_t); if (k.hapf_term_off) CPY(o.hapf_term_off, d.o_hapfoff, k.hapf_term_off,
			k.n_hapf + 1, int32_t); CPY(o.hapf_t1, d.o_hapft, k.hapf_t1, k.n_hapf_terms, int32_t); CPY(o.hapf_t2, d.o_hapft, k.hapf_t2, k.n_hapf_terms, int32_t);
		CPY(o.hap_diag, d.o_unk, k.hap_diag, d.nK, int32_t); CPY(o.hf_diag, d.o_ulm, k.hf_diag, d.nF, int32_t);
		if (k.n_sch_terms > 0) {
			CPY(o.sch_term_off, d.o_hapoff, k.sch_term_off, k.n_hap + 1, int32_t); CPY(o.sch_b1, d.o_sch, k.sch_b1, k.n_sch_terms, int32_t); CPY(o.sch_b2, d.o_sch, k.sch_b2, k.n_sch_terms, int32_t);
				CPY(o.sch_lm, d.o_sch, k.sch_lm, k.n_sch_terms, int32_t);
			int32_t *yw = (int32_t *)(h + o.sch_yw) + d.o_sch, *tbk = (int32_t *)(h + o.sch_tblk) + d.o_sch; int cnt = 0;
			for (int b = 0; b < k.n_hap; b++) for (int t = k.sch_term_off[b]; t < k.sch_term_off[b + 1]; t++) { yw[t] = (k.hap_i[b] == k.hap_j[b]) ? cnt++ : -1; tbk[t] = b; }
			if (d.n_vb > 0) { // work records of kb_schur_reduce_wave {first term, end term, block, 0}, longest list first (stable: equal lengths keep the block order); packed term records
				int32_t *vb = (int32_t *)(h + o.sch_vb) + 4 * d.o_vb, *rec = (int32_t *)(h + o.sch_rec) + 4 * d.o_sch;
				std::vector<int32_t> ord; ord.reserve(d.n_vb); for (int b = 0; b < k.n_hap; b++) if (k.sch_term_off[b + 1] > k.sch_term_off[b]) ord.push_back(b);
				if (c->sch_sort) std::stable_sort(ord.begin(), ord.end(), [&](int x, int y) { return k.sch_term_off[x + 1] - k.sch_term_off[x] > k.sch_term_off[y + 1] - k.sch_term_off[y]; });
				for (int q = 0; q < d.n_vb; q++) { const int b = ord[q]; vb[4 * q] = k.sch_term_off[b]; vb[4 * q + 1] = k.sch_term_off[b + 1]; vb[4 * q + 2] = b; vb[4 * q + 3] = 0; }
				for (int t = 0; t < k.n_sch_terms; t++) { rec[4 * t] = k.sch_lm[t]; rec[4 * t + 1] = k.sch_b1[t]; rec[4 * t + 2] = k.sch_b2[t]; rec[4 * t + 3] = yw[t]; }
			}
		} // else: zeros = empty term lists
		if (k.lm_hapf_off) CPY(o.lm_hapf_off, d.o_lmoff, k.lm_hapf_off, d.nF + 1, int32_t); CPY(o.lm_hapf_idx, d.o_hapf, k.lm_hapf_idx, k.n_hapf, int32_t);
		{ int32_t *nd = (int32_t *)(h + o.need_idx) + d.o_pair, *nr = (int32_t *)(h + o.need_rec) + 5 * d.o_pair; int cnt = 0, flat = 1;
		  for (int i = 0; i < k.n_pairs; i++) if (k.pair_needed[i]) {
			const int pb = k.pair_path_off[i], pl = k.pair_path_off[i + 1] - pb; if (pl > 4) flat = 0;
			nr[5 * cnt] = i; for (int u = 0; u < 4; u++) nr[5 * cnt + 1 + u] = u < pl ? k.path_edge[pb + u] : -1;
			nd[cnt++] = i; }
		  c->desc[p].n_need = cnt; c->desc[p].need_flat = flat; }
		{ int32_t *orc = (int32_t *)(h + o.obs_rec) + 5 * d.o_obs; // per observation: how its pose is obtained inside a trial (Worker::phase_residuals_fused)
		  for (int i = 0; i < k.n_obs; i++) { const int ip = k.obs_pose[i]; int32_t *r = orc + 5 * i; r[1] = r[2] = r[3] = r[4] = -1;
			if (ip < 0) { r[0] = -2; continue; }
			const int pr = ip >> 1, pb = k.pair_path_off[pr], pl = k.pair_path_off[pr + 1] - pb;
			if (k.pair_needed[pr] && pl <= 4) { r[0] = ip & 1; for (int u = 0; u < pl; u++) r[1 + u] = k.path_edge[pb + u]; } else { r[0] = -1; r[1] = ip; } } }
		{ // one record per DISTINCT observation (validity slot): its rows (at most three, else the capsule keeps the row-by-row residual phase), its pose, the obs_rec of the first row
		  int32_t *od = (int32_t *)(h + o.od_tab) + 12 * d.o_valid; const int32_t *orc = (const int32_t *)(h + o.obs_rec) + 5 * d.o_obs; bool ok = k.n_valid > 0;
		  for (int v = 0; v < k.n_valid; v++) { int32_t *r = od + 12 * v; r[0] = r[1] = r[2] = -1; for (int q = 3; q < 12; q++) r[q] = 0; }
		  for (int i = 0; i < k.n_obs && ok; i++) { const int v = k.obs_valid[i]; if (v < 0 || v >= k.n_valid) { ok = false; break; } int32_t *r = od + 12 * v;
			const int q = r[0] < 0 ? 0 : r[1] < 0 ? 1 : r[2] < 0 ? 2 : 3; if (q == 3) { ok = false; break; }
			if (q == 0) { r[3] = k.obs_pose[i]; for (int u = 0; u < 5; u++) r[4 + u] = orc[5 * i + u]; }
			else { const int f = r[0]; if (k.obs_pose[i] != k.obs_pose[f] || k.obs_lm[i] != k.obs_lm[f] || std::memcmp(k.obs_z + (size_t)i * O, k.obs_z + (size_t)f * O, sizeof(double) * O) != 0) { ok = false;
				break; } } // (rows of one observation are copies of one another: anything else keeps the row-by-row phase)
			r[q] = i; }
		  for (int v = 0; v < k.n_valid && ok; v++) if (od[12 * v] < 0) ok = false; // a slot without a row
		  c->desc[p].od_ok = (ok && c->od_on) ? 1 : 0; }
		{ int32_t *rq = (int32_t *)(h + o.req_idx) + d.o_req; int cnt = 0; for (int i = 0; i < 2 * k.n_pairs; i++) if (k.pose_required[i]) rq[cnt++] = i; }
		CPY(o.pair_needed, d.o_pair, k.pair_needed, k.n_pairs, uint8_t); CPY(o.pose_req, 2 * d.o_pair, k.pose_required, 2 * (size_t)k.n_pairs, uint8_t); CPY(o.bp_normal, d.o_bp, k.bp_normal, k.n_bp,
			uint8_t);
		CPY(o.sp_col_off, d.o_spcol, sym[p].col_off.data(), d.nb + 1, int32_t); CPY(o.sp_row, d.o_sprow, sym[p].row.data(), sym[p].row.size(), int32_t);
		CPY(o.sp_item_off, d.o_spcol, sym[p].item_off.data(), d.nb + 1, int32_t); CPY(o.sp_perm, d.o_spperm, sym[p].perm.data(), d.nb, int32_t); { std::vector<int32_t> &tg = sym[p].tgt;
			const std::vector<int32_t> &ab = sym[p].ab; for (size_t q = 0; q < tg.size(); q++) tg[q] = (int32_t)((((unsigned)(tg[q] >= 0 ? d.nb + tg[q] : -1 - tg[q])) << 18) | (((unsigned)ab[q] >>
			16) << 9) | ((unsigned)ab[q] & 0xffffu)); } /* packed words (only capsules with packable indices reach the kernels that read them) */ CPY(o.sp_tgt, d.o_spitem, sym[p].tgt.data(),
			sym[p].tgt.size(), int32_t);
		CPY(o.sp_rptr, d.o_spcol, sym[p].rptr.data(), d.nb + 1, int32_t); { std::vector<int32_t> &rc = sym[p].rcol; const std::vector<int32_t> &rbk = sym[p].rblk; for (size_t q = 0; q < rc.size();
			q++) rc[q] = (int32_t)(((unsigned)rc[q] << 14) | (unsigned)rbk[q]); } CPY(o.sp_rcol, d.o_sprow, sym[p].rcol.data(), sym[p].rcol.size(), int32_t); CPY(o.sp_fill, d.o_spfill,
			sym[p].fill.data(), d.n_fill, int32_t);
		{ std::vector<int32_t> ho(k.n_hap); for (int b = 0; b < k.n_hap; b++) ho[b] = b;
		  std::stable_sort(ho.begin(), ho.end(), [&](int x, int y) { return k.hap_term_off[x + 1] - k.hap_term_off[x] > k.hap_term_off[y + 1] - k.hap_term_off[y]; });
		  int32_t *hr = (int32_t *)(h + o.hap_rec) + 3 * d.o_hrec; int nr = 0;
		  for (int i = 0; i < k.n_hap; i++) { const int b = ho[i], tb = k.hap_term_off[b], te = k.hap_term_off[b + 1];
			hr[3 * nr] = b; hr[3 * nr + 1] = tb; hr[3 * nr + 2] = te; nr++; } }
		if (d.hs_lds) { // the term lists of the LDS path, panel by panel (block ranges that fit the LDS: one for most windows): K6 terms by observation, Schur terms by landmark
			// (stable: a block's terms keep their order inside an observation / a landmark), and the table of the panels' bounds
			const int np = d.n_panel, psize = (k.n_hap + np - 1) / np; auto panel_of = [&](int b) { return b / psize; };
			int32_t *pt = (int32_t *)(h + o.ptab) + d.o_ptab; for (int q = 0; q <= np; q++) pt[q] = std::min(k.n_hap, q * psize);
			std::vector<int32_t> ix(k.n_hap_terms), tb(k.n_hap_terms); for (int b = 0; b < k.n_hap; b++) for (int t = k.hap_term_off[b]; t < k.hap_term_off[b + 1]; t++) tb[t] = b;
			for (int t = 0; t < k.n_hap_terms; t++) ix[t] = t;
			std::stable_sort(ix.begin(), ix.end(), [&](int x, int y) { const int px = panel_of(tb[x]), py = panel_of(tb[y]); return px != py ? px < py : k.bp_res[k.hap_t1[x]] < k.bp_res[k.hap_t1[y]];
				});
			int32_t *ho = (int32_t *)(h + o.hapo) + 3 * d.o_hapo; for (int q = 0; q <= np; q++) pt[np + 1 + q] = 0;
			for (int q = 0; q < k.n_hap_terms; q++) { const int t = ix[q]; ho[3 * q] = k.hap_t1[t]; ho[3 * q + 1] = k.hap_t2[t]; ho[3 * q + 2] = tb[t]; pt[np + 2 + panel_of(tb[t])] = q + 1; }
			for (int q = 1; q <= np; q++) pt[np + 1 + q] = std::max(pt[np + 1 + q], pt[np + q]); // (a panel without terms: empty range)
			std::vector<int32_t> sx(k.n_sch_terms), sb(k.n_sch_terms); for (int b = 0; b < k.n_hap; b++) for (int t = k.sch_term_off[b]; t < k.sch_term_off[b + 1]; t++) sb[t] = b;
			for (int t = 0; t < k.n_sch_terms; t++) sx[t] = t;
			std::stable_sort(sx.begin(), sx.end(), [&](int x, int y) { const int px = panel_of(sb[x]), py = panel_of(sb[y]); return px != py ? px < py : k.sch_lm[x] < k.sch_lm[y]; });
			int32_t *so = (int32_t *)(h + o.schl) + 4 * d.o_schl; for (int q = 0; q <= np; q++) pt[2 * np + 2 + q] = 0;
			for (int q = 0; q < k.n_sch_terms; q++) { const int t = sx[q], b = sb[t]; const bool dg = k.hap_i[b] == k.hap_j[b];
				so[4 * q] = k.sch_lm[t]; so[4 * q + 1] = k.sch_b1[t]; so[4 * q + 2] = k.sch_b2[t]; so[4 * q + 3] = (int32_t)((uint32_t)b | ((uint32_t)k.hap_i[b] << 16) | (dg ? 0x80000000u : 0u));
					pt[2 * np + 3 + panel_of(b)] = q + 1; }
			for (int q = 1; q <= np; q++) pt[2 * np + 2 + q] = std::max(pt[2 * np + 2 + q], pt[2 * np + 1 + q]);
		}
		if (asm_fam) asm_rounds[p] = srbadev::asm_pack(k, (srbadev::AsmRec *)(h + o.asm_rec) + asm_ro[p]); // packed row records of the fused normal-equations kernel
		CPY(o.hap_dst, d.o_hap * (P / 3) * (P / 3), sym[p].hap_dst.data(), sym[p].hap_dst.size(), int32_t); CPY(o.hapf_dst, d.o_hapf * (P / 3), sym[p].hapf_dst.data(), sym[p].hapf_dst.size(),
			int32_t); CPY(o.hf_dst, d.o_hf, sym[p].hf_dst.data(), sym[p].hf_dst.size(), int32_t);
		acc_blocks[thread] += d.nb + d.nnzoff; acc_items[thread] += (int64_t)sym[p].tgt.size();
	}
	});
	for (size_t t = 0; t < acc_blocks.size(); t++) { st.n_chol_blocks += acc_blocks[t]; st.n_chol_items += acc_items[t]; }
#undef CPY
	std::memcpy(h + o.desc, c->desc.data(), sizeof(ProbDesc) * n);
	c->cls_of = cls; c->big_ld = big_lds;
	{ // launch order: capsules grouped by LDS size class
		int32_t *ord = (int32_t *)(h + o.order); int pos = 0;
		for (int k = 0; k < SRBA_NCLS; k++) {
			c->cls_first[k] = pos; for (int p = 0; p < n; p++) if (cls[p] == k) ord[pos++] = p; c->cls_count[k] = pos - c->cls_first[k];
			c->cls_lds[k] = k < SRBA_NLDS ? (size_t)cls_nbmax[k] * 8 : (k < SRBA_NCLS - 1 ? std::max(wg_lds[k - SRBA_NLDS], (size_t)srbadev::WG_LDS_DOUBLES * 8) : 0);
			// longest (most block updates per factorisation) first, dealt round-robin to the queue slices of lm_run_async
			int32_t *b = ord + c->cls_first[k]; const int cnt = c->cls_count[k], nq = c->n_queues;
			auto work = [&](int x) -> long long { const ProbDesc &dx = c->desc[x]; return dx.dense_blocks ? (long long)dx.nb * dx.nb * dx.nb / 6 : dx.n_items; }; // block updates per factorisation
			std::stable_sort(b, b + cnt, [&](int x, int y) { return work(x) > work(y); });
			if (c->sched == 0 && cnt >= 16 * nq) { std::vector<int32_t> t(b, b + cnt); for (int q = 0; q < nq; q++) { int i = slice_begin(cnt, q, nq); for (int src = q; src < cnt;
				src += nq) b[i++] = t[src]; } }
		}
		plan_launches(c, ord);
	}
	c->asm_ready = false; c->jp_stale = false;
	if (asm_fam) { // fused normal equations (srba_assemble.hpp): capsules packed into bins (one workgroup, a wavefront per capsule), descriptors in bin order
		std::vector<srbadev::AsmDesc> dsc(n);
		for (int p = 0; p < n; p++) { const ProbDesc &d = c->desc[p]; dsc[p] = {p, asm_rounds[p], d.n_hap, d.nK, 0, 0, asm_ro[p], d.o_pair * 2 * PDX, d.o_edge * PDX, d.o_obs * O, d.o_hap, d.o_scal, d.o_unk}; }
		srbadev::asm_config(c->asm_wpw, c->asm_bin_bytes);
		c->asm_bins = srbadev::asm_plan(n, dsc.data(), asm_rounds.data(), (size_t)c->asm_max_kb * 1024, c->asm_wpw, c->asm_bin_bytes, (srbadev::AsmDesc *)(h + o.asm_desc), (int32_t *)(h + o.asm_list),
			c->asm_rest);
		c->asm_ready = true;
	}
	if (host_timing) ht2 = now();
	// ---- work arena layout
	Arena wk; struct { size_t edge, ulm, pose, Jp, Jf, resid, resid2, HAp, HAp0, Hf, HApf, grad, delta, Hfinv, YW, Yh, old_edge, old_ulm, old_pose, dense, ulm_inf, valid, first_fail, hf_ok, bp_ok,
		bf_ok, ulm_inf_valid, results, lambda_io, chi2, notpd, phase_cycles, m_pair, grad0, edge1, ulm1, pose1; } w;
	w.results = wk.add(sizeof(srba_lm_result) * n); // (first: [result records | unknowns | spanning-tree poses] is one span -- srba_hip_optimize_capsule reads it back in one copy)
	w.edge = wk.add(8 * t_edge * PDX); w.ulm = wk.add(8 * t_ulm * L); w.pose = wk.add(8 * 2 * t_pair * PDX); w.Jp = wk.add(8 * t_bp * O * P); w.Jf = wk.add(8 * t_bf * O * L);
	w.resid = wk.add(8 * t_obs * O); w.resid2 = wk.add(8 * t_obs * O); w.HAp = wk.add(8 * t_hap * P * P); w.HAp0 = wk.add(8 * t_hap * P * P); w.Hf = wk.add(8 * t_hf * L * L);
		w.HApf = wk.add(8 * t_hapf * P * L);
	w.grad = wk.add(8 * t_scal); w.delta = wk.add(8 * t_scal); w.Hfinv = wk.add(8 * t_ulm * L * L); w.YW = wk.add(8 * t_yw * P * L); w.Yh = wk.add(c->wg_on ? 8 * t_hapf * P * L : 0);
		w.old_edge = wk.add(8 * t_unk * PDX); w.old_ulm = wk.add(8 * t_ulm * L); w.old_pose = wk.add(8 * t_req * PDX);
	w.dense = wk.add(8 * t_dense); w.ulm_inf = wk.add(8 * t_ulm * L * L); w.valid = wk.add(4 * t_valid); w.first_fail = wk.add(4 * t_valid); w.hf_ok = wk.add(4 * t_ulm); w.bp_ok = wk.add(t_bp);
		w.bf_ok = wk.add(t_bf); w.ulm_inf_valid = wk.add(t_ulm);
	w.lambda_io = wk.add(8 * n); w.chi2 = wk.add(8 * n); w.notpd = wk.add(4 * n); w.phase_cycles = wk.add(8 * 16 * (size_t)n);
	w.m_pair = wk.add(4 * t_pair);
	w.edge1 = wk.add(8 * t_edge * PDX); w.ulm1 = wk.add(8 * t_ulm * L); w.pose1 = wk.add(8 * 2 * t_pair * PDX); // second copy of the unknowns and of the spanning-tree poses: the fused loop is
		// double-buffered
	w.grad0 = wk.add((c->params.extensions & SRBA_EXT_SCHUR_KEEPS_GRADIENT) ? 8 * t_scal : 0);
	wk.add(0);
	if (in.size + 256 > c->cap_in) { const size_t old_cap = c->cap_in; if (c->d_in) hipFree(c->d_in); c->d_in = nullptr; c->cap_in = 0; const size_t want = n > 1 ? std::max(in.size + 256, old_cap + old_cap / 2) : (in.size + 256) * 4; /* (batches of a map sweep grow and shrink from round to round: no
		reallocation per round; the first batch of a context gets what it asks for) */ HIPCHK(c, hipMalloc((void **)&c->d_in, want)); c->cap_in = want; }
	// a batch of one relative-pose SE2 capsule whose system lives in LDS: spec_w replicas of the work arena for the lambda-ladder speculation (k_lm_spec)
	c->spec_ready = c->spec_on && n == 1 && c->params.family == SRBA_SE2_RELPOSE2D && c->two_on && c->sched == 3 && cls[0] < SRBA_NCLS - 1 && c->desc[0].dense_in_lds && c->desc[0].n_scal <=
		srba_hip_ctx::kSpecMaxN && c->desc[0].n_scal == c->desc[0].n_sys && c->params.max_iters <= 100 /* rounds <= trials <= ~ 70 per iteration (lambda *= nu,
		nu *= 2 reaches max_lambda within that): below the 8192 round numbers a launch owns (SpecCtl::round0) */;
	c->spec_stride = (wk.size + 255) & ~(size_t)255; const size_t wk_need = c->spec_ready ? c->spec_stride * (size_t)c->spec_w : wk.size;
	if (c->spec_ready && !c->d_spec) HIPCHK(c, hipMalloc((void **)&c->d_spec, srba_hip_ctx::kSpecBytes));
	if (wk_need + 256 > c->cap_wk) { const size_t old_cap = c->cap_wk; if (c->d_wk) hipFree(c->d_wk); c->d_wk = nullptr; c->cap_wk = 0; const size_t want = n > 1 ? std::max(wk_need + 256, old_cap + old_cap / 2) : (wk_need + 256) * 4; HIPCHK(c, hipMalloc((void **)&c->d_wk,
		want)); c->cap_wk = want; }
	HIPCHK(c, hipMemcpyAsync(c->d_in, h, in.size, hipMemcpyHostToDevice, c->stream));
	HIPCHK(c, hipMemsetAsync(c->d_wk, 0, wk_need, c->stream));
	// ---- batch struct
	Batch &B = c->B; std::memset(&B, 0, sizeof(B)); B.n_prob = n; B.max_lds_doubles = 0; B.hess_terms = c->lm_terms ? 1 : 0; B.dense_left = c->dense_left ? 1 : 0;
	char *di = c->d_in, *dw = c->d_wk;
#define DI(field, T) B.field = (const T *)(di + o.field)
	B.desc = (const ProbDesc *)(di + o.desc); DI(order, int); DI(sp_col_off, int); DI(sp_row, int); DI(sp_item_off, int); DI(sp_tgt, int); DI(sp_rptr, int); DI(sp_rcol, int); DI(sp_perm, int);
		DI(hap_rec, int); DI(hapo, int); DI(schl, int); DI(ptab, int); DI(sp_fill, int); DI(hap_dst, int); DI(hapf_dst, int); DI(hf_dst, int); DI(edge0, double); DI(ulm0, double); DI(klm, double);
		DI(obs_z, double);
	DI(pair_path_off, int); DI(path_edge, int); DI(obs_pose, int); DI(obs_lm, int); DI(obs_valid, int); DI(bp_col, int); DI(bp_res, int); DI(bp_A, int); DI(bp_D, int); DI(bp_lm, int); DI(colp_off,
		int);
	DI(bf_col, int); DI(bf_res, int); DI(bf_pose, int); DI(colf_off, int); DI(hap_i, int); DI(hap_j, int); DI(hap_term_off, int); DI(hap_t1, int); DI(hap_t2, int); DI(hap_tblk, int); DI(hf_i, int);
		DI(hf_j, int); DI(hf_term_off, int); DI(hf_t1, int); DI(hf_t2, int);
	DI(hapf_i, int); DI(hapf_j, int); DI(hapf_term_off, int); DI(hapf_t1, int); DI(hapf_t2, int); DI(hap_diag, int); DI(hf_diag, int); DI(sch_term_off, int); DI(sch_b1, int); DI(sch_b2, int);
		DI(sch_lm, int); DI(sch_yw, int); DI(sch_tblk, int); DI(sch_vb, int); DI(sch_rec, int);
	DI(lm_hapf_off, int); DI(lm_hapf_idx, int); DI(req_idx, int); DI(need_idx, int); DI(need_rec, int); DI(obs_rec, int); DI(od_tab, int); DI(pair_needed, unsigned char); DI(pose_req, unsigned char); DI(bp_normal,
		unsigned char);
	c->asm_tab.rec = asm_fam ? (const srbadev::AsmRec *)(di + o.asm_rec) : nullptr;
	c->asm_tab.desc = asm_fam ? (const srbadev::AsmDesc *)(di + o.asm_desc) : nullptr; c->asm_list = asm_fam ? (const int *)(di + o.asm_list) : nullptr;
#undef DI
#define DW(field, T) B.field = (T *)(dw + w.field)
	DW(edge, double); DW(ulm, double); DW(pose, double); DW(Jp, double); DW(Jf, double); DW(resid, double); DW(resid2, double); DW(HAp, double); DW(HAp0, double); DW(Hf, double); DW(HApf, double);
		DW(grad, double); DW(grad0, double); DW(delta, double); DW(edge1, double); DW(ulm1, double); DW(pose1, double);
	DW(Hfinv, double); DW(YW, double); DW(Yh, double); DW(old_edge, double); DW(old_ulm, double); DW(old_pose, double); DW(dense, double); DW(ulm_inf, double); DW(valid, int); DW(first_fail, int);
		DW(hf_ok, int); DW(bp_ok, unsigned char); DW(bf_ok, unsigned char); DW(ulm_inf_valid, unsigned char);
	DW(results, srba_lm_result); DW(lambda_io, double); DW(chi2, double); DW(notpd, int);
	c->flat.pair = (int *)(dw + w.m_pair); c->flat.n_pair = t_pair; c->flat_ready = false;
	c->off_phase = w.phase_cycles; B.phase_cycles = c->phase_timing ? (long long *)(dw + w.phase_cycles) : nullptr;
#undef DW
	c->off_edge = w.edge; c->off_ulm = w.ulm; c->off_pose = w.pose; c->off_inf = w.ulm_inf; c->off_infv = w.ulm_inf_valid; c->off_res = w.results;
	const size_t dbg_off[10] = {w.resid, w.Jp, w.Jf, w.HAp, w.Hf, w.HApf, w.grad, w.delta, w.valid, w.pose};
	const int64_t dbg_len[10] = {t_obs * O, t_bp * O * P, t_bf * O * L, t_hap * P * P, t_hf * L * L, t_hapf * P * L, t_scal, t_scal, t_valid, 2 * t_pair * PD};
	c->n_pose_total = 2 * t_pair;
	for (int i = 0; i < 10; i++) { c->off_dbg[i] = dbg_off[i]; c->len_dbg[i] = dbg_len[i]; }
	c->batch_copied = false; // (the device copy of the batch record is made by the first launch that needs it: the speculative single-capsule kernel takes the record by value)
	c->n_prob = n; st.device_bytes = (int64_t)(in.size + wk.size);
	c->off_valid = w.valid; c->off_bp_ok = w.bp_ok; c->n_valid_total = t_valid; c->n_bp_total = t_bp; c->asm_flags_set = false;
	if (c->asm_ready && !c->defer_upload_sync && set_asm_flags(c) != 0) return -1; // (srba_hip_optimize_capsule runs the LM loop only,
		// whose Jacobian phase writes the flags itself: srba_hip_linearize sets them on demand)
	if (srba_hip_reset_state(c) != 0) return -1;
	if (c->defer_upload_sync && c->h_in.pinned) { if (!c->ev_h2d) HIPCHK(c, hipEventCreateWithFlags(&c->ev_h2d, hipEventDisableTiming)); HIPCHK(c, hipEventRecord(c->ev_h2d, c->stream));
		c->h2d_pending = true; } // (srba_hip_optimize_capsule waits once, at its end)
	else HIPCHK(c, hipStreamSynchronize(c->stream)); // the staging buffer is reused by the next upload
	if (host_timing) { const double ht3 = now(); acc[0] += ht1 - ht0; acc[1] += ht2 - ht1; acc[2] += ht3 - ht2; if (++calls % 1000 == 0) { std::fprintf(stderr,
		"[upload] per call: descriptors + symbolic factorisation %.1f us, packing %.1f us, arena + copies queued %.1f us\n", acc[0] / 1000, acc[1] / 1000, acc[2] / 1000);
		acc[0] = acc[1] = acc[2] = 0; } }
	return 0;
}

int srba_hip_reset_state(srba_hip_ctx *c) {
	if (!c || !c->n_prob) return -1;
	HIPCHK(c, hipSetDevice(c->device));
	HIPCHK(c, hipMemcpyAsync(c->d_wk + c->off_edge, c->d_in + c->in_off_edge0, 8 * (size_t)c->tot_edge * c->dm.PDX(), hipMemcpyDeviceToDevice, c->stream));
	if (c->tot_ulm) HIPCHK(c, hipMemcpyAsync(c->d_wk + c->off_ulm, c->d_in + c->in_off_ulm0, 8 * (size_t)c->tot_ulm * c->dm.L, hipMemcpyDeviceToDevice, c->stream));
	return 0;
}

int srba_hip_big_path_stats(srba_hip_ctx *c, double out[4]) { if (!c || !out) return -1; out[0] = c->big_chol_ms; out[1] = c->big_chol_flops; out[2] = (double)c->big_chol_count;
	out[3] = c->big_chol_nmax; return 0; }
int srba_hip_big_path_stats2(srba_hip_ctx *c, double out[8]) { if (!c || !out) return -1; out[0] = c->big_chol_ms; out[1] = c->big_chol_flops; out[2] = (double)c->big_chol_count;
	out[3] = c->big_chol_nmax; out[4] = (double)c->big_chol_seqs; out[5] = c->big_gang && !c->big_persistent ? 1 : 0; out[6] = (double)c->big_t_seqs; out[7] = c->big_t_flops; return 0; }
int srba_hip_launch_order(srba_hip_ctx *c, int64_t *stamp, int32_t *workgroups, int32_t *delay_us, int n) { // see srba_hip.h
	if (!c || !stamp || n < 0) return -1;
	HIPCHK(c, hipSetDevice(c->device)); HIPCHK(c, hipStreamSynchronize(c->stream));
	const int m = (int)std::min<size_t>(std::min<size_t>((size_t)n, c->plan.size()), kMaxJobs); std::vector<int32_t> rec(4 * (size_t)std::max(m, 1));
	if (m > 0) HIPCHK(c, hipMemcpy(rec.data(), c->d_next, sizeof(int32_t) * 4 * (size_t)m, hipMemcpyDeviceToHost));
	for (int j = 0; j < m; j++) { int64_t t; std::memcpy(&t, &rec[4 * j + 2], 8); stamp[j] = t; if (workgroups) workgroups[j] = c->plan[j].grid; if (delay_us) delay_us[j] = c->plan[j].delay_us; }
	return m;
}
int64_t srba_hip_debug_assemble_records(const srba_problem_capsule *cap, uint32_t *words, int64_t cap_records) {
	if (!cap || !words) return 0;
	try { const int64_t room = srbadev::asm_rec_room(cap->n_obs, cap->n_bp); if (room > cap_records) return -1 - room;
		std::memset(words, 0, sizeof(srbadev::AsmRec) * (size_t)room); return srbadev::asm_pack(*cap, (srbadev::AsmRec *)words); }
	catch (...) { return 0; }
}
int srba_hip_spec_stats(srba_hip_ctx *c, int64_t out[2]) { if (!c || !out) return -1; out[0] = c->spec_launches; out[1] = c->spec_fallbacks; return 0; }
int srba_hip_batch_stats(srba_hip_ctx *c, srba_batch_stats *out) { if (!c || !out) return -1; *out = c->stats; return 0; }

} // extern "C"

#define SRBA_DISPATCH_N(c, KERNEL, nblocks, lds, ...) with_family((c)->params.family, [&](auto fam_) { \
	hipLaunchKernelGGL((srbadev::KERNEL<decltype(fam_)::value>), dim3(nblocks), dim3(SRBA_WG), (lds), (c)->stream, (c)->B, (c)->dp, ##__VA_ARGS__); })
#define SRBA_DISPATCH(c, KERNEL, lds, ...) SRBA_DISPATCH_N(c, KERNEL, (c)->n_prob, lds, ##__VA_ARGS__)
#define SRBA_DISPATCH_LDS(c, KERNEL, nblocks, lds, ...) with_family((c)->params.family, [&](auto fam_) { \
	hipLaunchKernelGGL((srbadev::KERNEL<decltype(fam_)::value>), dim3(nblocks), dim3(SRBA_WG), (lds), launch_stream, (c)->B, (c)->dp, ##__VA_ARGS__); })

template <class K> static int allow_big_lds(srba_hip_ctx *c, K kernel, size_t bytes) {
	if (bytes > 64 * 1024) { hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes); if (e != hipSuccess) {
		c->fail(std::string("hipFuncSetAttribute(MaxDynamicSharedMemorySize): ") + hipGetErrorString(e)); return -1; } }
	return 0;
}
static int prep_lds(srba_hip_ctx *c, bool for_lm) {
	size_t b = 0; for (int k = 0; k < SRBA_NCLS; k++) if (c->cls_count[k]) b = std::max(b, c->cls_lds[k]);
	b += c->lds_pad;
	if (b <= 64 * 1024) return 0;
	int rc = -1;
	with_family(c->params.family, [&](auto fam_) { constexpr int F = decltype(fam_)::value; rc = for_lm ? allow_big_lds(c, srbadev::k_lm_run<F>, b) : allow_big_lds(c, srbadev::k_solve<F>, b); });
	if (rc == 0 && for_lm && c->params.family == SRBA_SE2_RELPOSE2D) rc = allow_big_lds(c, srbadev::k_lm_run2<SRBA_SE2_RELPOSE2D>, b + 64);
	if (rc == 0 && for_lm && c->params.family == SRBA_SE2_RELPOSE2D) rc = allow_big_lds(c, srbadev::k_lm_spec<SRBA_SE2_RELPOSE2D>, b + 64);
	return rc;
}

// The HIP runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and reads the variable when it initialises: ask for 16
// (one per concurrent launch stream of the plan) unless the user chose a value. No effect if the process already initialised HIP.
__attribute__((constructor)) static void srba_hip_runtime_defaults() { setenv("GPU_MAX_HW_QUEUES", "16", 0); }



extern "C" {

int srba_hip_sync(srba_hip_ctx *c) { if (!c) return -1; HIPCHK(c, hipStreamSynchronize(c->stream)); return 0; }

static int lm_run_async_impl(srba_hip_ctx *c);
int srba_hip_lm_run_async(srba_hip_ctx *c) {
	if (c) c->jp_stale = false; // the LM loop writes the Jacobian blocks of its own linearisation points
 // no C++ exception crosses the C ABI
	try { return lm_run_async_impl(c); }
	catch (const std::exception &e) { if (c) c->fail(std::string("lm_run: ") + e.what()); return -1; }
	catch (...) { if (c) c->fail("lm_run: unknown exception"); return -1; }
}
static int lm_run_async_impl(srba_hip_ctx *c) {
	if (!c || !c->n_prob) { if (c) c->fail("lm_run: no batch uploaded"); return -1; }
	HIPCHK(c, hipSetDevice(c->device));
	if (prep_lds(c, true) != 0) return -1;
	{ const int slot = (int)(c->n_launches % srba_hip_ctx::kRing);
	  if (!c->ring0[slot]) { HIPCHK(c, hipEventCreate(&c->ring0[slot])); HIPCHK(c, hipEventCreate(&c->ring1[slot])); }
	  c->ev0 = c->ring0[slot]; c->ev1 = c->ring1[slot]; c->n_launches++; }
	HIPCHK(c, hipEventRecord(c->ev0, c->stream));
	if (c->spec_ready && !c->spec_suppress && c->plan.size() == 1 && c->cls_count[SRBA_NCLS - 1] == 0) { // a batch of one capsule: its lambda ladder on spec_w workgroups
		const int k = c->plan[0].cls; const size_t lds1 = (c->cls_lds[k] + c->lds_pad + 7) & ~(size_t)7; const int W = c->spec_w;
		srbadev::SpecCtl sc; sc.w = 0; sc.W = W; sc.round0 = (int)((c->spec_launches++ % 200000) * 8192); if (sc.round0 == 0) HIPCHK(c, hipMemsetAsync(c->d_spec, 0, 256, c->stream));
			sc.flag = (int *)c->d_spec; sc.box = (double *)(c->d_spec + 256); sc.xdelta = sc.box + 2 * srba_hip_ctx::kSpecMaxW * 4; sc.xstride = srba_hip_ctx::kSpecMaxN;
			sc.edge_backup = (double *)(c->d_spec + srba_hip_ctx::kSpecBackupOff); // (the round numbers of a launch continue where no earlier launch has been: the flags are cleared once per 200 000
			// launches, not per launch)
		const bool test_drop = c->spec_test_drop; /* test knob (SRBA_HIP_SPEC_TEST_DROP, read when the context is created): the last replica is never launched -- the others give up after the spin
			bound, status 2, and the host falls back (tests/test_gpu_parity.py) */
		hipLaunchKernelGGL((srbadev::k_lm_spec<SRBA_SE2_RELPOSE2D>), dim3(test_drop ? W - 1 : W), dim3(2 * SRBA_WG), lds1 + 32, c->stream, c->B, c->dp, (int)(lds1 / 8), (long long)c->spec_stride,
			sc); HIPCHK(c, hipGetLastError());
		HIPCHK(c, hipEventRecord(c->ev1, c->stream));
		return 0;
	}
	if (!c->batch_copied) { if (!c->d_batch) HIPCHK(c, hipMalloc((void **)&c->d_batch, sizeof(Batch) + sizeof(DevParams))); HIPCHK(c, hipMemcpyAsync(c->d_batch, &c->B, sizeof(Batch),
		hipMemcpyHostToDevice, c->stream)); HIPCHK(c, hipMemcpyAsync(c->d_batch + 1, &c->dp, sizeof(DevParams), hipMemcpyHostToDevice, c->stream)); c->batch_copied = true; }
		// the kernels read the batch record from device memory (c->B lives as long as the context and only changes at an upload)
	HIPCHK(c, hipMemsetAsync(c->d_next, 0, sizeof(int) * 4 * std::max<size_t>(1, std::min<size_t>(kMaxJobs, c->plan.size())), c->stream)); // per launch {work counter, pad,
		// device time stamp of its first capsule}
	// fork/join: the launch plan (made at upload) spreads the size classes over a few streams; see plan_launches()
	const int nq = c->plan.size() <= 1 ? 1 : c->n_streams_used; // a single launch (the per-key-frame use) stays on the context stream: no fork / join
	if (nq > 1) HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));
	for (int q = 1; q < nq; q++) HIPCHK(c, hipStreamWaitEvent(c->cls_stream[q], c->ev_fork, 0));
	for (size_t j = 0; j < c->plan.size(); j++) {
		const LaunchJob &J = c->plan[j]; const int k = J.cls;
		hipStream_t launch_stream = (J.queue && nq > 1) ? c->cls_stream[J.queue] : c->stream;
		if (J.delay_us > 0) hipLaunchKernelGGL(srbadev::k_delay, dim3(1), dim3(1), 0, launch_stream, J.delay_us);
		if (k >= SRBA_NLDS) { // landmark windows on a workgroup (k_lm_wg)
			int rc_attr = 0;
			with_family(c->params.family, [&](auto fam_) { constexpr int F = decltype(fam_)::value; if constexpr (srbadev::Tr<F>::SE3 && !srbadev::Tr<F>::REL) {
				if (k == SRBA_CLS_WG512) { if ((rc_attr = allow_big_lds(c, srbadev::k_lm_wg<F, SRBA_WG_TOP>, c->cls_lds[k])) == 0) hipLaunchKernelGGL((srbadev::k_lm_wg<F, SRBA_WG_TOP>), dim3(J.grid),
					dim3(SRBA_WG_TOP), c->cls_lds[k], launch_stream, SRBA_WG_BATCH_VAL(c), J.first, J.count, c->d_next + 4 * j); }
				else if (k == SRBA_CLS_WG256) { if ((rc_attr = allow_big_lds(c, srbadev::k_lm_wg<F, 256>, c->cls_lds[k])) == 0) hipLaunchKernelGGL((srbadev::k_lm_wg<F, 256>), dim3(J.grid), dim3(256),
					c->cls_lds[k], launch_stream, SRBA_WG_BATCH_VAL(c), J.first, J.count, c->d_next + 4 * j); }
				else hipLaunchKernelGGL((srbadev::k_lm_wg<F, 128>), dim3(J.grid), dim3(128), c->cls_lds[k], launch_stream, SRBA_WG_BATCH_VAL(c), J.first, J.count, c->d_next + 4 * j); } });
			if (rc_attr != 0) return -1;
			HIPCHK(c, hipGetLastError()); continue; }
		if (J.two) { const size_t lds1 = (c->cls_lds[k] + c->lds_pad + 7) & ~(size_t)7; hipLaunchKernelGGL((srbadev::k_lm_run2<SRBA_SE2_RELPOSE2D>), dim3(J.grid), dim3(2 * SRBA_WG), lds1 + 32,
			launch_stream, SRBA_LM_BATCH_VAL(c), J.first, J.count, c->d_next + 4 * j, (int)(lds1 / 8)); HIPCHK(c, hipGetLastError()); continue; }
		if (J.lean) { hipLaunchKernelGGL((srbadev::k_lm_run_lean<SRBA_SE2_RELPOSE2D>), dim3(J.grid), dim3(SRBA_WG), c->cls_lds[k] + c->lds_pad, launch_stream, SRBA_LM_BATCH_VAL(c), J.first, J.count,
			c->d_next + 4 * j); HIPCHK(c, hipGetLastError()); continue; }
		with_family(c->params.family, [&](auto fam_) { hipLaunchKernelGGL((srbadev::k_lm_run<decltype(fam_)::value>), dim3(J.grid), dim3(SRBA_WG), c->cls_lds[k] + c->lds_pad, launch_stream,
			SRBA_LM_BATCH_VAL(c), J.first, J.count, c->d_next + 4 * j); }); HIPCHK(c, hipGetLastError());
	}
	// capsules too large for one wavefront's LDS: the multi-workgroup path, one capsule after the other on the context stream (host-driven LM loop: this part
	// of the call synchronises with the device once per LM trial), overlapping with the persistent launches of the other classes on their own streams
	{ const int32_t *ord = (const int32_t *)(c->h_in.get() + c->h_off_order);
	  const int big_rc = big_run_class(c, ord + c->cls_first[SRBA_NCLS - 1], c->cls_count[SRBA_NCLS - 1]);
	  // join the class streams and close the timing pair whatever the large-capsule path returned: the launches above are in flight either way
	  for (int q = 1; q < nq; q++) { HIPCHK(c, hipEventRecord(c->cls_done[q], c->cls_stream[q])); HIPCHK(c, hipStreamWaitEvent(c->stream, c->cls_done[q], 0)); }
	  HIPCHK(c, hipEventRecord(c->ev1, c->stream));
	  if (big_rc != 0) { hipStreamSynchronize(c->stream); return -1; } }
	return 0;
}

int srba_hip_eval_overall_sqr_error(srba_hip_ctx *c, const srba_overall_problem *q, double *out) {
	if (!c || !q || !out) return -1;
	if (q->n_edges < 0 || q->n_pairs < 0 || q->n_obs < 0 || q->n_lms < 0 || (q->n_obs > 0 && (!q->obs_pose || !q->obs_lm || !q->obs_z)) || (q->n_pairs > 0 && (!q->pair_path_off || (q->n_path > 0 &&
		!q->path_edge)))) { c->fail("eval_overall_sqr_error: malformed problem"); return -1; }
	*out = 0; if (q->n_obs == 0) return 0;
	HIPCHK(c, hipSetDevice(c->device));
	const int L = c->dm.L, O = c->dm.O, PD = c->dm.PD, PDX = c->dm.PDX();
	for (int i = 0; i < q->n_obs; i++) if (q->obs_pose[i] >= 2 * q->n_pairs || q->obs_lm[i] < 0 || q->obs_lm[i] >= q->n_lms) { c->fail("eval_overall_sqr_error: observation index out of range");
		return -1; }
	for (int k = 0; k < q->n_path; k++) if ((q->path_edge[k] >> 1) >= q->n_edges || q->path_edge[k] < 0) { c->fail("eval_overall_sqr_error: path edge out of range"); return -1; }
	// one host arena -> one H2D copy
	Arena in; const size_t o_desc = in.add(sizeof(ProbDesc)), o_edge = in.add(8 * (size_t)q->n_edges * PDX), o_ppo = in.add(4 * ((size_t)q->n_pairs + 1)),
		o_path = in.add(4 * (size_t)std::max(q->n_path, 1)),
		o_op = in.add(4 * (size_t)q->n_obs), o_ol = in.add(4 * (size_t)q->n_obs), o_z = in.add(8 * (size_t)q->n_obs * O), o_lm = in.add(8 * (size_t)std::max(q->n_lms, 1) * L);
	const int nblk = std::min(1024, (q->n_obs + 255) / 256);
	Arena wk; const size_t o_pose = wk.add(8 * 2 * (size_t)std::max(q->n_pairs, 1) * PDX), o_part = wk.add(8 * (size_t)nblk);
	std::vector<char> h(in.size, 0);
	ProbDesc d; std::memset(&d, 0, sizeof(d)); d.n_edges = q->n_edges; d.n_pairs = q->n_pairs; d.n_obs = q->n_obs; d.nF = q->n_lms;
	std::memcpy(h.data() + o_desc, &d, sizeof(d));
	{ double *e = (double *)(h.data() + o_edge);
	  for (int i = 0; i < q->n_edges; i++) { const double *s = q->edge_pose + (size_t)i * PD; double *t = e + (size_t)i * PDX; for (int k = 0; k < PD; k++) t[k] = s[k]; if (PDX == 5) {
	  	t[3] = std::cos(s[2]); t[4] = std::sin(s[2]); } } }
	if (q->n_pairs) std::memcpy(h.data() + o_ppo, q->pair_path_off, 4 * ((size_t)q->n_pairs + 1));
	if (q->n_path) std::memcpy(h.data() + o_path, q->path_edge, 4 * (size_t)q->n_path);
	std::memcpy(h.data() + o_op, q->obs_pose, 4 * (size_t)q->n_obs); std::memcpy(h.data() + o_ol, q->obs_lm, 4 * (size_t)q->n_obs);
	std::memcpy(h.data() + o_z, q->obs_z, 8 * (size_t)q->n_obs * O); if (q->n_lms) std::memcpy(h.data() + o_lm, q->lm_pos, 8 * (size_t)q->n_lms * L);
	char *di = nullptr, *dw = nullptr;
	HIPCHK(c, hipMalloc(&di, in.size)); if (hipMalloc(&dw, wk.size) != hipSuccess) { hipFree(di); c->fail("eval_overall_sqr_error: hipMalloc failed"); return -1; }
	int rc = 0; std::vector<double> part(nblk, 0.0);
	do {
		if (hipMemcpyAsync(di, h.data(), in.size, hipMemcpyHostToDevice, c->stream) != hipSuccess) { rc = -1; break; }
		Batch B; std::memset(&B, 0, sizeof(B)); B.n_prob = 1;
		B.desc = (const ProbDesc *)(di + o_desc); B.edge = (double *)(di + o_edge); B.pair_path_off = (const int *)(di + o_ppo); B.path_edge = (const int *)(di + o_path);
		B.obs_pose = (const int *)(di + o_op); B.obs_lm = (const int *)(di + o_ol); B.obs_z = (const double *)(di + o_z); B.ulm = (double *)(di + o_lm); B.pose = (double *)(dw + o_pose);
		DevParams dp = c->dp; dp.use_robust_kernel = 0; // the reference sums plain squared norms (eval_overall_error.h:129)
		const int pblk = std::max(1, std::min(1024, (q->n_pairs + 255) / 256));
		with_family(c->params.family, [&](auto fam_) { constexpr int F = decltype(fam_)::value;
			if (q->n_pairs) hipLaunchKernelGGL((srbadev::k_overall_pairs<F>), dim3(pblk), dim3(256), 0, c->stream, B, dp);
			hipLaunchKernelGGL((srbadev::k_overall_residuals<F>), dim3(nblk), dim3(256), 0, c->stream, B, dp, (double *)(dw + o_part)); });
		if (hipGetLastError() != hipSuccess) { rc = -1; break; }
		if (hipMemcpyAsync(part.data(), dw + o_part, 8 * (size_t)nblk, hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) { rc = -1; break; }
	} while (0);
	hipFree(di); hipFree(dw);
	if (rc != 0) { c->fail("eval_overall_sqr_error: HIP error"); return -1; }
	double s = 0; for (int b = 0; b < nblk; b++) s += part[b];
	*out = s; return 0;
}
// The replicas of a speculative single-capsule run lost step with each other (status 2: one of them was not resident within the spin bound of spec_exchange -- a shared
// GPU, another context filling the CUs): put the unknown edges back as the launch found them (SpecCtl::edge_backup, written by replica 0 before anything else) and run
// the capsule once on the sequential path (k_lm_run2: recomputes every pose, Jacobian and result field from the edges). The stream is idle on entry and on return.
static int spec_fallback(srba_hip_ctx *c) {
	const ProbDesc &d = c->desc[0];
	HIPCHK(c, hipMemcpyAsync(c->d_wk + c->off_edge + 8 * (size_t)d.o_edge * c->dm.PDX(), c->d_spec + srba_hip_ctx::kSpecBackupOff, 8 * (size_t)d.nK * c->dm.PDX(), hipMemcpyDeviceToDevice, c->stream));
	c->spec_suppress = true; const int rc = lm_run_async_impl(c); c->spec_suppress = false; c->spec_fallbacks++;
	if (rc != 0) return -1;
	HIPCHK(c, hipStreamSynchronize(c->stream));
	float ms = 0; if (hipEventElapsedTime(&ms, c->ev0, c->ev1) == hipSuccess) c->last_ms = ms;
	return 0;
}
int srba_hip_download_results(srba_hip_ctx *c, srba_lm_result *results, int n) {
	if (!c || !results || n > c->n_prob) return -1;
	HIPCHK(c, hipStreamSynchronize(c->stream));
	HIPCHK(c, hipMemcpyAsync(results, c->d_wk + c->off_res, sizeof(srba_lm_result) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
	HIPCHK(c, hipStreamSynchronize(c->stream));
	if (c->spec_ready && n >= 1 && results[0].status == 3) { c->fail("lm_run: the replicas of the speculative single-capsule run disagree (protocol error, not a time-out)"); return -1; }
	if (c->spec_ready && n >= 1 && results[0].status == 2) { // the speculative run gave up: once more on the sequential path
		if (spec_fallback(c) != 0) return -1;
		HIPCHK(c, hipMemcpyAsync(results, c->d_wk + c->off_res, sizeof(srba_lm_result) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
		HIPCHK(c, hipStreamSynchronize(c->stream));
	}
	return 0;
}
int srba_hip_lm_run(srba_hip_ctx *c, srba_lm_result *results) {
	if (srba_hip_lm_run_async(c) != 0) return -1;
	HIPCHK(c, hipStreamSynchronize(c->stream));
	float ms = 0; if (hipEventElapsedTime(&ms, c->ev0, c->ev1) == hipSuccess) c->last_ms = ms;
	if (results) return srba_hip_download_results(c, results, c->n_prob);
	return 0;
}

// the flat (one thread per item of the batch) launches of srba_flat.hpp
static inline int flat_grid(long long items, int block) { return (int)std::max<long long>(1, std::min<long long>((items + block - 1) / block, 1 << 20)); }
#define FLATK(KERNEL, items, block, ...) with_family(c->params.family, [&](auto fam_) { hipLaunchKernelGGL((srbadev::KERNEL<decltype(fam_)::value>), dim3(flat_grid((items), (block))), \
	dim3(block), 0, c->stream, c->B, c->dp, ##__VA_ARGS__); })
static int flat_prepare(srba_hip_ctx *c) {
	if (c->flat_ready) return 0;
	hipLaunchKernelGGL(srbadev::kf_fill_maps, dim3(c->n_prob), dim3(256), 0, c->stream, c->B, c->flat); HIPCHK(c, hipGetLastError());
	c->flat_ready = true; return 0;
}
int srba_hip_update_spantree(srba_hip_ctx *c, int only_needed) {
	if (!c || !c->n_prob) return -1; HIPCHK(c, hipSetDevice(c->device));
	if (c->use_flat) { if (flat_prepare(c) != 0) return -1; FLATK(kf_spantree, c->flat.n_pair, 256, c->flat, only_needed); }
	else SRBA_DISPATCH(c, k_spantree, 0, only_needed);
	HIPCHK(c, hipGetLastError()); return 0;
}
int srba_hip_eval_residuals(srba_hip_ctx *c, double *chi2_out) {
	if (!c || !c->n_prob) return -1; HIPCHK(c, hipSetDevice(c->device));
	SRBA_DISPATCH(c, k_residuals, 16 * 8); HIPCHK(c, hipGetLastError());
	if (chi2_out) { HIPCHK(c, hipMemcpyAsync(chi2_out, c->B.chi2, 8 * (size_t)c->n_prob, hipMemcpyDeviceToHost, c->stream)); HIPCHK(c, hipStreamSynchronize(c->stream)); }
	return 0;
}
int srba_hip_linearize(srba_hip_ctx *c) {
	if (!c || !c->n_prob) return -1; HIPCHK(c, hipSetDevice(c->device));
	const int lds_doubles = c->lin_terms ? 16 + 1536 : 16; // 12 KB of Hessian accumulators per wavefront: U_Ap of up to 170 SE2 / 42 SE3 blocks (bigger capsules take the per-block path)
	bool lam_sym = true; // the fused kernel sums the upper triangle of J^t Lambda J only: an information matrix set after the upload (srba_hip_set_params) is checked again here
	if (c->dp.noise == SRBA_NOISE_CONSTANT_MATRIX) for (int i = 0; i < 3; i++) for (int j = 0; j < i; j++) if (c->dp.lambda[3 * i + j] != c->dp.lambda[3 * j + i]) lam_sym = false;
	if (c->asm_ready && lam_sym && set_asm_flags(c) != 0) return -1;
	if (c->asm_ready && lam_sym) { // relative-pose SE2: fused, Jacobian blocks never leave the chip (srba_assemble.hpp): one launch, a workgroup per bin of capsules;
		// what does not fit a bin takes the unfused kernel
		if (c->asm_bins > 0 && srbadev::asm_launch(c->dp.noise != SRBA_NOISE_CONSTANT_MATRIX ? 0 : (c->dp.lambda[1] == 0 && c->dp.lambda[2] == 0 && c->dp.lambda[5] == 0 && c->dp.lambda[3] == 0 &&
			c->dp.lambda[6] == 0 && c->dp.lambda[7] == 0) ? 1 : 2, c->asm_wpw, c->asm_bins, (size_t)c->asm_bin_bytes, c->stream, c->B, c->dp, c->asm_tab) != 0) { c->fail("k_assemble_se2rel: launch failed");
			return -1; }
		if (c->asm_rest > 0) { with_family(c->params.family, [&](auto fam_) { hipLaunchKernelGGL((srbadev::k_linearize<decltype(fam_)::value>), dim3(c->asm_rest), dim3(SRBA_WG),
			(size_t)lds_doubles * 8, c->stream, c->B, c->dp, lds_doubles, c->asm_list); }); HIPCHK(c, hipGetLastError()); }
		c->jp_stale = c->asm_bins > 0; return 0;
	}
	SRBA_DISPATCH(c, k_linearize, (size_t)lds_doubles * 8, lds_doubles, (const int *)nullptr); HIPCHK(c, hipGetLastError()); c->jp_stale = false; return 0;
}
int srba_hip_solve(srba_hip_ctx *c, const double *lambda, int32_t *not_pd_out) {
	if (!c || !c->n_prob) return -1; HIPCHK(c, hipSetDevice(c->device));
	if (lambda) HIPCHK(c, hipMemcpyAsync(c->B.lambda_io, lambda, 8 * (size_t)c->n_prob, hipMemcpyHostToDevice, c->stream)); // else: use the lambda guess left by srba_hip_linearize
	if (prep_lds(c, false) != 0) return -1;
	for (int k = 0; k < SRBA_NLDS; k++) if (c->cls_count[k]) { hipStream_t launch_stream = c->stream; SRBA_DISPATCH_LDS(c, k_solve, c->cls_count[k], c->cls_lds[k], c->cls_first[k]); HIPCHK(c,
		hipGetLastError()); }
	for (int k = SRBA_NLDS; k < SRBA_NCLS - 1; k++) if (c->cls_count[k]) { // landmark windows on a workgroup
		with_family(c->params.family, [&](auto fam_) { constexpr int F = decltype(fam_)::value; if constexpr (srbadev::Tr<F>::SE3 && !srbadev::Tr<F>::REL) {
			if (k == SRBA_CLS_WG512) { if (allow_big_lds(c, srbadev::k_solve_wg<F, SRBA_WG_TOP>, c->cls_lds[k]) == 0) hipLaunchKernelGGL((srbadev::k_solve_wg<F, SRBA_WG_TOP>), dim3(c->cls_count[k]),
				dim3(SRBA_WG_TOP), c->cls_lds[k], c->stream, c->B, c->dp, c->cls_first[k]); }
			else if (k == SRBA_CLS_WG256) { if (allow_big_lds(c, srbadev::k_solve_wg<F, 256>, c->cls_lds[k]) == 0) hipLaunchKernelGGL((srbadev::k_solve_wg<F, 256>), dim3(c->cls_count[k]), dim3(256),
				c->cls_lds[k], c->stream, c->B, c->dp, c->cls_first[k]); }
			else hipLaunchKernelGGL((srbadev::k_solve_wg<F, 128>), dim3(c->cls_count[k]), dim3(128), c->cls_lds[k], c->stream, c->B, c->dp, c->cls_first[k]); } });
		HIPCHK(c, hipGetLastError()); }
	{ const int32_t *ord = (const int32_t *)(c->h_in.get() + c->h_off_order); // dense multi-workgroup solver for the capsules of the big class
	  for (int i = 0; i < c->cls_count[SRBA_NCLS - 1]; i++) { const int p = ord[c->cls_first[SRBA_NCLS - 1] + i]; double lam = 0; HIPCHK(c, hipMemcpyAsync(&lam, c->B.lambda_io + p, 8,
	  	hipMemcpyDeviceToHost, c->stream)); HIPCHK(c, hipStreamSynchronize(c->stream));
		bool pd = true; if (big_prepare_lanes(c, 1) < 1) return -1; if (big_solve(c, &c->lanes[0], p, lam, &pd) != 0) { c->fail(c->lanes[0].error); return -1; } big_collect_lane_stats(c);
			const int np = pd ? 0 : 1; HIPCHK(c, hipMemcpyAsync(c->B.notpd + p, &np, 4, hipMemcpyHostToDevice, c->stream)); HIPCHK(c, hipStreamSynchronize(c->stream)); } }
	if (not_pd_out) { HIPCHK(c, hipMemcpyAsync(not_pd_out, c->B.notpd, 4 * (size_t)c->n_prob, hipMemcpyDeviceToHost, c->stream)); }
	HIPCHK(c, hipStreamSynchronize(c->stream)); return 0;
}
int srba_hip_hessian_from_jacobians(srba_hip_ctx *c) { if (!c || !c->n_prob) return -1; HIPCHK(c, hipSetDevice(c->device));
	if (c->jp_stale) { SRBA_DISPATCH(c, k_jacobians_only, 0); HIPCHK(c, hipGetLastError()); c->jp_stale = false; } // the blocks of a fused srba_hip_linearize were never written: do it now
	SRBA_DISPATCH(c, k_hessian_only, 0); HIPCHK(c, hipGetLastError()); HIPCHK(c, hipStreamSynchronize(c->stream)); return 0; }
int srba_hip_debug_write(srba_hip_ctx *c, int what, const double *in, int64_t n_doubles) {
	if (!c || !in || !(what == 1 || what == 2 || what == 6) || n_doubles != c->len_dbg[what]) {
		if (c) c->fail("debug_write: only the Jacobian blocks (1, 2) and the minus-gradient (6) can be written, with their exact sizes"); return -1; }
	HIPCHK(c, hipSetDevice(c->device));
	HIPCHK(c, hipMemcpyAsync(c->d_wk + c->off_dbg[what], in, 8 * (size_t)n_doubles, hipMemcpyHostToDevice, c->stream));
	if (what == 1) c->jp_stale = false; // the caller's blocks are the current ones
	if (what == 6 && (c->params.extensions & SRBA_EXT_SCHUR_KEEPS_GRADIENT)) HIPCHK(c, hipMemcpyAsync(c->B.grad0, in, 8 * (size_t)n_doubles, hipMemcpyHostToDevice, c->stream));
	HIPCHK(c, hipStreamSynchronize(c->stream)); return 0;
}
int srba_hip_apply_update(srba_hip_ctx *c) { if (!c || !c->n_prob) return -1; HIPCHK(c, hipSetDevice(c->device)); SRBA_DISPATCH(c, k_apply, 0); HIPCHK(c, hipGetLastError()); return 0; }
int srba_hip_rollback(srba_hip_ctx *c) { if (!c || !c->n_prob) return -1; HIPCHK(c, hipSetDevice(c->device)); SRBA_DISPATCH(c, k_rollback, 0); HIPCHK(c, hipGetLastError()); return 0; }

int srba_hip_download_state(srba_hip_ctx *c, srba_problem_capsule *caps, int n) {
	if (!c || !caps || n != c->n_prob) { if (c) c->fail("download_state: capsule count differs from the uploaded batch"); return -1; }
	HIPCHK(c, hipSetDevice(c->device));
	const int L = c->dm.L, PD = c->dm.PD, PDX = c->dm.PDX();
	std::vector<double> edge((size_t)c->tot_edge * PDX), ulm((size_t)c->tot_ulm * L), inf((size_t)c->tot_ulm * L * L), pose((size_t)2 * c->stats.n_pairs * PDX);
		std::vector<uint8_t> infv((size_t)c->tot_ulm);
	HIPCHK(c, hipMemcpyAsync(edge.data(), c->d_wk + c->off_edge, 8 * edge.size(), hipMemcpyDeviceToHost, c->stream));
	if (!ulm.empty()) { HIPCHK(c, hipMemcpyAsync(ulm.data(), c->d_wk + c->off_ulm, 8 * ulm.size(), hipMemcpyDeviceToHost, c->stream)); HIPCHK(c, hipMemcpyAsync(inf.data(), c->d_wk + c->off_inf,
		8 * inf.size(), hipMemcpyDeviceToHost, c->stream)); HIPCHK(c, hipMemcpyAsync(infv.data(), c->d_wk + c->off_infv, infv.size(), hipMemcpyDeviceToHost, c->stream)); }
	if (!pose.empty()) HIPCHK(c, hipMemcpyAsync(pose.data(), c->d_wk + c->off_pose, 8 * pose.size(), hipMemcpyDeviceToHost, c->stream));
	HIPCHK(c, hipStreamSynchronize(c->stream));
	for (int p = 0; p < n; p++) {
		const ProbDesc &d = c->desc[p]; srba_problem_capsule &k = caps[p];
		if (k.n_unk_edges != d.nK || k.n_unk_lms != d.nF || k.n_pairs != d.n_pairs) { c->fail("download_state: capsule layout differs from the uploaded batch"); return -1; }
		for (int q = 0; q < d.nK; q++) std::memcpy(k.edge_pose + (size_t)q * PD, &edge[((size_t)d.o_edge + q) * PDX], 8 * (size_t)PD); // SE2: drop the cached cos/sin
		if (d.nF) std::memcpy(k.ulm_pos, &ulm[(size_t)d.o_ulm * L], 8 * (size_t)d.nF * L);
		if (k.pose) for (long long q = 0; q < 2LL * d.n_pairs; q++) std::memcpy(k.pose + (size_t)q * PD, &pose[((size_t)d.o_pair * 2 + q) * PDX], 8 * (size_t)PD);
		if (k.ulm_inf && d.nF) std::memcpy(k.ulm_inf, &inf[(size_t)d.o_ulm * L * L], 8 * (size_t)d.nF * L * L);
		if (k.ulm_inf_valid && d.nF) std::memcpy(k.ulm_inf_valid, &infv[(size_t)d.o_ulm], (size_t)d.nF);
	}
	return 0;
}

// The per-key-frame use of the engine in one call: srba_hip_upload_problems(ctx, capsule, 1) + srba_hip_lm_run(ctx, result) + srba_hip_download_state(ctx, capsule, 1), i.e. one
// optimize_edges() call of the reference (optimize_edges.h:256-751) with its in-place write-back (526, 538), with ONE wait for the device instead of four: the input arena leaves from
// page-locked memory without a wait, the [result record | unknowns | spanning-tree poses] head of the work arena comes back in one copy queued behind the kernel.
int srba_hip_optimize_capsule(srba_hip_ctx *c, srba_problem_capsule *cap, srba_lm_result *res) {
	if (!c || !cap || !res) { if (c) c->fail("optimize_capsule: bad arguments"); return -1; }
	static const bool host_timing = getenv("SRBA_HIP_HOST_TIMING") != nullptr; static double acc[5] = {0, 0, 0, 0, 0}; static long long calls = 0; // (diagnostic: where the host side of a call goes;
		// printed every 1000 calls)
	auto now = []() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }; const double t0 = host_timing ? now() : 0;
	c->defer_upload_sync = true; const int rc_up = srba_hip_upload_problems(c, cap, 1); c->defer_upload_sync = false; const double t1 = host_timing ? now() : 0;
	if (rc_up != 0) return rc_up;
	const bool one_wait = c->h2d_pending && c->tot_ulm == 0 && c->cls_count[SRBA_NCLS - 1] == 0; // relative-pose families, system in LDS; everything else takes the three calls
	if (!one_wait) { if (srba_hip_lm_run(c, res) != 0) return -1; return srba_hip_download_state(c, cap, 1); }
	if (srba_hip_lm_run_async(c) != 0) return -1;
	const int PD = c->dm.PD, PDX = c->dm.PDX(); const ProbDesc &d = c->desc[0];
	const size_t span = c->off_pose + 8 * (size_t)2 * d.n_pairs * PDX - c->off_res; // [result record | unknowns | spanning-tree poses]: the head of the work arena
	if (c->off_res > c->off_edge || c->off_edge > c->off_pose) { c->fail("optimize_capsule: unexpected work arena layout"); return -1; }
	if (c->h_out_cap < span) { if (c->h_out) hipHostFree(c->h_out); c->h_out = nullptr; c->h_out_cap = 0; HIPCHK(c, hipHostMalloc((void **)&c->h_out, 2 * span, hipHostMallocDefault));
		c->h_out_cap = 2 * span; }
	HIPCHK(c, hipMemcpyAsync(c->h_out, c->d_wk + c->off_res, span, hipMemcpyDeviceToHost, c->stream));
	const double t2 = host_timing ? now() : 0;
	HIPCHK(c, hipStreamSynchronize(c->stream)); c->h2d_pending = false; const double t3 = host_timing ? now() : 0;
	float ms = 0; if (hipEventElapsedTime(&ms, c->ev0, c->ev1) == hipSuccess) c->last_ms = ms;
	if (host_timing) { acc[0] += t1 - t0; acc[1] += t2 - t1; acc[2] += t3 - t2; acc[3] += 1e3 * ms; if (++calls % 1000 == 0) { std::fprintf(stderr,
		"[optimize_capsule] per call: upload (host) %.1f us, launch + copies queued %.1f us, wait %.1f us (kernel %.1f us)\n", acc[0] / 1000, acc[1] / 1000, acc[2] / 1000, acc[3] / 1000);
		acc[0] = acc[1] = acc[2] = acc[3] = 0; } }
	std::memcpy(res, c->h_out, sizeof(srba_lm_result));
	if (c->spec_ready && res->status == 3) { c->fail("optimize_capsule: the replicas of the speculative single-capsule run disagree (protocol error, not a time-out)"); return -1; }
	if (c->spec_ready && res->status == 2) { // the speculative run gave up: once more on the sequential path, read back the same span
		if (spec_fallback(c) != 0) return -1;
		HIPCHK(c, hipMemcpyAsync(c->h_out, c->d_wk + c->off_res, span, hipMemcpyDeviceToHost, c->stream)); HIPCHK(c, hipStreamSynchronize(c->stream));
		std::memcpy(res, c->h_out, sizeof(srba_lm_result));
	}
	if (cap->n_unk_edges != d.nK || cap->n_pairs != d.n_pairs) { c->fail("optimize_capsule: capsule layout differs from the uploaded one"); return -1; }
	const double *edge = (const double *)(c->h_out + (c->off_edge - c->off_res)), *pose = (const double *)(c->h_out + (c->off_pose - c->off_res));
	for (int q = 0; q < d.nK; q++) std::memcpy(cap->edge_pose + (size_t)q * PD, edge + ((size_t)d.o_edge + q) * PDX, 8 * (size_t)PD); // SE2: drop the cached cos/sin
	if (cap->pose) for (long long q = 0; q < 2LL * d.n_pairs; q++) std::memcpy(cap->pose + (size_t)q * PD, pose + ((size_t)d.o_pair * 2 + q) * PDX, 8 * (size_t)PD);
	return 0;
}

int srba_hip_set_phase_timing(srba_hip_ctx *c, int on) { if (!c) return -1; c->phase_timing = on != 0; return 0; } // takes effect at the next upload (the counters are part of the work arena layout)
int64_t srba_hip_debug_size(srba_hip_ctx *c, int what) { if (c && what == 10) return c->phase_timing ? 16 * (int64_t)c->n_prob : 0; if (c && what == 11) return 4 * (int64_t)c->n_prob;
	return (c && what >= 0 && what < 10) ? c->len_dbg[what] : -1; }
int srba_hip_debug_read(srba_hip_ctx *c, int what, double *out, int64_t n_doubles) {
	if (c && what == 11) { // per-capsule solver shape: [LDS bytes reserved by its launch, nb, off-diagonal blocks, block updates per factorisation]
		if (n_doubles < 4 * (int64_t)c->n_prob) return -1;
		for (int p = 0; p < c->n_prob; p++) { const ProbDesc &d = c->desc[p]; out[4 * p] = (double)c->cls_lds[c->cls_of[p]]; out[4 * p + 1] = d.nb; out[4 * p + 2] = d.nnzoff;
			out[4 * p + 3] = d.n_items; }
		return 0;
	}
	if (c && what == 10) { // per-capsule phase cycle counters (100 MHz wall clock ticks), as doubles
		if (!c->phase_timing) return -1;
		std::vector<long long> v(16 * (size_t)c->n_prob); HIPCHK(c, hipMemcpyAsync(v.data(), c->d_wk + c->off_phase, 8 * v.size(), hipMemcpyDeviceToHost, c->stream)); HIPCHK(c,
			hipStreamSynchronize(c->stream));
		for (size_t i = 0; i < v.size() && (int64_t)i < n_doubles; i++) out[i] = (double)v[i]; return 0;
	}
	if (!c || what < 0 || what >= 10 || n_doubles < c->len_dbg[what]) return -1;
	HIPCHK(c, hipSetDevice(c->device));
	if (what == 1 && c->jp_stale) { SRBA_DISPATCH(c, k_jacobians_only, 0); HIPCHK(c, hipGetLastError()); c->jp_stale = false; } // the fused linearisation keeps the Jacobian blocks on the chip:
		// materialise them for the reader
	if (what == 9 && c->dm.PDX() != c->dm.PD) { // ST poses: strip the cached cos/sin of the device layout
		const int PD = c->dm.PD, PDX = c->dm.PDX(); std::vector<double> v((size_t)c->n_pose_total * PDX);
		HIPCHK(c, hipMemcpyAsync(v.data(), c->d_wk + c->off_dbg[9], 8 * v.size(), hipMemcpyDeviceToHost, c->stream)); HIPCHK(c, hipStreamSynchronize(c->stream));
		for (long long q = 0; q < c->n_pose_total; q++) for (int k = 0; k < PD; k++) out[q * PD + k] = v[(size_t)q * PDX + k];
		return 0;
	}
	if (what == 8) { std::vector<int> v((size_t)c->len_dbg[8]); HIPCHK(c, hipMemcpyAsync(v.data(), c->d_wk + c->off_dbg[8], 4 * v.size(), hipMemcpyDeviceToHost, c->stream)); HIPCHK(c,
		hipStreamSynchronize(c->stream)); for (size_t i = 0; i < v.size(); i++) out[i] = v[i]; return 0; }
	HIPCHK(c, hipMemcpyAsync(out, c->d_wk + c->off_dbg[what], 8 * (size_t)c->len_dbg[what], hipMemcpyDeviceToHost, c->stream));
	HIPCHK(c, hipStreamSynchronize(c->stream)); return 0;
}

} // extern "C"
