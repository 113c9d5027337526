/*
 * srba_big.hip -- the multi-workgroup path for large capsules: the grid-wide kernels of srba_big.hpp and the host code that drives them (lock-step gangs, LM control on the
 * host). Its own translation unit since round 5: it compiles beside srba_hip.hip instead of inside it. Entry points: srba_ctx.hpp.
 */
#include <hip/hip_runtime.h>
#include "../../include/srba_hip.h"
#include "srba_device.hpp"
#include "srba_big.hpp"
#define SRBA_FLAT_DECLS_ONLY
#include "srba_flat.hpp"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <memory>
#include <thread>
#include "srba_ctx.hpp"

using namespace srbadev;

// =================================================================================================== the multi-workgroup path for large capsules (srba_big.hpp)
static inline int big_grid(long long items, int block) { return (int)std::max<long long>(1, std::min<long long>((items + block - 1) / block, 4096)); }
// The LM loop of a large capsule is driven from the host. Since round 4 a batch that holds several such capsules runs them as a GANG in lock-step (big_gang_run): every
// grid-wide phase is ONE launch for all windows that need it (blockIdx.y = slot of the gang, srbadev::Gang::mask = who takes part), the host reads the scalars of all
// windows back with one stream synchronisation per phase group and walks the control flow of optimize_edges.h:454-696 for each window on its own; a window that finishes
// hands its slot to the next capsule of the class. The ~60 short dependent launches of a factorisation are shared by up to 16 windows instead of being queued 16 times.
// SRBA_HIP_BIG_GANG=0 selects the earlier scheme: up to kBigLanes host threads, each driving a gang of one on its own stream with its own scalar / partial-sum buffers.
using srbadev::BS_CHI2; using srbadev::BS_MAXDIAG; using srbadev::BS_DEN; using srbadev::BS_NINF; using srbadev::BS_LAMBDA;
static void gang_set(srba_hip_ctx *c, srbadev::Gang &G, int w, int p) {
	const ProbDesc &d = c->desc[p]; G.p[w] = p; G.ld[w] = c->big_ld[p]; G.nsys[w] = d.n_sys; G.A[w] = c->B.dense + d.o_dense; G.n = std::max(G.n, w + 1);
}
static srbadev::Gang gang_of(const BigLane *ln) { srbadev::Gang G; std::memset(&G, 0, sizeof(G)); G.part = ln->d_part; G.scal = ln->d_scal; G.iscal = ln->d_iscal; return G; }
static srbadev::Gang gang_masked(const srbadev::Gang &G, unsigned mask) { srbadev::Gang H = G; H.mask = mask; return H; }
// grid of a gang launch: x = the largest grid any participating window would have alone, y = slots
template <class ItemsF> static dim3 gang_grid(srba_hip_ctx *c, const srbadev::Gang &G, int block, bool one_workgroup_per_item, ItemsF &&items) {
	long long gx = 1; int gy = 1;
	for (int w = 0; w < G.n; w++) if ((G.mask >> w) & 1u) { const long long it = items(c->desc[G.p[w]], w); gx = std::max<long long>(gx, one_workgroup_per_item ? std::max<long long>(1,
		it) : big_grid(it, block)); gy = w + 1; }
	return dim3((unsigned)gx, (unsigned)gy);
}
#define BIGKG(KERNEL, ITEMS, block, ...) do { if (G.mask) with_family(c->params.family, [&](auto fam_) { hipLaunchKernelGGL((srbadev::KERNEL<decltype(fam_)::value>), gang_grid(c, G, \
	(block), true, [&](const ProbDesc &d, int) -> long long { return (ITEMS); }), dim3(block), 0, st, c->B, c->dp, G, ##__VA_ARGS__); }); } while (0)
#define BIGK(KERNEL, ITEMS, block, ...) do { if (G.mask) with_family(c->params.family, [&](auto fam_) { hipLaunchKernelGGL((srbadev::KERNEL<decltype(fam_)::value>), gang_grid(c, G, \
	(block), false, [&](const ProbDesc &d, int) -> long long { return (ITEMS); }), dim3(block), 0, st, c->B, c->dp, G, ##__VA_ARGS__); }); } while (0)
// deterministic reduction of per-workgroup partials into scal[slot] of every participating window
static void big_reduce(srba_hip_ctx *c, hipStream_t st, const srbadev::Gang &G, int which, int kind, int slot, int is_max) { if (G.mask) hipLaunchKernelGGL(srbadev::kb_reduce, dim3(1, G.n),
	dim3(256), 0, st, c->B, G, which, kind, slot, is_max); }
static bool big_schur(const srba_hip_ctx *c, const ProbDesc &d) { return c->params.solver != SRBA_SOLVER_NO_SCHUR_SPARSE_CHOL && d.nF > 0 && d.nK > 0; }
static unsigned gang_schur_mask(srba_hip_ctx *c, const srbadev::Gang &G) { unsigned m = 0; for (int w = 0; w < G.n; w++) if (((G.mask >> w) & 1u) && big_schur(c, c->desc[G.p[w]])) m |= 1u << w;
	return m; }
static void big_copy_vec(srba_hip_ctx *c, hipStream_t st, const srbadev::Gang &G, int kind) {
	if (!G.mask) return;
	hipLaunchKernelGGL(srbadev::kb_copy_vec, gang_grid(c, G, 256, false, [&](const ProbDesc &d, int) -> long long { return kind == 2 ? (long long)d.n_obs * c->dm.O : d.n_scal; }), dim3(256), 0, st,
		c->B, G, kind, c->dm.O);
}
static void big_set_lambda(hipStream_t st, const srbadev::Gang &G, const srbadev::GangLambda &lam) { if (G.mask) hipLaunchKernelGGL(srbadev::kb_set_lambda, dim3(1), dim3(64), 0, st, G, lam); }
// solve(lambda) of lev-marq_solvers.h for the windows of G.mask, lambda in scal[BS_LAMBDA] of each: (a) Schur reduction (if the solver has one) + dense assembly,
// (b) blocked Cholesky, (c) back-substitution + landmark increments. The not-positive-definite verdict of a window stays in its device flag.
static void big_enqueue_assemble(srba_hip_ctx *c, hipStream_t st, const srbadev::Gang &Gall) {
	const int P = c->dm.P;
	{ const srbadev::Gang G = gang_masked(Gall, gang_schur_mask(c, Gall));
	  BIGK(kb_schur_inv, std::max<long long>(d.nF, (long long)d.n_hap * P * P), 128); if (c->sch_wave) { int gy = 0; for (int w = 0; w < G.n; w++) if ((G.mask >> w) & 1u) gy = w + 1; /* = the grid's y (gang_grid) */
	    BIGKG(kb_schur_reduce_wave, (d.n_vb + 3) / 4, 256, (c->sch_xcd && gy % 8 == 0) ? 1 : 0); } else BIGKG(kb_schur_reduce, d.n_hap, 256);
	  BIGKG(kb_schur_grad, d.nK, 256); }
	if (!Gall.mask) return;
	hipLaunchKernelGGL(srbadev::kb_dense_clear, gang_grid(c, Gall, 256, false, [&](const ProbDesc &, int w) -> long long { return (long long)Gall.ld[w] * Gall.ld[w]; }), dim3(256), 0, st, Gall);
	const unsigned ms = gang_schur_mask(c, Gall);
	{ const srbadev::Gang G = gang_masked(Gall, ms); BIGK(kb_dense_assemble, d.n_hap + d.n_hapf + d.n_hf, 128, 0); }
	{ const srbadev::Gang G = gang_masked(Gall, Gall.mask & ~ms); BIGK(kb_dense_assemble, d.n_hap + d.n_hapf + d.n_hf, 128, 1); }
}
static srbadev::BigSys big_sys(const srbadev::Gang &G, int w) { srbadev::BigSys S; const int ld = G.ld[w]; S.A = G.A[w]; S.Ldiag = S.A + (size_t)ld * ld; S.rhs = S.Ldiag + (size_t)ld * srbadev::CB;
	S.y = S.rhs + ld; S.flag = G.iscal + w * 8 + 1; S.n = G.nsys[w]; S.ld = ld; return S; }
static void big_enqueue_cholesky(srba_hip_ctx *c, hipStream_t st, const srbadev::Gang &G) {
	if (!G.mask) return;
	int ldmax = 0, nw = 0, w1 = 0; for (int w = 0; w < G.n; w++) if ((G.mask >> w) & 1u) { ldmax = std::max(ldmax, G.ld[w]); nw++; w1 = w; }
	if (c->big_persistent && nw == 1) { // the whole factorisation in one launch: panel steps and trailing updates separated by grid barriers (srba_big.hpp, k_chol_persistent); one window per launch
		const srbadev::BigSys S = big_sys(G, w1);
		const int below0 = S.ld - srbadev::CB, nt0 = below0 > 0 ? (below0 + srbadev::CT - 1) / srbadev::CT : 0, resident = 4 * c->n_cu / std::max(1,
			c->n_lanes_ready) /* the grid barriers need every workgroup of every window in flight resident: 4 workgroups per CU (34 KB of LDS, 256 threads each) shared by the lanes */,
		          Gn = std::max(1, std::min(std::min(120, resident), std::max(nt0 * (nt0 + 1) / 2, 1 + (below0 > 0 ? (below0 + 63) / 64 : 0))));
		unsigned *bar = (unsigned *)(G.iscal + w1 * 8 + 4);
		(void)hipMemsetAsync(bar, 0, 4, st);
		hipLaunchKernelGGL(srbadev::k_chol_persistent, dim3(Gn), dim3(256), 0, st, S, bar);
		return;
	}
	if (c->big_fused_step) { // one launch per 32 columns: the panel step and, beside it, the trailing update of the step before (srba_big.hpp, k_chol_step)
		for (int k0 = 0; k0 < ldmax; k0 += srbadev::CB) { const int below = ldmax - k0 - srbadev::CB, nt = below > 0 ? (below + srbadev::CT - 1) / srbadev::CT : 0;
			hipLaunchKernelGGL(srbadev::k_chol_step, dim3(1 + (below + 63) / 64 + (k0 > 0 ? nt * (nt + 1) / 2 : 0), G.n), dim3(256), 0, st, G, k0); }
		return;
	}
	for (int k0 = 0; k0 < ldmax; k0 += srbadev::CB) { // windows smaller than the largest of the gang drop out of the later steps inside the kernels
		const int below = ldmax - k0 - srbadev::CB;
		hipLaunchKernelGGL(srbadev::k_chol_panel, dim3(1 + (below + 63) / 64, G.n), dim3(64), 0, st, G, k0);
		if (below > 0) { const int nt = (below + srbadev::CT - 1) / srbadev::CT; hipLaunchKernelGGL(srbadev::k_chol_update, dim3(nt * (nt + 1) / 2, G.n), dim3(256), 0, st, G, k0); }
	}
}
static void big_enqueue_backsub(srba_hip_ctx *c, hipStream_t st, const srbadev::Gang &Gall) {
	if (!Gall.mask) return;
	hipLaunchKernelGGL(srbadev::k_chol_bsub, dim3(1, Gall.n), dim3(256), 0, st, Gall);
	hipLaunchKernelGGL(srbadev::kb_take_delta, gang_grid(c, Gall, 256, false, [&](const ProbDesc &d, int) -> long long { return d.n_scal; }), dim3(256), 0, st, c->B, Gall);
	{ const srbadev::Gang G = gang_masked(Gall, gang_schur_mask(c, Gall)); BIGK(kb_schur_features, d.nF, 128, 1); }
}
// The factorisation sequence between two timing events -- on every big_time_every-th sequence of a lane only (SRBA_HIP_BIG_TIME_EVERY, default 8): an event with a time stamp
// drains the queue before and after it, 30 - 40 us per sequence in the kernel trace (4 - 5 % of a gang's time when every sequence was timed). The counts stay complete.
static int big_timed_cholesky(srba_hip_ctx *c, BigLane *ln, const srbadev::Gang &G) {
	ln->timed = (ln->seq_no++ % std::max(1, c->big_time_every)) == 0;
	if (!ln->timed) { big_enqueue_cholesky(c, ln->stream, G); return 0; }
	if (!ln->e0) { LNCHK(ln, hipEventCreate(&ln->e0)); LNCHK(ln, hipEventCreate(&ln->e1)); }
	LNCHK(ln, hipEventRecord(ln->e0, ln->stream));
	big_enqueue_cholesky(c, ln->stream, G);
	LNCHK(ln, hipEventRecord(ln->e1, ln->stream));
	return 0;
}
static void big_account_cholesky(srba_hip_ctx *c, BigLane *ln, const srbadev::Gang &G) { // after a stream synchronisation; chol_ms is the time of the TIMED launch sequences
	// (each shared by the windows of its gang), t_seqs / t_flops what they factored; chol_seqs / chol_count / chol_flops count every sequence
	float ms = 0; const bool timed = ln->timed && hipEventElapsedTime(&ms, ln->e0, ln->e1) == hipSuccess; double fl = 0;
	for (int w = 0; w < G.n; w++) if ((G.mask >> w) & 1u) { const double ld = G.ld[w]; fl += ld * ld * ld / 3.0; ln->chol_count++; ln->chol_nmax = std::max(ln->chol_nmax, G.nsys[w]); }
	ln->chol_flops += fl; ln->chol_seqs++;
	if (timed) { ln->chol_ms += ms; ln->t_seqs++; ln->t_flops += fl; }
}
int big_solve(srba_hip_ctx *c, BigLane *ln, int p, double lambda, bool *pos_def) { // the stepwise entry point (srba_hip_solve)
	hipStream_t st = ln->stream; srbadev::Gang G = gang_of(ln); gang_set(c, G, 0, p); G.mask = 1u;
	{ srbadev::GangLambda lam; std::memset(&lam, 0, sizeof(lam)); lam.v[0] = lambda; big_set_lambda(st, G, lam); }
	{ const ProbDesc &d = c->desc[p]; // (extension) start from the gradient as srba_hip_linearize left it
	  if ((c->params.extensions & SRBA_EXT_SCHUR_KEEPS_GRADIENT) && big_schur(c, d)) big_copy_vec(c, st, G, 0); }
	big_enqueue_assemble(c, st, G);
	if (big_timed_cholesky(c, ln, G) != 0) return -1;
	big_enqueue_backsub(c, st, G);
	int flag = 0; LNCHK(ln, hipMemcpyAsync(&flag, ln->d_iscal + 1, 4, hipMemcpyDeviceToHost, ln->stream)); LNCHK(ln, hipStreamSynchronize(ln->stream));
	big_account_cholesky(c, ln, G);
	if (flag == 2) { ln->error = "k_chol_persistent: a grid barrier timed out (the workgroups of the factorisation were not all resident)"; return -1; }
	*pos_def = (flag == 0);
	LNCHK(ln, hipGetLastError());
	return 0;
}
// optimize_edges S5..S17 for the large capsules caps[next++ ...]: the control flow of k_lm_run (optimize_edges.h:256-751) on the host for up to `nslots` windows at once,
// every phase a grid-wide launch over the windows that are at that point of their loop. `ln` supplies the stream and buffers with room for nslots slots.
namespace {
enum { GS_IDLE = 0, GS_NEW, GS_TRIAL, GS_ACCEPT, GS_FINAL };
struct GangSlot {
	int p = -1, st = GS_IDLE; bool schur = false, keep_g = false, s11 = false, relin = false, restore = false, stop = false;
	srba_lm_result out; double lambda = 0, nu = 2, total_err = 0, RMSE = 0, rho = 0, new_err = 0, new_RMSE = 0; int iter = 0, trials = 0, tr = 0, n_notpd = 0, n_acc = 0, n_relin = 0, stopmask = 0;
};
}
// host control flow of one window from the head of the inner loop (`while (rho <= 0 && !stop)`) to its next trial or to the end of the run
static void gang_advance(const srba_hip_params &prm, GangSlot &s, bool from_iter_head) {
	for (;;) {
		if (!from_iter_head) {
			if (s.rho <= 0 && !s.stop) { s.tr = s.trials++; if (s.tr < SRBA_TRACE_LEN) s.out.trace_lambda[s.tr] = s.lambda; s.out.lambda_last_trial = s.lambda; s.st = GS_TRIAL; return; }
			s.iter++;
		}
		from_iter_head = false;
		if (!(s.iter < prm.max_iters && !s.stop)) { if (!s.stop) s.stopmask |= 1 << SRBA_STOP_MAX_ITERS; s.st = GS_FINAL; return; }
		s.rho = 0;
		if (s.lambda >= prm.max_lambda) { s.stop = true; s.stopmask |= 1 << SRBA_STOP_LAMBDA; }
		if (s.RMSE < prm.max_error_per_obs_to_stop) { s.stop = true; s.stopmask |= 1 << SRBA_STOP_RMSE; }
	}
}
static int big_gang_run(srba_hip_ctx *c, BigLane *ln, const int32_t *caps, int count, std::atomic<int> &next, int nslots) {
	const int O = c->dm.O, L = c->dm.L; const srba_hip_params &prm = c->params; hipStream_t st = ln->stream;
	nslots = std::max(1, std::min(nslots, ln->slots));
	std::vector<GangSlot> S(nslots); std::vector<std::unique_ptr<srba_lm_result>> results; // result records stay alive until the last copy has been waited for
	struct WaitOnExit { hipStream_t s; ~WaitOnExit() { (void)hipStreamSynchronize(s); } } wait_on_exit{st}; // (declared after `results`: destroyed before it, also on the error returns)
	srbadev::Gang G0 = gang_of(ln);
	const size_t fetch_bytes = (8 * 16 + 4 * 8) * (size_t)ln->slots; // scalars of all slots, then their flags: one allocation, one copy into page-locked memory
	if (!ln->h_fetch) LNCHK(ln, hipHostMalloc(&ln->h_fetch, fetch_bytes, hipHostMallocDefault));
	const double *hs = (const double *)ln->h_fetch; const int *hi = (const int *)(hs + 16 * (size_t)ln->slots);
	auto fetch = [&]() -> int { LNCHK(ln, hipMemcpyAsync(ln->h_fetch, ln->d_scal, fetch_bytes, hipMemcpyDeviceToHost, st)); LNCHK(ln, hipStreamSynchronize(st)); return 0; };
	auto mask_of = [&](auto pred) { unsigned m = 0; for (int w = 0; w < nslots; w++) if (S[w].p >= 0 && pred(S[w])) m |= 1u << w; return m; };
	auto enqueue_residuals = [&](const srbadev::Gang &G, int to_trial_copy, int use_skip) { BIGK(kb_residuals, d.n_obs, 256, to_trial_copy, use_skip); big_reduce(c, st, G, 0, 0, BS_CHI2, 0); };
	auto enqueue_linearize = [&](const srbadev::Gang &G) { BIGK(kb_jac_init, d.n_valid, 256); BIGK(kb_jac, d.n_bp + d.n_bf, 128); BIGK(kb_jac_post, d.n_bp + d.n_bf, 256); BIGK(kb_hessian,
		d.n_hap + d.n_hf + d.n_hapf, 128); BIGKG(kb_hessian_heavy, d.n_hap, 256); };
	// (extension) the Schur kernels reduce B.grad in place: keep what K5 produced and start every solve from it
	auto enqueue_gradient = [&](const srbadev::Gang &G) { BIGKG(kb_gradient, d.nK + (d.nF + 255) / 256, 256); big_copy_vec(c, st, gang_masked(G, G.mask & mask_of([](const GangSlot &s) {
		return s.keep_g; })), 1); };
	auto enqueue_dot = [&](const srbadev::Gang &G, int which, int slot, int is_max, int use_skip) { BIGK(kb_dot, d.n_scal, 256, use_skip); big_reduce(c, st, G, which, 2, slot, is_max); };
	for (;;) {
		// ---- windows that ended: covariance recovery (S17) and the result record; their slots take the next capsules of the class
		{ srbadev::Gang G = G0; for (int w = 0; w < nslots; w++) if (S[w].st == GS_FINAL) { gang_set(c, G, w, S[w].p); G.mask |= 1u << w; }
		  // a rejected last trial is undone first
		  { const srbadev::Gang Gr = gang_masked(G, G.mask & mask_of([](const GangSlot &s) { return s.restore; })); const srbadev::Gang &G = Gr; BIGK(kb_restore, d.nK + (long long)d.nF * L + d.n_req,
		  	128); }
		  if (G.mask) { const srbadev::Gang Gs = gang_masked(G, gang_schur_mask(c, G)), Gn = gang_masked(G, G.mask & ~Gs.mask);
		    { const srbadev::Gang &G = Gs; BIGK(kb_cov_recovery, std::max(d.nF, 1), 128, 1); } { const srbadev::Gang &G = Gn; BIGK(kb_cov_recovery, std::max(d.nF, 1), 128, 0); } }
		  for (int w = 0; w < nslots; w++) if (S[w].st == GS_FINAL) { GangSlot &s = S[w];
		    s.out.num_iters = s.iter; s.out.num_trials = s.trials; s.out.num_not_pd = s.n_notpd; s.out.num_accepted = s.n_acc; s.out.num_relinearized = s.n_relin; s.out.stop_reason = s.stopmask;
		    s.out.total_sqr_error_final = s.total_err; s.out.obs_rmse = s.RMSE; s.out.lambda_final = s.lambda;
		    results.emplace_back(new srba_lm_result(s.out)); LNCHK(ln, hipMemcpyAsync(c->B.results + s.p, results.back().get(), sizeof(srba_lm_result), hipMemcpyHostToDevice, st));
		    s = GangSlot(); } }
		for (int w = 0; w < nslots; w++) if (S[w].st == GS_IDLE) { const int i = next.fetch_add(1); if (i >= count) break; GangSlot &s = S[w]; s = GangSlot(); s.p = caps[i]; s.st = GS_NEW;
			const ProbDesc &d = c->desc[s.p]; s.schur = big_schur(c, d); s.keep_g = s.schur && (prm.extensions & SRBA_EXT_SCHUR_KEEPS_GRADIENT); s.s11 = (long long)O * d.n_obs < (long long)d.n_scal;
			std::memset(&s.out, 0, sizeof(s.out)); for (int k = 0; k < SRBA_TRACE_LEN; k++) { s.out.trace_chi2[k] = NAN; s.out.trace_lambda[k] = NAN; s.out.trace_rho[k] = NAN; }
				s.out.lambda_last_trial = NAN; }
		srbadev::Gang Gall = G0; bool any = false; for (int w = 0; w < nslots; w++) if (S[w].p >= 0) { gang_set(c, Gall, w, S[w].p); any = true; }
		if (!any) break;
		// ---- rejected trials: restore (K12); accepted trials: residuals of the trial become current, relinearise where the error moved enough, gradient, |g|_inf;
		//      new windows: S5 numeric spanning tree, S6/S7/S10 linearisation, S12 lambda_0, S13 residuals, S14 gradient
		const unsigned m_restore = mask_of([](const GangSlot &s) { return s.restore && s.st != GS_FINAL; }), m_accept = mask_of([](const GangSlot &s) { return s.st == GS_ACCEPT; }),
			m_relin = mask_of([](const GangSlot &s) { return s.st == GS_ACCEPT && s.relin; }),
		               m_new = mask_of([](const GangSlot &s) { return s.st == GS_NEW; }), m_new_full = mask_of([](const GangSlot &s) { return s.st == GS_NEW && !s.s11; });
		{ const srbadev::Gang G = gang_masked(Gall, m_restore); BIGK(kb_restore, d.nK + (long long)d.nF * L + d.n_req, 128); for (int w = 0; w < nslots; w++) S[w].restore = false; }
		for (int w = 0; w < nslots; w++) if ((m_new >> w) & 1u) LNCHK(ln, hipMemsetAsync(ln->d_iscal + w * 8, 0, 32, st));
		big_copy_vec(c, st, gang_masked(Gall, m_accept), 2);
		{ const srbadev::Gang G = gang_masked(Gall, m_new); BIGK(kb_spantree, d.n_pairs, 256, 0, 0); }
		enqueue_linearize(gang_masked(Gall, m_relin | m_new));
		{ const srbadev::Gang G = gang_masked(Gall, m_new_full); BIGK(kb_maxdiag, d.nK + d.nF, 256); big_reduce(c, st, G, 1, 1, BS_MAXDIAG, 1); enqueue_residuals(G, 0, 0); }
		enqueue_gradient(gang_masked(Gall, m_accept | m_new_full));
		enqueue_dot(gang_masked(Gall, m_accept), 2, BS_NINF, 1, 0);
		if (m_accept | m_new) {
			if (fetch() != 0) return -1;
			for (int w = 0; w < nslots; w++) { GangSlot &s = S[w]; const double *h = hs + (size_t)w * 16;
				if ((m_new >> w) & 1u) { const ProbDesc &d = c->desc[s.p];
					s.out.num_invalid_jacobs = hi[(size_t)w * 8]; s.out.num_observations = d.n_obs; s.out.num_jacobians = d.n_bp + d.n_bf; s.out.num_span_tree_numeric_updates = d.n_pairs;
					if (s.s11) { s.out.status = 1; results.emplace_back(new srba_lm_result(s.out)); LNCHK(ln, hipMemcpyAsync(c->B.results + s.p, results.back().get(), sizeof(srba_lm_result),
						hipMemcpyHostToDevice, st)); s = GangSlot(); continue; } // S11
					s.lambda = h[BS_MAXDIAG] * 1e-3; s.nu = 2.0; s.total_err = h[BS_CHI2]; s.RMSE = std::sqrt(s.total_err / d.n_obs); s.out.lambda_init = s.lambda;
						s.out.total_sqr_error_init = s.total_err;
					s.iter = 0; gang_advance(prm, s, true);
				} else if ((m_accept >> w) & 1u) {
					if (h[BS_NINF] <= 1e-15) { s.stop = true; s.stopmask |= 1 << SRBA_STOP_GRADIENT; }
					if (s.RMSE < prm.max_error_per_obs_to_stop) { s.stop = true; s.stopmask |= 1 << SRBA_STOP_RMSE; }
					if (s.rho > prm.max_rho) { s.stop = true; s.stopmask |= 1 << SRBA_STOP_RHO; }
					s.lambda *= 1.0 / 3.0; s.nu = 2.0; gang_advance(prm, s, false);
				} }
		}
		// ---- one trial of every window that is in its inner loop: solve, apply, numeric spanning tree of the poses in use, residuals, rho denominator; the kernels after
		//      the factorisation return at once for a window whose factorisation failed
		const unsigned m_trial = mask_of([](const GangSlot &s) { return s.st == GS_TRIAL; });
		if (m_trial) {
			const srbadev::Gang G = gang_masked(Gall, m_trial);
			{ srbadev::GangLambda lam; std::memset(&lam, 0, sizeof(lam)); for (int w = 0; w < nslots; w++) lam.v[w] = S[w].lambda; big_set_lambda(st, G, lam); }
			big_copy_vec(c, st, gang_masked(G, m_trial & mask_of([](const GangSlot &s) { return s.keep_g; })), 0);
			big_enqueue_assemble(c, st, G);
			if (big_timed_cholesky(c, ln, G) != 0) return -1;
			big_enqueue_backsub(c, st, G);
			BIGK(kb_apply, d.nK + (long long)d.nF * L + d.n_req, 128, 1);
			BIGK(kb_spantree, d.n_need, 256, 1, 1);
			enqueue_residuals(G, 1, 1);
			enqueue_dot(G, 1, BS_DEN, 0, 1);
			if (fetch() != 0) return -1;
			big_account_cholesky(c, ln, G);
			for (int w = 0; w < nslots; w++) if ((m_trial >> w) & 1u) { GangSlot &s = S[w]; const double *h = hs + (size_t)w * 16; const int hflag = hi[(size_t)w * 8 + 1];
				const ProbDesc &d = c->desc[s.p];
				if (hflag == 2) { ln->error = "k_chol_persistent: a grid barrier timed out (the workgroups of the factorisation were not all resident)"; return -1; }
				if (hflag) { s.n_notpd++; s.lambda *= s.nu; s.nu *= 2.0; s.stop = (s.lambda > prm.max_lambda); if (s.stop) s.stopmask |= 1 << SRBA_STOP_LAMBDA; gang_advance(prm, s, false); continue; }
				const double new_err = h[BS_CHI2], new_RMSE = std::sqrt(new_err / d.n_obs), err_red = s.total_err > 0 ? (s.total_err - new_err) / s.total_err : 0;
				s.rho = (s.total_err - new_err) / h[BS_DEN];
				if (s.tr < SRBA_TRACE_LEN) { s.out.trace_chi2[s.tr] = new_err; s.out.trace_rho[s.tr] = s.rho; }
				if (s.rho > 0) { s.n_acc++; s.relin = (err_red < 0 || err_red > prm.min_error_reduction_ratio_to_relinearize); s.total_err = new_err; s.RMSE = new_RMSE; if (s.relin) s.n_relin++;
					s.st = GS_ACCEPT; }
				else { s.restore = true; s.lambda *= s.nu; s.nu *= 2.0; s.stop = (s.lambda > prm.max_lambda); if (s.stop) s.stopmask |= 1 << SRBA_STOP_LAMBDA; gang_advance(prm, s, false); }
			}
		}
	}
	LNCHK(ln, hipStreamSynchronize(st));
	LNCHK(ln, hipGetLastError());
	return 0;
}
// lanes [0, n): lane 0 is the context's own stream with buffers for a whole gang, the others (one window each, SRBA_HIP_BIG_GANG=0) get theirs on first use
int big_prepare_lanes(srba_hip_ctx *c, int n) {
	n = std::max(1, std::min(n, kBigLanes));
	BigLane &l0 = c->lanes[0]; l0.id = 0; l0.stream = c->stream; l0.d_part = c->d_part; l0.d_scal = c->d_scal; l0.d_iscal = (int *)(c->d_scal + 16 * srbadev::kGang); l0.slots = srbadev::kGang;
	c->n_lanes_ready = std::max(c->n_lanes_ready, 1);
	for (int i = c->n_lanes_ready; i < n; i++) {
		BigLane &l = c->lanes[i]; l.id = i; l.slots = srbadev::kGang; // (every lane can hold a gang: SRBA_HIP_BIG_GANGS > 1 runs several gangs side by side)
		HIPCHK(c, hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking));
		HIPCHK(c, hipMalloc((void **)&l.d_part, 8 * 3 * kBigPart * srbadev::kGang)); HIPCHK(c, hipMalloc((void **)&l.d_scal, (8 * 16 + 4 * 8) * srbadev::kGang));
			l.d_iscal = (int *)(l.d_scal + 16 * srbadev::kGang);
		c->n_lanes_ready = i + 1;
	}
	return n;
}
void big_collect_lane_stats(srba_hip_ctx *c) { for (int i = 0; i < c->n_lanes_ready; i++) { BigLane &l = c->lanes[i]; c->big_chol_ms += l.chol_ms; c->big_chol_flops += l.chol_flops;
	c->big_chol_count += l.chol_count; c->big_chol_seqs += l.chol_seqs; c->big_chol_nmax = std::max(c->big_chol_nmax, l.chol_nmax); c->big_t_seqs += l.t_seqs; c->big_t_flops += l.t_flops;
	l.chol_ms = l.chol_flops = l.t_flops = 0; l.chol_count = l.chol_seqs = l.t_seqs = 0;
	l.chol_nmax = 0; } }
// all capsules of the big class: a gang on lane 0 (default), or dealt to several lanes (host threads) with one window each
int big_run_class(srba_hip_ctx *c, const int32_t *caps, int count) {
	if (count <= 0) return 0;
	const bool gang = c->big_gang && !c->big_persistent;
	// several gangs side by side (big_gangs > 1): the windows of the class are dealt to that many lanes, each a lock-step gang on its own stream and host thread -- the latency-bound
	// phases of one gang (the panel chains of the factorisation use a few CUs) run under the throughput-bound ones of the others (Schur reduction, Hessian)
	const int ngang = gang ? std::max(1, std::min(std::min(c->big_gangs, kBigLanes), count)) : 1;
	const int per_gang = gang ? std::max(1, std::min(c->big_gang_slots, (count + ngang - 1) / ngang)) : 1;
	// lane 0 is the context stream; SRBA_HIP_BIG_FRESH=1 keeps the gangs off it: their streams are then created one after the other, and the runtime deals streams to its hardware
	// queues round-robin -- consecutive streams never share a queue, while the context stream (created long before, dozens of streams ago) may share one with a lane
	const int l0 = (gang && ngang > 1 && c->big_fresh) ? 1 : 0;
	const int n = big_prepare_lanes(c, l0 + (gang ? ngang : (c->big_lanes_max <= 1 ? 1 : std::min(count, c->big_lanes_max)))); if (n < 1) return -1;
	int rc = 0; std::atomic<int> next(0);
	if (n == 1) { rc = big_gang_run(c, &c->lanes[0], caps, count, next, gang ? c->big_gang_slots : 1); }
	else {
		// the lanes start after everything already queued on the context stream (uploads, state resets)
		hipEvent_t ready = nullptr; HIPCHK(c, hipEventCreateWithFlags(&ready, hipEventDisableTiming));
		hipError_t e = hipEventRecord(ready, c->stream);
		for (int i = 1; i < n && e == hipSuccess; i++) e = hipStreamWaitEvent(c->lanes[i].stream, ready, 0);
		if (e != hipSuccess) { hipEventDestroy(ready); c->fail(std::string("large-capsule path: ") + hipGetErrorString(e)); return -1; }
		std::vector<int> rcs(n, 0); std::vector<std::thread> th;
		auto work = [&](int li) { // no exception leaves a worker (std::terminate otherwise) nor this function (it is reached from an extern "C" entry)
			try { hipSetDevice(c->device); BigLane *ln = &c->lanes[li]; rcs[li] = big_gang_run(c, ln, caps, count, next, per_gang); hipStreamSynchronize(ln->stream); }
			catch (const std::exception &ex) { rcs[li] = -1; c->lanes[li].error = std::string("large-capsule path: ") + ex.what(); }
			catch (...) { rcs[li] = -1; c->lanes[li].error = "large-capsule path: unknown exception"; } };
		try { for (int i = l0 + 1; i < n; i++) th.emplace_back(work, i); } catch (...) { /* fewer threads than lanes: the ones that started (and this thread) share the capsules */ }
		work(l0);
		for (auto &t : th) t.join();
		hipEventDestroy(ready);
		for (int i = 0; i < n; i++) if (rcs[i] != 0 && rc == 0) { rc = -1; c->fail(c->lanes[i].error.empty() ? std::string("large-capsule path failed") : c->lanes[i].error); }
	}
	if (rc != 0 && c->error.empty()) c->fail(c->lanes[0].error);
	big_collect_lane_stats(c);
	return rc;
}
#undef BIGK
#undef BIGKG
unsigned long long big_layout_signature() { return layout_signature(); }
