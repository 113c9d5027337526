/*
 * srba_rounds.hpp -- the Levenberg-Marquardt loop of optimize_edges() (impl/optimize_edges.h:361-696) as ROUNDS over a batch: every capsule that still iterates
 * does exactly one LM trial per round, and a round is three persistent launches over the list of those capsules
 *
 *   kr_solve : K7..K10 + K11   schur / assemble (H + lambda I) / block-sparse LL^t / substitutions in LDS, then the trial unknowns exp(delta) (+) edge, lm + delta
 *   kr_eval  : K1(needed) + K4 the spanning-tree poses the trial moved, residuals, chi2, rho, accept / reject and the lambda schedule
 *   kr_lin   : K2 K3 K6 K5     (accepted trials only) Jacobians + Hessian blocks when the step asks for a relinearisation, minus-gradient, the stop tests
 *
 * instead of the whole loop inside one kernel (k_lm_run). Why: the fused kernel needs 234 VGPRs because its memory phases keep many gathers in flight, so the LDS-bound
 * solver -- more than half of a trial -- runs with two wavefronts per SIMD (5.4 resident per CU on the benchmark batch); the solver alone needs ~60 registers and is
 * limited by the 160 KB of LDS only (13 wavefronts per CU for the typical window), and the memory phases need no LDS at all. Groups of capsules (LDS size classes,
 * the big ones split) run their rounds on separate streams, so one group's LDS-bound solves overlap another group's HBM-bound evaluations.
 *
 * No state has to be backed up or restored: unknowns and spanning-tree poses exist twice, a trial reads the accepted copy and writes the other one, an accepted trial
 * flips which copy is the accepted one (LmState::cur), a rejected trial needs nothing. The reference's partial restore (optimize_edges.h:664-670: only the poses that
 * Jacobian blocks read go back to their old values, their twins keep the rejected trial's) is reproduced at the end (kr_finish), where it becomes visible.
 * Every phase is the device function the fused kernel uses, on the same values in the same order: both paths give bit-identical results.
 */
#pragma once

namespace srbadev {

__device__ __forceinline__ int rounds_pull(int *ctr) { int i = 0; if (threadIdx.x == 0) i = atomicAdd(ctr, 1); return __builtin_amdgcn_readfirstlane(i); }
__device__ __forceinline__ void rounds_append(const Rounds &R, int g, int slot, int first, int pidx) { if (threadIdx.x == 0) { const int q = atomicAdd(R.count + 3 * g + slot, 1); R.list[(size_t)slot * R.n_prob + first + q] = pidx; } }
// head of one pass of `for (iter...; iter < max_iters && !stop; iter++)` (optimize_edges.h:454-470): the two tests made before the inner loop
__device__ __forceinline__ void rounds_enter_iteration(LmState &s, const DevParams &prm) {
	if (s.iter < prm.max_iters && !s.stop) {
		if (s.lambda >= prm.max_lambda) { s.stop = 1; s.stopmask |= 1 << SRBA_STOP_LAMBDA; }
		if (s.rmse < prm.max_err) { s.stop = 1; s.stopmask |= 1 << SRBA_STOP_RMSE; }
		if (s.stop) { s.iter++; s.phase = 1; } else s.phase = 0; // (with stop set the inner loop is skipped, the pass ends, iter++ runs and the loop condition fails)
	} else s.phase = 1;
}
template <int FAM> __device__ __forceinline__ bool rounds_hess_terms(const Batch &B, const ProbDesc &d) { return B.hess_terms && d.dense_in_lds && d.n_hap * Solver<FAM>::P * Solver<FAM>::P <= 9 * (d.nb + d.nnzoff); }

// ---- S5 .. S14 of optimize_edges() for every capsule of a group, then the head of the first LM pass
template <int FAM> __global__ void __launch_bounds__(SRBA_WG) kr_init(const Batch B, const DevParams prm, const Rounds R, int g, int first, int count) {
	typedef Solver<FAM> SV; constexpr int P = SV::P, L = SV::L, O = SV::O, PD = SV::PD;
	for (int i = blockIdx.x; i < count; i = gridDim.x + rounds_pull(R.ctr + 5 * g + 3)) { // first capsule: one per workgroup, no atomic; then the shared counter
		const int pidx = B.order[first + i]; const ProbDesc &d = B.desc[pidx]; const int tid = threadIdx.x;
		SV S(B, d, prm); srba_lm_result *out = B.results + pidx; double *red = nullptr;
		S.phase_spantree(false); __syncthreads();
		// the second copy of the unknowns and of the spanning-tree poses (pairs no trial refreshes stay equal in both for good)
		for (int k = tid; k < d.n_edges * PD; k += SRBA_WG) B.edge1[d.o_edge * PD + k] = B.edge[d.o_edge * PD + k];
		for (int k = tid; k < d.nF * L; k += SRBA_WG) B.ulm1[d.o_ulm * L + k] = B.ulm[d.o_ulm * L + k];
		for (long long k = tid; k < 2LL * d.n_pairs * PD; k += SRBA_WG) B.pose1[d.o_pair * 2 * PD + k] = B.pose[d.o_pair * 2 * PD + k];
		S.phase_jacobians();
		const bool terms = rounds_hess_terms<FAM>(B, d);
		const int ninv = (int)block_sum((double)(terms ? S.phase_hessian_terms(srba_lds) + S.phase_hessian_landmark_blocks() : S.phase_hessian()), red);
		__syncthreads();
		if (tid == 0) {
			out->status = 0; out->num_iters = 0; out->num_trials = 0; out->num_not_pd = 0; out->num_accepted = 0; out->num_relinearized = 0; out->stop_reason = 0;
			out->num_invalid_jacobs = ninv; out->num_observations = d.n_obs; out->num_jacobians = d.n_bp + d.n_bf; out->num_span_tree_numeric_updates = d.n_pairs;
			for (int k = 0; k < SRBA_TRACE_LEN; k++) { out->trace_chi2[k] = NAN; out->trace_lambda[k] = NAN; out->trace_rho[k] = NAN; }
		}
		LmState s; s.lambda = 0; s.nu = 2.0; s.total_err = 0; s.rmse = 0; s.rho_last = 0; s.iter = s.trials = s.n_notpd = s.n_acc = s.n_relin = s.stopmask = s.stop = 0; s.phase = 1; s.cur = s.rcur = s.solved = s.need_lin = s.last_rejected = s.pad = 0;
		if ((long long)O * d.n_obs < (long long)d.n_scal) { if (tid == 0) { out->status = 1; R.st[pidx] = s; } } // S11
		else {
		s.lambda = S.lambda_guess(red);
		s.total_err = S.phase_residuals(B.resid, red); s.rmse = sqrt(s.total_err / d.n_obs);
		if (tid == 0) { out->lambda_init = s.lambda; out->total_sqr_error_init = s.total_err; }
		__syncthreads();
		S.phase_gradient(B.resid); __syncthreads(); S.keep_gradient();
		rounds_enter_iteration(s, prm);
		if (tid == 0) R.st[pidx] = s;
		if (s.phase == 0) rounds_append(R, g, 0, first, pidx);
		}
		__syncthreads(); // the LDS accumulators are reused by the next capsule
	}
}

// ---- one LM trial, first third: solve (H + lambda I) delta = -g for every capsule of the round's list and write the trial unknowns into the non-accepted copy
template <int FAM> __global__ void __launch_bounds__(SRBA_WG) kr_solve(const Batch B, const DevParams prm, const Rounds R, int g, int first, int round) {
	typedef Solver<FAM> SV; constexpr int P = SV::P, L = SV::L, PD = SV::PD; typedef typename SV::W::PO PO; typedef typename SV::W::pose_t pose_t;
	const int rd = round % 3, wr = (round + 1) % 3, zr = (round + 2) % 3; int *cnt = R.count + 3 * g, *ctr = R.ctr + 5 * g;
	const int n = cnt[rd]; const int *list = R.list + (size_t)rd * R.n_prob + first;
	if (blockIdx.x == 0 && threadIdx.x == 0) { ctr[1] = 0; cnt[zr] = 0; if (round < SRBA_ROUNDS_HIST) R.hist[g * SRBA_ROUNDS_HIST + round] = n; } // the counter of this round's kr_eval; the list the NEXT round will fill; how many capsules this round has (sizes the launches of the next run) const int *list = R.list + (size_t)rd * R.n_prob + first;
	for (int i = blockIdx.x; i < n; i = gridDim.x + rounds_pull(ctr + 0)) { // (a workgroup beyond the list leaves without touching the counter: late rounds launch far more workgroups than capsules)
		const int pidx = rounds_uni(list[i]); const ProbDesc &d = B.desc[pidx]; const int tid = threadIdx.x;
		LmState s = rounds_state(R.st + pidx); srba_lm_result *out = B.results + pidx;
		const Batch Ba = rounds_view(B, s.cur); SV S(Ba, d, prm);
		const SparseSys A = S.make_sys(srba_lds);
		const int tr = s.trials++;
		if (tid == 0 && tr < SRBA_TRACE_LEN) out->trace_lambda[tr] = s.lambda;
		const bool solved = S.solve(A, s.lambda);
		s.solved = solved ? 1 : 0; s.need_lin = 0;
		if (!solved) { // optimize_edges.h:476-485
			s.n_notpd++; s.lambda *= s.nu; s.nu *= 2.0; s.stop = (s.lambda > prm.max_lambda) ? 1 : 0;
			if (s.stop) { s.stopmask |= 1 << SRBA_STOP_LAMBDA; s.iter++; s.phase = 1; }
		} else { // K11 (optimize_edges.h:508-539) into the other copy: nothing to back up, nothing to restore
			const Batch Bt = rounds_view(B, s.cur ^ 1); const double *dl = B.delta + d.o_scal;
			for (int e = tid; e < d.nK; e += SRBA_WG) {
				const pose_t cur = PO::ld(Ba.edge + (d.o_edge + e) * PD); double inc[P];
#pragma unroll
				for (int k = 0; k < P; k++) { const int q = e * P + k; inc[k] = A.rhs[3 * A.perm[q / 3] + q % 3]; }
				PO::st(Bt.edge + (d.o_edge + e) * PD, comp(PO::expm(inc), cur));
			}
			for (int k = tid; k < d.nF * L; k += SRBA_WG) Bt.ulm[d.o_ulm * L + k] = Ba.ulm[d.o_ulm * L + k] + dl[d.nK * P + k];
		}
		if (tid == 0) R.st[pidx] = s;
		if (s.phase == 0 && !solved) rounds_append(R, g, wr, first, pidx); // (a solved trial is carried on by kr_eval / kr_lin)
		__syncthreads(); // the LDS image and the symbolic copy are rebuilt by the next capsule
	}
}

// ---- second third: evaluate the trial point, decide (optimize_edges.h:562-604, 658-690)
template <int FAM> __global__ void __launch_bounds__(SRBA_WG) kr_eval(const Batch B, const DevParams prm, const Rounds R, int g, int first, int round) {
	typedef Solver<FAM> SV;
	const int rd = round % 3, wr = (round + 1) % 3; int *cnt = R.count + 3 * g, *ctr = R.ctr + 5 * g;
	if (blockIdx.x == 0 && threadIdx.x == 0) ctr[2] = 0;
	const int n = cnt[rd]; const int *list = R.list + (size_t)rd * R.n_prob + first;
	for (int i = blockIdx.x; i < n; i = gridDim.x + rounds_pull(ctr + 1)) { // (a workgroup beyond the list leaves without touching the counter: late rounds launch far more workgroups than capsules)
		const int pidx = rounds_uni(list[i]); LmState s = rounds_state(R.st + pidx);
		if (s.solved) { // (no `continue` in these loops: one structured body per capsule)
		const ProbDesc &d = B.desc[pidx]; const int tid = threadIdx.x; srba_lm_result *out = B.results + pidx; double *red = nullptr;
		const Batch Bt = rounds_view(B, s.cur ^ 1); SV S(Bt, d, prm);
		S.phase_spantree(true); __syncthreads();
		double *rt = s.rcur ? B.resid : B.resid2; // the residual buffer that is NOT the accepted one
		const double new_err = S.phase_residuals(rt, red), new_rmse = sqrt(new_err / d.n_obs);
		const double err_red = s.total_err > 0 ? (s.total_err - new_err) / s.total_err : 0;
		double den = 0; { const double *dl = B.delta + d.o_scal, *gr = B.grad + d.o_scal; for (int k = tid; k < d.n_scal; k += SRBA_WG) den += dl[k] * (s.lambda * dl[k] + gr[k]); }
		den = block_sum(den, red);
		const double rho = (s.total_err - new_err) / den; const int tr = s.trials - 1;
		if (tid == 0 && tr < SRBA_TRACE_LEN) { out->trace_chi2[tr] = new_err; out->trace_rho[tr] = rho; }
		s.solved = 0;
		if (rho > 0) {
			s.n_acc++; s.need_lin = (err_red < 0 || err_red > prm.min_relin) ? 3 : 1;
			s.rcur ^= 1; s.cur ^= 1; s.total_err = new_err; s.rmse = new_rmse; s.rho_last = rho; s.last_rejected = 0;
		} else {
			s.last_rejected = 1; s.lambda *= s.nu; s.nu *= 2.0; s.stop = (s.lambda > prm.max_lambda) ? 1 : 0;
			if (s.stop) { s.stopmask |= 1 << SRBA_STOP_LAMBDA; s.iter++; s.phase = 1; }
		}
		if (tid == 0) R.st[pidx] = s;
		if (s.phase == 0 && !s.need_lin) rounds_append(R, g, wr, first, pidx);
		}
	}
}

// ---- last third, accepted trials only: relinearise if asked, new minus-gradient, stop tests, lambda / 3 (optimize_edges.h:606-656), head of the next pass
template <int FAM> __global__ void __launch_bounds__(SRBA_WG) kr_lin(const Batch B, const DevParams prm, const Rounds R, int g, int first, int round) {
	typedef Solver<FAM> SV;
	const int rd = round % 3, wr = (round + 1) % 3; int *cnt = R.count + 3 * g, *ctr = R.ctr + 5 * g;
	if (blockIdx.x == 0 && threadIdx.x == 0) ctr[0] = 0; // the counter of the next round's kr_solve
	const int n = cnt[rd]; const int *list = R.list + (size_t)rd * R.n_prob + first;
	for (int i = blockIdx.x; i < n; i = gridDim.x + rounds_pull(ctr + 2)) { // (a workgroup beyond the list leaves without touching the counter: late rounds launch far more workgroups than capsules)
		const int pidx = rounds_uni(list[i]); LmState s = rounds_state(R.st + pidx);
		if (s.need_lin) {
		const ProbDesc &d = B.desc[pidx]; const int tid = threadIdx.x; double *red = nullptr;
		const Batch Ba = rounds_view(B, s.cur); SV S(Ba, d, prm);
		const double *resid = s.rcur ? B.resid2 : B.resid;
		if (s.need_lin & 2) { s.n_relin++; S.phase_jacobians(); if (rounds_hess_terms<FAM>(B, d)) { S.phase_hessian_terms(srba_lds); S.phase_hessian_landmark_blocks(); } else S.phase_hessian(); __syncthreads(); }
		S.phase_gradient(resid); __syncthreads(); S.keep_gradient();
		double ninf = 0; { const double *gr = B.grad + d.o_scal; for (int k = tid; k < d.n_scal; k += SRBA_WG) ninf = fmax(ninf, fabs(gr[k])); }
		ninf = block_max(ninf, red);
		if (ninf <= 1e-15) { s.stop = 1; s.stopmask |= 1 << SRBA_STOP_GRADIENT; }
		if (s.rmse < prm.max_err) { s.stop = 1; s.stopmask |= 1 << SRBA_STOP_RMSE; }
		if (s.rho_last > prm.max_rho) { s.stop = 1; s.stopmask |= 1 << SRBA_STOP_RHO; }
		s.lambda *= 1.0 / 3.0; s.nu = 2.0; s.need_lin = 0;
		s.iter++; rounds_enter_iteration(s, prm); // the inner loop ended with rho > 0: the pass is over
		if (tid == 0) R.st[pidx] = s;
		if (s.phase == 0) rounds_append(R, g, wr, first, pidx);
		}
		__syncthreads(); // the LDS accumulators are reused by the next capsule
	}
}

// ---- hand-over: the capsules still iterating at round `round` finish their loop inside ONE launch of the fused loop (lm_one<FAM, true>), resumed from their LmState.
// Why: a late round holds a few per cent of the capsules and costs three launches per group whatever it holds; the fused loop has no such floor (and with few capsules left,
// residency does not matter any more).
template <int FAM> __global__ void __launch_bounds__(SRBA_WG) kr_tail(const Batch B, const DevParams prm, const Rounds R, int g, int first, int round) {
	const int rd = round % 3; int *cnt = R.count + 3 * g, *ctr = R.ctr + 5 * g;
	const int n = cnt[rd]; const int *list = R.list + (size_t)rd * R.n_prob + first;
	if (blockIdx.x == 0 && threadIdx.x == 0 && round < SRBA_ROUNDS_HIST) R.hist[g * SRBA_ROUNDS_HIST + round] = n;
	for (int i = blockIdx.x; i < n; i = gridDim.x + rounds_pull(ctr + 0)) {
		const int pidx = rounds_uni(list[i]);
		lm_one<FAM, true>(B, prm, pidx, R.st + pidx);
		__syncthreads(); // the LDS image and the symbolic copy are rebuilt by the next capsule
	}
}

// ---- S17 + results; the accepted state goes back to the primary arrays (what srba_hip_download_state reads), with the reference's twin semantics. One workgroup per capsule.
template <int FAM> __global__ void __launch_bounds__(SRBA_WG) kr_finish(const Batch B, const DevParams prm, const Rounds R, int g, int first, int count) {
	typedef Solver<FAM> SV; constexpr int L = SV::L, PD = SV::PD;
	if ((int)blockIdx.x >= count) return;
	const int pidx = B.order[first + blockIdx.x]; const ProbDesc &d = B.desc[pidx]; const int tid = threadIdx.x; srba_lm_result *out = B.results + pidx;
	LmState s = rounds_state(R.st + pidx);
	if (s.phase == 0) { if (tid == 0) atomicAdd(R.unfinished, 1); return; } // still iterating: the host enqueues more rounds (rounds_complete)
	if (s.phase == 2 || out->status != 0) return;                           // results already written by an earlier kr_finish / under-determined problem
	if (!s.stop) s.stopmask |= 1 << SRBA_STOP_MAX_ITERS;
	const Batch Ba = rounds_view(B, s.cur), Bt = rounds_view(B, s.cur ^ 1);
	// a rejected trial refreshed BOTH poses of every pair in use, the reference then restored only the ones Jacobian blocks read (optimize_edges.h:664-670)
	if (s.last_rejected) for (int q = tid; q < 2 * d.n_need; q += SRBA_WG) {
		const long long ps = 2LL * B.need_idx[d.o_pair + (q >> 1)] + (q & 1);
		if (!B.pose_req[d.o_pair * 2 + ps]) { double v[PD]; ldn<PD>(v, Bt.pose + (d.o_pair * 2 + ps) * PD); stn<PD>(Ba.pose + (d.o_pair * 2 + ps) * PD, v); }
	}
	__syncthreads();
	if (s.cur) { // the accepted copy is the second one: bring what can differ (unknowns, refreshed pairs) back to the primary arrays
		for (int k = tid; k < d.nK * PD; k += SRBA_WG) B.edge[d.o_edge * PD + k] = B.edge1[d.o_edge * PD + k];
		for (int k = tid; k < d.nF * L; k += SRBA_WG) B.ulm[d.o_ulm * L + k] = B.ulm1[d.o_ulm * L + k];
		for (int q = tid; q < 2 * d.n_need; q += SRBA_WG) { const long long ps = 2LL * B.need_idx[d.o_pair + (q >> 1)] + (q & 1); double v[PD]; ldn<PD>(v, B.pose1 + (d.o_pair * 2 + ps) * PD); stn<PD>(B.pose + (d.o_pair * 2 + ps) * PD, v); }
	}
	if constexpr (!SV::W::T::REL) { // S17: crpLandmarksApprox
		SV S(Ba, d, prm);
		for (int l = tid; l < d.nF; l += SRBA_WG) {
			const bool ok = prm.cov_recovery == 1 && (S.schur_active() ? (B.hf_ok[d.o_ulm + l] != 0) : true);
			B.ulm_inf_valid[d.o_ulm + l] = ok ? 1 : 0;
			if (ok) for (int k = 0; k < L * L; k++) B.ulm_inf[(d.o_ulm + l) * L * L + k] = B.Hf[(d.o_hf + B.hf_diag[d.o_ulm + l]) * L * L + k];
		}
	}
	if (tid == 0) {
		out->num_iters = s.iter; out->num_trials = s.trials; out->num_not_pd = s.n_notpd; out->num_accepted = s.n_acc; out->num_relinearized = s.n_relin; out->stop_reason = s.stopmask;
		out->total_sqr_error_final = s.total_err; out->obs_rmse = s.rmse; out->lambda_final = s.lambda;
		s.phase = 2; R.st[pidx] = s;
	}
}

} // namespace srbadev
