/*
 * hip_backend.h -- the default numeric back-end of srba::RbaEngine<>: one optimize_edges() call = one capsule
 * uploaded to the GPU, optimised by srba_hip_lm_run() (whole Levenberg-Marquardt loop device-resident) and read back.
 * Replaces the CPU body of the reference's optimize_edges() S5-S17 (include/srba/impl/optimize_edges.h:256-751).
 * No CPU fallback: if libsrba_hip / a HIP device is unavailable, construction throws.
 */
#pragma once
#include "RbaEngine.h"
#include <cstdlib>

#ifndef SRBA_DETAILED_TIME_PROFILING
#	define SRBA_DETAILED_TIME_PROFILING 0   // as in the reference (impl/optimize_edges.h:16-27): define it to 1 before including <srba.h> to get the per-stage "opt.*" sections
#endif

namespace srba {

class hip_backend : public numeric_backend {
public:
	explicit hip_backend(int device) : m_ctx(NULL), m_device(device), m_prof(NULL) { std::memset(&m_params, 0, sizeof(m_params)); }
	~hip_backend() { if (m_ctx) srba_hip_destroy(m_ctx); }
	const char *name() const { return "hip-gfx950"; }
	void ensure(const srba_hip_params &p) {
		if (!m_ctx) {
			m_ctx = srba_hip_create(m_device, &p);
			if (!m_ctx) throw std::runtime_error(std::string("srba::hip_backend: cannot create the HIP context: ") + srba_hip_last_error(NULL));
#if SRBA_DETAILED_TIME_PROFILING
			srba_hip_set_phase_timing(m_ctx, 1); // the fused kernel keeps one cycle counter per stage and capsule of THIS context (read below); no process-wide setting
#endif
			m_params = p;
		} else if (std::memcmp(&m_params, &p, sizeof(p)) != 0) {
			check(srba_hip_set_params(m_ctx, &p), "srba_hip_set_params"); m_params = p;
		}
	}
	void run(const srba_hip_params &p, srba_problem_capsule &c, srba_lm_result &r) {
		ensure(p);
		if (m_prof) m_prof->enter("opt.backend.optimize_capsule"); // upload + LM loop + write-back, one wait for the device (srba_hip_optimize_capsule)
		check(srba_hip_optimize_capsule(m_ctx, &c, &r), "srba_hip_optimize_capsule");
		if (m_prof) { m_prof->leave("opt.backend.optimize_capsule"); m_prof->registerUserMeasure("opt.backend.lm_run.kernel", 1e-3 * srba_hip_last_kernel_ms(m_ctx)); }
#if SRBA_DETAILED_TIME_PROFILING
		if (m_prof) report_stages(p);
#endif
	}
	/** n independent capsules as ONE batch: one upload, one fused launch per size class, one read-back (RbaEngine<>::optimize_local_areas_batch) */
	void run_batch(const srba_hip_params &p, srba_problem_capsule *caps, int n, srba_lm_result *res) {
		if (n <= 0) return; ensure(p);
		if (m_prof) m_prof->enter("opt.backend.optimize_batch");
		check(srba_hip_upload_problems(m_ctx, caps, n), "srba_hip_upload_problems"); check(srba_hip_lm_run(m_ctx, res), "srba_hip_lm_run"); check(srba_hip_download_state(m_ctx, caps, n),
			"srba_hip_download_state");
		if (m_prof) { m_prof->leave("opt.backend.optimize_batch"); m_prof->registerUserMeasure("opt.backend.lm_run.kernel", 1e-3 * srba_hip_last_kernel_ms(m_ctx)); }
	}
	void set_profiler(mrpt::utils::CTimeLogger *p) { m_prof = p; }
	double eval_overall(const srba_hip_params &p, const srba_overall_problem &q) {
		ensure(p);
		double v = 0; check(srba_hip_eval_overall_sqr_error(m_ctx, &q, &v), "srba_hip_eval_overall_sqr_error"); return v;
	}
	bool read_blocks(int what, std::vector<double> &out) { // the arrays of the capsule of the last run(), still on the device
		if (!m_ctx || !(what == 3 || what == 4 || what == 5)) return false;
		const int64_t n = srba_hip_debug_size(m_ctx, what); if (n < 0) return false;
		out.assign((size_t)std::max<int64_t>(n, 1), 0.0); if (n > 0) check(srba_hip_debug_read(m_ctx, what, out.data(), n), "srba_hip_debug_read"); out.resize((size_t)n); return true;
	}
	srba_hip_ctx *context() { return m_ctx; }
private:
	/** The reference's detailed sections (impl/optimize_edges.h, impl/lev-marq_solvers.h: DETAILED_PROFILING_ENTER) fed from the stage counters of the fused kernel
	 *  (100 MHz ticks per stage of the capsule just run). Stages the device fuses are reported under the name of the first one: the backups of optimize_edges.h:486-557 are
	 *  part of "opt.add_se3_deltas_to_frames", the triplet compression of the sparse solvers does not exist (the symbolic factorisation is done at upload). INTEGRATION.md
	 *  has the section -> kernel / device function table. */
	void report_stages(const srba_hip_params &p) {
		double t[16]; if (srba_hip_debug_size(m_ctx, 10) < 16 || srba_hip_debug_read(m_ctx, 10, t, 16) != 0) return;
		const bool dense = p.solver == SRBA_SOLVER_SCHUR_DENSE_CHOL;
		const struct { int slot; const char *name; } map[] = {{0, "opt.update_spanning_tree_num"}, {7, "opt.update_spanning_tree_num"}, {1, "opt.recompute_all_Jacobians"}, {2,
			"opt.sparse_hessian_update_numeric"},
			{3, "opt.reprojection_residuals"}, {4, "opt.compute_minus_gradient"}, {6, "opt.add_se3_deltas_to_frames"}, {8, "opt.failedstep_restore_backup"}, {9, "opt.schur_build_reduced"},
			{10, dense ? "opt.DenseFill" : "opt.SparseTripletFill"}, {11, dense ? "opt.DenseChol" : "opt.SparseChol"}, {12, "opt.backsub"}, {13, "opt.schur_features"}};
		for (const auto &m : map) if (t[m.slot] > 0) m_prof->registerUserMeasure(m.name, 1e-8 * t[m.slot]);
	}
	void check(int rc, const char *what) { if (rc != 0) throw std::runtime_error(std::string("srba::hip_backend: ") + what + " failed: " + srba_hip_last_error(m_ctx)); }
	srba_hip_ctx *m_ctx; int m_device; srba_hip_params m_params; mrpt::utils::CTimeLogger *m_prof;
};

inline std::shared_ptr<numeric_backend> make_hip_backend(int device) { return std::shared_ptr<numeric_backend>(new hip_backend(device)); }

} // namespace srba
