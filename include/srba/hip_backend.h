/*
 * hip_backend.h -- the default numeric back-end of srba::RbaEngine<>: one optimize_edges() call = one capsule
 * uploaded to the GPU, optimised by srba_hip_lm_run() (whole Levenberg-Marquardt loop device-resident) and read back.
 * Replaces the CPU body of the reference's optimize_edges() S5-S17 (include/srba/impl/optimize_edges.h:256-751).
 * No CPU fallback: if libsrba_hip / a HIP device is unavailable, construction throws.
 */
#pragma once
#include "RbaEngine.h"

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
			m_params = p;
		} else if (std::memcmp(&m_params, &p, sizeof(p)) != 0) {
			check(srba_hip_set_params(m_ctx, &p), "srba_hip_set_params"); m_params = p;
		}
	}
	void run(const srba_hip_params &p, srba_problem_capsule &c, srba_lm_result &r) {
		ensure(p);
		if (m_prof) m_prof->enter("opt.backend.upload");
		check(srba_hip_upload_problems(m_ctx, &c, 1), "srba_hip_upload_problems");
		if (m_prof) { m_prof->leave("opt.backend.upload"); m_prof->enter("opt.backend.lm_run"); }
		check(srba_hip_lm_run(m_ctx, &r), "srba_hip_lm_run");
		if (m_prof) { m_prof->leave("opt.backend.lm_run"); m_prof->registerUserMeasure("opt.backend.lm_run.kernel", 1e-3 * srba_hip_last_kernel_ms(m_ctx)); m_prof->enter("opt.backend.download"); }
		check(srba_hip_download_state(m_ctx, &c, 1), "srba_hip_download_state");
		if (m_prof) m_prof->leave("opt.backend.download");
	}
	void set_profiler(mrpt::utils::CTimeLogger *p) { m_prof = p; }
	double eval_overall(const srba_hip_params &p, const srba_overall_problem &q) {
		ensure(p);
		double v = 0; check(srba_hip_eval_overall_sqr_error(m_ctx, &q, &v), "srba_hip_eval_overall_sqr_error"); return v;
	}
	srba_hip_ctx *context() { return m_ctx; }
private:
	void check(int rc, const char *what) { if (rc != 0) throw std::runtime_error(std::string("srba::hip_backend: ") + what + " failed: " + srba_hip_last_error(m_ctx)); }
	srba_hip_ctx *m_ctx; int m_device; srba_hip_params m_params; mrpt::utils::CTimeLogger *m_prof;
};

inline std::shared_ptr<numeric_backend> make_hip_backend(int device) { return std::shared_ptr<numeric_backend>(new hip_backend(device)); }

} // namespace srba
