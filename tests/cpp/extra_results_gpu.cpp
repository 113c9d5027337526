// GPU: what the reference returns next to the optimised unknowns -- TOptimizeExtraOutputInfo::extra_results.hessian (impl/lev-marq_solvers.h:586-590), HAp_condition_number
// (impl/optimize_edges.h:753-766) and, with SRBA_DETAILED_TIME_PROFILING, the per-stage "opt.*" sections of the time profiler (impl/optimize_edges.h:16-27) -- through the
// header-only front-end with the HIP back-end. The problem is the literal two-key-frame range-bearing dataset of tutorial-srba-range-bearing-se2.cpp:50-78 plus a third key-frame.
// Prints "name value" lines that tests/test_frontend_exports.py checks.
#define SRBA_DETAILED_TIME_PROFILING 1
#include <srba.h>
#include <cstdio>
using namespace srba;

struct OPTS : public RBA_OPTIONS_DEFAULT { typedef options::sensor_pose_on_robot_none sensor_pose_on_robot_t; typedef options::solver_LM_schur_dense_cholesky solver_t; };
typedef RbaEngine<kf2kf_poses::SE2, landmarks::Euclidean2D, observations::RangeBearing_2D, OPTS> rba_t;
struct Obs { int lm; double range, yaw; };
static const Obs kf0[] = {{0, 1.75, 0.96}, {1, 2.21, -0.70}, {2, 3.05, 0.32}, {3, 2.56, -0.21}, {4, 1.62, 0.11}, {5, 2.84, 0.71}};
static const Obs kf1[] = {{0, 1.43, 1.48}, {1, 1.71, -1.10}, {2, 2.46, 0.55}, {3, 1.95, -0.38}, {4, 0.94, 0.38}, {5, 2.43, 1.03}, {6, 2.10, -0.05}};
static const Obs kf2[] = {{2, 1.93, 0.91}, {3, 1.41, -0.66}, {5, 2.12, 1.47}, {6, 1.52, -0.10}, {7, 1.80, 0.40}};

template <size_t N> static void add_kf(rba_t &rba, const Obs (&o)[N], rba_t::TNewKeyFrameInfo &info) {
	rba_t::new_kf_observations_t obs;
	for (size_t i = 0; i < N; i++) { rba_t::new_kf_observation_t k; k.is_fixed = false; k.is_unknown_with_init_val = false; k.obs.feat_id = o[i].lm; k.obs.obs_data.range = o[i].range;
		k.obs.obs_data.yaw = o[i].yaw; obs.push_back(k); }
	rba.define_new_keyframe(obs, info, true);
}
int main() {
	rba_t rba; rba.setVerbosityLevel(0);
	rba.parameters.srba.max_tree_depth = 3; rba.parameters.srba.max_optimize_depth = 3; rba.parameters.obs_noise.std_noise_observations = 0.03;
	rba.parameters.srba.compute_condition_number = true; rba.parameters.srba.return_hessian = true;
	rba_t::TNewKeyFrameInfo info; add_kf(rba, kf0, info); add_kf(rba, kf1, info); add_kf(rba, kf2, info);
	const rba_t::TOptimizeExtraOutputInfo &r = info.optimize_results;
	const size_t n = 3 * r.num_kf2kf_edges_optimized; const std::vector<double> &H = r.extra_results.hessian;
	std::printf("edges %zu\nhessian_valid %d\nhessian_size %zu\n", r.num_kf2kf_edges_optimized, (int)r.extra_results.hessian_valid, H.size());
	double asym = 0, mind = 1e300; bool pd = H.size() == n * n && n > 0;
	if (pd) { std::vector<double> Lc(H); // symmetric and positive definite (it is the matrix the last LM trial factored)
		for (size_t i = 0; i < n; i++) { mind = std::min(mind, H[i * n + i]); for (size_t j = 0; j < n; j++) asym = std::max(asym, std::fabs(H[i * n + j] - H[j * n + i])); }
		for (size_t k = 0; k < n && pd; k++) { if (!(Lc[k * n + k] > 0)) { pd = false; break; } const double d = std::sqrt(Lc[k * n + k]); for (size_t i = k; i < n; i++) Lc[i * n + k] /= d;
			for (size_t j = k + 1; j < n; j++) for (size_t i = j; i < n; i++) Lc[i * n + j] -= Lc[i * n + k] * Lc[j * n + k]; } }
	std::printf("hessian_asym %.3e\nhessian_pd %d\nhessian_min_diag %.6g\ncondition_number %.6g\nrmse %.6g\n", asym, (int)pd, mind, r.HAp_condition_number, r.obs_rmse);
	const char *secs[] = {"opt", "opt.update_spanning_tree_num", "opt.recompute_all_Jacobians", "opt.sparse_hessian_update_numeric", "opt.reprojection_residuals", "opt.compute_minus_gradient",
		"opt.schur_build_reduced", "opt.DenseFill", "opt.DenseChol", "opt.backsub", "opt.schur_features", "opt.add_se3_deltas_to_frames"};
	for (const char *s : secs) std::printf("section %s %.3e\n", s, rba.get_time_profiler().getMeanTime(s));
	return 0;
}
