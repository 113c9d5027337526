/*
 * srba_options.h -- compile-time policy structs selectable through RBA_OPTIONS, with the reference's names:
 *   srba::options::observation_noise_identity / observation_noise_constant_matrix<OBS>   (srba_options_noise.h:25-133)
 *   srba::options::sensor_pose_on_robot_none / sensor_pose_on_robot_se3                  (srba_options_sensor_pose.h:32-135)
 *   srba::options::solver_LM_{schur_dense,schur_sparse,no_schur_sparse}_cholesky         (srba_options_solver.h:24-76)
 * In the reference these policies are evaluated inside the CPU hot loops; here each one only marshals itself into
 * srba_hip_params and the arithmetic runs in the HIP kernels.
 */
#pragma once
#include "models.h"

namespace srba {
namespace options {

struct observation_noise_identity {
	struct parameters_t { double std_noise_observations; parameters_t() : std_noise_observations(1.) {} };
	static void fill_params(srba_hip_params &hp, const parameters_t &p) { hp.noise = SRBA_NOISE_IDENTITY; hp.std_noise_observations = p.std_noise_observations; }
};
template <class obs_t> struct observation_noise_constant_matrix {
	static const size_t OBS_DIMS = obs_t::OBS_DIMS;
	typedef mrpt::math::CMatrixFixed<OBS_DIMS, OBS_DIMS> obs_noise_matrix_t;
	struct parameters_t { obs_noise_matrix_t lambda; parameters_t() : lambda(obs_noise_matrix_t::Identity()) {} };
	static void fill_params(srba_hip_params &hp, const parameters_t &p) { hp.noise = SRBA_NOISE_CONSTANT_MATRIX; for (size_t i = 0; i < OBS_DIMS * OBS_DIMS; i++) hp.lambda[i] = p.lambda.m[i]; }
};

struct sensor_pose_on_robot_none {
	struct parameters_t {};
	static void fill_params(srba_hip_params &hp, const parameters_t &) { hp.sensor_pose = SRBA_SENSOR_POSE_NONE; }
	template <class LANDMARK_T, class ARR> static void sensor2robot_point(ARR &, const parameters_t &) {}
	static mrpt::poses::CPose3D sensor_pose_as_3d(const parameters_t &) { return mrpt::poses::CPose3D(); }
};
struct sensor_pose_on_robot_se3 {
	struct parameters_t { mrpt::poses::CPose3D relative_pose; };
	static void fill_params(srba_hip_params &hp, const parameters_t &p) { hp.sensor_pose = SRBA_SENSOR_POSE_SE3; p.relative_pose.storeTo(hp.sensor_pose_se3); }
	template <class LANDMARK_T, class ARR> static void sensor2robot_point(ARR &pt, const parameters_t &p) { LANDMARK_T::composePosePoint(pt, p.relative_pose); } // srba_options_sensor_pose.h:131-134
	static mrpt::poses::CPose3D sensor_pose_as_3d(const parameters_t &p) { return p.relative_pose; }
};

/** Dense copy of the last (reduced) Hessian, as returned by the reference solvers' get_extra_results (lev-marq_solvers.h:204-208,586-590).
 * The device keeps it in HBM; it is not downloaded unless asked for, so hessian_valid stays false by default. */
struct hessian_result_t { bool hessian_valid; std::vector<double> hessian; hessian_result_t() { clear(); } void clear() { hessian_valid = false; } };

struct solver_LM_schur_dense_cholesky { static const bool USE_SCHUR = true; static const bool DENSE_CHOLESKY = true; static const int solver_id = SRBA_SOLVER_SCHUR_DENSE_CHOL;
	typedef hessian_result_t extra_results_t; };
struct solver_LM_schur_sparse_cholesky { static const bool USE_SCHUR = true; static const bool DENSE_CHOLESKY = false; static const int solver_id = SRBA_SOLVER_SCHUR_SPARSE_CHOL;
	typedef hessian_result_t extra_results_t; };
struct solver_LM_no_schur_sparse_cholesky { static const bool USE_SCHUR = false; static const bool DENSE_CHOLESKY = false; static const int solver_id = SRBA_SOLVER_NO_SCHUR_SPARSE_CHOL;
	typedef hessian_result_t extra_results_t; };

} // namespace options
} // namespace srba
