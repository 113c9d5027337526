/*
 * engine_capi.h -- C view of the srba::RbaEngine<> front-end for script drivers (ctypes).
 * One handle = one RbaEngine<KF2KF,LM,OBS,OPTS> instance (reference include/srba/RbaEngine.h:66-816).
 */
#ifndef SRBA_ENGINE_CAPI_H
#define SRBA_ENGINE_CAPI_H
#include "../../include/srba_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct srba_engine_config {
	int32_t family, solver, noise, sensor_pose;          /* template arguments of RbaEngine<> / RBA_OPTIONS (RbaEngine.h:39-45) */
	double  std_noise_observations;                      /* parameters.obs_noise (identity policy) */
	double  lambda[36];                                  /* parameters.obs_noise.lambda (constant-matrix policy), O x O row-major */
	double  sensor_pose_xyzypr[6];                       /* parameters.sensor_pose.relative_pose as CPose3D(x,y,z,yaw,pitch,roll) */
	double  cam_left[4], cam_right[4], right_cam_pose[7];/* parameters.sensor.camera_calib */
	uint64_t max_tree_depth, max_optimize_depth;         /* parameters.srba.* (RbaEngine.h:424-460) */
	uint64_t submap_size, min_obs_to_loop_closure;       /* parameters.ecp.* (ecps/local_areas_fixed_size.h:24-33) */
	int32_t optimize_new_edges_alone, use_robust_kernel, use_robust_kernel_stage1, max_iters;
	double  kernel_param, max_error_per_obs_to_stop, max_rho, max_lambda, min_error_reduction_ratio_to_relinearize;
	int32_t cov_recovery;
	int32_t run_local_optimization;                      /* 3rd argument of define_new_keyframe */
	int32_t harvest;                                     /* bit0: keep a copy of every optimize_local_area capsule; bit1: of stage-1 capsules */
	int32_t verbose, enable_profiler, hip_device;
	int32_t refresh_all_read_poses;                      /* extensions (default 0 = reference behaviour), see RbaEngine.h TSRBAParameters: bit 0 refresh_all_read_poses,
		bit 1 restore_spanning_tree_twins, bit 2 schur_keeps_gradient, bit 3 consistent_loop_closure_init */
	int32_t ecp;                                         /* RBA_OPTIONS::edge_creation_policy_t: 0 ecps::local_areas_fixed_size (default), 1 ecps::classic_linear_rba */
} srba_engine_config;

typedef struct srba_kf_info {                            /* TNewKeyFrameInfo (RbaEngine.h:180-195) */
	uint64_t kf_id;
	int32_t  n_new_edges, reserved;
	uint64_t edge_id[4], lc_observer[4], lc_base[4];
	int32_t  edge_has_init[4];
	uint64_t num_observations, num_jacobians, num_k2k, num_k2f; /* TOptimizeExtraOutputInfo of the local-area optimisation */
	double   chi2_init, chi2_final, obs_rmse;
	srba_lm_result lm, lm_stage1;
} srba_kf_info;

typedef int (*srba_backend_fn)(const srba_hip_params *, srba_problem_capsule *, srba_lm_result *);

void  srba_engine_config_default(srba_engine_config *c, int family);
void *srba_engine_create(const srba_engine_config *c);
void  srba_engine_destroy(void *h);
const char *srba_engine_last_error(void *h);
int   srba_engine_set_backend_fn(void *h, srba_backend_fn fn, const char *name);
/* optional second entry point of a plugged back-end: whole-map squared error (include/srba_hip.h: srba_overall_problem) */
typedef int (*srba_overall_fn)(const srba_hip_params *, const srba_overall_problem *, double *);
int   srba_engine_set_overall_fn(void *h, srba_overall_fn fn);
/* RbaEngine<>::eval_overall_squared_error() (impl/eval_overall_error.h:15-137) */
int   srba_engine_eval_overall_sqr_error(void *h, double *out);
/* flags[i]: bit0 is_fixed, bit1 is_unknown_with_init_val (srba_types.h:473-495); z: n_obs x O; relpos: n_obs x L or NULL */
int   srba_engine_add_keyframe(void *h, int n_obs, const uint64_t *feat_id, const double *z, const uint8_t *flags, const double *relpos, srba_kf_info *out);
int   srba_engine_optimize_local_area(void *h, uint64_t root, unsigned win, srba_kf_info *out);
int64_t srba_engine_num_edges(void *h);
int   srba_engine_get_edge(void *h, int64_t i, uint64_t *from, uint64_t *to, double *pose /*PD*/);
int64_t srba_engine_num_unknown_lms(void *h);
int   srba_engine_get_unknown_lms(void *h, uint64_t *ids, uint64_t *base, double *pos);
int64_t srba_engine_st_dump(void *h, int what, int64_t *out, int64_t cap);
int   srba_engine_get_rel_pose(void *h, uint64_t query, uint64_t reference, double *pose);
double srba_engine_profiler_mean(void *h, const char *name);
/* Low-level graph construction (RbaEngine.h:334-347), used by the spanning-tree property tests */
uint64_t srba_engine_alloc_keyframe(void *h);
int64_t srba_engine_create_edge(void *h, uint64_t new_kf, uint64_t from, uint64_t to, const double *pose /*PD or NULL*/);
int64_t srba_engine_harvest_count(void *h);
srba_problem_capsule *srba_engine_harvest_capsules(void *h);
uint64_t srba_engine_harvest_kf(void *h, int64_t i);
void  srba_engine_harvest_clear(void *h);
int   srba_engine_get_hip_params(void *h, srba_hip_params *out);
int   srba_engine_harvest_save(void *h, const char *path, int64_t first, int64_t count);
void *srba_capsule_file_load(const char *path);
int64_t srba_capsule_file_count(void *h);
srba_problem_capsule *srba_capsule_file_capsules(void *h);
int   srba_capsule_file_params(void *h, srba_hip_params *out);
void  srba_capsule_file_free(void *h);
/* Map sweeps (RbaEngine<>::plan_local_area_sweep / optimize_local_areas_batch; no counterpart in the reference: SURVEY 8e "new mode"). plan: round_of[n] (-1: nothing to optimise at that root);
 * touch_off[n + 1] / touch[cap]: per root the kf2kf edges its window touches, id | 0x80000000 when it writes them; returns the number of rounds, < 0 on error (touch too small: -2 - needed). */
int64_t srba_engine_plan_sweep(void *h, const uint64_t *roots, int64_t n, unsigned win, int32_t *round_of, int64_t *touch_off, uint32_t *touch, int64_t cap);
/* optimize_local_area() of n mutually independent roots as one batch of the numeric back-end; out: n records (may be NULL) */
int   srba_engine_optimize_batch(void *h, const uint64_t *roots, int64_t n, unsigned win, srba_kf_info *out);
/* the landmarks the windows of the LAST srba_engine_plan_sweep touch: off[n + 1], touch[cap] = landmark id | 0x80000000 when the window optimises it; returns the entry count (-2 - needed: touch too small) */
int64_t srba_engine_plan_sweep_lms(void *h, int64_t *off, uint32_t *touch, int64_t cap);
int   srba_engine_get_lm_positions(void *h, const uint64_t *ids, int64_t n, double *out /* n x L */);
int   srba_engine_set_lm_positions(void *h, const uint64_t *ids, int64_t n, const double *in);
int   srba_engine_get_edge_poses(void *h, const uint64_t *ids, int64_t n, double *out /* n x PD */);
int   srba_engine_set_edge_poses(void *h, const uint64_t *ids, int64_t n, const double *in);
/* RbaEngine<>::get_global_graphslam_problem(): global node poses (complete breadth-first spanning tree from `root`) + one constraint per kf2kf edge as (to, from, inv_pose). Returns the node count. */
int64_t srba_engine_export_graphslam(void *h, uint64_t root, uint64_t *node_id, double *node_pose, int64_t node_cap, uint64_t *edge_from_to, double *edge_pose, int64_t edge_cap);
void *srba_capsule_clone(const srba_problem_capsule *caps, int64_t n, int family);
/* one capsule from per-observation Jacobian columns (identity poses; Hessian / Schur plan by CapsuleData::build_plan) -- replays tests/schur_unittest.cpp */
void *srba_capsule_from_blocks(int family, int nK, int nF, int n_obs, const int32_t *row_bp_col, const int32_t *row_bf_col, int with_schur);

#ifdef __cplusplus
}
#endif
#endif
