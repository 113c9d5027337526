/*
 * srba_hip.h -- C ABI of the MI355X (gfx950) back-end for SRBA's local-optimisation hot path.
 *
 * The reference (MRPT/srba) has no FFI: its optimiser is the protected template method
 *   RbaEngine<>::optimize_edges()            (include/srba/impl/optimize_edges.h:44-793)
 * reached from
 *   RbaEngine<>::optimize_local_area()       (include/srba/impl/optimize_local_area.h:15-58)
 *   RbaEngine<>::define_new_keyframe()       (include/srba/impl/define_new_keyframe.h:16-116)
 * and its only extension seam is the compile-time solver_engine<SCHUR,DENSE,ENGINE>
 * (include/srba/impl/optimize_edges.h:32-36, include/srba/srba_options_solver.h:24-76).
 *
 * This header is what a binding of that path talks to: one optimize_edges() call is flattened by the
 * host front-end (the include/srba/ headers of THIS repo) into a "problem capsule" -- index-linked flat arrays,
 * no pointers between items -- and a batch of capsules is solved on the GPU.  Every entry point cites
 * the reference code it replaces.  Plain C: pointers, sizes, int status codes; no exceptions cross it.
 *
 * Threading: a context is not thread-safe (the reference engine is single-threaded).
 * Ownership: caller owns every host array in a capsule; the context owns all device memory.
 */
#ifndef SRBA_HIP_H
#define SRBA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ----------------------------------------------------------------------------------------------
 * Model families (= the reference's <KF2KF_POSE, LANDMARK, OBSERVATION> template triples).
 * P = REL_POSE_DIMS, L = LM_DIMS, O = OBS_DIMS (include/srba/srba_types.h:393-395).
 * -------------------------------------------------------------------------------------------- */
enum srba_family {
	SRBA_SE2_RELPOSE2D = 0, /* <SE2,RelativePoses2D,RelativePoses_2D> P3 L3 O3  models/sensors.h:744 ; jacobians.h:645-744 */
	SRBA_SE2_RB2D      = 1, /* <SE2,Euclidean2D,RangeBearing_2D>      P3 L2 O2  models/sensors.h:635 ; jacobians.h:501-634 */
	SRBA_SE2_CART2D    = 2, /* <SE2,Euclidean2D,Cartesian_2D>         P3 L2 O2  models/sensors.h:421 */
	SRBA_SE3_STEREO    = 3, /* <SE3,Euclidean3D,StereoCamera>         P6 L3 O4  models/sensors.h:149 ; jacobians.h:364-494 */
	SRBA_SE3_MONO      = 4, /* <SE3,Euclidean3D,MonocularCamera>      P6 L3 O2  models/sensors.h:24 */
	SRBA_SE3_CART3D    = 5, /* <SE3,Euclidean3D,Cartesian_3D>         P6 L3 O3  models/sensors.h:323 */
	SRBA_SE3_RB3D      = 6, /* <SE3,Euclidean3D,RangeBearing_3D>      P6 L3 O3  models/sensors.h:517 (range, yaw, pitch) */
	SRBA_SE3_RELPOSE3D = 7, /* <SE3,RelativePoses3D,RelativePoses_3D> P6 L6 O6  models/sensors.h:842 ; jacobians.h:748-873 (x y z yaw pitch roll) */
	SRBA_SE2_STEREO    = 8, /* <SE2,Euclidean3D,StereoCamera>         P3 L3 O4  SE(2) key-frames with 3D points: jacobians.h:501-641 (POINT_DIMS = 3) */
	SRBA_NUM_FAMILIES  = 9
};

/* Pose storage at the boundary ("PD" doubles per pose):
 *   SE2: PD=3  [x, y, phi]                                   (mrpt CPose2D)
 *   SE3: PD=12 [x, y, z, r00 r01 r02 r10 r11 r12 r20 r21 r22] (mrpt CPose3D keeps t + 3x3 R; the LM update
 *        R <- exp(w)*R is never re-orthonormalised in the reference, optimize_edges.h:515-521, so R itself is state) */
int srba_family_dims(int family, int *P, int *L, int *O, int *PD);

/* Solver selection = RBA_OPTIONS::solver_t (srba_options_solver.h:24-76) */
enum srba_solver {
	SRBA_SOLVER_SCHUR_DENSE_CHOL     = 0, /* solver_LM_schur_dense_cholesky   lev-marq_solvers.h:410-591 (default, RbaEngine.h:44) */
	SRBA_SOLVER_SCHUR_SPARSE_CHOL    = 1, /* solver_LM_schur_sparse_cholesky  lev-marq_solvers.h:214-405 */
	SRBA_SOLVER_NO_SCHUR_SPARSE_CHOL = 2  /* solver_LM_no_schur_sparse_cholesky lev-marq_solvers.h:28-209 (srba-slam --graph-slam) */
};

/* Noise policy = RBA_OPTIONS::obs_noise_matrix_t (srba_options_noise.h) */
enum srba_noise {
	SRBA_NOISE_IDENTITY        = 0, /* observation_noise_identity: H*=1/sigma, g*=1/sigma (sic)  :25-73 */
	SRBA_NOISE_CONSTANT_MATRIX = 1  /* observation_noise_constant_matrix: J^t Lambda J           :78-133 */
};

/* Sensor pose policy = RBA_OPTIONS::sensor_pose_on_robot_t (srba_options_sensor_pose.h) */
enum srba_sensor_pose {
	SRBA_SENSOR_POSE_NONE = 0, /* :32-78  */
	SRBA_SENSOR_POSE_SE3  = 1  /* :92-135 */
};

/* Everything the reference keeps in RbaEngine<>::parameters that the hot path reads
 * (RbaEngine.h:424-473; defaults impl/rba_problem_common.h:35-56). */
typedef struct srba_hip_params {
	int32_t family;              /* enum srba_family */
	int32_t solver;              /* enum srba_solver */
	int32_t noise;               /* enum srba_noise */
	int32_t sensor_pose;         /* enum srba_sensor_pose */
	double  std_noise_observations; /* identity policy sigma (srba_options_noise.h:29-33) */
	double  lambda[36];          /* constant-matrix policy: O x O row-major information matrix (:91-97) */
	double  sensor_pose_se3[12]; /* robot->sensor pose, PD=12 layout (srba_options_sensor_pose.h:94-97) */
	/* camera calibration (mrpt TCamera fx fy cx cy; TStereoCamera left/right + rightCameraPose as CPose3DQuat
	 * [x y z qr qx qy qz], models/sensors.h:193-203) */
	double  cam_left[4];         /* fx fy cx cy (monocular camera uses this one) */
	double  cam_right[4];
	double  right_cam_pose[7];
	/* TSRBAParameters fields read by optimize_edges (optimize_edges.h:440,454,465,592,642,647) */
	int32_t max_iters;                  /* default 20 */
	int32_t use_robust_kernel;          /* pseudo-Huber (reprojection_residuals.h:67-72, RbaEngine.h:810-813) */
	double  kernel_param;               /* default 3 */
	double  max_error_per_obs_to_stop;  /* default 1e-6 */
	double  max_rho;                    /* default 10 */
	double  max_lambda;                 /* default 1e20 */
	double  min_error_reduction_ratio_to_relinearize; /* default 0.01 */
	int32_t cov_recovery;               /* 0 crpNone, 1 crpLandmarksApprox (optimize_edges.h:728-750) */
	int32_t extensions;                 /* bit mask of SRBA_EXT_*; 0 = the reference's behaviour to the letter (default) */
} srba_hip_params;

/* Opt-in deviations from the reference (all default off). Each one repairs a behaviour of the reference that loses maps on landmark problems:
 * SRBA_EXT_SCHUR_KEEPS_GRADIENT: the Schur solvers reduce minus_grad IN PLACE (impl/schur.h:248-265, :294) and the LM loop does not recompute it after a rejected
 *   step (impl/optimize_edges.h:658-690), so every retry of a rejected trial starts from a gradient that has already been reduced once per earlier retry. With this
 *   bit every solve starts from the gradient that compute_minus_gradient produced (the reduction works on it afresh). */
enum srba_extensions { SRBA_EXT_SCHUR_KEEPS_GRADIENT = 1 };

/* Fill with the reference's defaults for a family (rba_problem_common.h:35-56). */
void srba_hip_params_default(srba_hip_params *p, int family);

/* ----------------------------------------------------------------------------------------------
 * Problem capsule: one optimize_edges() call, flattened.  All indices are LOCAL to the capsule and
 * int32.  "lm ref" encoding: v>=0 -> unknown landmark slot v ; v<0 -> known landmark (-1-v).
 * "pose idx": index into the capsule's ST pose table (2 poses per ST pair), -1 = identity (NULL in the
 * reference: TJacobianSymbolicInfo_dh_dAp::rel_pose_d1_from_obs, srba_types.h:212).
 * -------------------------------------------------------------------------------------------- */
typedef struct srba_problem_capsule {
	/* sizes */
	int32_t n_edges;      /* local k2k-edge table; [0,n_unk_edges) = unknowns in optimisation order, rest = fixed edges met on ST paths */
	int32_t n_unk_edges;  /* nUnknowns_k2k (optimize_edges.h:125) */
	int32_t n_unk_lms;    /* nUnknowns_k2f (optimize_edges.h:126) */
	int32_t n_known_lms;
	int32_t n_pairs;      /* ST pairs: every all_edges[r][t] row of every root r in kfs_num_spantrees_to_update (optimize_edges.h:245-256) */
	int32_t n_path;       /* total path entries */
	int32_t n_obs;        /* involved_obs, WITH the reference's duplicates (optimize_edges.h:177-193) */
	int32_t n_valid;      /* distinct observations = validity flags (srba_types.h:722) */
	int32_t n_bp;         /* dh_dAp blocks of the selected columns */
	int32_t n_bf;         /* dh_df blocks of the selected columns */
	int32_t n_hap, n_hap_terms;   /* HAp upper blocks (incl. Schur fill-in blocks, which carry no J terms) */
	int32_t n_hf, n_hf_terms;
	int32_t n_hapf, n_hapf_terms;
	int32_t n_sch_terms;          /* Schur reduction terms (schur.h:56-157) ; 0 when solver has no Schur or n_unk_lms==0 */
	int32_t reserved0;

	/* unknowns and constants */
	double  *edge_pose;   /* [n_edges*PD]   k2k_edge_t::inv_pose (srba_types.h:82-91); first n_unk_edges are read AND written back */
	double  *ulm_pos;     /* [n_unk_lms*L]  TRelativeLandmarkPos::pos of unknown landmarks, read and written back */
	double  *klm_pos;     /* [n_known_lms*L] */

	/* numeric spanning tree (TSpanningTree::num, srba_types.h:611-620; spantree_update_numeric.h:19-81) */
	int32_t *pair_path_off;  /* [n_pairs+1] CSR into path_edge */
	int32_t *path_edge;      /* [n_path] (local edge index << 1) | inv ; inv=1: compose with the INVERSE of inv_pose (:57-65) */
	uint8_t *pair_needed;    /* [n_pairs] 1: one of its two poses is in list_of_required_num_poses -> refreshed in every LM trial (App. B-12) */
	uint8_t *pose_required;  /* [2*n_pairs] pose is in list_of_required_num_poses (jacobians.h:225-230,900-901): backed up / restored (optimize_edges.h:550-557,664-670) */
	double  *pose;           /* [2*n_pairs*PD] OUT: final num[root][target] (2p) and num[target][root] (2p+1) */

	/* observations = rows of the residual vector (reprojection_residuals.h:26-78) */
	int32_t *obs_pose;    /* [n_obs] pose idx of num[obs_kf][base_kf] or -1 when obs_kf==base_kf */
	int32_t *obs_lm;      /* [n_obs] lm ref */
	int32_t *obs_valid;   /* [n_obs] validity slot in [0,n_valid) */
	double  *obs_z;       /* [n_obs*O] kf_observation_t::obs_arr */

	/* dh_dAp blocks, ordered by (unknown column, ascending global observation index) = the reference sweep order (jacobians.h:1094-1110) */
	int32_t *bp_col;      /* [n_bp] unknown edge slot */
	int32_t *bp_res;      /* [n_bp] residual row (obs_global_idx2residual_idx: last duplicate wins, optimize_edges.h:187) */
	int32_t *bp_A;        /* [n_bp] pose idx of rel_pose_d1_from_obs or -1 */
	int32_t *bp_D;        /* [n_bp] pose idx of rel_pose_base_from_d1 */
	int32_t *bp_lm;       /* [n_bp] lm ref of feat_rel_pos */
	uint8_t *bp_normal;   /* [n_bp] edge_normal_dir (add-observations.h:172) */
	int32_t *colp_off;    /* [n_unk_edges+1] CSR: blocks of each unknown edge column */

	/* dh_df blocks ordered by (unknown landmark column, ascending observation index) (jacobians.h:50-65) */
	int32_t *bf_col;      /* [n_bf] unknown lm slot */
	int32_t *bf_res;      /* [n_bf] residual row */
	int32_t *bf_pose;     /* [n_bf] pose idx of rel_pose_base_from_obs or -1 */
	int32_t *colf_off;    /* [n_unk_lms+1] */

	/* Hessian plan = output of sparse_hessian_build_symbolic (sparse_hessian_build_symbolic.h:22-237) */
	int32_t *hap_i, *hap_j;         /* [n_hap] block (row i <= col j), ordered by (col j, row i) like getCol(j)[i] */
	int32_t *hap_term_off;          /* [n_hap+1] */
	int32_t *hap_t1, *hap_t2;       /* [n_hap_terms] dh_dAp block indices: H_ij += J_t1^t Lambda J_t2 */
	int32_t *hf_i, *hf_j;           /* [n_hf] */
	int32_t *hf_term_off;
	int32_t *hf_t1, *hf_t2;         /* dh_df block indices */
	int32_t *hapf_i, *hapf_j;       /* [n_hapf] (edge slot i, lm slot j), ordered by (i, j): stored by rows (:189-235) */
	int32_t *hapf_term_off;
	int32_t *hapf_t1, *hapf_t2;     /* t1: dh_dAp block, t2: dh_df block */
	int32_t *hap_diag;              /* [n_unk_edges] index of HAp block (i,i) */
	int32_t *hf_diag;               /* [n_unk_lms]   index of Hf block (i,i) */

	/* Schur plan = SchurComplement ctor (schur.h:25-159): per HAp block the landmarks shared by its two edges */
	int32_t *sch_term_off;          /* [n_hap+1] (NULL when n_sch_terms==0) */
	int32_t *sch_b1, *sch_b2;       /* [n_sch_terms] HApf block of (hap_i, lm) and of (hap_j, lm) */
	int32_t *sch_lm;                /* [n_sch_terms] lm slot */
	int32_t *lm_hapf_off;           /* [n_unk_lms+1] CSR lm -> HApf blocks in ascending edge slot (schur.h:278-296 order) */
	int32_t *lm_hapf_idx;           /* [n_hapf] */

	/* OUT: crpLandmarksApprox (optimize_edges.h:732-746): Hf_ii of landmarks whose (Hf_ii+lambda I) was invertible */
	double  *ulm_inf;               /* [n_unk_lms*L*L] may be NULL */
	uint8_t *ulm_inf_valid;         /* [n_unk_lms]     may be NULL */
} srba_problem_capsule;

/* What optimize_edges reports in TOptimizeExtraOutputInfo (RbaEngine.h:125-177) + LM bookkeeping. */
enum srba_stop_reason {
	SRBA_STOP_MAX_ITERS = 0, SRBA_STOP_LAMBDA = 1, SRBA_STOP_RMSE = 2, SRBA_STOP_GRADIENT = 3, SRBA_STOP_RHO = 4
};
#define SRBA_TRACE_LEN 48
typedef struct srba_lm_result {
	int32_t status;           /* 0 ok; 1 rank assert failed (optimize_edges.h:355): problem left untouched; 2 internal, never returned: the replicas of a speculative single-capsule run lost step --
		the library re-runs the capsule on the sequential path before it hands the record out */
	int32_t num_iters;        /* value of "iter" at loop exit (optimize_edges.h:452-454) */
	int32_t num_trials;       /* passes of the inner while (optimize_edges.h:471-692) = "LM trials" */
	int32_t num_not_pd;       /* solve() returned false (:476-485) */
	int32_t num_accepted;     /* rho>0 */
	int32_t num_relinearized;
	int32_t num_invalid_jacobs; /* nInvalidJacobs at S10 (:327-338) */
	int32_t stop_reason;      /* bitmask of (1<<srba_stop_reason) that fired */
	int32_t num_observations; /* nObs (with duplicates) */
	int32_t num_jacobians;    /* count_jacobians (:276) */
	int32_t num_span_tree_numeric_updates; /* (:256) */
	int32_t reserved;
	double  total_sqr_error_init, total_sqr_error_final, obs_rmse;
	double  lambda_init, lambda_final;
	/* per-trial trace (first SRBA_TRACE_LEN trials): chi2 of the trial point (NaN when solve failed), lambda used, rho */
	double  trace_chi2[SRBA_TRACE_LEN];
	double  trace_lambda[SRBA_TRACE_LEN];
	double  trace_rho[SRBA_TRACE_LEN];
	double  lambda_last_trial; /* lambda of the last trial (the one the solver's extra_results refer to, lev-marq_solvers.h:204-208); NaN when no trial ran. Unlike the trace it is not limited to
		SRBA_TRACE_LEN trials */
} srba_lm_result;

typedef struct srba_hip_ctx srba_hip_ctx;

/* ---- life cycle ---- */
/* device<0: use the current HIP device. Returns NULL on failure (no GPU, bad params): srba_hip_last_error(NULL). */
srba_hip_ctx *srba_hip_create(int device, const srba_hip_params *params);
int  srba_hip_destroy(srba_hip_ctx *ctx);
/* Replace the parameter block (same family): e.g. define_new_keyframe toggles use_robust_kernel for its stage-1
 * optimisation (define_new_keyframe.h:67-87). */
int  srba_hip_set_params(srba_hip_ctx *ctx, const srba_hip_params *params);
const char *srba_hip_last_error(const srba_hip_ctx *ctx);

/* Replaces the reference's per-call gathering of unknown pointers and column lists
 * (optimize_edges.h:141-163): packs n capsules into SoA device buffers (HBM), keeps a pristine copy of the
 * unknowns so the same batch can be re-run. Host arrays may be freed afterwards. */
int  srba_hip_upload_problems(srba_hip_ctx *ctx, const srba_problem_capsule *capsules, int n);
/* Resets unknowns to the uploaded values (device-to-device). */
int  srba_hip_reset_state(srba_hip_ctx *ctx);

/* ---- stepwise API: one streaming launch per reference hot loop, batch-wide ---- */
int  srba_hip_update_spantree(srba_hip_ctx *ctx, int only_needed);          /* K1  spantree_update_numeric.h:19-81 */
int  srba_hip_eval_residuals(srba_hip_ctx *ctx, double *chi2_out /*[n] host, may be NULL*/); /* K4 reprojection_residuals.h:16-81 */
int  srba_hip_linearize(srba_hip_ctx *ctx);  /* K2,K3 (jacobians.h:1083-1117) + K6 (sparse_hessian_update_numeric.h:22-60) + K5 (compute_minus_gradient.h:20-91).
                                                For <SE2, RelativePoses2D> one fused launch that keeps the Jacobian blocks on the chip (srba_assemble.hpp): Hessian blocks, gradient and the
                                                lambda guess come out as always, the dh_dAp blocks are written to device memory only when srba_hip_debug_read(ctx, 1, ...) or
                                                srba_hip_hessian_from_jacobians() asks for them (from the state of that moment: read them before the next srba_hip_apply_update / rollback) */
int  srba_hip_solve(srba_hip_ctx *ctx, const double *lambda /*[n] host*/, int32_t *not_pd_out /*[n] host*/); /* K7-K10 lev-marq_solvers.h solve() */
int  srba_hip_apply_update(srba_hip_ctx *ctx);   /* K11+K12 backup then x <- x (+) delta (optimize_edges.h:491-539) */
int  srba_hip_rollback(srba_hip_ctx *ctx);       /* K12 restore (optimize_edges.h:664-680) */

/* ---- fused API: the whole of optimize_edges S5..S17 on the device, one workgroup per capsule ---- */
int  srba_hip_lm_run(srba_hip_ctx *ctx, srba_lm_result *results /*[n] host, may be NULL*/);
/* One optimize_edges() call of the reference (impl/optimize_edges.h:256-751, write-back in place: 526, 538) in one call: srba_hip_upload_problems(ctx, capsule, 1) +
 * srba_hip_lm_run(ctx, result) + srba_hip_download_state(ctx, capsule, 1) with a single wait for the device. What RbaEngine<>::optimize_edges() binds per key-frame. */
int  srba_hip_optimize_capsule(srba_hip_ctx *ctx, srba_problem_capsule *capsule, srba_lm_result *result);
/* Same, asynchronous on the context's stream, no host copies: for timing loops. */
int  srba_hip_lm_run_async(srba_hip_ctx *ctx);
int  srba_hip_sync(srba_hip_ctx *ctx);
void *srba_hip_stream(srba_hip_ctx *ctx);   /* hipStream_t the kernels are launched on */

/* ---- whole-map squared error: RbaEngine<>::eval_overall_squared_error() (impl/eval_overall_error.h:15-137) ----
 * The host front-end lists, for every distinct (observer KF, base KF) pair of the map, the breadth-first path between the two key-frames
 * (impl/spantree_create_complete.h:18-126; root = the smaller id) and every observation of the map; the device composes the poses along
 * the paths (root towards the leaf, same association order as the reference) and evaluates sum ||z - h(pose (+) landmark)||^2 without
 * robust kernel. All arrays are host pointers, copied per call; the context's family / sensor parameters apply. */
typedef struct srba_overall_problem {
	int32_t n_edges, n_pairs, n_path, n_obs, n_lms, reserved;
	const double  *edge_pose;      /* [n_edges * PD] inv_pose of every kf2kf edge */
	const int32_t *pair_path_off;  /* [n_pairs + 1] CSR into path_edge */
	const int32_t *path_edge;      /* [n_path] (edge << 1) | use_inverse, in composition order from the root of the pair */
	const int32_t *obs_pose;       /* [n_obs] 2*pair (pose of the leaf seen from the root) or 2*pair+1 (its inverse); -1 = identity */
	const int32_t *obs_lm;         /* [n_obs] index into lm_pos */
	const double  *obs_z;          /* [n_obs * O] */
	const double  *lm_pos;         /* [n_lms * L] landmark position relative to its base key-frame */
} srba_overall_problem;
int  srba_hip_eval_overall_sqr_error(srba_hip_ctx *ctx, const srba_overall_problem *prob, double *sqr_error_out);

/* ---- read back ---- */
/* Writes unknown edge poses / landmark positions / ST poses / ulm_inf back into the arrays of the SAME capsule
 * structs (host pointers) that describe the batch layout (optimize_edges.h:526,538 write in place in the reference). */
int  srba_hip_download_state(srba_hip_ctx *ctx, srba_problem_capsule *capsules, int n);
int  srba_hip_download_results(srba_hip_ctx *ctx, srba_lm_result *results, int n);
/* Debug / parity read-backs of intermediate arrays, concatenated over the batch in capsule order.
 * what: 0 residuals [n_obs*O], 1 dh_dAp blocks [n_bp*O*P] row-major, 2 dh_df blocks [n_bf*O*L], 3 HAp blocks [n_hap*P*P],
 *       4 Hf blocks [n_hf*L*L], 5 HApf blocks [n_hapf*P*L], 6 minus_grad [P*nK+L*nF], 7 delta_eps, 8 validity bytes as doubles,
 *       9 ST poses [2*n_pairs*PD] */
int64_t srba_hip_debug_size(srba_hip_ctx *ctx, int what);
int  srba_hip_debug_read(srba_hip_ctx *ctx, int what, double *out, int64_t n_doubles);
/* Write twin of srba_hip_debug_read for what = 1 (dh_dAp blocks), 2 (dh_df blocks), 6 (minus_grad), and K6 alone on the blocks in device memory
 * (every row valid): together with srba_hip_solve they replay the reference's SchurTests (tests/schur_unittest.cpp:71-279 -- Hessians and Schur
 * complement from GIVEN Jacobian blocks) on the device. After srba_hip_solve, what = 3 reads the Schur-reduced HAp and what = 6 the reduced gradient. */
int  srba_hip_debug_write(srba_hip_ctx *ctx, int what, const double *in, int64_t n_doubles);
int  srba_hip_hessian_from_jacobians(srba_hip_ctx *ctx);

/* Totals over the uploaded batch: used by bench.py for the algorithmic-bytes roofline. */
typedef struct srba_batch_stats {
	int64_t n_problems, n_edges, n_unk_edges, n_unk_lms, n_pairs, n_pairs_needed, n_path, n_path_needed,
	        n_obs, n_bp, n_bf, n_hap, n_hap_terms, n_hf_terms, n_hapf_terms, n_sch_terms, n_scalars,
	        n_chol_blocks /* 3x3 blocks of the symbolic Cholesky factors */, n_chol_items /* block updates per factorisation */;
	int64_t device_bytes;   /* HBM held by the context for this batch */
} srba_batch_stats;
int  srba_hip_batch_stats(srba_hip_ctx *ctx, srba_batch_stats *out);

/* Large-window path (capsules whose system does not fit one wavefront's LDS; srba_amd/csrc/srba_big.hpp): dense Cholesky factorisations since the last upload.
 * out = { milliseconds inside the TIMED factorisation sequences (HIP events on the lane's stream; since round 5 every 8th sequence of a lane is timed, SRBA_HIP_BIG_TIME_EVERY:
 * a time-stamp event drains the queue around it), total flops ld^3/3 of ALL factorisations, number of factorisations, largest system }. */
int    srba_hip_big_path_stats(srba_hip_ctx *ctx, double out[4]);
/* The same with the launch sequences counted: since round 4 the large windows of a batch run in lock-step and ONE sequence of panel / update launches factors the systems of all
 * windows that are in a trial (srba_big.hpp, Gang). out = { ms of the timed sequences, flops, factorisations, largest system, launch sequences, 1 if the lock-step gang is on,
 * timed launch sequences, flops of the timed sequences }: achieved rate = out[7] / out[0], time of a sequence = out[0] / out[6]. */
int    srba_hip_big_path_stats2(srba_hip_ctx *ctx, double out[8]);
/* Order in which the class launches of the last srba_hip_lm_run* started on the device. The fused LM kernel is one persistent launch per size class, all enqueued at once on their own
 * streams; the plan holds each stream back so that the launches start largest-footprint-first (DESIGN 4a "staggered start"). For plan job j (in plan order = the intended order):
 * stamp[j] = device time (ticks of the constant 100 MHz counter) at which its first capsule was taken, 0 if it never started; workgroups[j], delay_us[j] (either may be NULL) = its grid and
 * the delay its stream was held back by. Returns the number of jobs written (<= n). The order held iff the non-zero stamps are non-decreasing. Synchronises the stream. */
int    srba_hip_launch_order(srba_hip_ctx *ctx, int64_t *stamp, int32_t *workgroups, int32_t *delay_us, int n);
/* Single-capsule batches of the relative-pose SE2 family (the per-key-frame use: RbaEngine<>::optimize_edges, impl/optimize_edges.h:471-692) speculate on the lambda ladder with several
 * workgroups. out = { speculative launches since the context was created, launches whose replicas lost step (one of them not resident within the spin bound: another context holding
 * the CUs) and that were therefore run again on the sequential path }. */
int    srba_hip_spec_stats(srba_hip_ctx *ctx, int64_t out[2]);
/* Host only (no device, no context): the packed row records of the fused normal-equations kernel of <SE2, RelativePoses2D> for ONE capsule (srba_amd/csrc/srba_assemble.hpp: 16 bytes =
 * four words per observation row that has Jacobian blocks, rows dealt to the lanes the kernel will run them on) -- what srba_hip_upload_problems builds for srba_hip_linearize, exposed
 * so that the packing can be checked against the capsule's own Hessian plan without a GPU (tests/test_assemble_records.py). words: room for 4 * cap_records; returns the number of
 * records (a multiple of 16), 0 if the capsule does not fit the kernel (it then takes the unfused one), -1 - needed if cap_records is too small. */
int64_t srba_hip_debug_assemble_records(const srba_problem_capsule *capsule, uint32_t *words, int64_t cap_records);
/* Time (ms) spent inside the last srba_hip_lm_run* kernel launch, measured with HIP events on the context stream. */
double srba_hip_last_kernel_ms(srba_hip_ctx *ctx);
/* Durations (ms, most recent first) of the last `n` srba_hip_lm_run / srba_hip_lm_run_async launches: HIP events recorded on the context
 * stream around the launch (fork to join of the size-class kernels). Synchronises the stream; returns how many were written (<= 64). */
int    srba_hip_kernel_ms_history(srba_hip_ctx *ctx, double *out_ms, int n);
/* Per-stage cycle counters of the fused kernel (the reference's SRBA_DETAILED_TIME_PROFILING sections, impl/optimize_edges.h:16-27), read with srba_hip_debug_read(ctx, 10, ...): on != 0 switches
 * them on for the batches uploaded to THIS context from now on (the environment variable SRBA_HIP_PHASE_TIMING=1 sets the default of new contexts). */
int    srba_hip_set_phase_timing(srba_hip_ctx *ctx, int on);

#ifdef __cplusplus
}
#endif
#endif /* SRBA_HIP_H */
