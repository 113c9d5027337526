/*
 * engine_capi.cpp -- thin C wrapper over the header-only srba::RbaEngine<> front-end (include/srba.h) so that the
 * Python drivers (bench.py, tests/) can feed keyframes through exactly the code path a C++ user of the reference's
 * API would (rba.define_new_keyframe(obs, info), apps/srba-slam/srba-run-generic-impl.h:461), harvest the
 * problem capsules of every optimize_local_area() call, and read back edges / landmarks / spanning-tree tables.
 *
 * The numeric back-end is the GPU (srba::hip_backend) unless a plain C function is plugged in with
 * srba_engine_set_backend_fn(): tests use that to run the same front-end against the CPU oracle.  This file never
 * references anything under oracle/.
 */
#include "../../include/srba.h"
#include "engine_capi.h"

using namespace srba;

namespace {

template <class OBS> struct obs_io;
template <> struct obs_io<observations::RelativePoses_2D> { static void set(observations::RelativePoses_2D::obs_data_t &o, const double *z) { o.x = z[0]; o.y = z[1]; o.yaw = z[2]; } };
template <> struct obs_io<observations::RangeBearing_2D> { static void set(observations::RangeBearing_2D::obs_data_t &o, const double *z) { o.range = z[0]; o.yaw = z[1]; } };
template <> struct obs_io<observations::Cartesian_2D> { static void set(observations::Cartesian_2D::obs_data_t &o, const double *z) { o.pt.x = z[0]; o.pt.y = z[1]; } };
template <> struct obs_io<observations::Cartesian_3D> { static void set(observations::Cartesian_3D::obs_data_t &o, const double *z) { o.pt.x = z[0]; o.pt.y = z[1]; o.pt.z = z[2]; } };
template <> struct obs_io<observations::RangeBearing_3D> { static void set(observations::RangeBearing_3D::obs_data_t &o, const double *z) { o.range = z[0]; o.yaw = z[1]; o.pitch = z[2]; } };
template <> struct obs_io<observations::RelativePoses_3D> { static void set(observations::RelativePoses_3D::obs_data_t &o, const double *z) { o.x = z[0]; o.y = z[1]; o.z = z[2]; o.yaw = z[3];
	o.pitch = z[4]; o.roll = z[5]; } };
template <> struct obs_io<observations::MonocularCamera> { static void set(observations::MonocularCamera::obs_data_t &o, const double *z) { o.px.x = (float)z[0]; o.px.y = (float)z[1]; } };
template <> struct obs_io<observations::StereoCamera> { static void set(observations::StereoCamera::obs_data_t &o, const double *z) { o.left_px.x = (float)z[0]; o.left_px.y = (float)z[1];
	o.right_px.x = (float)z[2]; o.right_px.y = (float)z[3]; } };

template <class NOISE> struct noise_io;
template <> struct noise_io<options::observation_noise_identity> { template <class P> static void set(P &p, const srba_engine_config &c) { p.std_noise_observations = c.std_noise_observations; } };
template <class OBS> struct noise_io<options::observation_noise_constant_matrix<OBS> > { template <class P> static void set(P &p, const srba_engine_config &c) { for (size_t i = 0;
	i < OBS::OBS_DIMS * OBS::OBS_DIMS; i++) p.lambda.m[i] = c.lambda[i]; } };
template <class SP> struct spose_io;
template <> struct spose_io<options::sensor_pose_on_robot_none> { template <class P> static void set(P &, const srba_engine_config &) {} };
template <> struct spose_io<options::sensor_pose_on_robot_se3> { template <class P> static void set(P &p, const srba_engine_config &c) {
	p.relative_pose = mrpt::poses::CPose3D(c.sensor_pose_xyzypr[0], c.sensor_pose_xyzypr[1], c.sensor_pose_xyzypr[2], c.sensor_pose_xyzypr[3], c.sensor_pose_xyzypr[4], c.sensor_pose_xyzypr[5]); } };
template <class OBS> struct sensor_io { template <class P> static void set(P &, const srba_engine_config &) {} };
template <> struct sensor_io<observations::MonocularCamera> { template <class P> static void set(P &p, const srba_engine_config &c) { p.camera_calib.setIntrinsicParamsFromValues(c.cam_left[0],
	c.cam_left[1], c.cam_left[2], c.cam_left[3]); } };
template <> struct sensor_io<observations::StereoCamera> { template <class P> static void set(P &p, const srba_engine_config &c) {
	p.camera_calib.leftCamera.setIntrinsicParamsFromValues(c.cam_left[0], c.cam_left[1], c.cam_left[2], c.cam_left[3]);
	p.camera_calib.rightCamera.setIntrinsicParamsFromValues(c.cam_right[0], c.cam_right[1], c.cam_right[2], c.cam_right[3]);
	p.camera_calib.rightCameraPose = mrpt::poses::CPose3DQuat(c.right_cam_pose[0], c.right_cam_pose[1], c.right_cam_pose[2], mrpt::math::CQuaternionDouble(c.right_cam_pose[3], c.right_cam_pose[4],
		c.right_cam_pose[5], c.right_cam_pose[6])); } };

struct Harvest {
	std::deque<CapsuleData> data; std::vector<srba_problem_capsule> views; std::vector<uint64_t> kf_of; bool dirty = true;
};

struct EngineBase {
	srba_engine_config cfg; Harvest harvest; srba_hip_params last_params; std::string error;
	virtual ~EngineBase() {}
	virtual int add_keyframe(int n_obs, const uint64_t *feat_id, const double *z, const uint8_t *flags, const double *relpos, srba_kf_info *out) = 0;
	virtual void set_backend(const std::shared_ptr<numeric_backend> &b) = 0;
	virtual int optimize_local_area(uint64_t root, unsigned win, srba_kf_info *out) = 0;
	virtual int64_t num_edges() const = 0; virtual int get_edge(int64_t i, uint64_t *from, uint64_t *to, double *pose) const = 0;
	virtual int64_t num_unknown_lms() const = 0; virtual int get_unknown_lms(uint64_t *ids, uint64_t *base, double *pos) const = 0;
	virtual int64_t st_dump(int what, int64_t *out, int64_t cap) const = 0;
	virtual int get_rel_pose(uint64_t query, uint64_t reference, double *pose) const = 0;
	virtual double profiler_mean(const char *name) const = 0;
	virtual int eval_overall(double *out) = 0;
	std::shared_ptr<function_backend> fn_backend; // set when a C function was plugged in
	virtual uint64_t alloc_keyframe() = 0;
	virtual int64_t export_graphslam(uint64_t root, uint64_t *node_id, double *node_pose, int64_t node_cap, uint64_t *edge_from_to, double *edge_pose, int64_t edge_cap) const = 0;
	virtual int64_t plan_sweep(const uint64_t *roots, int64_t n, unsigned win, int32_t *round_of, int64_t *touch_off, uint32_t *touch, int64_t cap) = 0;
	virtual int optimize_batch(const uint64_t *roots, int64_t n, unsigned win, srba_kf_info *out) = 0;
	virtual int edge_poses(const uint64_t *ids, int64_t n, double *out, const double *in) = 0;
	virtual int64_t plan_sweep_lms(int64_t *off, uint32_t *touch, int64_t cap) = 0;
	virtual int lm_positions(const uint64_t *ids, int64_t n, double *out, const double *in) = 0;
	virtual int64_t create_edge(uint64_t new_kf, uint64_t from, uint64_t to, const double *pose) = 0;
};

template <class ECP> struct ecp_io;
template <> struct ecp_io<ecps::local_areas_fixed_size> { template <class P> static void set(P &p, const srba_engine_config &c) { p.submap_size = c.submap_size;
	p.min_obs_to_loop_closure = c.min_obs_to_loop_closure; } };
template <> struct ecp_io<ecps::classic_linear_rba> { template <class P> static void set(P &p, const srba_engine_config &c) { p.min_obs_to_loop_closure = c.min_obs_to_loop_closure; } };

template <class KF, class LM, class OBS, class NOISE, class SPOSE, class SOLVER, class ECP = ecps::local_areas_fixed_size>
struct EngineImpl : public EngineBase {
	struct OPTS : public RBA_OPTIONS_DEFAULT { typedef ECP edge_creation_policy_t; typedef SPOSE sensor_pose_on_robot_t; typedef NOISE obs_noise_matrix_t; typedef SOLVER solver_t; };
	typedef RbaEngine<KF, LM, OBS, OPTS> rba_t;
	rba_t rba; uint64_t cur_kf = 0;

	explicit EngineImpl(const srba_engine_config &c) {
		cfg = c;
		rba.setVerbosityLevel(c.verbose); rba.get_time_profiler().enable(c.enable_profiler != 0);
		typename rba_t::TSRBAParameters &s = rba.parameters.srba;
		s.max_tree_depth = c.max_tree_depth; s.max_optimize_depth = c.max_optimize_depth; s.optimize_new_edges_alone = c.optimize_new_edges_alone != 0;
		s.use_robust_kernel = c.use_robust_kernel != 0; s.use_robust_kernel_stage1 = c.use_robust_kernel_stage1 != 0; s.kernel_param = c.kernel_param; s.max_iters = c.max_iters;
		s.max_error_per_obs_to_stop = c.max_error_per_obs_to_stop; s.max_rho = c.max_rho; s.max_lambda = c.max_lambda;
			s.min_error_reduction_ratio_to_relinearize = c.min_error_reduction_ratio_to_relinearize;
		s.cov_recovery = c.cov_recovery ? crpLandmarksApprox : crpNone; s.refresh_all_read_poses = (c.refresh_all_read_poses & 1) != 0; s.restore_spanning_tree_twins = (c.refresh_all_read_poses & 2)
			!= 0; s.schur_keeps_gradient = (c.refresh_all_read_poses & 4) != 0; s.consistent_loop_closure_init = (c.refresh_all_read_poses & 8) != 0;
			// bit 0 / bit 1 of the config field: the two extensions of SURVEY App. B-12
		ecp_io<ECP>::set(rba.parameters.ecp, c);
		noise_io<NOISE>::set(rba.parameters.obs_noise, c); spose_io<SPOSE>::set(rba.parameters.sensor_pose, c); sensor_io<OBS>::set(rba.parameters.sensor, c);
		rba.set_hip_device(c.hip_device);
		rba.on_capsule = [this](const srba_hip_params &hp, CapsuleData &cd, int stage) {
			last_params = hp;
			const bool stage1 = (stage == 1); // define_new_keyframe's single-edge optimisations, tagged by the engine itself
			if ((cfg.harvest & 1) && !stage1) { harvest.data.push_back(cd); harvest.kf_of.push_back(cur_kf); harvest.dirty = true; }
			if ((cfg.harvest & 2) && stage1) { harvest.data.push_back(cd); harvest.kf_of.push_back(cur_kf); harvest.dirty = true; }
		};
	}
	void set_backend(const std::shared_ptr<numeric_backend> &b) { rba.set_numeric_backend(b); }
	static void fill_info(srba_kf_info *out, const typename rba_t::TOptimizeExtraOutputInfo &r, const typename rba_t::TOptimizeExtraOutputInfo *s1) {
		out->num_observations = r.num_observations; out->num_jacobians = r.num_jacobians; out->num_k2k = r.num_kf2kf_edges_optimized; out->num_k2f = r.num_kf2lm_edges_optimized;
		out->chi2_init = r.total_sqr_error_init; out->chi2_final = r.total_sqr_error_final; out->obs_rmse = r.obs_rmse; out->lm = r.lm;
		if (s1) out->lm_stage1 = s1->lm;
	}
	int add_keyframe(int n_obs, const uint64_t *feat_id, const double *z, const uint8_t *flags, const double *relpos, srba_kf_info *out) {
		try {
			typename rba_t::new_kf_observations_t list;
			for (int i = 0; i < n_obs; i++) {
				typename rba_t::new_kf_observation_t o;
				o.obs.feat_id = feat_id[i]; obs_io<OBS>::set(o.obs.obs_data, z + (size_t)i * OBS::OBS_DIMS);
				o.is_fixed = (flags && (flags[i] & 1)); o.is_unknown_with_init_val = (flags && (flags[i] & 2));
				if (relpos) for (size_t k = 0; k < LM::LM_DIMS; k++) o.feat_rel_pos[k] = relpos[(size_t)i * LM::LM_DIMS + k];
				list.push_back(o);
			}
			typename rba_t::TNewKeyFrameInfo info;
			cur_kf = rba.get_rba_state().keyframes.size();
			rba.define_new_keyframe(list, info, cfg.run_local_optimization != 0);
			if (out) {
				std::memset(out, 0, sizeof(*out));
				out->kf_id = info.kf_id; out->n_new_edges = (int)std::min<size_t>(info.created_edge_ids.size(), 4);
				for (int i = 0; i < out->n_new_edges; i++) { const TNewEdgeInfo &e = info.created_edge_ids[i]; out->edge_id[i] = e.id; out->edge_has_init[i] = e.has_approx_init_val;
					out->lc_observer[i] = e.loopclosure_observer_kf; out->lc_base[i] = e.loopclosure_base_kf; }
				fill_info(out, info.optimize_results, &info.optimize_results_stg1);
			}
			return 0;
		} catch (std::exception &e) { error = e.what(); return -1; }
	}
	int optimize_local_area(uint64_t root, unsigned win, srba_kf_info *out) {
		try { typename rba_t::TOptimizeExtraOutputInfo r; rba.optimize_local_area(root, win, r); if (out) { std::memset(out, 0, sizeof(*out)); out->kf_id = root; fill_info(out, r, NULL); } return 0; }
		catch (std::exception &e) { error = e.what(); return -1; }
	}
	int64_t num_edges() const { return (int64_t)rba.get_k2k_edges().size(); }
	int get_edge(int64_t i, uint64_t *from, uint64_t *to, double *pose) const {
		if (i < 0 || i >= num_edges()) return -1;
		const typename rba_t::k2k_edge_t &e = rba.get_k2k_edges()[i]; *from = e.from; *to = e.to; e.inv_pose.storeTo(pose); return 0;
	}
	int64_t num_unknown_lms() const { return (int64_t)rba.get_unknown_feats().size(); }
	int get_unknown_lms(uint64_t *ids, uint64_t *base, double *pos) const {
		size_t i = 0;
		for (typename rba_t::TRelativeLandmarkPosMap::const_iterator it = rba.get_unknown_feats().begin(); it != rba.get_unknown_feats().end(); ++it, ++i) { ids[i] = it->first;
			base[i] = it->second.id_frame_base; for (size_t k = 0; k < LM::LM_DIMS; k++) pos[i * LM::LM_DIMS + k] = it->second.pos[k]; }
		return 0;
	}
	/** what=0: next_edge rows [src trg next dist]; what=1: all_edges rows [from to len e0 e1 ...] ; returns the number of int64 needed */
	int64_t st_dump(int what, int64_t *out, int64_t cap) const {
		int64_t n = 0; const graph::topology &T = rba.get_rba_state().topo;
		auto put = [&](int64_t v) { if (out && n < cap) out[n] = v; n++; };
		for (size_t s = 0; s < T.st.rows(); s++) {
			const graph::st_entry *r = T.st.row((graph::id32)s);
			for (size_t i = 0; i < T.st.len((graph::id32)s); i++) {
				if (what == 0) { put((int64_t)s); put(r[i].trg); put(r[i].next); put(r[i].dist); }
				else if (r[i].trg < s && r[i].path != graph::NIL) { put((int64_t)s); put(r[i].trg); put(r[i].path_len); for (uint32_t k = 0; k < r[i].path_len; k++) put(T.path_pool[r[i].path + k]); }
			}
		}
		return n;
	}
	int get_rel_pose(uint64_t query, uint64_t reference, double *pose) const { const typename rba_t::pose_t *p = rba.get_kf_relative_pose(query, reference); if (!p) return -1; p->storeTo(pose);
		return 0; }
	double profiler_mean(const char *name) const { return const_cast<rba_t &>(rba).get_time_profiler().getMeanTime(name); }
	int eval_overall(double *out) { try { *out = rba.eval_overall_squared_error(); return 0; } catch (std::exception &e) { error = e.what(); return -1; } }
	uint64_t alloc_keyframe() { return rba.alloc_keyframe(); }
	/** RbaEngine<>::get_global_graphslam_problem() into a minimal pose-graph container (what mrpt::graphs::CNetworkOfPoses offers the reference: clear(), nodes[id], insertEdgeAtEnd) */
	struct pose_graph_t {
		std::map<uint64_t, typename rba_t::pose_t> nodes; std::vector<std::pair<std::pair<uint64_t, uint64_t>, typename rba_t::pose_t> > edges;
		void clear() { nodes.clear(); edges.clear(); }
		void insertEdgeAtEnd(uint64_t from, uint64_t to, const typename rba_t::pose_t &p) { edges.push_back(std::make_pair(std::make_pair(from, to), p)); }
	};
	int64_t plan_sweep(const uint64_t *roots, int64_t n, unsigned win, int32_t *round_of, int64_t *touch_off, uint32_t *touch, int64_t cap) {
		std::vector<TKeyFrameID> r(roots, roots + n); typename rba_t::TSweepPlan &plan = last_plan; rba.plan_local_area_sweep(r, win, plan);
		if ((int64_t)plan.touch.size() > cap) return -2 - (int64_t)plan.touch.size();
		for (int64_t i = 0; i < n; i++) round_of[i] = plan.round_of[i];
		for (int64_t i = 0; i <= n; i++) touch_off[i] = plan.touch_off[i];
		std::copy(plan.touch.begin(), plan.touch.end(), touch);
		return plan.n_rounds;
	}
	int optimize_batch(const uint64_t *roots, int64_t n, unsigned win, srba_kf_info *out) {
		std::vector<TKeyFrameID> r(roots, roots + n); std::vector<typename rba_t::TOptimizeExtraOutputInfo> res; cur_kf = n > 0 ? roots[0] : 0;
		rba.optimize_local_areas_batch(r, win, res);
		if (out) for (int64_t i = 0; i < n; i++) { std::memset(&out[i], 0, sizeof(out[i])); out[i].kf_id = roots[i]; fill_info(&out[i], res[i], NULL); }
		return 0;
	}
	typename rba_t::TSweepPlan last_plan;
	int64_t plan_sweep_lms(int64_t *off, uint32_t *touch, int64_t cap) { // the landmarks the windows of the LAST plan_sweep touch (id | 0x80000000: written)
		const typename rba_t::TSweepPlan &plan = last_plan; if ((int64_t)plan.touch_lm.size() > cap) return -2 - (int64_t)plan.touch_lm.size();
		std::copy(plan.touch_lm_off.begin(), plan.touch_lm_off.end(), off); std::copy(plan.touch_lm.begin(), plan.touch_lm.end(), touch); return (int64_t)plan.touch_lm.size();
	}
	int lm_positions(const uint64_t *ids, int64_t n, double *out, const double *in) {
		const size_t nL = rba.lm_table_size(); std::vector<size_t> id(n); for (int64_t i = 0; i < n; i++) { if (ids[i] >= nL) throw std::out_of_range("landmark id out of range"); id[i] = (size_t)ids[i]; }
		if (out) rba.get_lm_positions(id.data(), (size_t)n, out); if (in) rba.set_lm_positions(id.data(), (size_t)n, in);
		return 0;
	}
	int edge_poses(const uint64_t *ids, int64_t n, double *out, const double *in) {
		const size_t nE = rba.get_k2k_edges().size(); std::vector<size_t> id(n); for (int64_t i = 0; i < n; i++) { if (ids[i] >= nE) throw std::out_of_range("kf2kf edge id out of range"); id[i] = (size_t)ids[i]; }
		if (out) rba.get_k2k_edge_poses(id.data(), (size_t)n, out); if (in) rba.set_k2k_edge_poses(id.data(), (size_t)n, in);
		return 0;
	}
	int64_t export_graphslam(uint64_t root, uint64_t *node_id, double *node_pose, int64_t node_cap, uint64_t *edge_from_to, double *edge_pose, int64_t edge_cap) const {
		pose_graph_t g; typename rba_t::ExportGraphSLAM_Params prm; prm.root_kf_id = root;
		rba.get_global_graphslam_problem(g, prm);
		const size_t PD = rba_t::pose_t::storage_doubles(); int64_t i = 0;
		for (typename std::map<uint64_t, typename rba_t::pose_t>::const_iterator it = g.nodes.begin(); it != g.nodes.end() && i < node_cap; ++it, ++i) { node_id[i] = it->first;
			it->second.storeTo(node_pose + (size_t)i * PD); }
		for (size_t e = 0; e < g.edges.size() && (int64_t)e < edge_cap; e++) { edge_from_to[2 * e] = g.edges[e].first.first; edge_from_to[2 * e + 1] = g.edges[e].first.second;
			g.edges[e].second.storeTo(edge_pose + e * PD); }
		return (int64_t)g.nodes.size();
	}
	int64_t create_edge(uint64_t new_kf, uint64_t from, uint64_t to, const double *pose) {
		try { typename rba_t::pose_t p; if (pose) p.loadFrom(pose); typename rba_t::new_kf_observations_t dummy; return (int64_t)rba.create_kf2kf_edge(new_kf, TPairKeyFrameID(from, to), dummy, p); }
		catch (std::exception &e) { error = e.what(); return -1; }
	}
};

typedef options::observation_noise_identity N_ID;
typedef options::sensor_pose_on_robot_none SP_NONE;
typedef options::sensor_pose_on_robot_se3 SP_SE3;
typedef options::solver_LM_schur_dense_cholesky S_SD;
typedef options::solver_LM_schur_sparse_cholesky S_SS;
typedef options::solver_LM_no_schur_sparse_cholesky S_NS;

template <class KF, class LM, class OBS, class NOISE, class SPOSE>
EngineBase *make_solver(const srba_engine_config &c) {
	switch (c.solver) {
		case SRBA_SOLVER_SCHUR_DENSE_CHOL: return new EngineImpl<KF, LM, OBS, NOISE, SPOSE, S_SD>(c);
		case SRBA_SOLVER_SCHUR_SPARSE_CHOL: return new EngineImpl<KF, LM, OBS, NOISE, SPOSE, S_SS>(c);
		case SRBA_SOLVER_NO_SCHUR_SPARSE_CHOL: return new EngineImpl<KF, LM, OBS, NOISE, SPOSE, S_NS>(c);
	}
	return NULL;
}
template <class KF, class LM, class OBS>
EngineBase *make_point_engine(const srba_engine_config &c) { // landmark families: identity noise (the only policy the reference apps use for them)
	if (c.noise != SRBA_NOISE_IDENTITY) return NULL;
	if (c.sensor_pose == SRBA_SENSOR_POSE_SE3) return make_solver<KF, LM, OBS, N_ID, SP_SE3>(c);
	return make_solver<KF, LM, OBS, N_ID, SP_NONE>(c);
}
template <class KF, class LM, class OBS>
EngineBase *make_point_engine_2d(const srba_engine_config &c) {
	if (c.noise != SRBA_NOISE_IDENTITY || c.sensor_pose != SRBA_SENSOR_POSE_NONE) return NULL;
	return make_solver<KF, LM, OBS, N_ID, SP_NONE>(c);
}

std::string g_error;
} // namespace

extern "C" {

void srba_engine_config_default(srba_engine_config *c, int family) {
	std::memset(c, 0, sizeof(*c));
	c->family = family; c->solver = SRBA_SOLVER_SCHUR_DENSE_CHOL; c->noise = SRBA_NOISE_IDENTITY; c->sensor_pose = SRBA_SENSOR_POSE_NONE;
	c->std_noise_observations = 1.0; for (int i = 0; i < 6; i++) c->lambda[i * 7 % 36] = 0; for (int i = 0; i < 6; i++) c->lambda[i * 6 + i] = 1.0;
	c->right_cam_pose[3] = 1.0; c->cam_left[0] = c->cam_left[1] = c->cam_right[0] = c->cam_right[1] = 1.0;
	c->max_tree_depth = 4; c->max_optimize_depth = 4; c->submap_size = 15; c->min_obs_to_loop_closure = 4; // rba_problem_common.h:35-56, ecps/local_areas_fixed_size.h:24-33
	c->optimize_new_edges_alone = 1; c->use_robust_kernel = 0; c->use_robust_kernel_stage1 = 0; c->kernel_param = 3.0; c->max_iters = 20;
	c->max_error_per_obs_to_stop = 1e-6; c->max_rho = 10.0; c->max_lambda = 1e20; c->min_error_reduction_ratio_to_relinearize = 0.01; c->cov_recovery = 1;
	c->run_local_optimization = 1; c->harvest = 0; c->verbose = 0; c->enable_profiler = 0; c->hip_device = -1;
}

void *srba_engine_create(const srba_engine_config *c) {
	EngineBase *e = NULL;
	try {
		if (c->ecp == 1) { // classic linear RBA: the two problem types the reference tutorials use it with
			if (c->family == SRBA_SE2_RELPOSE2D && c->noise == SRBA_NOISE_CONSTANT_MATRIX && c->sensor_pose == SRBA_SENSOR_POSE_NONE && c->solver == SRBA_SOLVER_NO_SCHUR_SPARSE_CHOL)
				e = new EngineImpl<kf2kf_poses::SE2, landmarks::RelativePoses2D, observations::RelativePoses_2D, options::observation_noise_constant_matrix<observations::RelativePoses_2D>, SP_NONE,
					S_NS, ecps::classic_linear_rba>(*c);
			else if (c->family == SRBA_SE3_CART3D && c->noise == SRBA_NOISE_IDENTITY && c->sensor_pose == SRBA_SENSOR_POSE_NONE && c->solver == SRBA_SOLVER_SCHUR_DENSE_CHOL)
				e = new EngineImpl<kf2kf_poses::SE3, landmarks::Euclidean3D, observations::Cartesian_3D, N_ID, SP_NONE, S_SD, ecps::classic_linear_rba>(*c);
		} else
		switch (c->family) {
			case SRBA_SE2_RELPOSE2D:
				if (c->noise == SRBA_NOISE_CONSTANT_MATRIX && c->sensor_pose == SRBA_SENSOR_POSE_NONE)
					e = make_solver<kf2kf_poses::SE2, landmarks::RelativePoses2D, observations::RelativePoses_2D, options::observation_noise_constant_matrix<observations::RelativePoses_2D>,
						SP_NONE>(*c);
				else if (c->noise == SRBA_NOISE_IDENTITY && c->sensor_pose == SRBA_SENSOR_POSE_NONE)
					e = make_solver<kf2kf_poses::SE2, landmarks::RelativePoses2D, observations::RelativePoses_2D, N_ID, SP_NONE>(*c);
				break;
			case SRBA_SE2_RB2D: e = make_point_engine_2d<kf2kf_poses::SE2, landmarks::Euclidean2D, observations::RangeBearing_2D>(*c); break;
			case SRBA_SE2_CART2D: e = make_point_engine_2d<kf2kf_poses::SE2, landmarks::Euclidean2D, observations::Cartesian_2D>(*c); break;
			case SRBA_SE3_STEREO: e = make_point_engine<kf2kf_poses::SE3, landmarks::Euclidean3D, observations::StereoCamera>(*c); break;
			case SRBA_SE3_MONO: e = make_point_engine<kf2kf_poses::SE3, landmarks::Euclidean3D, observations::MonocularCamera>(*c); break;
			case SRBA_SE3_CART3D: e = make_point_engine<kf2kf_poses::SE3, landmarks::Euclidean3D, observations::Cartesian_3D>(*c); break;
			case SRBA_SE3_RB3D: e = make_point_engine<kf2kf_poses::SE3, landmarks::Euclidean3D, observations::RangeBearing_3D>(*c); break;
			case SRBA_SE3_RELPOSE3D: // SE(3) relative graph-SLAM (tutorial-srba-relative-graph-slam-se3.cpp): constant 6x6 information matrix, or identity noise
				if (c->noise == SRBA_NOISE_CONSTANT_MATRIX && c->sensor_pose == SRBA_SENSOR_POSE_NONE)
					e = make_solver<kf2kf_poses::SE3, landmarks::RelativePoses3D, observations::RelativePoses_3D, options::observation_noise_constant_matrix<observations::RelativePoses_3D>,
						SP_NONE>(*c);
				else if (c->noise == SRBA_NOISE_IDENTITY && c->sensor_pose == SRBA_SENSOR_POSE_NONE)
					e = make_solver<kf2kf_poses::SE3, landmarks::RelativePoses3D, observations::RelativePoses_3D, N_ID, SP_NONE>(*c);
				break;
			case SRBA_SE2_STEREO: // SE(2) key-frames + 3D landmarks + stereo camera (tutorial-srba-stereo-se2.cpp)
				if (c->noise == SRBA_NOISE_IDENTITY) e = (c->sensor_pose == SRBA_SENSOR_POSE_SE3) ? make_solver<kf2kf_poses::SE2, landmarks::Euclidean3D, observations::StereoCamera, N_ID, SP_SE3>(*c)
				                                                                                     : make_solver<kf2kf_poses::SE2, landmarks::Euclidean3D, observations::StereoCamera, N_ID,
				                                                                                     	SP_NONE>(*c);
				break;
		}
	} catch (std::exception &ex) { g_error = ex.what(); return NULL; }
	if (!e) g_error = "srba_engine_create: unsupported family / policy combination";
	return e;
}
void srba_engine_destroy(void *h) { delete static_cast<EngineBase *>(h); }
const char *srba_engine_last_error(void *h) { return h ? static_cast<EngineBase *>(h)->error.c_str() : g_error.c_str(); }
int srba_engine_set_backend_fn(void *h, srba_backend_fn fn, const char *name) {
	EngineBase *e = static_cast<EngineBase *>(h);
	e->fn_backend.reset(new function_backend(fn, name ? name : "external")); e->set_backend(e->fn_backend); return 0;
}
int srba_engine_set_overall_fn(void *h, srba_overall_fn fn) { EngineBase *e = static_cast<EngineBase *>(h); if (!e->fn_backend) { e->error = "set_overall_fn: plug a back-end function first";
	return -1; } e->fn_backend->overall_fn = fn; return 0; }
int srba_engine_eval_overall_sqr_error(void *h, double *out) { return static_cast<EngineBase *>(h)->eval_overall(out); }
int srba_engine_add_keyframe(void *h, int n_obs, const uint64_t *feat_id, const double *z, const uint8_t *flags, const double *relpos, srba_kf_info *out) {
	return static_cast<EngineBase *>(h)->add_keyframe(n_obs, feat_id, z, flags, relpos, out);
}
int srba_engine_optimize_local_area(void *h, uint64_t root, unsigned win, srba_kf_info *out) { return static_cast<EngineBase *>(h)->optimize_local_area(root, win, out); }
int64_t srba_engine_num_edges(void *h) { return static_cast<EngineBase *>(h)->num_edges(); }
int srba_engine_get_edge(void *h, int64_t i, uint64_t *from, uint64_t *to, double *pose) { return static_cast<EngineBase *>(h)->get_edge(i, from, to, pose); }
int64_t srba_engine_num_unknown_lms(void *h) { return static_cast<EngineBase *>(h)->num_unknown_lms(); }
int srba_engine_get_unknown_lms(void *h, uint64_t *ids, uint64_t *base, double *pos) { return static_cast<EngineBase *>(h)->get_unknown_lms(ids, base, pos); }
int64_t srba_engine_st_dump(void *h, int what, int64_t *out, int64_t cap) { return static_cast<EngineBase *>(h)->st_dump(what, out, cap); }
int srba_engine_get_rel_pose(void *h, uint64_t query, uint64_t reference, double *pose) { return static_cast<EngineBase *>(h)->get_rel_pose(query, reference, pose); }
uint64_t srba_engine_alloc_keyframe(void *h) { return static_cast<EngineBase *>(h)->alloc_keyframe(); }
int64_t srba_engine_create_edge(void *h, uint64_t new_kf, uint64_t from, uint64_t to, const double *pose) { return static_cast<EngineBase *>(h)->create_edge(new_kf, from, to, pose); }
#define SRBA_ENGINE_GUARD(label, expr, fail) EngineBase *e = static_cast<EngineBase *>(h); try { return (expr); } catch (const std::exception &ex) { e->error = std::string(label ": ") + ex.what(); return (fail); } \
	catch (...) { e->error = label ": unknown exception"; return (fail); }
int64_t srba_engine_plan_sweep(void *h, const uint64_t *roots, int64_t n, unsigned win, int32_t *round_of, int64_t *touch_off, uint32_t *touch, int64_t cap) {
	SRBA_ENGINE_GUARD("plan_sweep", e->plan_sweep(roots, n, win, round_of, touch_off, touch, cap), -1) }
int srba_engine_optimize_batch(void *h, const uint64_t *roots, int64_t n, unsigned win, srba_kf_info *out) { SRBA_ENGINE_GUARD("optimize_batch", e->optimize_batch(roots, n, win, out), -1) }
int srba_engine_get_edge_poses(void *h, const uint64_t *ids, int64_t n, double *out) { SRBA_ENGINE_GUARD("get_edge_poses", e->edge_poses(ids, n, out, NULL), -1) }
int srba_engine_set_edge_poses(void *h, const uint64_t *ids, int64_t n, const double *in) { SRBA_ENGINE_GUARD("set_edge_poses", e->edge_poses(ids, n, NULL, in), -1) }
int64_t srba_engine_plan_sweep_lms(void *h, int64_t *off, uint32_t *touch, int64_t cap) { SRBA_ENGINE_GUARD("plan_sweep_lms", e->plan_sweep_lms(off, touch, cap), -1) }
int srba_engine_get_lm_positions(void *h, const uint64_t *ids, int64_t n, double *out) { SRBA_ENGINE_GUARD("get_lm_positions", e->lm_positions(ids, n, out, NULL), -1) }
int srba_engine_set_lm_positions(void *h, const uint64_t *ids, int64_t n, const double *in) { SRBA_ENGINE_GUARD("set_lm_positions", e->lm_positions(ids, n, NULL, in), -1) }
#undef SRBA_ENGINE_GUARD
int64_t srba_engine_export_graphslam(void *h, uint64_t root, uint64_t *node_id, double *node_pose, int64_t node_cap, uint64_t *edge_from_to, double *edge_pose, int64_t edge_cap) {
	EngineBase *e = static_cast<EngineBase *>(h); // no C++ exception crosses the C ABI (narrow() of an out-of-range id, allocation failures)
	try { return e->export_graphslam(root, node_id, node_pose, node_cap, edge_from_to, edge_pose, edge_cap); }
	catch (const std::exception &ex) { e->error = std::string("export_graphslam: ") + ex.what(); return -1; }
	catch (...) { e->error = "export_graphslam: unknown exception"; return -1; } }
double srba_engine_profiler_mean(void *h, const char *name) { return static_cast<EngineBase *>(h)->profiler_mean(name); }

int64_t srba_engine_harvest_count(void *h) { return (int64_t)static_cast<EngineBase *>(h)->harvest.data.size(); }
/* Array of capsule views over the harvested (pre-optimisation) problems; valid until the next add_keyframe / clear. */
srba_problem_capsule *srba_engine_harvest_capsules(void *h) {
	Harvest &hv = static_cast<EngineBase *>(h)->harvest;
	if (hv.dirty) { hv.views.clear(); for (size_t i = 0; i < hv.data.size(); i++) hv.views.push_back(hv.data[i].view()); hv.dirty = false; }
	return hv.views.empty() ? NULL : &hv.views[0];
}
uint64_t srba_engine_harvest_kf(void *h, int64_t i) { return static_cast<EngineBase *>(h)->harvest.kf_of[i]; }
void srba_engine_harvest_clear(void *h) { Harvest &hv = static_cast<EngineBase *>(h)->harvest; hv.data.clear(); hv.views.clear(); hv.kf_of.clear(); hv.dirty = true; }
int srba_engine_get_hip_params(void *h, srba_hip_params *out) { *out = static_cast<EngineBase *>(h)->last_params; return 0; }
/* Serialise / load harvested capsules (golden fixtures, capsule caches). */
int srba_engine_harvest_save(void *h, const char *path, int64_t first, int64_t count) {
	EngineBase *e = static_cast<EngineBase *>(h); FILE *f = fopen(path, "wb"); if (!f) return -1;
	const int64_t n = std::min<int64_t>(count, (int64_t)e->harvest.data.size() - first);
	const uint64_t magic = 0x53524241434150ULL; fwrite(&magic, 8, 1, f); fwrite(&n, 8, 1, f); fwrite(&e->last_params, sizeof(srba_hip_params), 1, f);
	for (int64_t i = 0; i < n; i++) e->harvest.data[first + i].write(f);
	fclose(f); return 0;
}

/* Stand-alone capsule file reader (no engine needed). */
struct CapsuleFile { std::deque<CapsuleData> data; std::vector<srba_problem_capsule> views; srba_hip_params params; };
void *srba_capsule_file_load(const char *path) {
	FILE *f = fopen(path, "rb"); if (!f) { g_error = std::string("cannot open ") + path; return NULL; }
	CapsuleFile *cf = new CapsuleFile();
	try {
		uint64_t magic = 0; int64_t n = 0;
		if (fread(&magic, 8, 1, f) != 1 || magic != 0x53524241434150ULL || fread(&n, 8, 1, f) != 1 || fread(&cf->params, sizeof(srba_hip_params), 1,
			f) != 1) throw std::runtime_error("bad capsule file header");
		for (int64_t i = 0; i < n; i++) { cf->data.push_back(CapsuleData()); cf->data.back().read(f); }
		for (size_t i = 0; i < cf->data.size(); i++) cf->views.push_back(cf->data[i].view());
	} catch (std::exception &e) { g_error = e.what(); delete cf; fclose(f); return NULL; }
	fclose(f); return cf;
}
int64_t srba_capsule_file_count(void *h) { return (int64_t)static_cast<CapsuleFile *>(h)->views.size(); }
srba_problem_capsule *srba_capsule_file_capsules(void *h) { CapsuleFile *cf = static_cast<CapsuleFile *>(h); return cf->views.empty() ? NULL : &cf->views[0]; }
int srba_capsule_file_params(void *h, srba_hip_params *out) { *out = static_cast<CapsuleFile *>(h)->params; return 0; }
void srba_capsule_file_free(void *h) { delete static_cast<CapsuleFile *>(h); }
/* A capsule whose Jacobian structure is given directly (no graph): observation `r` has an optional dh_dAp block in column bp_col and a dh_df block in column
 * bf_col (-1 = none); every pose is the identity. The Hessian / Schur plan comes from the product's CapsuleData::build_plan. Used to replay the reference's
 * SchurTests, whose Jacobians are filled by hand (tests/schur_unittest.cpp:96-139). Rows must be listed in ascending order per column: callers pass rows 0..n-1. */
void *srba_capsule_from_blocks(int family, int nK, int nF, int n_obs, const int32_t *row_bp_col, const int32_t *row_bf_col, int with_schur) {
	int P, L, O, PD; if (srba_family_dims(family, &P, &L, &O, &PD) != 0 || nK < 0 || nF < 0 || n_obs <= 0) return NULL;
	CapsuleFile *cf = new CapsuleFile(); std::memset(&cf->params, 0, sizeof(cf->params)); cf->params.family = family;
	cf->data.push_back(CapsuleData()); CapsuleData &d = cf->data.back();
	d.P = P; d.L = L; d.O = O; d.PD = PD; d.n_unk_edges = nK; d.n_unk_lms = nF; d.n_valid = n_obs;
	d.edge_pose.assign((size_t)nK * PD, 0.0); if (PD == 12) for (int e = 0; e < nK; e++) for (int k = 0; k < 3; k++) d.edge_pose[(size_t)e * 12 + 3 + 4 * k] = 1.0;
	d.ulm_pos.assign((size_t)nF * L, 0.0); d.obs_z.assign((size_t)n_obs * O, 0.0); d.pair_path_off.assign(1, 0);
	std::vector<uint64_t> bp_row, bf_row;
	for (int r = 0; r < n_obs; r++) { d.obs_pose.push_back(-1); d.obs_lm.push_back(row_bf_col[r] >= 0 ? row_bf_col[r] : 0); d.obs_valid.push_back(r); }
	d.colp_off.assign(1, 0);
	for (int c = 0; c < nK; c++) { for (int r = 0; r < n_obs; r++) if (row_bp_col[r] == c) { d.bp_col.push_back(c); d.bp_res.push_back(r); d.bp_A.push_back(-1); d.bp_D.push_back(-1);
		d.bp_lm.push_back(d.obs_lm[r]); d.bp_normal.push_back(1); bp_row.push_back(r); } d.colp_off.push_back((int32_t)d.bp_col.size()); }
	d.colf_off.assign(1, 0);
	for (int c = 0; c < nF; c++) { for (int r = 0; r < n_obs; r++) if (row_bf_col[r] == c) { d.bf_col.push_back(c); d.bf_res.push_back(r); d.bf_pose.push_back(-1); bf_row.push_back(r); }
		d.colf_off.push_back((int32_t)d.bf_col.size()); }
	d.build_plan(bp_row, bf_row, with_schur != 0);
	cf->views.push_back(d.view());
	return cf;
}

/* Deep copy of a capsule array (so that a pristine batch can be re-optimised by several back-ends). */
void *srba_capsule_clone(const srba_problem_capsule *caps, int64_t n, int family) {
	int P, L, O, PD; if (srba_family_dims(family, &P, &L, &O, &PD) != 0) return NULL;
	CapsuleFile *cf = new CapsuleFile(); std::memset(&cf->params, 0, sizeof(cf->params)); cf->params.family = family;
	for (int64_t i = 0; i < n; i++) {
		const srba_problem_capsule &c = caps[i]; cf->data.push_back(CapsuleData()); CapsuleData &d = cf->data.back();
		d.P = P; d.L = L; d.O = O; d.PD = PD; d.n_unk_edges = c.n_unk_edges; d.n_unk_lms = c.n_unk_lms; d.n_valid = c.n_valid;
#define CP(field, ptr, count) d.field.assign(ptr, ptr + (size_t)(count))
		CP(edge_pose, c.edge_pose, (size_t)c.n_edges * PD); CP(ulm_pos, c.ulm_pos, (size_t)c.n_unk_lms * L); CP(klm_pos, c.klm_pos, (size_t)c.n_known_lms * L); CP(obs_z, c.obs_z, (size_t)c.n_obs * O);
		CP(pair_path_off, c.pair_path_off, c.n_pairs + 1); CP(path_edge, c.path_edge, c.n_path); CP(pair_needed, c.pair_needed, c.n_pairs); CP(pose_required, c.pose_required, 2 * c.n_pairs);
		CP(obs_pose, c.obs_pose, c.n_obs); CP(obs_lm, c.obs_lm, c.n_obs); CP(obs_valid, c.obs_valid, c.n_obs);
		CP(bp_col, c.bp_col, c.n_bp); CP(bp_res, c.bp_res, c.n_bp); CP(bp_A, c.bp_A, c.n_bp); CP(bp_D, c.bp_D, c.n_bp); CP(bp_lm, c.bp_lm, c.n_bp); CP(bp_normal, c.bp_normal, c.n_bp); CP(colp_off,
			c.colp_off, c.n_unk_edges + 1);
		CP(bf_col, c.bf_col, c.n_bf); CP(bf_res, c.bf_res, c.n_bf); CP(bf_pose, c.bf_pose, c.n_bf); CP(colf_off, c.colf_off, c.n_unk_lms + 1);
		CP(hap_i, c.hap_i, c.n_hap); CP(hap_j, c.hap_j, c.n_hap); CP(hap_term_off, c.hap_term_off, c.n_hap + 1); CP(hap_t1, c.hap_t1, c.n_hap_terms); CP(hap_t2, c.hap_t2, c.n_hap_terms);
		CP(hf_i, c.hf_i, c.n_hf); CP(hf_j, c.hf_j, c.n_hf); CP(hf_term_off, c.hf_term_off, c.n_hf + 1); CP(hf_t1, c.hf_t1, c.n_hf_terms); CP(hf_t2, c.hf_t2, c.n_hf_terms);
		CP(hapf_i, c.hapf_i, c.n_hapf); CP(hapf_j, c.hapf_j, c.n_hapf); CP(hapf_term_off, c.hapf_term_off, c.n_hapf + 1); CP(hapf_t1, c.hapf_t1, c.n_hapf_terms); CP(hapf_t2, c.hapf_t2,
			c.n_hapf_terms);
		CP(hap_diag, c.hap_diag, c.n_unk_edges); CP(hf_diag, c.hf_diag, c.n_unk_lms);
		if (c.sch_term_off) { CP(sch_term_off, c.sch_term_off, c.n_hap + 1); CP(sch_b1, c.sch_b1, c.n_sch_terms); CP(sch_b2, c.sch_b2, c.n_sch_terms); CP(sch_lm, c.sch_lm, c.n_sch_terms); }
		CP(lm_hapf_off, c.lm_hapf_off, c.n_unk_lms + 1); CP(lm_hapf_idx, c.lm_hapf_idx, c.n_hapf);
#undef CP
	}
	for (size_t i = 0; i < cf->data.size(); i++) cf->views.push_back(cf->data[i].view());
	return cf;
}

} // extern "C"
