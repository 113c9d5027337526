/*
 * RbaEngine.h -- source-compatible front-end of srba::RbaEngine<KF2KF,LM,OBS,RBA_OPTIONS> whose numeric optimiser
 * runs on an MI355X through the C ABI of include/srba_hip.h.
 *
 * Public surface kept from the reference (include/srba/RbaEngine.h): nested typedefs :72-118, TOptimizeExtraOutputInfo
 * :125-177, TNewKeyFrameInfo :180-195, define_new_keyframe :207-211, TOptimizeLocalAreaParams :214-227,
 * optimize_local_area :235-241, create_kf2kf_edge :343-347, get_kf_relative_pose :385-396, bfs_visitor :408-415,
 * TSRBAParameters :424-460 (defaults impl/rba_problem_common.h:35-56), TAllParameters :463-473, getters :483-495.
 *
 * Host-side (integer / bookkeeping) algorithms restated here, with identical iteration orders so that unknown
 * numbering and spanning-tree tables come out the same:
 *   alloc_keyframe / alloc_kf2kf_edge / create_kf2kf_edge      impl/alloc_keyframe.h:19, impl/alloc_kf2kf_edge.h:17-61, impl/create_kf2kf_edge.h:15-37
 *   TSpanningTree::update_symbolic_new_node + find_path_bfs     impl/spantree_update_symbolic.h:19-211, :227-300
 *   add_observation                                             impl/add-observations.h:17-264
 *   determine_kf2kf_edges_to_create                             impl/determine_kf2kf_edges_to_create.h:17-266
 *   define_new_keyframe                                         impl/define_new_keyframe.h:16-116
 *   optimize_local_area + bfs_visitor + VisitorOptimizeLocalArea impl/optimize_local_area.h:15-58, impl/bfs_visitor.h:21-177, RbaEngine.h:543-618
 *   optimize_edges S1-S4 (unknown filtering, involved_obs, ST roots)  impl/optimize_edges.h:71-246, impl/jacobians.h:1020-1076
 * optimize_edges S5-S17 (numeric) is NOT done here: the call is flattened into a srba_problem_capsule and handed to the
 * numeric back-end (srba::hip_backend, GPU).  There is no CPU fallback: without the HIP library/GPU the call throws.
 */
#pragma once
#include "capsule.h"
#include "srba_options.h"
#include <cstring>
#include <functional>
#include <limits>
#include <memory>
#include <queue>

namespace srba {

// -------------------------------------------------------------------------------------------------
// Numeric back-end seam (replaces the reference's compile-time internal::solver_engine<> + the CPU loops)
// -------------------------------------------------------------------------------------------------
struct numeric_backend {
	virtual ~numeric_backend() {}
	/** Optimise the capsule in place (unknowns, ST poses, landmark information matrices) and fill the result. Throws on failure. */
	virtual void run(const srba_hip_params &params, srba_problem_capsule &capsule, srba_lm_result &result) = 0;
	virtual const char *name() const = 0;
	/** Optional: where to record the back-end's own stage timings ("opt.backend.*"). */
	virtual void set_profiler(mrpt::utils::CTimeLogger *) {}
	/** Whole-map squared error over prepared path lists (eval_overall_squared_error). */
	virtual double eval_overall(const srba_hip_params &, const srba_overall_problem &) { throw std::runtime_error(std::string("numeric back-end '") + name() + "' does not implement eval_overall"); }
};
/** Adapter over a plain C function (used by tests to plug the CPU oracle in from outside the product). */
struct function_backend : public numeric_backend {
	typedef int (*fn_t)(const srba_hip_params *, srba_problem_capsule *, srba_lm_result *);
	fn_t fn; std::string nm;
	function_backend(fn_t f, const std::string &n) : fn(f), nm(n) {}
	void run(const srba_hip_params &p, srba_problem_capsule &c, srba_lm_result &r) { if (fn(&p, &c, &r) != 0) throw std::runtime_error("numeric back-end '" + nm + "' failed"); }
	const char *name() const { return nm.c_str(); }
	typedef int (*overall_fn_t)(const srba_hip_params *, const srba_overall_problem *, double *);
	overall_fn_t overall_fn = NULL;
	double eval_overall(const srba_hip_params &p, const srba_overall_problem &q) { double v = 0; if (!overall_fn || overall_fn(&p, &q, &v) != 0) throw std::runtime_error("numeric back-end '" + nm + "': eval_overall failed or not provided"); return v; }
};
std::shared_ptr<numeric_backend> make_hip_backend(int device); // srba/hip_backend.h

/** map_as_vector-like container indexed by keyframe id (reference: mrpt::utils::map_as_vector). */
template <class V> struct kf_indexed {
	std::deque<V> data; std::deque<char> present;
	V &operator[](size_t k) { if (k >= data.size()) { data.resize(k + 1); present.resize(k + 1, 0); } present[k] = 1; return data[k]; }
	V *find(size_t k) { return (k < data.size() && present[k]) ? &data[k] : (V *)0; }
	const V *find(size_t k) const { return (k < data.size() && present[k]) ? &data[k] : (const V *)0; }
	size_t size() const { return data.size(); }
	void clear() { data.clear(); present.clear(); }
};

namespace ecps { struct local_areas_fixed_size; struct classic_linear_rba; }

/** Default RBA_OPTIONS (reference RbaEngine.h:39-45) */
struct RBA_OPTIONS_DEFAULT;

// -------------------------------------------------------------------------------------------------
// Problem state (reference TRBA_Problem_state, srba_types.h:548-785)
// -------------------------------------------------------------------------------------------------
template <class kf2kf_pose_t, class landmark_t, class obs_t, class RBA_OPTIONS>
struct TRBA_Problem_state {
	typedef typename kf2kf_pose_t::pose_t pose_t;
	typedef typename kf2kf_pose_traits<kf2kf_pose_t>::k2k_edge_t k2k_edge_t;
	typedef typename kf2kf_pose_traits<kf2kf_pose_t>::frameid2pose_map_t frameid2pose_map_t;
	typedef typename kf2kf_pose_traits<kf2kf_pose_t>::pose_flag_t pose_flag_t;
	typedef typename landmark_traits<landmark_t>::TRelativeLandmarkPosMap TRelativeLandmarkPosMap;
	typedef typename landmark_traits<landmark_t>::TRelativeLandmarkPos TRelativeLandmarkPos;
	typedef typename landmark_traits<landmark_t>::TLandmarkEntry TLandmarkEntry;
	typedef rba_joint_parameterization_traits_t<kf2kf_pose_t, landmark_t, obs_t> traits_t;
	typedef typename traits_t::keyframe_info keyframe_info;
	typedef typename traits_t::k2f_edge_t k2f_edge_t;
	typedef std::deque<k2k_edge_t> k2k_edges_deque_t;
	typedef std::deque<k2f_edge_t> all_observations_deque_t;
	typedef std::deque<keyframe_info> keyframe_vector_t;
	typedef std::vector<size_t> k2k_edge_path_t; //!< a path as a list of edge ids
	typedef mrpt::math::CMatrixFixed<landmark_t::LM_DIMS, landmark_t::LM_DIMS> lm_inf_matrix_t;

	struct TSpanningTree {
		typedef kf_indexed<std::map<TKeyFrameID, TSpanTreeEntry> > next_edge_maps_t;
		typedef kf_indexed<std::map<TKeyFrameID, k2k_edge_path_t> > all_edges_maps_t;
		const TRBA_Problem_state *m_parent;
		struct TSpanningTreeSym {
			next_edge_maps_t next_edge; //!< [SOURCE][TARGET] -> next node + distance (both directions stored)
			all_edges_maps_t all_edges; //!< [i][j], i>j -> edges of the shortest path
		} sym;
		kf_indexed<frameid2pose_map_t> num; //!< num[SOURCE][TARGET] = pose of TARGET as seen from SOURCE (filled from the back-end's results)
		void clear() { sym.next_edge.clear(); sym.all_edges.clear(); num.clear(); }

		/** Incremental update after inserting ONE edge touching new_node_id (impl/spantree_update_symbolic.h:19-211) */
		void update_symbolic_new_node(const TKeyFrameID new_node_id, const TPairKeyFrameID &new_edge, const topo_dist_t max_depth) {
			ASSERT_(max_depth >= 1);
			std::set<TPairKeyFrameID> kfs_with_modified_next_edge;
			const TKeyFrameID ik = getTheOtherFromPair(new_node_id, new_edge);
			// tk = all nodes within distance <= max_depth-1 of ik, + ik itself  (:39-44)
			std::vector<std::pair<TKeyFrameID, topo_dist_t> > tk;
			{
				const std::map<TKeyFrameID, TSpanTreeEntry> &st_ik = sym.next_edge[ik];
				for (std::map<TKeyFrameID, TSpanTreeEntry>::const_iterator it = st_ik.begin(); it != st_ik.end(); ++it)
					if (it->second.distance < max_depth) tk.push_back(std::make_pair(it->first, it->second.distance));
				tk.push_back(std::make_pair(ik, (topo_dist_t)0));
			}
			// STn = all nodes currently in the ST of new_node_id, + itself (:48-52); entries are read "live"
			std::vector<TKeyFrameID> STn;
			{
				const std::map<TKeyFrameID, TSpanTreeEntry> &st_n = sym.next_edge[new_node_id];
				for (std::map<TKeyFrameID, TSpanTreeEntry>::const_iterator it = st_n.begin(); it != st_n.end(); ++it) STn.push_back(it->first);
				STn.push_back(new_node_id);
			}
			for (size_t r_idx = 0; r_idx < STn.size(); r_idx++) {
				const TKeyFrameID r = STn[r_idx];
				const topo_dist_t dist_r2n = (r == new_node_id) ? 0 : sym.next_edge[new_node_id][r].distance; // ste_n2r->distance (:58-59)
				std::map<TKeyFrameID, TSpanTreeEntry> &st_r = sym.next_edge[r];
				TSpanTreeEntry *ste_r2n = NULL;
				if (r != new_node_id) { std::map<TKeyFrameID, TSpanTreeEntry>::iterator it = st_r.find(new_node_id); ASSERT_(it != st_r.end()); ste_r2n = &it->second; }
				for (size_t s_idx = 0; s_idx < tk.size(); s_idx++) {
					const TKeyFrameID s = tk[s_idx].first; if (r == s) continue;
					const topo_dist_t dist_s2ik = tk[s_idx].second;
					std::map<TKeyFrameID, TSpanTreeEntry> &st_s = sym.next_edge[s];
					TSpanTreeEntry *ste_s2ik = NULL;
					if (s != ik) { std::map<TKeyFrameID, TSpanTreeEntry>::iterator it2 = st_s.find(ik); ASSERT_(it2 != st_s.end()); ste_s2ik = &it2->second; }
					const topo_dist_t new_dist = dist_r2n + dist_s2ik + 1;
					std::map<TKeyFrameID, TSpanTreeEntry>::iterator it_s_inSTr = st_r.find(s);
					if (it_s_inSTr != st_r.end()) {
						if (new_dist < it_s_inSTr->second.distance) { // strictly shorter (:98)
							it_s_inSTr->second.distance = new_dist; it_s_inSTr->second.next = ste_r2n ? ste_r2n->next : ik;
							TSpanTreeEntry &ste_r_inSTs = st_s[r]; ste_r_inSTs.distance = new_dist; ste_r_inSTs.next = ste_s2ik ? ste_s2ik->next : new_node_id;
							kfs_with_modified_next_edge.insert(std::make_pair(s, r)); kfs_with_modified_next_edge.insert(std::make_pair(r, s));
						}
					} else if (new_dist <= max_depth) { // newly reachable (:124-147)
						TSpanTreeEntry &ste_s_inSTr = st_r[s]; ste_s_inSTr.distance = new_dist; ste_s_inSTr.next = ste_r2n ? ste_r2n->next : ik;
						TSpanTreeEntry &ste_r_inSTs = st_s[r]; ste_r_inSTs.distance = new_dist; ste_r_inSTs.next = ste_s2ik ? ste_s2ik->next : new_node_id;
						kfs_with_modified_next_edge.insert(std::make_pair(r, s)); kfs_with_modified_next_edge.insert(std::make_pair(s, r));
					}
				}
			}
			// rebuild all_edges of the modified pairs by BFS over the whole graph (:168-190)
			for (std::set<TPairKeyFrameID>::const_iterator it = kfs_with_modified_next_edge.begin(); it != kfs_with_modified_next_edge.end(); ++it) {
				const TKeyFrameID from = std::max(it->first, it->second), to = std::min(it->first, it->second);
				k2k_edge_path_t &path = sym.all_edges[from][to];
				path.clear();
				const bool found = m_parent->find_path_bfs(from, to, NULL, &path);
				ASSERT_(found);
			}
		}
	};

	/** Symbolic Jacobian structure: one column per kf2kf edge / per unknown landmark (reference TLinearSystem :682-698) */
	struct TLinearSystem {
		std::deque<std::vector<TJacobianSymbolicInfo_dh_dAp> > dh_dAp; //!< [edge id] -> blocks, ascending obs_idx
		std::deque<std::vector<TJacobianSymbolicInfo_dh_df> > dh_df;   //!< [column] -> blocks
		kf_indexed<size_t> dh_df_remap;                                //!< landmark id -> column of dh_df
		void clear() { dh_dAp.clear(); dh_df.clear(); dh_df_remap.clear(); }
	};

	keyframe_vector_t keyframes;
	k2k_edges_deque_t k2k_edges;
	TRelativeLandmarkPosMap unknown_lms, known_lms;
	std::map<TLandmarkID, lm_inf_matrix_t> unknown_lms_inf_matrices;
	std::deque<TLandmarkEntry> all_lms;
	TSpanningTree spanning_tree;
	all_observations_deque_t all_observations;
	TLinearSystem lin_system;
	std::deque<char> all_observations_Jacob_validity;
	std::set<size_t> last_timestep_touched_kfs;

	TRBA_Problem_state() { spanning_tree.m_parent = this; }
	void clear() {
		keyframes.clear(); k2k_edges.clear(); unknown_lms.clear(); unknown_lms_inf_matrices.clear(); known_lms.clear(); all_lms.clear();
		spanning_tree.clear(); all_observations.clear(); lin_system.clear(); all_observations_Jacob_validity.clear(); last_timestep_touched_kfs.clear();
	}

	/** Unbounded BFS over the KF graph, first-found predecessors (impl/spantree_update_symbolic.h:227-300) */
	bool find_path_bfs(const TKeyFrameID cur_node, const TKeyFrameID trg_node, std::vector<TKeyFrameID> *out_path_IDs, k2k_edge_path_t *out_path_edges = NULL) const {
		if (out_path_IDs) out_path_IDs->clear();
		if (out_path_edges) out_path_edges->clear();
		if (cur_node == trg_node) return true;
		struct TBFSEntry { TKeyFrameID prev; size_t prev_edge; topo_dist_t dist; TBFSEntry() : prev(0), prev_edge(SRBA_INVALID_INDEX), dist(std::numeric_limits<topo_dist_t>::max()) {} };
		std::set<TKeyFrameID> visited; std::queue<TKeyFrameID> pending; std::map<TKeyFrameID, TBFSEntry> preceding;
		pending.push(cur_node); visited.insert(cur_node); preceding[cur_node].dist = 0;
		while (!pending.empty()) {
			const TKeyFrameID next_kf = pending.front(); pending.pop();
			const topo_dist_t cur_dist = preceding[next_kf].dist;
			if (next_kf == trg_node) {
				topo_dist_t dist = cur_dist, dist2 = cur_dist;
				if (out_path_IDs) out_path_IDs->resize(dist);
				if (out_path_edges) out_path_edges->resize(dist);
				TKeyFrameID path_node = trg_node;
				while (path_node != cur_node) {
					if (out_path_IDs) (*out_path_IDs)[--dist] = path_node;
					const TBFSEntry &e = preceding.find(path_node)->second;
					path_node = e.prev;
					if (out_path_edges) (*out_path_edges)[--dist2] = e.prev_edge;
				}
				return true;
			}
			const keyframe_info &kfi = keyframes[next_kf];
			for (size_t i = 0; i < kfi.adjacent_k2k_edges.size(); i++) {
				const k2k_edge_t *ed = kfi.adjacent_k2k_edges[i];
				const TKeyFrameID new_kf = getTheOtherFromPair2(next_kf, *ed);
				if (!visited.count(new_kf)) {
					pending.push(new_kf); visited.insert(new_kf);
					TBFSEntry &p = preceding[new_kf];
					if (p.dist > cur_dist + 1) { p.dist = cur_dist + 1; p.prev = next_kf; p.prev_edge = ed->id; }
				}
			}
		}
		return false;
	}

	/** impl/alloc_kf2kf_edge.h:17-61 */
	size_t alloc_kf2kf_edge(const TPairKeyFrameID &ids, const pose_t &init_inv_pose_val = pose_t()) {
		k2k_edges.push_back(k2k_edge_t());
		k2k_edge_t &e = k2k_edges.back();
		e.from = ids.first; e.to = ids.second; ASSERT_(e.from != e.to);
		e.inv_pose = init_inv_pose_val; e.id = k2k_edges.size() - 1;
		keyframes[ids.first].adjacent_k2k_edges.push_back(&e); keyframes[ids.second].adjacent_k2k_edges.push_back(&e);
		lin_system.dh_dAp.push_back(std::vector<TJacobianSymbolicInfo_dh_dAp>());
		return e.id;
	}
	bool are_keyframes_connected(const TKeyFrameID id1, const TKeyFrameID id2) const {
		const std::deque<k2k_edge_t *> &adj = keyframes[id1].adjacent_k2k_edges;
		for (size_t i = 0; i < adj.size(); i++) if (id2 == getTheOtherFromPair2(id1, *adj[i])) return true;
		return false;
	}
private:
	TRBA_Problem_state(const TRBA_Problem_state &); TRBA_Problem_state &operator=(const TRBA_Problem_state &);
};

namespace internal {
/** impl/make_ordered_list_base_kfs.h:16-45 */
template <class traits_t, class rba_problem_state_t>
void make_ordered_list_base_kfs(const typename traits_t::new_kf_observations_t &obs, const rba_problem_state_t &rba_state, base_sorted_lst_t &obs_for_each_base_sorted, std::map<TKeyFrameID, size_t> *out_obs_for_each_base = NULL) {
	std::map<TKeyFrameID, size_t> obs_for_each_base;
	for (typename traits_t::new_kf_observations_t::const_iterator itObs = obs.begin(); itObs != obs.end(); ++itObs) {
		const TLandmarkID lm_id = itObs->obs.feat_id;
		if (lm_id >= rba_state.all_lms.size()) continue;
		if (!rba_state.all_lms[lm_id].rfp) continue;
		obs_for_each_base[rba_state.all_lms[lm_id].rfp->id_frame_base]++;
	}
	for (std::map<TKeyFrameID, size_t>::const_iterator it = obs_for_each_base.begin(); it != obs_for_each_base.end(); ++it) obs_for_each_base_sorted.insert(std::make_pair(it->second, it->first));
	if (out_obs_for_each_base) out_obs_for_each_base->swap(obs_for_each_base);
}
} // namespace internal

namespace ecps {
/** Edge creation policy: fixed-size sub-maps (reference ecps/local_areas_fixed_size.h:22-213) */
struct local_areas_fixed_size {
	struct parameters_t { size_t submap_size, min_obs_to_loop_closure; parameters_t() : submap_size(15), min_obs_to_loop_closure(4) {} };
	TKeyFrameID get_center_kf_for_kf(const TKeyFrameID kf_id, const parameters_t &params) const { return params.submap_size * (kf_id / params.submap_size); }

	template <class traits_t, class rba_engine_t>
	void eval(const TKeyFrameID new_kf_id, const typename traits_t::new_kf_observations_t &obs, std::vector<TNewEdgeInfo> &new_k2k_edge_ids, rba_engine_t &rba_engine, const parameters_t &params) {
		using namespace std;
		ASSERT_(new_kf_id >= 1);
		const size_t MINIMUM_OBS_TO_LOOP_CLOSURE = params.min_obs_to_loop_closure;
		const TKeyFrameID current_center_kf_id = get_center_kf_for_kf(new_kf_id, params);
		const topo_dist_t min_dist_for_loop_closure = rba_engine.parameters.srba.max_tree_depth + 1;
		base_sorted_lst_t obs_for_each_base_sorted;
		srba::internal::make_ordered_list_base_kfs<traits_t, typename rba_engine_t::rba_problem_state_t>(obs, rba_engine.get_rba_state(), obs_for_each_base_sorted);
		map<TKeyFrameID, size_t> obs_for_each_area; map<TKeyFrameID, bool> base_is_center_for_all_obs_in_area; map<TKeyFrameID, map<TKeyFrameID, size_t> > obs_for_base_KF_grouped_by_area;
		for (base_sorted_lst_t::const_iterator it = obs_for_each_base_sorted.begin(); it != obs_for_each_base_sorted.end(); ++it) {
			const size_t num_obs_this_base = it->first; const TKeyFrameID base_id = it->second;
			const TKeyFrameID this_localmap_center = get_center_kf_for_kf(base_id, params);
			obs_for_each_area[this_localmap_center] += num_obs_this_base;
			obs_for_base_KF_grouped_by_area[this_localmap_center][base_id] += num_obs_this_base;
			if (base_is_center_for_all_obs_in_area.find(this_localmap_center) == base_is_center_for_all_obs_in_area.end()) base_is_center_for_all_obs_in_area[this_localmap_center] = true;
			if (base_id != this_localmap_center) base_is_center_for_all_obs_in_area[this_localmap_center] = false;
		}
		base_sorted_lst_t obs_for_each_area_sorted;
		for (map<TKeyFrameID, size_t>::const_iterator it = obs_for_each_area.begin(); it != obs_for_each_area.end(); ++it) obs_for_each_area_sorted.insert(make_pair(it->second, it->first));
		map<TKeyFrameID, base_sorted_lst_t> obs_for_base_KF_grouped_by_area_sorted;
		for (map<TKeyFrameID, map<TKeyFrameID, size_t> >::const_iterator it = obs_for_base_KF_grouped_by_area.begin(); it != obs_for_base_KF_grouped_by_area.end(); ++it) {
			base_sorted_lst_t &bsl = obs_for_base_KF_grouped_by_area_sorted[it->first];
			for (map<TKeyFrameID, size_t>::const_iterator it2 = it->second.begin(); it2 != it->second.end(); ++it2) bsl.insert(make_pair(it2->second, it2->first));
		}
		// always one edge: new KF <- its area centre, unless it IS a new centre (:114-131)
		if (current_center_kf_id != new_kf_id) {
			TNewEdgeInfo nei;
			nei.id = rba_engine.create_kf2kf_edge(new_kf_id, TPairKeyFrameID(current_center_kf_id, new_kf_id), obs);
			nei.has_approx_init_val = false;
			new_k2k_edge_ids.push_back(nei);
		}
		// loop closures towards other areas (:134-198)
		for (base_sorted_lst_t::const_iterator it = obs_for_each_area_sorted.begin(); it != obs_for_each_area_sorted.end(); ++it) {
			const size_t num_obs_this_base = it->first; const TKeyFrameID remote_center_kf_id = it->second;
			const TKeyFrameID from_id = current_center_kf_id, to_id = remote_center_kf_id;
			if (from_id == to_id) continue;
			topo_dist_t found_distance = numeric_limits<topo_dist_t>::max();
			const map<TKeyFrameID, TSpanTreeEntry> *from_Ds = rba_engine.get_rba_state().spanning_tree.sym.next_edge.find(from_id);
			if (from_Ds) { map<TKeyFrameID, TSpanTreeEntry>::const_iterator it_to = from_Ds->find(to_id); if (it_to != from_Ds->end()) found_distance = it_to->second.distance; }
			topo_dist_t dist_extra_edges = 2;
			if (current_center_kf_id == new_kf_id) dist_extra_edges--;
			if (base_is_center_for_all_obs_in_area[remote_center_kf_id]) dist_extra_edges--;
			if (found_distance >= min_dist_for_loop_closure - dist_extra_edges) {
				if (num_obs_this_base >= MINIMUM_OBS_TO_LOOP_CLOSURE) {
					TNewEdgeInfo nei;
					nei.id = rba_engine.create_kf2kf_edge(from_id, TPairKeyFrameID(to_id, from_id), obs);
					nei.has_approx_init_val = false;
					nei.loopclosure_observer_kf = new_kf_id;
					const base_sorted_lst_t &bsl = obs_for_base_KF_grouped_by_area_sorted[remote_center_kf_id]; ASSERT_(!bsl.empty());
					nei.loopclosure_base_kf = bsl.begin()->second;
					new_k2k_edge_ids.push_back(nei);
				}
			}
		}
		ASSERTMSG_(new_k2k_edge_ids.size() >= 1, "Error for new KF: no suitable linking KF found with the minimum number of common observations: the node becomes isolated of the graph!");
	}
};

/** Edge creation policy: the classic linear graph -- always an edge (n-1) -> n, plus loop-closure edges towards the base key-frames of
  * re-observed landmarks that are farther than max_tree_depth (reference ecps/classic_linear_rba.h:22-118) */
struct classic_linear_rba {
	struct parameters_t { size_t min_obs_to_loop_closure; parameters_t() : min_obs_to_loop_closure(4) {} };

	template <class traits_t, class rba_engine_t>
	void eval(const TKeyFrameID new_kf_id, const typename traits_t::new_kf_observations_t &obs, std::vector<TNewEdgeInfo> &new_k2k_edge_ids, rba_engine_t &rba_engine, const parameters_t &params) {
		using namespace std;
		ASSERT_(new_kf_id >= 1);
		// (1/2) always an edge (n-1) => (n), initialised at the null pose: each key-frame starts at the pose of the previous one (:61-69)
		const typename traits_t::original_kf2kf_pose_t::pose_t init_inv_pose;
		TNewEdgeInfo nei1; nei1.has_approx_init_val = true;
		nei1.id = rba_engine.create_kf2kf_edge(new_kf_id, TPairKeyFrameID(new_kf_id - 1, new_kf_id), obs, init_inv_pose);
		new_k2k_edge_ids.push_back(nei1);
		// (2/2) loop closures (:71-114)
		const topo_dist_t min_dist_for_loop_closure = rba_engine.parameters.srba.max_tree_depth + 1;
		base_sorted_lst_t obs_for_each_base_sorted;
		srba::internal::make_ordered_list_base_kfs<traits_t, typename rba_engine_t::rba_problem_state_t>(obs, rba_engine.get_rba_state(), obs_for_each_base_sorted);
		for (base_sorted_lst_t::const_iterator it = obs_for_each_base_sorted.begin(); it != obs_for_each_base_sorted.end(); ++it) {
			const size_t num_obs_this_base = it->first; const TKeyFrameID from_id = new_kf_id, to_id = it->second;
			topo_dist_t found_distance = numeric_limits<topo_dist_t>::max();
			const map<TKeyFrameID, TSpanTreeEntry> *from_Ds = rba_engine.get_rba_state().spanning_tree.sym.next_edge.find(from_id);
			if (from_Ds) { map<TKeyFrameID, TSpanTreeEntry>::const_iterator it_to = from_Ds->find(to_id); if (it_to != from_Ds->end()) found_distance = it_to->second.distance; }
			if (found_distance >= min_dist_for_loop_closure && num_obs_this_base >= params.min_obs_to_loop_closure) {
				TNewEdgeInfo nei;
				nei.id = rba_engine.create_kf2kf_edge(new_kf_id, TPairKeyFrameID(to_id, new_kf_id), obs);
				nei.has_approx_init_val = false;
				new_k2k_edge_ids.push_back(nei);
			}
		}
	}
};
} // namespace ecps

struct RBA_OPTIONS_DEFAULT {
	typedef ecps::local_areas_fixed_size edge_creation_policy_t;
	typedef options::sensor_pose_on_robot_none sensor_pose_on_robot_t;
	typedef options::observation_noise_identity obs_noise_matrix_t;
	typedef options::solver_LM_schur_dense_cholesky solver_t;
};

// -------------------------------------------------------------------------------------------------
// The engine
// -------------------------------------------------------------------------------------------------
template <class KF2KF_POSE_TYPE, class LM_TYPE, class OBS_TYPE, class RBA_OPTIONS = RBA_OPTIONS_DEFAULT>
class RbaEngine {
public:
	typedef RbaEngine<KF2KF_POSE_TYPE, LM_TYPE, OBS_TYPE, RBA_OPTIONS> rba_engine_t;
	typedef KF2KF_POSE_TYPE kf2kf_pose_t; typedef LM_TYPE landmark_t; typedef OBS_TYPE obs_t; typedef RBA_OPTIONS rba_options_t;
	static const size_t REL_POSE_DIMS = kf2kf_pose_t::REL_POSE_DIMS, LM_DIMS = landmark_t::LM_DIMS, OBS_DIMS = obs_t::OBS_DIMS;
	typedef typename kf2kf_pose_t::se_traits_t se_traits_t;
	typedef rba_joint_parameterization_traits_t<kf2kf_pose_t, landmark_t, obs_t> traits_t;
	typedef kf2kf_pose_traits<kf2kf_pose_t> kf2kf_pose_traits_t; typedef landmark_traits<landmark_t> landmark_traits_t; typedef observation_traits<obs_t> observation_traits_t;
	typedef sensor_model<landmark_t, obs_t> sensor_model_t;
	typedef typename kf2kf_pose_t::pose_t pose_t;
	typedef TRBA_Problem_state<KF2KF_POSE_TYPE, LM_TYPE, OBS_TYPE, RBA_OPTIONS> rba_problem_state_t;
	typedef typename rba_problem_state_t::k2f_edge_t k2f_edge_t; typedef typename rba_problem_state_t::k2k_edge_t k2k_edge_t;
	typedef typename rba_problem_state_t::k2k_edges_deque_t k2k_edges_deque_t;
	typedef typename kf2kf_pose_traits_t::pose_flag_t pose_flag_t; typedef typename kf2kf_pose_traits_t::frameid2pose_map_t frameid2pose_map_t;
	typedef typename landmark_traits_t::TRelativeLandmarkPosMap TRelativeLandmarkPosMap; typedef typename landmark_traits_t::TRelativeLandmarkPos TRelativeLandmarkPos;
	typedef typename traits_t::keyframe_info keyframe_info; typedef typename traits_t::new_kf_observation_t new_kf_observation_t; typedef typename traits_t::new_kf_observations_t new_kf_observations_t;
	typedef typename kf2kf_pose_traits_t::array_pose_t array_pose_t; typedef typename landmark_traits_t::array_landmark_t array_landmark_t; typedef typename observation_traits_t::array_obs_t array_obs_t;

	RbaEngine() : m_verbose_level(1), m_profiler(true), m_hip_device(-1) { clear(); }

	/** reference RbaEngine.h:125-177 */
	struct TOptimizeExtraOutputInfo {
		TOptimizeExtraOutputInfo() { clear(); }
		size_t num_observations, num_jacobians, num_kf2kf_edges_optimized, num_kf2lm_edges_optimized, num_total_scalar_optimized, num_kf_optimized, num_lm_optimized, num_span_tree_numeric_updates;
		double obs_rmse, total_sqr_error_init, total_sqr_error_final, HAp_condition_number;
		size_t sparsity_dh_dAp_nnz, sparsity_dh_dAp_max_size, sparsity_dh_df_nnz, sparsity_dh_df_max_size, sparsity_HAp_nnz, sparsity_HAp_max_size, sparsity_Hf_nnz, sparsity_Hf_max_size, sparsity_HApf_nnz, sparsity_HApf_max_size;
		std::vector<size_t> optimized_k2k_edge_indices, optimized_landmark_indices;
		typename RBA_OPTIONS::solver_t::extra_results_t extra_results;
		srba_lm_result lm; //!< (extension) raw result record of the numeric back-end: LM trials, lambda, per-trial chi2 trace
		void clear() {
			num_observations = num_jacobians = num_kf2kf_edges_optimized = num_kf2lm_edges_optimized = num_total_scalar_optimized = num_kf_optimized = num_lm_optimized = num_span_tree_numeric_updates = 0;
			obs_rmse = 0; total_sqr_error_init = total_sqr_error_final = HAp_condition_number = 0;
			sparsity_dh_dAp_nnz = sparsity_dh_dAp_max_size = sparsity_dh_df_nnz = sparsity_dh_df_max_size = sparsity_HAp_nnz = sparsity_HAp_max_size = sparsity_Hf_nnz = sparsity_Hf_max_size = sparsity_HApf_nnz = sparsity_HApf_max_size = 0;
			optimized_k2k_edge_indices.clear(); optimized_landmark_indices.clear(); extra_results.clear(); std::memset(&lm, 0, sizeof(lm));
		}
	};
	/** reference RbaEngine.h:180-195 */
	struct TNewKeyFrameInfo {
		TKeyFrameID kf_id; std::vector<TNewEdgeInfo> created_edge_ids; TOptimizeExtraOutputInfo optimize_results, optimize_results_stg1;
		void clear() { kf_id = static_cast<TKeyFrameID>(-1); created_edge_ids.clear(); optimize_results.clear(); } // (sic: stg1 not cleared, App. B-8)
	};
	struct TOptimizeLocalAreaParams {
		bool optimize_k2k_edges, optimize_landmarks; TKeyFrameID max_visitable_kf_id; size_t dont_optimize_landmarks_seen_less_than_n_times;
		TOptimizeLocalAreaParams() : optimize_k2k_edges(true), optimize_landmarks(true), max_visitable_kf_id(static_cast<TKeyFrameID>(-1)), dont_optimize_landmarks_seen_less_than_n_times(2) {}
	};
	/** reference RbaEngine.h:424-460, defaults impl/rba_problem_common.h:35-56 */
	struct TSRBAParameters {
		topo_dist_t max_tree_depth, max_optimize_depth;
		bool optimize_new_edges_alone, use_robust_kernel, use_robust_kernel_stage1;
		double kernel_param; size_t max_iters; double max_error_per_obs_to_stop, max_rho, max_lambda, min_error_reduction_ratio_to_relinearize;
		bool numeric_jacobians; void (*feedback_user_iteration)(unsigned int, const double, const double);
		bool compute_condition_number, compute_sparsity_stats; double max_rmse_show_red_warning; TCovarianceRecoveryPolicy cov_recovery;
		/** (extension, default false = reference behaviour) The reference refreshes, inside the LM loop, only the spanning-tree poses that Jacobian blocks of the
		 *  optimised columns reference (optimize_edges.h:550-566, spantree_update_numeric.h:34-35); residuals of observations whose observer-side edge is not being
		 *  optimised are then evaluated with pre-step poses. Set to true to refresh every pose a residual reads as well. */
		bool refresh_all_read_poses;
		TSRBAParameters() : max_tree_depth(4), max_optimize_depth(4), optimize_new_edges_alone(true), use_robust_kernel(false), use_robust_kernel_stage1(false), kernel_param(3.), max_iters(20),
			max_error_per_obs_to_stop(1e-6), max_rho(10.0), max_lambda(1e20), min_error_reduction_ratio_to_relinearize(0.01), numeric_jacobians(false), feedback_user_iteration(NULL),
			compute_condition_number(false), compute_sparsity_stats(false), max_rmse_show_red_warning(0.5), cov_recovery(crpLandmarksApprox), refresh_all_read_poses(false) {}
	};
	struct TAllParameters {
		TSRBAParameters srba;
		typename obs_t::TObservationParams sensor;
		typename RBA_OPTIONS::sensor_pose_on_robot_t::parameters_t sensor_pose;
		typename RBA_OPTIONS::obs_noise_matrix_t::parameters_t obs_noise;
		typename RBA_OPTIONS::edge_creation_policy_t::parameters_t ecp;
	};
	TAllParameters parameters;
	typename RBA_OPTIONS::edge_creation_policy_t edge_creation_policy;

	// ----------------------------------------------------------------------------------- main API
	/** impl/define_new_keyframe.h:16-116 */
	void define_new_keyframe(const new_kf_observations_t &obs, TNewKeyFrameInfo &out_new_kf_info, const bool run_local_optimization = true) {
		m_profiler.enter("define_new_keyframe");
		out_new_kf_info.clear();
		const TKeyFrameID new_kf_id = alloc_keyframe();
		std::vector<TNewEdgeInfo> new_k2k_edge_ids;
		m_profiler.enter("define_new_keyframe.determine_edges");
		determine_kf2kf_edges_to_create(new_kf_id, obs, new_k2k_edge_ids);
		m_profiler.leave("define_new_keyframe.determine_edges");
		m_profiler.enter("define_new_keyframe.add_observations");
		for (typename new_kf_observations_t::const_iterator it_obs = obs.begin(); it_obs != obs.end(); ++it_obs) {
			const array_landmark_t *fixed_rel_pos = it_obs->is_fixed ? &it_obs->feat_rel_pos : NULL;
			const array_landmark_t *unk_rel_pos_initval = it_obs->is_unknown_with_init_val ? &it_obs->feat_rel_pos : NULL;
			this->add_observation(new_kf_id, it_obs->obs, fixed_rel_pos, unk_rel_pos_initval);
		}
		m_profiler.leave("define_new_keyframe.add_observations");
		if (run_local_optimization) {
			if (parameters.srba.optimize_new_edges_alone && !new_k2k_edge_ids.empty()) { // stage 1 (:59-91)
				m_profiler.enter("define_new_keyframe.opt_new_edges");
				const bool old_kernel = parameters.srba.use_robust_kernel;
				parameters.srba.use_robust_kernel = parameters.srba.use_robust_kernel_stage1;
				std::vector<size_t> k2f_edges_to_opt, k2k_edges_to_opt(1);
				for (size_t i = 0; i < new_k2k_edge_ids.size(); i++) {
					if (new_k2k_edge_ids[i].has_approx_init_val) continue;
					k2k_edges_to_opt[0] = new_k2k_edge_ids[i].id;
					this->optimize_edges(k2k_edges_to_opt, k2f_edges_to_opt, out_new_kf_info.optimize_results_stg1);
				}
				parameters.srba.use_robust_kernel = old_kernel;
				m_profiler.leave("define_new_keyframe.opt_new_edges");
			}
			m_profiler.enter("define_new_keyframe.optimize");
			TOptimizeLocalAreaParams opt_params;
			this->optimize_local_area(new_kf_id, parameters.srba.max_optimize_depth, out_new_kf_info.optimize_results, opt_params);
			m_profiler.leave("define_new_keyframe.optimize");
		}
		out_new_kf_info.kf_id = new_kf_id;
		out_new_kf_info.created_edge_ids.swap(new_k2k_edge_ids);
		m_profiler.leave("define_new_keyframe");
		if (m_verbose_level >= 1) std::cout << "[define_new_keyframe] Done. New KF #" << out_new_kf_info.kf_id << " with " << out_new_kf_info.created_edge_ids.size() << " new edges.\n";
	}

	/** impl/optimize_local_area.h:15-58 */
	void optimize_local_area(const TKeyFrameID root_id, const unsigned int win_size, TOptimizeExtraOutputInfo &out_info, const TOptimizeLocalAreaParams &params = TOptimizeLocalAreaParams(), const std::vector<size_t> &observation_indices_to_optimize = std::vector<size_t>()) {
		m_profiler.enter("optimize_local_area");
		const bool use_prebuilt_st = (win_size <= parameters.srba.max_tree_depth);
		if (!use_prebuilt_st && m_verbose_level >= 1) std::cout << "[optimize_local_area] *WARNING* Optimize win_size > max_tree_depth of prebuilt spanning trees. This is not efficient!\n";
		VisitorOptimizeLocalArea my_visitor(this->rba_state, params);
		this->bfs_visitor(root_id, win_size, use_prebuilt_st, my_visitor, my_visitor, my_visitor, my_visitor);
		if (!my_visitor.k2k_edges_to_optimize.empty() || !my_visitor.lm_IDs_to_optimize.empty())
			this->optimize_edges(my_visitor.k2k_edges_to_optimize, my_visitor.lm_IDs_to_optimize, out_info, observation_indices_to_optimize);
		m_profiler.leave("optimize_local_area");
	}

	void clear() { rba_state.clear(); }
	/** impl/alloc_keyframe.h:19-29 */
	TKeyFrameID alloc_keyframe() { const TKeyFrameID id = rba_state.keyframes.size(); rba_state.keyframes.push_back(keyframe_info()); return id; }
	/** impl/create_kf2kf_edge.h:15-37 */
	size_t create_kf2kf_edge(const TKeyFrameID new_kf_id, const TPairKeyFrameID &new_edge, const new_kf_observations_t &obs, const pose_t &init_inv_pose_val = pose_t()) {
		(void)obs;
		const size_t ed_id = rba_state.alloc_kf2kf_edge(new_edge, init_inv_pose_val);
		m_profiler.enter("define_new_keyframe.st.update_symbolic");
		rba_state.spanning_tree.update_symbolic_new_node(new_kf_id, new_edge, parameters.srba.max_tree_depth);
		m_profiler.leave("define_new_keyframe.st.update_symbolic");
		return ed_id;
	}
	bool find_path_bfs(const TKeyFrameID src_kf, const TKeyFrameID trg_kf, std::vector<TKeyFrameID> &found_path) const { return rba_state.find_path_bfs(src_kf, trg_kf, &found_path); }
	/** reference RbaEngine.h:385-396 */
	const pose_t *get_kf_relative_pose(const TKeyFrameID kf_query, const TKeyFrameID kf_reference) const {
		const frameid2pose_map_t *m = rba_state.spanning_tree.num.find(kf_reference);
		if (!m) return NULL;
		typename frameid2pose_map_t::const_iterator it = m->find(kf_query);
		return it != m->end() ? &it->second.pose : NULL;
	}

	/** impl/bfs_visitor.h:21-177 */
	template <class KF_VISITOR, class FEAT_VISITOR, class K2K_EDGE_VISITOR, class K2F_EDGE_VISITOR>
	void bfs_visitor(const TKeyFrameID root_id, const topo_dist_t max_distance, const bool rely_on_prebuilt_spanning_trees, KF_VISITOR &kf_visitor, FEAT_VISITOR &feat_visitor, K2K_EDGE_VISITOR &k2k_edge_visitor, K2F_EDGE_VISITOR &k2f_edge_visitor) const {
		using namespace std;
		set<TLandmarkID> lm_visited; set<const k2k_edge_t *> k2k_visited; set<const k2f_edge_t *> k2f_visited;
		vector<pair<TKeyFrameID, topo_dist_t> > KFs; // visiting order
		if (!rely_on_prebuilt_spanning_trees) { // (:35-104) plain BFS, expanded lazily below
			set<TKeyFrameID> kf_visited; queue<TKeyFrameID> pending; map<TKeyFrameID, topo_dist_t> distances;
			pending.push(root_id); kf_visited.insert(root_id); distances[root_id] = 0;
			while (!pending.empty()) {
				const TKeyFrameID next_kf = pending.front(); pending.pop();
				const topo_dist_t cur_dist = distances[next_kf];
				kf_visitor.visit_kf(next_kf, cur_dist);
				const keyframe_info &kfi = rba_state.keyframes[next_kf];
				visit_k2f_edges(kfi, next_kf, cur_dist, lm_visited, k2f_visited, feat_visitor, k2f_edge_visitor);
				if (cur_dist >= max_distance) continue;
				for (size_t i = 0; i < kfi.adjacent_k2k_edges.size(); i++) {
					const k2k_edge_t *ed = kfi.adjacent_k2k_edges[i];
					const TKeyFrameID new_kf = getTheOtherFromPair2(next_kf, *ed);
					if (!kf_visited.count(new_kf)) { if (kf_visitor.visit_filter_kf(new_kf, cur_dist)) { pending.push(new_kf); distances[new_kf] = cur_dist + 1; } kf_visited.insert(new_kf); }
					if (!k2k_visited.count(ed)) { if (k2k_edge_visitor.visit_filter_k2k(next_kf, new_kf, ed, cur_dist)) k2k_edge_visitor.visit_k2k(next_kf, new_kf, ed, cur_dist); k2k_visited.insert(ed); }
				}
			}
			return;
		}
		// prebuilt spanning trees (:105-176): root first, then ascending KF id
		const map<TKeyFrameID, TSpanTreeEntry> *root_ST = rba_state.spanning_tree.sym.next_edge.find(root_id);
		if (!root_ST) return;
		KFs.push_back(make_pair(root_id, (topo_dist_t)0));
		for (map<TKeyFrameID, TSpanTreeEntry>::const_iterator it = root_ST->begin(); it != root_ST->end(); ++it) KFs.push_back(make_pair(it->first, it->second.distance));
		for (size_t i = 0; i < KFs.size(); i++) {
			const TKeyFrameID kf_id = KFs[i].first; const topo_dist_t cur_dist = KFs[i].second;
			if (cur_dist > max_distance) continue;
			if (kf_visitor.visit_filter_kf(kf_id, cur_dist)) kf_visitor.visit_kf(kf_id, cur_dist);
			const keyframe_info &kfi = rba_state.keyframes[kf_id];
			visit_k2f_edges(kfi, kf_id, cur_dist, lm_visited, k2f_visited, feat_visitor, k2f_edge_visitor);
			for (size_t k = 0; k < kfi.adjacent_k2k_edges.size(); k++) {
				const k2k_edge_t *ed = kfi.adjacent_k2k_edges[k];
				const TKeyFrameID new_kf = getTheOtherFromPair2(kf_id, *ed);
				if (!k2k_visited.count(ed)) { if (k2k_edge_visitor.visit_filter_k2k(kf_id, new_kf, ed, cur_dist)) k2k_edge_visitor.visit_k2k(kf_id, new_kf, ed, cur_dist); k2k_visited.insert(ed); }
			}
		}
	}

	void enable_time_profiler(bool enable = true) { m_profiler.enable(enable); }
	const k2k_edges_deque_t &get_k2k_edges() const { return rba_state.k2k_edges; }
	const TRelativeLandmarkPosMap &get_known_feats() const { return rba_state.known_lms; }
	const TRelativeLandmarkPosMap &get_unknown_feats() const { return rba_state.unknown_lms; }
	const rba_problem_state_t &get_rba_state() const { return rba_state; }
	rba_problem_state_t &get_rba_state() { return rba_state; }
	mrpt::utils::CTimeLogger &get_time_profiler() { return m_profiler; }
	void setVerbosityLevel(int level) { m_verbose_level = level; }

	// ----------------------------------------------------------------------------------- MI355X-specific additions
	/** Select the numeric back-end. Default (lazy): srba::make_hip_backend(device) -- the GPU. */
	void set_numeric_backend(const std::shared_ptr<numeric_backend> &b) { m_backend = b; }
	void set_hip_device(int device) { m_hip_device = device; }
	/** Called with every capsule right before it is optimised (pre-optimisation values): capsule harvesting for batch replay. */
	std::function<void(const srba_hip_params &, CapsuleData &)> on_capsule;


	/** Squared error of ALL observations of the map with the current estimate (impl/eval_overall_error.h:15-137).
	  * Host: for every root = min(observer, base) a breadth-first search over the k2k graph (impl/spantree_create_complete.h:18-126; stopped as
	  * soon as every wanted target has been discovered -- the paths found are those of the complete search) gives the edge path to each target;
	  * the numeric back-end composes the poses along the paths and evaluates the residuals (GPU: srba_hip_eval_overall_sqr_error). */
	double eval_overall_squared_error() const {
		m_profiler.enter("eval_overall_squared_error");
		const size_t nObs = rba_state.all_observations.size();
		if (!nObs) { m_profiler.leave("eval_overall_squared_error"); return 0; }
		std::map<TKeyFrameID, std::set<TKeyFrameID> > ob_pairs; // minimum id first (:29-38)
		for (size_t i = 0; i < nObs; i++) { const k2f_edge_t &o = rba_state.all_observations[i]; const TKeyFrameID a = o.obs.kf_id, b = o.feat_rel_pos->id_frame_base; if (a != b) ob_pairs[std::min(a, b)].insert(std::max(a, b)); }
		// pairs + paths
		std::map<std::pair<TKeyFrameID, TKeyFrameID>, int32_t> pair_index; // (root, target) -> pair
		std::vector<int32_t> pair_path_off(1, 0), path_edge;
		std::vector<TKeyFrameID> prev(rba_state.keyframes.size()); std::vector<const k2k_edge_t *> via(rba_state.keyframes.size()); std::vector<char> seen(rba_state.keyframes.size(), 0); std::vector<TKeyFrameID> touched;
		for (typename std::map<TKeyFrameID, std::set<TKeyFrameID> >::const_iterator it1 = ob_pairs.begin(); it1 != ob_pairs.end(); ++it1) {
			const TKeyFrameID root = it1->first; size_t missing = it1->second.size();
			std::queue<TKeyFrameID> pending; pending.push(root); seen[root] = 1; touched.clear(); touched.push_back(root);
			while (!pending.empty() && missing) {
				const TKeyFrameID cur = pending.front(); pending.pop();
				const typename rba_problem_state_t::keyframe_info &kfi = rba_state.keyframes[cur];
				for (size_t i = 0; i < kfi.adjacent_k2k_edges.size() && missing; i++) {
					const k2k_edge_t *ed = kfi.adjacent_k2k_edges[i]; const TKeyFrameID nk = getTheOtherFromPair2(cur, *ed);
					if (seen[nk]) continue;
					seen[nk] = 1; touched.push_back(nk); prev[nk] = cur; via[nk] = ed; pending.push(nk);
					if (it1->second.count(nk)) missing--;
				}
			}
			for (typename std::set<TKeyFrameID>::const_iterator itT = it1->second.begin(); itT != it1->second.end(); ++itT) {
				if (!seen[*itT]) { for (size_t k = 0; k < touched.size(); k++) seen[touched[k]] = 0; throw std::runtime_error("eval_overall_squared_error: an observation relates two key-frames that are not connected"); }
				std::vector<int32_t> rev; // leaf -> root
				for (TKeyFrameID k = *itT; k != root; k = prev[k]) { const k2k_edge_t *ed = via[k]; rev.push_back((int32_t)((ed->id << 1) | (ed->to == k ? 1 : 0))); } // parent->me edge: my pose = parent (+) (-inv_pose) (:111-116)
				pair_index[std::make_pair(root, *itT)] = (int32_t)pair_path_off.size() - 1;
				path_edge.insert(path_edge.end(), rev.rbegin(), rev.rend()); pair_path_off.push_back((int32_t)path_edge.size());
			}
			for (size_t k = 0; k < touched.size(); k++) seen[touched[k]] = 0;
		}
		// observations + landmark table
		std::vector<int32_t> obs_pose(nObs), obs_lm(nObs); std::vector<double> obs_z(nObs * OBS_DIMS), lm_pos;
		typedef std::map<const void *, int32_t> lm_index_t; lm_index_t lm_index;
		for (size_t i = 0; i < nObs; i++) {
			const k2f_edge_t &o = rba_state.all_observations[i]; const TKeyFrameID a = o.obs.kf_id, b = o.feat_rel_pos->id_frame_base;
			// pose of the base as seen from the observer: root==observer -> stored pose of the target; root==base -> its inverse (:68-71)
			obs_pose[i] = a == b ? -1 : (a < b ? 2 * pair_index[std::make_pair(a, b)] : 2 * pair_index[std::make_pair(b, a)] + 1);
			lm_index_t::const_iterator itL = lm_index.find((const void *)o.feat_rel_pos);
			if (itL == lm_index.end()) { itL = lm_index.insert(std::make_pair((const void *)o.feat_rel_pos, (int32_t)(lm_pos.size() / LM_DIMS))).first; for (size_t k = 0; k < LM_DIMS; k++) lm_pos.push_back(o.feat_rel_pos->pos[k]); }
			obs_lm[i] = itL->second;
			for (size_t k = 0; k < OBS_DIMS; k++) obs_z[i * OBS_DIMS + k] = o.obs.obs_arr[k];
		}
		std::vector<double> edge_pose(rba_state.k2k_edges.size() * pose_t::storage_doubles());
		for (size_t e = 0; e < rba_state.k2k_edges.size(); e++) rba_state.k2k_edges[e].inv_pose.storeTo(&edge_pose[e * pose_t::storage_doubles()]);
		srba_overall_problem q; std::memset(&q, 0, sizeof(q));
		q.n_edges = (int32_t)rba_state.k2k_edges.size(); q.n_pairs = (int32_t)pair_path_off.size() - 1; q.n_path = (int32_t)path_edge.size(); q.n_obs = (int32_t)nObs; q.n_lms = (int32_t)(lm_pos.size() / LM_DIMS);
		q.edge_pose = edge_pose.data(); q.pair_path_off = pair_path_off.data(); q.path_edge = path_edge.data(); q.obs_pose = obs_pose.data(); q.obs_lm = obs_lm.data(); q.obs_z = obs_z.data(); q.lm_pos = lm_pos.data();
		srba_hip_params hp; fill_hip_params(hp);
		if (!m_backend) m_backend = make_hip_backend(m_hip_device);
		const double sqerr = m_backend->eval_overall(hp, q);
		m_profiler.leave("eval_overall_squared_error");
		return sqerr;
	}

	/** Fill the back-end parameter block from `parameters` (what the reference hot loops read from RbaEngine::parameters). */
	void fill_hip_params(srba_hip_params &hp) const {
		std::memset(&hp, 0, sizeof(hp));
		hp.family = sensor_model_t::family; hp.solver = RBA_OPTIONS::solver_t::solver_id;
		hp.std_noise_observations = 1.0; for (int i = 0; i < 9; i++) hp.sensor_pose_se3[3 + i] = (i % 4 == 0) ? 1.0 : 0.0; hp.right_cam_pose[3] = 1.0;
		RBA_OPTIONS::obs_noise_matrix_t::fill_params(hp, parameters.obs_noise);
		RBA_OPTIONS::sensor_pose_on_robot_t::fill_params(hp, parameters.sensor_pose);
		sensor_model_t::fill_params(hp, parameters.sensor);
		hp.max_iters = (int)parameters.srba.max_iters; hp.use_robust_kernel = parameters.srba.use_robust_kernel ? 1 : 0; hp.kernel_param = parameters.srba.kernel_param;
		hp.max_error_per_obs_to_stop = parameters.srba.max_error_per_obs_to_stop; hp.max_rho = parameters.srba.max_rho; hp.max_lambda = parameters.srba.max_lambda;
		hp.min_error_reduction_ratio_to_relinearize = parameters.srba.min_error_reduction_ratio_to_relinearize; hp.cov_recovery = (parameters.srba.cov_recovery == crpLandmarksApprox) ? 1 : 0;
	}

protected:
	int m_verbose_level;

	template <class FEAT_VISITOR, class K2F_EDGE_VISITOR>
	static void visit_k2f_edges(const keyframe_info &kfi, const TKeyFrameID kf_id, const topo_dist_t cur_dist, std::set<TLandmarkID> &lm_visited, std::set<const k2f_edge_t *> &k2f_visited, FEAT_VISITOR &feat_visitor, K2F_EDGE_VISITOR &k2f_edge_visitor) {
		for (size_t i = 0; i < kfi.adjacent_k2f_edges.size(); i++) {
			const k2f_edge_t *ed = kfi.adjacent_k2f_edges[i];
			const TLandmarkID lm_ID = ed->obs.obs.feat_id;
			if (!lm_visited.count(lm_ID)) { if (feat_visitor.visit_filter_feat(lm_ID, cur_dist)) feat_visitor.visit_feat(lm_ID, cur_dist); lm_visited.insert(lm_ID); }
			if (!k2f_visited.count(ed)) { if (k2f_edge_visitor.visit_filter_k2f(kf_id, ed, cur_dist)) k2f_edge_visitor.visit_k2f(kf_id, ed, cur_dist); k2f_visited.insert(ed); }
		}
	}

	/** reference RbaEngine.h:543-618 */
	struct VisitorOptimizeLocalArea {
		VisitorOptimizeLocalArea(const rba_problem_state_t &rba_state_, const TOptimizeLocalAreaParams &params_) : rba_state(rba_state_), params(params_) {}
		const rba_problem_state_t &rba_state; const TOptimizeLocalAreaParams &params;
		std::vector<size_t> k2k_edges_to_optimize, lm_IDs_to_optimize; std::map<TLandmarkID, size_t> lm_times_seen;
		bool visit_filter_feat(const TLandmarkID, const topo_dist_t) { return false; }
		void visit_feat(const TLandmarkID, const topo_dist_t) {}
		bool visit_filter_kf(const TKeyFrameID kf_ID, const topo_dist_t) { return kf_ID <= params.max_visitable_kf_id; }
		void visit_kf(const TKeyFrameID, const topo_dist_t) {}
		bool visit_filter_k2k(const TKeyFrameID, const TKeyFrameID, const k2k_edge_t *, const topo_dist_t) { return true; }
		void visit_k2k(const TKeyFrameID, const TKeyFrameID, const k2k_edge_t *edge, const topo_dist_t) { if (params.optimize_k2k_edges) k2k_edges_to_optimize.push_back(edge->id); }
		bool visit_filter_k2f(const TKeyFrameID, const k2f_edge_t *, const topo_dist_t) { return params.optimize_landmarks; }
		void visit_k2f(const TKeyFrameID, const k2f_edge_t *edge, const topo_dist_t) {
			if (!edge->feat_has_known_rel_pos) { const TLandmarkID lm_ID = edge->obs.obs.feat_id; if (++lm_times_seen[lm_ID] == params.dont_optimize_landmarks_seen_less_than_n_times) lm_IDs_to_optimize.push_back(lm_ID); }
		}
	};

	/** impl/determine_kf2kf_edges_to_create.h:17-266 */
	void determine_kf2kf_edges_to_create(const TKeyFrameID new_kf_id, const new_kf_observations_t &obs, std::vector<TNewEdgeInfo> &new_k2k_edge_ids) {
		new_k2k_edge_ids.clear();
		if (rba_state.keyframes.size() == 1) return;
		edge_creation_policy.template eval<traits_t, rba_engine_t>(new_kf_id, obs, new_k2k_edge_ids, *this, parameters.ecp);
		typedef std::vector<typename obs_t::obs_data_t> obs_vec_t;
		const size_t nNEI = new_k2k_edge_ids.size();
		for (size_t i = 0; i < nNEI; i++) {
			TNewEdgeInfo &nei = new_k2k_edge_ids[i];
			if (nei.has_approx_init_val) continue;
			k2k_edge_t &nei_edge = rba_state.k2k_edges[nei.id];
			const bool nei_edge_does_not_touch_cur_kf = (nei_edge.to != new_kf_id) && (nei_edge.from != new_kf_id);
			// Method #1: relative pose of the previous KF wrt the same "from" KF (:48-60)
			if (!nei_edge_does_not_touch_cur_kf && rba_state.last_timestep_touched_kfs.count(nei_edge.from) != 0) {
				const pose_t *rel_pose = get_kf_relative_pose(new_kf_id - 1, nei_edge.from);
				if (rel_pose) {
					if (nei_edge.to == new_kf_id) nei_edge.inv_pose = -(*rel_pose); else nei_edge.inv_pose = *rel_pose;
					nei.has_approx_init_val = true;
				}
			}
			// Method #2: sensor-specific landmark matcher (:64-255)
			if (!nei.has_approx_init_val) {
				TKeyFrameID last_kf_id, other_kf_id;
				if (nei_edge_does_not_touch_cur_kf) { last_kf_id = nei_edge.from; other_kf_id = nei_edge.to; }
				else { last_kf_id = new_kf_id; other_kf_id = (nei_edge.to == new_kf_id) ? nei_edge.from : nei_edge.to; }
				obs_vec_t new_kf_obs, old_kf_obs;
				gather_matching_obs(obs, !nei_edge_does_not_touch_cur_kf, last_kf_id, other_kf_id, new_kf_obs, old_kf_obs);
				pose_t pose_new_kf_wrt_old_kf;
				bool found_ok = observations::landmark_matcher<obs_t>::find_relative_pose(new_kf_obs, old_kf_obs, parameters.sensor, pose_new_kf_wrt_old_kf);
				if (!found_ok && nei.loopclosure_observer_kf != SRBA_INVALID_KEYFRAMEID && nei.loopclosure_base_kf != SRBA_INVALID_KEYFRAMEID) { // 2nd attempt (:137-189)
					last_kf_id = nei.loopclosure_observer_kf; other_kf_id = nei.loopclosure_base_kf;
					gather_matching_obs(obs, last_kf_id == new_kf_id, last_kf_id, other_kf_id, new_kf_obs, old_kf_obs);
					found_ok = observations::landmark_matcher<obs_t>::find_relative_pose(new_kf_obs, old_kf_obs, parameters.sensor, pose_new_kf_wrt_old_kf);
				}
				if (found_ok) {
					// sensor pose on the robot (:193-197)
					const mrpt::poses::CPose3D sensor_pose = RBA_OPTIONS::sensor_pose_on_robot_t::sensor_pose_as_3d(parameters.sensor_pose);
					pose_new_kf_wrt_old_kf = pose_t((sensor_pose + mrpt::poses::CPose3D(pose_new_kf_wrt_old_kf)) + (-sensor_pose));
					const bool edge_dir_to_newkf = (nei_edge.to == new_kf_id);
					nei.has_approx_init_val = true;
					if (!nei_edge_does_not_touch_cur_kf) { if (edge_dir_to_newkf) nei_edge.inv_pose = -pose_new_kf_wrt_old_kf; else nei_edge.inv_pose = pose_new_kf_wrt_old_kf; }
					else { // loop closure between two older centre KFs (:209-246)
						const pose_t default_identity_pose;
						const pose_t *pose_observer_wrt_to = (nei.loopclosure_observer_kf == nei_edge.to) ? &default_identity_pose : get_kf_relative_pose(nei.loopclosure_observer_kf, nei_edge.to);
						const pose_t *pose_base_wrt_to = (nei.loopclosure_base_kf == nei_edge.to) ? &default_identity_pose : get_kf_relative_pose(nei.loopclosure_base_kf, nei_edge.to);
						const pose_t *pose_observer_wrt_from = (nei.loopclosure_observer_kf == nei_edge.from) ? &default_identity_pose : get_kf_relative_pose(nei.loopclosure_observer_kf, nei_edge.from);
						const pose_t *pose_base_wrt_from = (nei.loopclosure_base_kf == nei_edge.from) ? &default_identity_pose : get_kf_relative_pose(nei.loopclosure_base_kf, nei_edge.from);
						const bool observer_is_near_to = (pose_observer_wrt_to || pose_base_wrt_from) || !(pose_observer_wrt_from || pose_base_wrt_to);
						const pose_t *pose_observer_wrt_local = observer_is_near_to ? (pose_observer_wrt_to ? pose_observer_wrt_to : &default_identity_pose) : (pose_observer_wrt_from ? pose_observer_wrt_from : &default_identity_pose);
						const pose_t *pose_base_wrt_remote = observer_is_near_to ? (pose_base_wrt_from ? pose_base_wrt_from : &default_identity_pose) : (pose_base_wrt_to ? pose_base_wrt_to : &default_identity_pose);
						const pose_t pose_local_wrt_remote = ((*pose_base_wrt_remote) + (pose_new_kf_wrt_old_kf)) + (-(*pose_observer_wrt_local));
						if (edge_dir_to_newkf) nei_edge.inv_pose = -pose_local_wrt_remote; else nei_edge.inv_pose = pose_local_wrt_remote;
					}
				}
				if (!nei.has_approx_init_val && m_verbose_level >= 2) std::cout << "[determine_kf2kf_edges_to_create] Could not provide initial value to relative pose " << nei_edge.from << "<=>" << nei_edge.to << "\n";
			}
		}
		rba_state.last_timestep_touched_kfs.clear();
		for (size_t i = 0; i < nNEI; i++) { const k2k_edge_t &e = rba_state.k2k_edges[new_k2k_edge_ids[i].id]; rba_state.last_timestep_touched_kfs.insert(e.from); rba_state.last_timestep_touched_kfs.insert(e.to); }
	}
	/** The two correspondence lists of determine_kf2kf_edges_to_create.h:81-119 / :154-183 */
	void gather_matching_obs(const new_kf_observations_t &obs, const bool last_is_new_kf, const TKeyFrameID last_kf_id, const TKeyFrameID other_kf_id,
	                         std::vector<typename obs_t::obs_data_t> &new_kf_obs, std::vector<typename obs_t::obs_data_t> &old_kf_obs) const {
		new_kf_obs.clear(); old_kf_obs.clear();
		const std::deque<k2f_edge_t *> &other_k2f_edges = rba_state.keyframes[other_kf_id].adjacent_k2f_edges;
		std::map<TLandmarkID, size_t> newkf_obs_feats;
		const std::deque<k2f_edge_t *> *last_k2f_edges = NULL;
		if (!last_is_new_kf) { last_k2f_edges = &rba_state.keyframes[last_kf_id].adjacent_k2f_edges; for (size_t i = 0; i < last_k2f_edges->size(); i++) newkf_obs_feats[(*last_k2f_edges)[i]->obs.obs.feat_id] = i; }
		else for (size_t i = 0; i < obs.size(); i++) newkf_obs_feats[obs[i].obs.feat_id] = i;
		for (size_t i = 0; i < other_k2f_edges.size(); i++) {
			std::map<TLandmarkID, size_t>::const_iterator it_id = newkf_obs_feats.find(other_k2f_edges[i]->obs.obs.feat_id);
			if (it_id == newkf_obs_feats.end()) continue;
			old_kf_obs.push_back(other_k2f_edges[i]->obs.obs.obs_data);
			if (!last_is_new_kf) new_kf_obs.push_back((*last_k2f_edges)[it_id->second]->obs.obs.obs_data); else new_kf_obs.push_back(obs[it_id->second].obs.obs_data);
		}
	}

	/** impl/add-observations.h:17-264 */
	size_t add_observation(const TKeyFrameID observing_kf_id, const typename observation_traits_t::observation_t &new_obs, const array_landmark_t *fixed_relative_position = NULL, const array_landmark_t *unknown_relative_position_init_val = NULL) {
		ASSERT_(!(fixed_relative_position != NULL && unknown_relative_position_init_val != NULL));
		const bool is_1st_time_seen = (new_obs.feat_id >= rba_state.all_lms.size() || rba_state.all_lms[new_obs.feat_id].rfp == NULL);
		const bool is_fixed = (fixed_relative_position != NULL) || (!is_1st_time_seen && rba_state.all_lms[new_obs.feat_id].has_known_pos);
		const size_t new_obs_idx = rba_state.all_observations.size();
		rba_state.all_observations.push_back(k2f_edge_t());
		rba_state.all_observations_Jacob_validity.push_back(1);
		k2f_edge_t &new_k2f_edge = rba_state.all_observations.back();
		if (is_1st_time_seen) {
			TRelativeLandmarkPos new_rfp; new_rfp.id_frame_base = observing_kf_id;
			if (is_fixed) {
				new_rfp.pos = *fixed_relative_position;
				typename TRelativeLandmarkPosMap::iterator it_new = rba_state.known_lms.insert(rba_state.known_lms.end(), typename TRelativeLandmarkPosMap::value_type(new_obs.feat_id, new_rfp));
				if (new_obs.feat_id >= rba_state.all_lms.size()) rba_state.all_lms.resize(new_obs.feat_id + 1);
				rba_state.all_lms[new_obs.feat_id] = typename landmark_traits_t::TLandmarkEntry(true, &it_new->second);
			} else {
				if (unknown_relative_position_init_val) new_rfp.pos = *unknown_relative_position_init_val;
				else {
					sensor_model_t::inverse_sensor_model(new_rfp.pos, new_obs.obs_data, this->parameters.sensor);
					RBA_OPTIONS::sensor_pose_on_robot_t::template sensor2robot_point<landmark_t>(new_rfp.pos, this->parameters.sensor_pose);
				}
				typename TRelativeLandmarkPosMap::iterator it_new = rba_state.unknown_lms.insert(rba_state.unknown_lms.end(), typename TRelativeLandmarkPosMap::value_type(new_obs.feat_id, new_rfp));
				if (new_obs.feat_id >= rba_state.all_lms.size()) rba_state.all_lms.resize(new_obs.feat_id + 1);
				rba_state.all_lms[new_obs.feat_id] = typename landmark_traits_t::TLandmarkEntry(false, &it_new->second);
				rba_state.lin_system.dh_df_remap[new_obs.feat_id] = rba_state.lin_system.dh_df.size();
				rba_state.lin_system.dh_df.push_back(std::vector<TJacobianSymbolicInfo_dh_df>());
			}
		}
		TRelativeLandmarkPos *lm_rel_pos = rba_state.all_lms[new_obs.feat_id].rfp;
		const TKeyFrameID base_id = lm_rel_pos->id_frame_base;
		new_k2f_edge.obs.kf_id = observing_kf_id; new_k2f_edge.obs.obs = new_obs; new_k2f_edge.obs.obs.obs_data.getAsArray(new_k2f_edge.obs.obs_arr);
		new_k2f_edge.is_first_obs_of_unknown = is_1st_time_seen && !is_fixed; new_k2f_edge.feat_has_known_rel_pos = is_fixed; new_k2f_edge.feat_rel_pos = lm_rel_pos;
		rba_state.keyframes[observing_kf_id].adjacent_k2f_edges.push_back(&new_k2f_edge);
		// dh_dAp: one block per edge on the ST path observer -> base (:139-215)
		bool graph_says_ignore_this_obs = false;
		if (base_id != observing_kf_id) {
			const TKeyFrameID from = std::max(observing_kf_id, base_id), to = std::min(observing_kf_id, base_id);
			const bool all_edges_inverse = (from != observing_kf_id);
			const std::map<TKeyFrameID, typename rba_problem_state_t::k2k_edge_path_t> *it_map = rba_state.spanning_tree.sym.all_edges.find(from);
			ASSERTMSG_(it_map != NULL, "No ST.all_edges found for observing/base keyframes");
			typename std::map<TKeyFrameID, typename rba_problem_state_t::k2k_edge_path_t>::const_iterator it_obs_ed = it_map->find(to);
			if (it_obs_ed != it_map->end()) {
				const typename rba_problem_state_t::k2k_edge_path_t &obs_edges = it_obs_ed->second; ASSERT_(!obs_edges.empty());
				TKeyFrameID curKF = observing_kf_id;
				for (size_t j = 0; j < obs_edges.size(); j++) {
					const size_t i = all_edges_inverse ? (obs_edges.size() - j - 1) : j;
					const k2k_edge_t &e = rba_state.k2k_edges[obs_edges[i]];
					const bool normal_dir = (e.to == curKF);
					TJacobianSymbolicInfo_dh_dAp sym;
					sym.obs_idx = new_obs_idx; sym.k2k_edge_id = e.id; sym.kf_d = curKF; sym.kf_base = base_id; sym.edge_normal_dir = normal_dir; sym.has_A = (curKF != observing_kf_id);
					rba_state.lin_system.dh_dAp[e.id].push_back(sym);
					// placeholders in the numeric ST (the reference takes pointers to them here, :199-204)
					rba_state.spanning_tree.num[curKF][base_id];
					if (sym.has_A) rba_state.spanning_tree.num[observing_kf_id][curKF];
					curKF = normal_dir ? e.from : e.to;
				}
			} else graph_says_ignore_this_obs = true;
		}
		// dh_df (:220-257)
		if (!is_fixed && !graph_says_ignore_this_obs) {
			const size_t *col_idx = rba_state.lin_system.dh_df_remap.find(new_obs.feat_id); ASSERT_(col_idx != NULL);
			TJacobianSymbolicInfo_dh_df sym; sym.obs_idx = new_obs_idx; sym.has_pose = !new_k2f_edge.is_first_obs_of_unknown;
			rba_state.lin_system.dh_df[*col_idx].push_back(sym);
			if (sym.has_pose) rba_state.spanning_tree.num[observing_kf_id][base_id];
		}
		return new_obs_idx;
	}

	/** impl/optimize_edges.h:44-793.  S1-S4 and the symbolic planning run here; S5-S17 on the numeric back-end. */
	void optimize_edges(const std::vector<size_t> &run_k2k_edges_in, const std::vector<size_t> &run_feat_ids_in, TOptimizeExtraOutputInfo &out_info, const std::vector<size_t> &in_observation_indices_to_optimize = std::vector<size_t>()) {
		m_profiler.enter("opt");
		out_info.clear();
		const int P = REL_POSE_DIMS, L = LM_DIMS, O = OBS_DIMS; const int PD = (int)pose_t::storage_doubles();
		// S1: drop unknowns without observations (:71-119)
		std::vector<size_t> run_k2k_edges, run_feat_ids; std::set<TKeyFrameID> touched_KFs; std::set<TLandmarkID> touched_LMs;
		for (size_t i = 0; i < run_k2k_edges_in.size(); i++) {
			if (!rba_state.lin_system.dh_dAp[run_k2k_edges_in[i]].empty()) { run_k2k_edges.push_back(run_k2k_edges_in[i]); touched_KFs.insert(rba_state.k2k_edges[run_k2k_edges_in[i]].from); touched_KFs.insert(rba_state.k2k_edges[run_k2k_edges_in[i]].to); }
			else std::cerr << "[RbaEngine::optimize_edges] *Warning*: Skipping optimization of k2k edge #" << run_k2k_edges_in[i] << " (" << rba_state.k2k_edges[run_k2k_edges_in[i]].from << "->" << rba_state.k2k_edges[run_k2k_edges_in[i]].to << ") since no observation depends on it.\n";
		}
		std::vector<size_t> run_feat_cols;
		for (size_t i = 0; i < run_feat_ids_in.size(); i++) {
			const TLandmarkID feat_id = run_feat_ids_in[i]; ASSERT_(feat_id < rba_state.all_lms.size());
			const typename rba_problem_state_t::TLandmarkEntry &lm_e = rba_state.all_lms[feat_id];
			ASSERTMSG_(lm_e.rfp != NULL, "Trying to optimize an unknown feature ID"); ASSERTMSG_(!lm_e.has_known_pos, "Trying to optimize a feature with fixed (known) value");
			const size_t *col = rba_state.lin_system.dh_df_remap.find(feat_id); ASSERT_(col != NULL);
			if (!rba_state.lin_system.dh_df[*col].empty()) { run_feat_ids.push_back(feat_id); run_feat_cols.push_back(*col); touched_LMs.insert(feat_id); }
			else std::cerr << "[RbaEngine::optimize_edges] *Warning*: Skipping optimization of k2f edge #" << feat_id << " since no observation depends on it.\n";
		}
		const size_t nUnknowns_k2k = run_k2k_edges.size(), nUnknowns_k2f = run_feat_ids.size();
		out_info.num_kf_optimized = touched_KFs.size(); out_info.num_lm_optimized = touched_LMs.size();
		const size_t nUnknowns_scalars = P * nUnknowns_k2k + L * nUnknowns_k2f;
		if (!nUnknowns_scalars) { std::cerr << "[RbaEngine::optimize_edges] *Warning*: Skipping optimization since no observation depends on any of the given variables.\n"; m_profiler.leave("opt"); return; }

		CapsuleData cd; cd.P = P; cd.L = L; cd.O = O; cd.PD = PD; cd.n_unk_edges = (int)nUnknowns_k2k; cd.n_unk_lms = (int)nUnknowns_k2f;
		for (size_t i = 0; i < nUnknowns_k2k; i++) cd.unk_edge_ids.push_back(run_k2k_edges[i]);
		for (size_t i = 0; i < nUnknowns_k2f; i++) cd.unk_lm_ids.push_back(run_feat_ids[i]);
		std::map<TLandmarkID, int> unk_lm_slot; for (size_t i = 0; i < nUnknowns_k2f; i++) unk_lm_slot[run_feat_ids[i]] = (int)i;

		m_profiler.enter("opt.capsule.s3s4");
		// S3: involved observations, with the reference's duplicates for k2k columns (:171-234)
		std::map<size_t, size_t> obs_global_idx2residual_idx; std::vector<size_t> involved_obs;
		if (in_observation_indices_to_optimize.empty()) {
			for (size_t i = 0; i < nUnknowns_k2k; i++) { const std::vector<TJacobianSymbolicInfo_dh_dAp> &col = rba_state.lin_system.dh_dAp[run_k2k_edges[i]];
				for (size_t b = 0; b < col.size(); b++) { obs_global_idx2residual_idx[col[b].obs_idx] = involved_obs.size(); involved_obs.push_back(col[b].obs_idx); } }
			for (size_t i = 0; i < nUnknowns_k2f; i++) { const std::vector<TJacobianSymbolicInfo_dh_df> &col = rba_state.lin_system.dh_df[run_feat_cols[i]];
				for (size_t b = 0; b < col.size(); b++) if (obs_global_idx2residual_idx.find(col[b].obs_idx) == obs_global_idx2residual_idx.end()) { obs_global_idx2residual_idx[col[b].obs_idx] = involved_obs.size(); involved_obs.push_back(col[b].obs_idx); } }
		} else for (size_t i = 0; i < in_observation_indices_to_optimize.size(); i++) { obs_global_idx2residual_idx[in_observation_indices_to_optimize[i]] = involved_obs.size(); involved_obs.push_back(in_observation_indices_to_optimize[i]); }
		const size_t nObs = involved_obs.size();

		// S4: roots whose numeric spanning trees are refreshed (jacobians.h:1020-1076), ascending (std::set)
		std::set<TKeyFrameID> kfs_num_spantrees_to_update;
		for (size_t i = 0; i < nUnknowns_k2k; i++) { const std::vector<TJacobianSymbolicInfo_dh_dAp> &col = rba_state.lin_system.dh_dAp[run_k2k_edges[i]];
			for (size_t b = 0; b < col.size(); b++) { kfs_num_spantrees_to_update.insert(col[b].kf_d); kfs_num_spantrees_to_update.insert(rba_state.all_observations[col[b].obs_idx].obs.kf_id); kfs_num_spantrees_to_update.insert(col[b].kf_base); } }
		for (size_t i = 0; i < nUnknowns_k2f; i++) { const std::vector<TJacobianSymbolicInfo_dh_df> &col = rba_state.lin_system.dh_df[run_feat_cols[i]];
			for (size_t b = 0; b < col.size(); b++) { const k2f_edge_t &k2f = rba_state.all_observations[col[b].obs_idx]; kfs_num_spantrees_to_update.insert(k2f.feat_rel_pos->id_frame_base); kfs_num_spantrees_to_update.insert(k2f.obs.kf_id); } }

		m_profiler.leave("opt.capsule.s3s4"); m_profiler.enter("opt.capsule.pairs");
		// ST pair table: every stored path of every root (spantree_update_numeric.h:111-127), local edge table, path flags
		std::map<size_t, int> local_edge; // global edge id -> local slot
		for (size_t i = 0; i < nUnknowns_k2k; i++) local_edge[run_k2k_edges[i]] = (int)i;
		std::vector<size_t> local_edge_ids(run_k2k_edges.begin(), run_k2k_edges.end());
		std::map<TPairKeyFrameID, int> pair_index; // (root > target) -> pair
		cd.pair_path_off.push_back(0);
		for (std::set<TKeyFrameID>::const_iterator it = kfs_num_spantrees_to_update.begin(); it != kfs_num_spantrees_to_update.end(); ++it) {
			const std::map<TKeyFrameID, typename rba_problem_state_t::k2k_edge_path_t> *row = rba_state.spanning_tree.sym.all_edges.find(*it);
			if (!row) continue;
			for (typename std::map<TKeyFrameID, typename rba_problem_state_t::k2k_edge_path_t>::const_iterator itE = row->begin(); itE != row->end(); ++itE) {
				pair_index[TPairKeyFrameID(*it, itE->first)] = (int)cd.pair_kfs.size();
				cd.pair_kfs.push_back(std::make_pair((uint64_t)*it, (uint64_t)itE->first));
				TKeyFrameID curKF = *it;
				for (size_t k = 0; k < itE->second.size(); k++) {
					const k2k_edge_t &e = rba_state.k2k_edges[itE->second[k]];
					std::map<size_t, int>::iterator le = local_edge.find(e.id);
					if (le == local_edge.end()) { le = local_edge.insert(std::make_pair(e.id, (int)local_edge_ids.size())).first; local_edge_ids.push_back(e.id); }
					int inv; if (e.to == curKF) { inv = 0; curKF = e.from; } else { inv = 1; curKF = e.to; } // spantree_update_numeric.h:49-65
					cd.path_edge.push_back((le->second << 1) | inv);
				}
				cd.pair_path_off.push_back((int32_t)cd.path_edge.size());
			}
		}
		const size_t nPairs = cd.pair_kfs.size();
		cd.pair_needed.assign(nPairs, 0); cd.pose_required.assign(2 * nPairs, 0);
		cd.edge_pose.resize(local_edge_ids.size() * PD);
		for (size_t i = 0; i < local_edge_ids.size(); i++) rba_state.k2k_edges[local_edge_ids[i]].inv_pose.storeTo(&cd.edge_pose[i * PD]);
		struct PoseIdx { const std::map<TPairKeyFrameID, int> &m; int operator()(TKeyFrameID src, TKeyFrameID trg) const { // index of num[src][trg]
			std::map<TPairKeyFrameID, int>::const_iterator it = m.find(src > trg ? TPairKeyFrameID(src, trg) : TPairKeyFrameID(trg, src));
			if (it == m.end()) throw std::logic_error("optimize_edges: numeric spanning-tree entry not available (graph deeper than max_tree_depth?)");
			return 2 * it->second + (src > trg ? 0 : 1); } } pose_idx = {pair_index};

		m_profiler.leave("opt.capsule.pairs"); m_profiler.enter("opt.capsule.tables");
		// landmarks: unknown slots / constants
		cd.ulm_pos.resize(nUnknowns_k2f * L);
		for (size_t i = 0; i < nUnknowns_k2f; i++) for (int k = 0; k < L; k++) cd.ulm_pos[i * L + k] = rba_state.all_lms[run_feat_ids[i]].rfp->pos[k];
		std::map<TLandmarkID, int> const_lm_slot;
		struct LmRef { std::map<TLandmarkID, int> &unk, &cst; CapsuleData &cd; const rba_problem_state_t &st; int L; int operator()(TLandmarkID id) {
			std::map<TLandmarkID, int>::const_iterator u = unk.find(id); if (u != unk.end()) return u->second;
			std::map<TLandmarkID, int>::const_iterator c = cst.find(id);
			if (c == cst.end()) { c = cst.insert(std::make_pair(id, (int)cst.size())).first; for (int k = 0; k < L; k++) cd.klm_pos.push_back(st.all_lms[id].rfp->pos[k]); }
			return -1 - c->second; } } lm_ref = {unk_lm_slot, const_lm_slot, cd, rba_state, L};

		// observation rows + validity slots
		std::map<size_t, int> valid_slot;
		for (size_t i = 0; i < nObs; i++) {
			const k2f_edge_t &k2f = rba_state.all_observations[involved_obs[i]];
			const TKeyFrameID obs_kf = k2f.obs.kf_id, base_kf = k2f.feat_rel_pos->id_frame_base;
			cd.obs_pose.push_back(obs_kf == base_kf ? -1 : pose_idx(obs_kf, base_kf));
			if (parameters.srba.refresh_all_read_poses && cd.obs_pose.back() >= 0) cd.pose_required[cd.obs_pose.back()] = 1;
			cd.obs_lm.push_back(lm_ref(k2f.obs.obs.feat_id));
			std::map<size_t, int>::iterator vs = valid_slot.find(involved_obs[i]); if (vs == valid_slot.end()) vs = valid_slot.insert(std::make_pair(involved_obs[i], (int)valid_slot.size())).first;
			cd.obs_valid.push_back(vs->second);
			for (int k = 0; k < O; k++) cd.obs_z.push_back(k2f.obs.obs_arr[k]);
		}
		cd.n_valid = (int)valid_slot.size();

		// Jacobian block tables in the reference sweep order (jacobians.h:1094-1114) + list_of_required_num_poses (:225-230,:900-901)
		std::vector<uint64_t> bp_row, bf_row;
		cd.colp_off.push_back(0);
		for (size_t i = 0; i < nUnknowns_k2k; i++) {
			const std::vector<TJacobianSymbolicInfo_dh_dAp> &col = rba_state.lin_system.dh_dAp[run_k2k_edges[i]];
			for (size_t b = 0; b < col.size(); b++) {
				const TJacobianSymbolicInfo_dh_dAp &s = col[b]; const k2f_edge_t &k2f = rba_state.all_observations[s.obs_idx];
				std::map<size_t, size_t>::const_iterator r = obs_global_idx2residual_idx.find(s.obs_idx); ASSERT_(r != obs_global_idx2residual_idx.end());
				const int A = s.has_A ? pose_idx(k2f.obs.kf_id, s.kf_d) : -1, D = pose_idx(s.kf_d, s.kf_base);
				cd.bp_col.push_back((int)i); cd.bp_res.push_back((int)r->second); cd.bp_A.push_back(A); cd.bp_D.push_back(D); cd.bp_lm.push_back(lm_ref(k2f.obs.obs.feat_id)); cd.bp_normal.push_back(s.edge_normal_dir ? 1 : 0);
				bp_row.push_back(s.obs_idx);
				if (A >= 0) cd.pose_required[A] = 1;
				cd.pose_required[D] = 1;
			}
			cd.colp_off.push_back((int32_t)cd.bp_col.size());
		}
		cd.colf_off.push_back(0);
		for (size_t i = 0; i < nUnknowns_k2f; i++) {
			const std::vector<TJacobianSymbolicInfo_dh_df> &col = rba_state.lin_system.dh_df[run_feat_cols[i]];
			for (size_t b = 0; b < col.size(); b++) {
				const k2f_edge_t &k2f = rba_state.all_observations[col[b].obs_idx];
				std::map<size_t, size_t>::const_iterator r = obs_global_idx2residual_idx.find(col[b].obs_idx); ASSERT_(r != obs_global_idx2residual_idx.end());
				const int pi = col[b].has_pose ? pose_idx(k2f.obs.kf_id, k2f.feat_rel_pos->id_frame_base) : -1;
				cd.bf_col.push_back((int)i); cd.bf_res.push_back((int)r->second); cd.bf_pose.push_back(pi); bf_row.push_back(col[b].obs_idx);
				if (pi >= 0) cd.pose_required[pi] = 1;
			}
			cd.colf_off.push_back((int32_t)cd.bf_col.size());
		}
		for (size_t p = 0; p < nPairs; p++) cd.pair_needed[p] = (cd.pose_required[2 * p] || cd.pose_required[2 * p + 1]) ? 1 : 0;
		m_profiler.leave("opt.capsule.tables");
		// S9 + S15: symbolic Hessian / Schur plan
		m_profiler.enter("opt.sparse_hessian_build_symbolic");
		cd.build_plan(bp_row, bf_row, RBA_OPTIONS::solver_t::USE_SCHUR);
		m_profiler.leave("opt.sparse_hessian_build_symbolic");

		// ---- numeric part: S5..S17 on the back-end ----
		srba_hip_params hp; fill_hip_params(hp);
		if (on_capsule) on_capsule(hp, cd);
		srba_problem_capsule cap = cd.view();
		srba_lm_result res; std::memset(&res, 0, sizeof(res));
		if (!m_backend) m_backend = make_hip_backend(m_hip_device);
		m_backend->set_profiler(&m_profiler);
		m_profiler.enter("opt.backend");
		m_backend->run(hp, cap, res);
		m_profiler.leave("opt.backend");
		if (res.status == 1) { m_profiler.leave("opt"); throw std::logic_error("optimize_edges: OBS_DIMS*nObs < number of unknown scalars (reference ASSERT_ABOVEEQ_, optimize_edges.h:355)"); }

		// write results back into the problem state (the reference optimises in place)
		for (size_t i = 0; i < nUnknowns_k2k; i++) rba_state.k2k_edges[run_k2k_edges[i]].inv_pose.loadFrom(&cd.edge_pose[i * PD]);
		for (size_t i = 0; i < nUnknowns_k2f; i++) for (int k = 0; k < L; k++) rba_state.all_lms[run_feat_ids[i]].rfp->pos[k] = cd.ulm_pos[i * L + k];
		for (size_t p = 0; p < nPairs; p++) {
			pose_flag_t &i2j = rba_state.spanning_tree.num[cd.pair_kfs[p].first][cd.pair_kfs[p].second], &j2i = rba_state.spanning_tree.num[cd.pair_kfs[p].second][cd.pair_kfs[p].first];
			i2j.pose.loadFrom(&cd.pose[(2 * p) * PD]); i2j.updated = true; j2i.pose.loadFrom(&cd.pose[(2 * p + 1) * PD]); j2i.updated = true;
		}
		rba_state.unknown_lms_inf_matrices.clear(); // S17 (:727-751)
		if (parameters.srba.cov_recovery == crpLandmarksApprox)
			for (size_t i = 0; i < nUnknowns_k2f; i++) if (cd.ulm_inf_valid[i]) { typename rba_problem_state_t::lm_inf_matrix_t &M = rba_state.unknown_lms_inf_matrices[run_feat_ids[i]]; for (int k = 0; k < L * L; k++) M.m[k] = cd.ulm_inf[i * L * L + k]; }

		out_info.num_observations = nObs; out_info.num_jacobians = res.num_jacobians; out_info.num_kf2kf_edges_optimized = run_k2k_edges.size(); out_info.num_kf2lm_edges_optimized = run_feat_ids.size();
		out_info.num_total_scalar_optimized = nUnknowns_scalars; out_info.num_span_tree_numeric_updates = res.num_span_tree_numeric_updates;
		out_info.total_sqr_error_init = res.total_sqr_error_init; out_info.total_sqr_error_final = res.total_sqr_error_final; out_info.obs_rmse = res.obs_rmse; out_info.lm = res;
		out_info.optimized_k2k_edge_indices.swap(run_k2k_edges); out_info.optimized_landmark_indices.swap(run_feat_ids);
		m_profiler.leave("opt");
		if (m_verbose_level >= 1) std::cout << "[OPT] Final RMSE=" << res.obs_rmse << " #iters=" << res.num_iters << "\n";
	}

private:
	rba_problem_state_t rba_state;
	mutable mrpt::utils::CTimeLogger m_profiler;
	mutable std::shared_ptr<numeric_backend> m_backend;
	int m_hip_device;
};

} // namespace srba
