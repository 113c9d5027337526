/*
 * RbaEngine.h -- srba::RbaEngine<KF2KF_POSE, LANDMARK, OBSERVATION, RBA_OPTIONS>: the reference's public engine API
 * (define_new_keyframe / optimize_local_area / parameters / getters; reference include/srba/RbaEngine.h:66-816) as a thin TYPED layer over
 *   graph::topology          all integer bookkeeping: graph, symbolic spanning trees, symbolic Jacobian structure (graph_topology.h)
 *   graph::capsule_builder   one optimize_edges() call -> flat integer tables of a srba_problem_capsule              (capsule_builder.h)
 *   numeric_backend          the Levenberg-Marquardt optimisation itself: on an MI355X through the C ABI of srba_hip.h (hip_backend.h)
 * This class owns only the payload the integer layer does not need -- edge poses, landmark coordinates, observation vectors, the numeric
 * spanning-tree poses -- in flat arrays indexed by the topology's ids, and moves numbers in and out of capsules.
 * There is no CPU optimiser behind it: without libsrba_hip / a GPU the first optimisation throws.
 *
 * Reference behaviour followed (what a caller can observe; the code organisation is not the reference's):
 *   define_new_keyframe          impl/define_new_keyframe.h:16-116     (incl. the stage-1 robust-kernel swap :67-87)
 *   edge initial values          impl/determine_kf2kf_edges_to_create.h:36-255
 *   edge-creation policies       ecps/local_areas_fixed_size.h:61-213, ecps/classic_linear_rba.h:50-118
 *   optimize_local_area          impl/optimize_local_area.h:15-58
 *   result write-back            impl/optimize_edges.h:526,538,727-751,768-792
 */
#pragma once
#include "capsule_builder.h"
#include "srba_options.h"
#include <cstring>
#include <fstream>
#include <functional>
#include <limits>
#include <memory>
#include <thread>

namespace srba {

// -------------------------------------------------------------------------------------------------
namespace internal {
/** sigma_max / sigma_min of a dense SYMMETRIC matrix (row-major n x n): its singular values are the moduli of its eigenvalues; cyclic Jacobi rotations.
 *  Stands where Eigen::JacobiSVD stands in the reference (optimize_edges.h:761-762). */
inline double symmetric_condition_number(std::vector<double> A, const size_t n) {
	if (n == 0) return 0;
	for (int sweep = 0; sweep < 60; sweep++) {
		double off = 0, diag = 0; for (size_t i = 0; i < n; i++) { diag += A[i * n + i] * A[i * n + i]; for (size_t j = i + 1; j < n; j++) off += A[i * n + j] * A[i * n + j]; }
		if (off <= 1e-30 * diag || off == 0) break;
		for (size_t p = 0; p + 1 < n; p++) for (size_t q = p + 1; q < n; q++) {
			const double apq = A[p * n + q]; if (apq == 0) continue;
			const double th = (A[q * n + q] - A[p * n + p]) / (2 * apq), t = (th >= 0 ? 1.0 : -1.0) / (std::fabs(th) + std::sqrt(th * th + 1)), c = 1 / std::sqrt(t * t + 1), sn = t * c;
			for (size_t k = 0; k < n; k++) { const double akp = A[k * n + p], akq = A[k * n + q]; A[k * n + p] = c * akp - sn * akq; A[k * n + q] = sn * akp + c * akq; }
			for (size_t k = 0; k < n; k++) { const double apk = A[p * n + k], aqk = A[q * n + k]; A[p * n + k] = c * apk - sn * aqk; A[q * n + k] = sn * apk + c * aqk; }
		}
	}
	double mx = 0, mn = std::numeric_limits<double>::infinity(); for (size_t i = 0; i < n; i++) { const double v = std::fabs(A[i * n + i]); mx = std::max(mx, v); mn = std::min(mn, v); }
	return mx / mn;
}
} // namespace internal
// Numeric back-end seam: replaces the reference's compile-time internal::solver_engine<> together with its CPU loops
// -------------------------------------------------------------------------------------------------
struct numeric_backend {
	virtual ~numeric_backend() {}
	/** Optimise the capsule in place (unknowns, spanning-tree poses, landmark information matrices) and fill the result. Throws on failure. */
	virtual void run(const srba_hip_params &params, srba_problem_capsule &capsule, srba_lm_result &result) = 0;
	/** Optimise n mutually independent capsules (RbaEngine<>::optimize_local_areas_batch). Default: one after the other; the GPU back-end uploads them as ONE batch. */
	virtual void run_batch(const srba_hip_params &params, srba_problem_capsule *capsules, int n, srba_lm_result *results) { for (int i = 0; i < n; i++) run(params, capsules[i], results[i]); }
	virtual const char *name() const = 0;
	/** Optional: where to record the back-end's own stage timings ("opt.backend.*"). */
	virtual void set_profiler(mrpt::utils::CTimeLogger *) {}
	/** Whole-map squared error over prepared path lists (eval_overall_squared_error). */
	virtual double eval_overall(const srba_hip_params &, const srba_overall_problem &) { throw std::runtime_error(std::string("numeric back-end '") + name() + "' does not implement eval_overall"); }
	/** Optional: one block array of the capsule just optimised as the back-end left it (3 = HAp, after the last Schur reduction for the Schur solvers; 4 = Hf; 5 = HApf).
	 *  Serves extra_results.hessian and HAp_condition_number; false if the back-end keeps nothing to read. */
	virtual bool read_blocks(int /*what*/, std::vector<double> & /*out*/) { return false; }
};
/** Adapter over a plain C function (tests plug the CPU oracle in from outside the product this way). */
struct function_backend : public numeric_backend {
	typedef int (*fn_t)(const srba_hip_params *, srba_problem_capsule *, srba_lm_result *);
	typedef int (*overall_fn_t)(const srba_hip_params *, const srba_overall_problem *, double *);
	fn_t fn; overall_fn_t overall_fn; std::string nm;
	function_backend(fn_t f, const std::string &n) : fn(f), overall_fn(NULL), nm(n) {}
	void run(const srba_hip_params &p, srba_problem_capsule &c, srba_lm_result &r) { if (fn(&p, &c, &r) != 0) throw std::runtime_error("numeric back-end '" + nm + "' failed"); }
	const char *name() const { return nm.c_str(); }
	double eval_overall(const srba_hip_params &p, const srba_overall_problem &q) { double v = 0; if (!overall_fn || overall_fn(&p, &q, &v) != 0) throw std::runtime_error("numeric back-end '" + nm +
		"': eval_overall failed or not provided"); return v; }
};
std::shared_ptr<numeric_backend> make_hip_backend(int device); // srba/hip_backend.h

namespace internal {
/** leaves a profiler section when the scope ends, also on exceptions */
struct profiler_scope {
	mrpt::utils::CTimeLogger &p; const char *name;
	profiler_scope(mrpt::utils::CTimeLogger &p_, const char *n) : p(p_), name(n) { p.enter(name); }
	~profiler_scope() { p.leave(name); }
};
} // namespace internal

// -------------------------------------------------------------------------------------------------
// Problem state: flat typed payload + the integer topology (reference TRBA_Problem_state, srba_types.h:548-785)
// -------------------------------------------------------------------------------------------------
template <class kf2kf_pose_t, class landmark_t, class obs_t, class RBA_OPTIONS>
struct TRBA_Problem_state {
	typedef typename kf2kf_pose_t::pose_t pose_t;
	typedef typename kf2kf_pose_traits<kf2kf_pose_t>::k2k_edge_t k2k_edge_t;
	typedef typename kf2kf_pose_traits<kf2kf_pose_t>::pose_flag_t pose_flag_t;
	typedef typename kf2kf_pose_traits<kf2kf_pose_t>::frameid2pose_map_t frameid2pose_map_t;
	typedef typename landmark_traits<landmark_t>::TRelativeLandmarkPos TRelativeLandmarkPos;
	typedef rba_joint_parameterization_traits_t<kf2kf_pose_t, landmark_t, obs_t> traits_t;
	typedef typename traits_t::k2f_edge_t k2f_edge_t;
	typedef typename traits_t::kf_observation_t kf_observation_t;
	typedef std::vector<k2k_edge_t> k2k_edges_deque_t; //!< (name kept from the reference; contiguous here)
	typedef mrpt::math::CMatrixFixed<landmark_t::LM_DIMS, landmark_t::LM_DIMS> lm_inf_matrix_t;

	graph::topology topo;                              //!< graph, symbolic spanning trees, symbolic Jacobians
	k2k_edges_deque_t k2k_edges;                       //!< [edge id] the unknown relative poses
	std::vector<TRelativeLandmarkPos> lm_table;        //!< [feature id] base key-frame + relative coordinates (id_frame_base invalid = never seen)
	std::vector<kf_observation_t> obs_table;           //!< [observation index]
	std::vector<pose_flag_t> num_pool;                 //!< numeric spanning-tree poses, slot = graph::st_entry::num
	std::vector<std::pair<TLandmarkID, lm_inf_matrix_t> > unknown_lms_inf_matrices; //!< landmarks of the last optimisation with an invertible Hf block, in unknown order

	/** ascending-id view over the known or the unknown landmarks (what the reference exposes as a std::map id -> TRelativeLandmarkPos) */
	class TRelativeLandmarkPosMap {
	public:
		struct item { TLandmarkID first; const TRelativeLandmarkPos &second; const item *operator->() const { return this; } };
		class const_iterator {
			const TRelativeLandmarkPosMap *m; size_t i;
		public:
			const_iterator(const TRelativeLandmarkPosMap *m_, size_t i_) : m(m_), i(i_) {}
			item operator*() const { const item it = {m->m_ids[i], (*m->m_table)[m->m_ids[i]]}; return it; }
			item operator->() const { return **this; }
			const_iterator &operator++() { ++i; return *this; }
			const_iterator operator++(int) { const_iterator c(*this); ++i; return c; }
			bool operator==(const const_iterator &o) const { return i == o.i; }
			bool operator!=(const const_iterator &o) const { return i != o.i; }
		};
		typedef const_iterator iterator;
		TRelativeLandmarkPosMap() : m_table(NULL) {}
		size_t size() const { return m_ids.size(); }
		bool empty() const { return m_ids.empty(); }
		const_iterator begin() const { return const_iterator(this, 0); }
		const_iterator end() const { return const_iterator(this, m_ids.size()); }
		const_iterator find(TLandmarkID id) const { const std::vector<TLandmarkID>::const_iterator p = std::lower_bound(m_ids.begin(), m_ids.end(), id);
			return (p != m_ids.end() && *p == id) ? const_iterator(this, (size_t)(p - m_ids.begin())) : end(); }
		// engine side
		void bind(const std::vector<TRelativeLandmarkPos> *t) { m_table = t; }
		void add(TLandmarkID id) { if (m_ids.empty() || m_ids.back() < id) m_ids.push_back(id); else m_ids.insert(std::lower_bound(m_ids.begin(), m_ids.end(), id), id); }
		void clear() { m_ids.clear(); }
	private:
		std::vector<TLandmarkID> m_ids; const std::vector<TRelativeLandmarkPos> *m_table;
	};
	TRelativeLandmarkPosMap known_lms, unknown_lms;

	/** reference: deque<keyframe_info>; user code only asks for its size */
	struct keyframes_view { const graph::topology *t; size_t size() const { return t->n_keyframes(); } bool empty() const { return !size(); } } keyframes;
	/** reference: deque<k2f_edge_t>; element access builds the record from the flat tables */
	struct observations_view {
		const TRBA_Problem_state *s;
		size_t size() const { return s->obs_table.size(); }
		bool empty() const { return s->obs_table.empty(); }
		k2f_edge_t operator[](size_t i) const { k2f_edge_t e; e.obs = s->obs_table[i]; e.feat_has_known_rel_pos = s->topo.obs_known[i] != 0;
			e.is_first_obs_of_unknown = s->topo.obs_first_of_unknown[i] != 0; e.feat_rel_pos = &s->lm_table[e.obs.obs.feat_id]; return e; }
	} all_observations;

	/** read access to the spanning trees in the reference's terms */
	struct TSpanningTree {
		const TRBA_Problem_state *s;
		/** next_edge[src][trg] (false if trg is not within max_tree_depth of src) */
		bool get_next_edge(TKeyFrameID src, TKeyFrameID trg, TSpanTreeEntry &out) const { const graph::st_entry *e = s->topo.st.find((graph::id32)src, (graph::id32)trg); if (!e) return false;
			out.next = e->next; out.distance = e->dist; return true; }
		/** all_edges[max(a,b)][min(a,b)]: edge ids of the stored shortest path, walking from the larger id */
		bool get_path(TKeyFrameID a, TKeyFrameID b, std::vector<size_t> &edge_ids) const {
			edge_ids.clear(); const graph::id32 hi = (graph::id32)std::max(a, b), lo = (graph::id32)std::min(a, b);
			const graph::st_entry *e = s->topo.st.find(hi, lo); if (!e || e->path == graph::NIL) return false;
			for (uint32_t k = 0; k < e->path_len; k++) edge_ids.push_back(s->topo.path_pool[e->path + k]);
			return true;
		}
		/** num[src][trg]: pose of trg as seen from src, NULL if never requested */
		const pose_flag_t *get_num(TKeyFrameID src, TKeyFrameID trg) const { const int32_t slot = s->topo.find_num((graph::id32)src, (graph::id32)trg);
			return (slot >= 0 && (size_t)slot < s->num_pool.size()) ? &s->num_pool[slot] : (const pose_flag_t *)0; }
		/** number of key-frames in each symbolic tree: min / max / mean / standard deviation (reference impl/spantree_misc.h) */
		void get_stats(size_t &num_nodes_min, size_t &num_nodes_max, double &num_nodes_mean, double &num_nodes_std) const {
			num_nodes_min = num_nodes_max = 0; num_nodes_mean = num_nodes_std = 0; const size_t n = s->topo.st.rows(); if (!n) return;
			num_nodes_min = std::numeric_limits<size_t>::max(); double sum = 0, sum2 = 0;
			for (size_t k = 0; k < n; k++) { const size_t c = s->topo.st.len((graph::id32)k); num_nodes_min = std::min(num_nodes_min, c); num_nodes_max = std::max(num_nodes_max, c); sum += c;
				sum2 += (double)c * c; }
			num_nodes_mean = sum / n; num_nodes_std = std::sqrt(std::max(0.0, sum2 / n - num_nodes_mean * num_nodes_mean));
		}
		/** plain-text listing of every tree: "src: trg(next,dist) ..." */
		bool dump_as_text_to_file(const std::string &file) const {
			FILE *f = std::fopen(file.c_str(), "wt"); if (!f) return false;
			for (size_t k = 0; k < s->topo.st.rows(); k++) { std::fprintf(f, "%zu:", k); const graph::st_entry *r = s->topo.st.row((graph::id32)k); for (size_t i = 0;
				i < s->topo.st.len((graph::id32)k); i++) std::fprintf(f, " %u(%u,%u)", r[i].trg, r[i].next, r[i].dist); std::fprintf(f, "\n"); }
			std::fclose(f); return true;
		}
		/** Graphviz file with one cluster per requested root (all roots if the list is empty) */
		bool save_as_dot_file(const std::string &file, const std::vector<TKeyFrameID> &roots = std::vector<TKeyFrameID>()) const {
			FILE *f = std::fopen(file.c_str(), "wt"); if (!f) return false;
			std::fprintf(f, "digraph G {\n");
			std::vector<TKeyFrameID> rs(roots); if (rs.empty()) for (size_t k = 0; k < s->topo.st.rows(); k++) rs.push_back(k);
			for (size_t q = 0; q < rs.size(); q++) {
				const graph::id32 root = (graph::id32)rs[q]; std::fprintf(f, " subgraph cluster_%u { label=\"root %u\";\n", root, root);
				const graph::st_entry *r = s->topo.st.row(root);
				for (size_t i = 0; i < s->topo.st.len(root); i++) { TSpanTreeEntry back; if (get_next_edge(r[i].trg, root, back)) std::fprintf(f, "  n%u_%u -> n%u_%llu;\n", root, r[i].trg, root,
					(unsigned long long)back.next); }
				std::fprintf(f, " }\n");
			}
			std::fprintf(f, "}\n"); std::fclose(f); return true;
		}
	} spanning_tree;

	TRBA_Problem_state() { wire(); }
	void clear() { topo.clear(); k2k_edges.clear(); lm_table.clear(); obs_table.clear(); num_pool.clear(); unknown_lms_inf_matrices.clear(); known_lms.clear(); unknown_lms.clear(); wire(); }
	bool are_keyframes_connected(const TKeyFrameID id1, const TKeyFrameID id2) const { return topo.connected((graph::id32)id1, (graph::id32)id2); }
	/** degree (number of kf2kf edges) statistics over all key-frames */
	void compute_all_node_degrees(double &out_mean_degree, double &out_std_degree, double &out_max_degree) const {
		out_mean_degree = out_std_degree = out_max_degree = 0; const size_t n = topo.n_keyframes(); if (!n) return;
		double sum = 0, sum2 = 0; for (size_t k = 0; k < n; k++) { const double d = topo.kf_degree[k]; sum += d; sum2 += d * d; out_max_degree = std::max(out_max_degree, d); }
		out_mean_degree = sum / n; out_std_degree = std::sqrt(std::max(0.0, sum2 / n - out_mean_degree * out_mean_degree));
	}
	const lm_inf_matrix_t *find_inf_matrix(TLandmarkID id) const { for (size_t i = 0; i < unknown_lms_inf_matrices.size();
		i++) if (unknown_lms_inf_matrices[i].first == id) return &unknown_lms_inf_matrices[i].second; return NULL; }
	pose_flag_t &num_at(int32_t slot) { if ((size_t)slot >= num_pool.size()) num_pool.resize((size_t)topo.num_slots); return num_pool[slot]; }
private:
	void wire() { known_lms.bind(&lm_table); unknown_lms.bind(&lm_table); keyframes.t = &topo; all_observations.s = this; spanning_tree.s = this; }
	TRBA_Problem_state(const TRBA_Problem_state &); TRBA_Problem_state &operator=(const TRBA_Problem_state &);
};

// -------------------------------------------------------------------------------------------------
// Edge-creation policies
// -------------------------------------------------------------------------------------------------
namespace ecps {
/** Sub-maps of fixed size: every key-frame hangs from the centre of its area (first key-frame of each block of submap_size ids); a key-frame that
 *  opens a new area, or that re-observes landmarks of another area far away in the graph, adds centre-to-centre edges (ecps/local_areas_fixed_size.h). */
struct local_areas_fixed_size {
	struct parameters_t : public mrpt::utils::CLoadableOptions {
		size_t submap_size, min_obs_to_loop_closure; parameters_t() : submap_size(15), min_obs_to_loop_closure(4) {}
		void loadFromConfigFile(const mrpt::utils::CConfigFileBase &source, const std::string &section) override { submap_size = (size_t)source.read<uint64_t>(section, "submap_size", submap_size);
			min_obs_to_loop_closure = (size_t)source.read<uint64_t>(section, "min_obs_to_loop_closure", min_obs_to_loop_closure); }
		void saveToConfigFile(mrpt::utils::CConfigFileBase &out, const std::string &section) const override { out.write(section, "submap_size", (uint64_t)submap_size, 30, 30,
			"Key-frames per sub-map"); out.write(section, "min_obs_to_loop_closure", (uint64_t)min_obs_to_loop_closure, 30, 30,
			"shared landmark observations needed before a loop-closure edge is created"); }
	};
	TKeyFrameID get_center_kf_for_kf(const TKeyFrameID kf_id, const parameters_t &params) const { return params.submap_size * (kf_id / params.submap_size); }

	template <class traits_t, class rba_engine_t>
	void eval(const TKeyFrameID new_kf_id, const typename traits_t::new_kf_observations_t &obs, std::vector<TNewEdgeInfo> &out_edges, rba_engine_t &rba_engine, const parameters_t &params) {
		ASSERT_(new_kf_id >= 1);
		const TKeyFrameID my_centre = get_center_kf_for_kf(new_kf_id, params);
		// votes: observations of already-known landmarks per base key-frame, most voted first (ties: lower id first)
		std::vector<std::pair<TKeyFrameID, size_t> > votes; rba_engine.count_observations_per_base_kf(obs, votes);
		struct area { TKeyFrameID centre, best_base; size_t votes; bool only_centre_is_base; };
		std::vector<area> areas;
		for (size_t v = 0; v < votes.size(); v++) {
			const TKeyFrameID c = get_center_kf_for_kf(votes[v].first, params);
			size_t a = 0; while (a < areas.size() && areas[a].centre != c) a++;
			if (a == areas.size()) { area n = {c, votes[v].first, 0, true}; areas.push_back(n); } // first base met in an area is its most voted one
			areas[a].votes += votes[v].second; if (votes[v].first != c) areas[a].only_centre_is_base = false;
		}
		struct by_centre { bool operator()(const area &x, const area &y) const { return x.centre < y.centre; } };
		struct by_votes { bool operator()(const area &x, const area &y) const { return x.votes > y.votes; } };
		std::sort(areas.begin(), areas.end(), by_centre()); std::stable_sort(areas.begin(), areas.end(), by_votes());
		// (1) a key-frame that does not open an area is linked to the centre of its area
		if (my_centre != new_kf_id) {
			TNewEdgeInfo link; link.has_approx_init_val = false;
			link.id = rba_engine.create_kf2kf_edge(new_kf_id, TPairKeyFrameID(my_centre, new_kf_id), obs);
			out_edges.push_back(link);
		}
		// (2) centre-to-centre edges towards areas whose centre is farther than what the spanning trees (plus the links of this very step) will cover
		const topo_dist_t reach = rba_engine.parameters.srba.max_tree_depth + 1;
		for (size_t a = 0; a < areas.size(); a++) {
			const TKeyFrameID remote = areas[a].centre; if (remote == my_centre) continue;
			const topo_dist_t known_dist = rba_engine.symbolic_distance(my_centre, remote);
			topo_dist_t slack = 2; if (my_centre == new_kf_id) slack--; if (areas[a].only_centre_is_base) slack--;
			if (known_dist < reach - slack || areas[a].votes < params.min_obs_to_loop_closure) continue;
			TNewEdgeInfo link; link.has_approx_init_val = false;
			link.id = rba_engine.create_kf2kf_edge(my_centre, TPairKeyFrameID(remote, my_centre), obs);
			link.loopclosure_observer_kf = new_kf_id; link.loopclosure_base_kf = areas[a].best_base;
			out_edges.push_back(link);
		}
		ASSERTMSG_(out_edges.size() >= 1, "Error for new KF: no suitable linking KF found with the minimum number of common observations: the node becomes isolated of the graph!");
	}
};

/** The classic linear graph: key-frame n hangs from n-1 (starting at the same pose), plus a direct edge to every base key-frame of re-observed
 *  landmarks that is out of reach of the spanning trees (ecps/classic_linear_rba.h). */
struct classic_linear_rba {
	struct parameters_t : public mrpt::utils::CLoadableOptions {
		size_t min_obs_to_loop_closure; parameters_t() : min_obs_to_loop_closure(4) {}
		void loadFromConfigFile(const mrpt::utils::CConfigFileBase &source, const std::string &section) override { min_obs_to_loop_closure = (size_t)source.read<uint64_t>(section,
			"min_obs_to_loop_closure", min_obs_to_loop_closure); }
		void saveToConfigFile(mrpt::utils::CConfigFileBase &out, const std::string &section) const override { out.write(section, "min_obs_to_loop_closure", (uint64_t)min_obs_to_loop_closure, 30, 30,
			"shared landmark observations needed before a loop-closure edge is created"); }
	};

	template <class traits_t, class rba_engine_t>
	void eval(const TKeyFrameID new_kf_id, const typename traits_t::new_kf_observations_t &obs, std::vector<TNewEdgeInfo> &out_edges, rba_engine_t &rba_engine, const parameters_t &params) {
		ASSERT_(new_kf_id >= 1);
		TNewEdgeInfo chain; chain.has_approx_init_val = true; // identity: the new key-frame starts where the previous one is
		chain.id = rba_engine.create_kf2kf_edge(new_kf_id, TPairKeyFrameID(new_kf_id - 1, new_kf_id), obs, typename traits_t::original_kf2kf_pose_t::pose_t());
		out_edges.push_back(chain);
		std::vector<std::pair<TKeyFrameID, size_t> > votes; rba_engine.count_observations_per_base_kf(obs, votes);
		const topo_dist_t reach = rba_engine.parameters.srba.max_tree_depth + 1;
		for (size_t v = 0; v < votes.size(); v++) {
			if (rba_engine.symbolic_distance(new_kf_id, votes[v].first) < reach || votes[v].second < params.min_obs_to_loop_closure) continue;
			TNewEdgeInfo link; link.has_approx_init_val = false;
			link.id = rba_engine.create_kf2kf_edge(new_kf_id, TPairKeyFrameID(votes[v].first, new_kf_id), obs);
			out_edges.push_back(link);
		}
	}
};
} // namespace ecps

/** Default compile-time options (reference RbaEngine.h:39-45) */
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
	typedef typename rba_problem_state_t::k2k_edges_deque_t k2k_edges_deque_t; typedef k2k_edges_deque_t k2k_edges_t;
	typedef typename kf2kf_pose_traits_t::pose_flag_t pose_flag_t; typedef typename kf2kf_pose_traits_t::frameid2pose_map_t frameid2pose_map_t;
	typedef typename rba_problem_state_t::TRelativeLandmarkPosMap TRelativeLandmarkPosMap; typedef typename landmark_traits_t::TRelativeLandmarkPos TRelativeLandmarkPos;
	typedef typename traits_t::new_kf_observation_t new_kf_observation_t; typedef typename traits_t::new_kf_observations_t new_kf_observations_t;
	typedef typename kf2kf_pose_traits_t::array_pose_t array_pose_t; typedef typename landmark_traits_t::array_landmark_t array_landmark_t;
		typedef typename observation_traits_t::array_obs_t array_obs_t;
	typedef typename observation_traits_t::residual_t residual_t; typedef typename observation_traits_t::vector_residuals_t vector_residuals_t;

	RbaEngine() : m_verbose_level(1), m_profiler(true), m_hip_device(-1) {}

	/** What one optimisation reports (reference RbaEngine.h:125-177) */
	struct TOptimizeExtraOutputInfo {
		TOptimizeExtraOutputInfo() { clear(); }
		size_t num_observations, num_jacobians, num_kf2kf_edges_optimized, num_kf2lm_edges_optimized, num_total_scalar_optimized, num_kf_optimized, num_lm_optimized, num_span_tree_numeric_updates;
		double obs_rmse, total_sqr_error_init, total_sqr_error_final, HAp_condition_number;
		size_t sparsity_dh_dAp_nnz, sparsity_dh_dAp_max_size, sparsity_dh_df_nnz, sparsity_dh_df_max_size, sparsity_HAp_nnz, sparsity_HAp_max_size, sparsity_Hf_nnz, sparsity_Hf_max_size,
			sparsity_HApf_nnz, sparsity_HApf_max_size;
		std::vector<size_t> optimized_k2k_edge_indices, optimized_landmark_indices;
		typename RBA_OPTIONS::solver_t::extra_results_t extra_results;
		srba_lm_result lm; //!< (extension) raw record of the numeric back-end: LM trials, lambda, per-trial chi2 / rho trace
		void clear() {
			num_observations = num_jacobians = num_kf2kf_edges_optimized = num_kf2lm_edges_optimized = num_total_scalar_optimized = num_kf_optimized = num_lm_optimized = num_span_tree_numeric_updates
				= 0;
			obs_rmse = 0; total_sqr_error_init = total_sqr_error_final = HAp_condition_number = 0;
			sparsity_dh_dAp_nnz = sparsity_dh_dAp_max_size = sparsity_dh_df_nnz = sparsity_dh_df_max_size = sparsity_HAp_nnz = sparsity_HAp_max_size = sparsity_Hf_nnz = sparsity_Hf_max_size =
				sparsity_HApf_nnz = sparsity_HApf_max_size = 0;
			optimized_k2k_edge_indices.clear(); optimized_landmark_indices.clear(); extra_results.clear(); std::memset(&lm, 0, sizeof(lm));
		}
	};
	/** What define_new_keyframe() reports (reference RbaEngine.h:180-195; clear() leaves the stage-1 results alone there too) */
	struct TNewKeyFrameInfo {
		TKeyFrameID kf_id; std::vector<TNewEdgeInfo> created_edge_ids; TOptimizeExtraOutputInfo optimize_results, optimize_results_stg1;
		void clear() { kf_id = static_cast<TKeyFrameID>(-1); created_edge_ids.clear(); optimize_results.clear(); }
	};
	struct TOptimizeLocalAreaParams {
		bool optimize_k2k_edges, optimize_landmarks; TKeyFrameID max_visitable_kf_id; size_t dont_optimize_landmarks_seen_less_than_n_times;
		TOptimizeLocalAreaParams() : optimize_k2k_edges(true), optimize_landmarks(true), max_visitable_kf_id(static_cast<TKeyFrameID>(-1)), dont_optimize_landmarks_seen_less_than_n_times(2) {}
	};
	/** Run-time parameters (reference RbaEngine.h:424-460; default VALUES are those of impl/rba_problem_common.h:35-56) */
	struct TSRBAParameters : public mrpt::utils::CLoadableOptions {
		topo_dist_t max_tree_depth, max_optimize_depth;
		bool optimize_new_edges_alone, use_robust_kernel, use_robust_kernel_stage1;
		double kernel_param; size_t max_iters; double max_error_per_obs_to_stop, max_rho, max_lambda, min_error_reduction_ratio_to_relinearize;
		bool numeric_jacobians; void (*feedback_user_iteration)(unsigned int iter, const double total_sq_err, const double mean_sqroot_error);
		bool compute_condition_number, compute_sparsity_stats; double max_rmse_show_red_warning; TCovarianceRecoveryPolicy cov_recovery;
		/** (extension, default false = reference behaviour) inside the LM loop the reference refreshes only the spanning-tree poses that Jacobian blocks of the
		 *  optimised columns read (optimize_edges.h:550-566); residuals of observations whose observer-side edge is not optimised are then evaluated with
		 *  pre-step poses. true: every pose a residual reads is refreshed as well. */
		bool refresh_all_read_poses;
		/** (extension, default false = reference behaviour) every LM trial recomputes BOTH poses num[r][t], num[t][r] of a refreshed spanning-tree pair, but a rejected step restores only
		 *  the ones Jacobian blocks read (optimize_edges.h:664-670): the twin keeps the value of the rejected trial until some later optimisation refreshes it, and meanwhile
		 *  determine_kf2kf_edges_to_create may seed the next edge from it. true: back up and restore both. */
		bool restore_spanning_tree_twins;
		/** (extension, default false = reference behaviour) the Schur solvers reduce minus_grad in place (impl/schur.h:248-265, :294) and nothing recomputes it after a rejected
		 *  LM trial (impl/optimize_edges.h:658-690): every retry then solves for an already reduced gradient, and a rejection in mid-descent sends the window (and the map) to
		 *  infinity. true: every solve starts from the gradient compute_minus_gradient produced (SRBA_EXT_SCHUR_KEEPS_GRADIENT of the numeric back-end). */
		bool schur_keeps_gradient;
		/** (extension, default false = reference behaviour) initial value of a loop-closure edge between two EXISTING area centres (impl/determine_kf2kf_edges_to_create.h:196-248):
		 *  the reference stores pose_local_wrt_remote, i.e. the pose of `to` with respect to `from`, unless `to` is the key-frame being inserted -- the inverse of what inv_pose means
		 *  everywhere else (:60-62, :204-206) -- and takes the observer's pose in its own area as the identity, because num[centre][new key-frame] does not exist before the first
		 *  numeric update. The edge then starts metres away, is marked has_approx_init_val (so it is not initialised alone, define_new_keyframe.h:75-76) and the landmark families
		 *  do not recover. true: inv_pose = pose of `from` w.r.t. `to`, and the observer's pose comes from the initial value just given to its own new edge. */
		bool consistent_loop_closure_init;
		/** (extension, default false) fill TOptimizeExtraOutputInfo::extra_results.hessian (dense, row-major, both triangles) after every optimisation: the reference's solvers hand their
		 *  last system matrix over for free (lev-marq_solvers.h:204-208, :586-590); here it has to be downloaded from the device, so it is done on request only. */
		bool return_hessian;
		TSRBAParameters() : max_tree_depth(4), max_optimize_depth(4), optimize_new_edges_alone(true), use_robust_kernel(false), use_robust_kernel_stage1(false), kernel_param(3.), max_iters(20),
			max_error_per_obs_to_stop(1e-6), max_rho(10.0), max_lambda(1e20), min_error_reduction_ratio_to_relinearize(0.01), numeric_jacobians(false), feedback_user_iteration(NULL),
			compute_condition_number(false), compute_sparsity_stats(false), max_rmse_show_red_warning(0.5), cov_recovery(crpLandmarksApprox), refresh_all_read_poses(false),
				restore_spanning_tree_twins(false), schur_keeps_gradient(false), consistent_loop_closure_init(false), return_hessian(false) {}
		/** keys of the reference's configuration files (impl/rba_problem_common.h:60-92); cov_recovery by enumerator name or number */
		void loadFromConfigFile(const mrpt::utils::CConfigFileBase &source, const std::string &section) override {
			max_tree_depth = (topo_dist_t)source.read<uint64_t>(section, "max_tree_depth", max_tree_depth); max_optimize_depth = (topo_dist_t)source.read<uint64_t>(section, "max_optimize_depth",
				max_optimize_depth);
			optimize_new_edges_alone = source.read<bool>(section, "optimize_new_edges_alone", optimize_new_edges_alone);
			use_robust_kernel = source.read<bool>(section, "use_robust_kernel", use_robust_kernel); use_robust_kernel_stage1 = source.read<bool>(section, "use_robust_kernel_stage1",
				use_robust_kernel_stage1);
			max_rho = source.read<double>(section, "max_rho", max_rho); max_lambda = source.read<double>(section, "max_lambda", max_lambda); kernel_param = source.read<double>(section,
				"kernel_param", kernel_param);
			max_iters = (size_t)source.read<uint64_t>(section, "max_iters", max_iters); max_error_per_obs_to_stop = source.read<double>(section, "max_error_per_obs_to_stop",
				max_error_per_obs_to_stop);
			const std::string cr = source.read<std::string>(section, "cov_recovery", cov_recovery == crpNone ? "crpNone" : "crpLandmarksApprox");
			cov_recovery = (cr == "crpNone" || cr == "0") ? crpNone : crpLandmarksApprox;
		}
		void saveToConfigFile(mrpt::utils::CConfigFileBase &out, const std::string &section) const override {
			out.write(section, "max_tree_depth", (uint64_t)max_tree_depth, 30, 30, "depth limit of the spanning trees kept per key-frame"); out.write(section, "max_optimize_depth",
				(uint64_t)max_optimize_depth, 30, 30, "radius (in kf2kf edges) of the local area that is optimised");
			out.write(section, "optimize_new_edges_alone", optimize_new_edges_alone, 30, 30, "first optimise each new kf2kf edge on its own");
			out.write(section, "use_robust_kernel", use_robust_kernel, 30, 30, "pseudo-Huber robust cost in the local-area optimisation"); out.write(section, "use_robust_kernel_stage1",
				use_robust_kernel_stage1, 30, 30, "pseudo-Huber robust cost while new edges are optimised alone");
			out.write(section, "kernel_param", kernel_param, 30, 30, "threshold of the pseudo-Huber cost"); out.write(section, "max_rho", max_rho, 30, 30,
				"LM stops once the gain ratio rho exceeds this");
			out.write(section, "max_lambda", max_lambda, 30, 30, "LM stops once the damping lambda exceeds this"); out.write(section, "max_iters", (uint64_t)max_iters, 30, 30,
				"upper bound on LM iterations");
			out.write(section, "max_error_per_obs_to_stop", max_error_per_obs_to_stop, 30, 30, "LM stops below this RMSE per observation");
			out.write(section, "cov_recovery", std::string(cov_recovery == crpNone ? "crpNone" : "crpLandmarksApprox"), 30, 30, "crpNone | crpLandmarksApprox");
		}
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

	// =================================================================================== main API
	/** New key-frame with its observations: graph + symbolic update, then (optionally) the two-stage local optimisation. */
	void define_new_keyframe(const new_kf_observations_t &obs, TNewKeyFrameInfo &out_new_kf_info, const bool run_local_optimization = true) {
		internal::profiler_scope ps(m_profiler, "define_new_keyframe");
		out_new_kf_info.clear();
		const TKeyFrameID new_kf_id = alloc_keyframe();
		std::vector<TNewEdgeInfo> created;
		{ internal::profiler_scope p2(m_profiler, "define_new_keyframe.determine_edges"); determine_kf2kf_edges_to_create(new_kf_id, obs, created); }
		{ internal::profiler_scope p2(m_profiler, "define_new_keyframe.add_observations");
		  for (typename new_kf_observations_t::const_iterator o = obs.begin(); o != obs.end(); ++o)
			add_observation(new_kf_id, o->obs, o->is_fixed ? &o->feat_rel_pos : NULL, o->is_unknown_with_init_val ? &o->feat_rel_pos : NULL); }
		if (run_local_optimization) {
			if (parameters.srba.optimize_new_edges_alone) { // stage 1: every new edge that got no initial value is optimised alone, with the stage-1 kernel switch
				internal::profiler_scope p2(m_profiler, "define_new_keyframe.opt_new_edges");
				struct kernel_swap { bool &flag; const bool saved; kernel_swap(bool &f, bool v) : flag(f), saved(f) { flag = v; } ~kernel_swap() { flag = saved; } }
					swap(parameters.srba.use_robust_kernel, parameters.srba.use_robust_kernel_stage1);
				std::vector<size_t> one_edge(1), no_landmarks;
				for (size_t i = 0; i < created.size(); i++) if (!created[i].has_approx_init_val) { one_edge[0] = created[i].id; m_capsule_stage = 1; optimize_edges(one_edge, no_landmarks,
					out_new_kf_info.optimize_results_stg1); }
			}
			internal::profiler_scope p2(m_profiler, "define_new_keyframe.optimize");
			optimize_local_area(new_kf_id, (unsigned int)parameters.srba.max_optimize_depth, out_new_kf_info.optimize_results, TOptimizeLocalAreaParams());
		}
		out_new_kf_info.kf_id = new_kf_id; out_new_kf_info.created_edge_ids.swap(created);
		if (m_verbose_level >= 1) std::cout << "[define_new_keyframe] Done. New KF #" << out_new_kf_info.kf_id << " with " << out_new_kf_info.created_edge_ids.size() << " new edges.\n";
	}

	/** Optimise every kf2kf edge touching, and every landmark seen often enough from, the key-frames within win_size of root_id. */
	void optimize_local_area(const TKeyFrameID root_id, const unsigned int win_size, TOptimizeExtraOutputInfo &out_info, const TOptimizeLocalAreaParams &params = TOptimizeLocalAreaParams(),
		const std::vector<size_t> &observation_indices_to_optimize = std::vector<size_t>()) {
		internal::profiler_scope ps(m_profiler, "optimize_local_area");
		const bool use_prebuilt_st = (win_size <= parameters.srba.max_tree_depth);
		if (!use_prebuilt_st && m_verbose_level >= 1) std::cout << "[optimize_local_area] *WARNING* Optimize win_size > max_tree_depth of prebuilt spanning trees. This is not efficient!\n";
		const graph::window_params wp = {params.optimize_k2k_edges, params.optimize_landmarks, params.dont_optimize_landmarks_seen_less_than_n_times, params.max_visitable_kf_id};
		rba_state.topo.select_local_area(graph::topology::narrow(root_id), win_size, use_prebuilt_st, wp, m_sel_edges, m_sel_lms);
		m_capsule_stage = 0;
		if (!m_sel_edges.empty() || !m_sel_lms.empty()) optimize_edges(m_sel_edges, m_sel_lms, out_info, observation_indices_to_optimize);
	}

	void clear() { rba_state.clear(); }
	TKeyFrameID alloc_keyframe() { return rba_state.topo.new_keyframe(); }
	/** New kf2kf edge new_edge = (from, to) attached to key-frame new_kf_id (one of its ends): O(1) allocation + symbolic spanning-tree update. */
	size_t create_kf2kf_edge(const TKeyFrameID new_kf_id, const TPairKeyFrameID &new_edge, const new_kf_observations_t &obs, const pose_t &init_inv_pose_val = pose_t()) {
		(void)obs;
		const graph::id32 from = graph::topology::narrow(new_edge.first), to = graph::topology::narrow(new_edge.second), n = graph::topology::narrow(new_kf_id);
		ASSERT_(n == from || n == to);
		const graph::id32 e = rba_state.topo.new_edge(from, to);
		k2k_edge_t rec; rec.from = new_edge.first; rec.to = new_edge.second; rec.inv_pose = init_inv_pose_val; rec.id = e; rba_state.k2k_edges.push_back(rec);
		internal::profiler_scope ps(m_profiler, "define_new_keyframe.st.update_symbolic");
		rba_state.topo.st_attach_edge(n, n == from ? to : from, (uint32_t)parameters.srba.max_tree_depth);
		return e;
	}
	/** key-frame ids of a shortest path src -> trg over the whole graph (breadth-first); false if they are not connected */
	bool find_path_bfs(const TKeyFrameID src_kf, const TKeyFrameID trg_kf, std::vector<TKeyFrameID> &found_path) const {
		std::vector<graph::id32> kfs; const bool ok = const_cast<graph::topology &>(rba_state.topo).shortest_path(graph::topology::narrow(src_kf), graph::topology::narrow(trg_kf), NULL, &kfs);
		found_path.assign(kfs.begin(), kfs.end()); return ok;
	}
	/** pose of kf_query as seen from kf_reference as last computed by an optimisation, or NULL */
	const pose_t *get_kf_relative_pose(const TKeyFrameID kf_query, const TKeyFrameID kf_reference) const {
		const pose_flag_t *p = rba_state.spanning_tree.get_num(kf_reference, kf_query); return p ? &p->pose : NULL;
	}
	/** symbolic spanning-tree distance, "infinite" when out of the trees' reach (what the edge-creation policies ask) */
	topo_dist_t symbolic_distance(const TKeyFrameID a, const TKeyFrameID b) const {
		if (a >= rba_state.topo.n_keyframes() || b >= rba_state.topo.n_keyframes()) return std::numeric_limits<topo_dist_t>::max();
		const uint32_t d = rba_state.topo.st_distance((graph::id32)a, (graph::id32)b); return d == 0xffffffffu ? std::numeric_limits<topo_dist_t>::max() : d;
	}
	/** (base key-frame, number of observations) over the observations of already-known landmarks, most observed first, ties by ascending id (impl/make_ordered_list_base_kfs.h) */
	void count_observations_per_base_kf(const new_kf_observations_t &obs, std::vector<std::pair<TKeyFrameID, size_t> > &out) const {
		out.clear(); std::vector<TKeyFrameID> bases;
		for (typename new_kf_observations_t::const_iterator o = obs.begin(); o != obs.end(); ++o) { const TLandmarkID id = o->obs.feat_id;
			if (id < rba_state.topo.lm_base.size() && rba_state.topo.lm_base[id] != graph::NIL) bases.push_back(rba_state.topo.lm_base[id]); }
		std::sort(bases.begin(), bases.end());
		for (size_t i = 0; i < bases.size();) { size_t j = i; while (j < bases.size() && bases[j] == bases[i]) j++; out.push_back(std::make_pair(bases[i], j - i)); i = j; }
		struct more_votes { bool operator()(const std::pair<TKeyFrameID, size_t> &a, const std::pair<TKeyFrameID, size_t> &b) const { return a.second > b.second; } };
		std::stable_sort(out.begin(), out.end(), more_votes());
	}

	/** Generic breadth-first visit of the neighbourhood of root_id up to max_distance (reference impl/bfs_visitor.h:21-177): key-frames in BFS order -- or, with
	 *  rely_on_prebuilt_spanning_trees, the root and then its spanning tree by ascending id -- each followed by its key-frame->feature edges and its kf2kf edges. */
	template <class KF_VISITOR, class FEAT_VISITOR, class K2K_EDGE_VISITOR, class K2F_EDGE_VISITOR>
	void bfs_visitor(const TKeyFrameID root_id, const topo_dist_t max_distance, const bool rely_on_prebuilt_spanning_trees, KF_VISITOR &kf_visitor, FEAT_VISITOR &feat_visitor,
		K2K_EDGE_VISITOR &k2k_edge_visitor, K2F_EDGE_VISITOR &k2f_edge_visitor) const {
		const graph::topology &T = rba_state.topo; const graph::id32 root = graph::topology::narrow(root_id); if (root >= T.n_keyframes()) return;
		std::vector<char> lm_done(T.lm_base.size(), 0), edge_done(T.n_edges(), 0), kf_done(T.n_keyframes(), 0);
		std::vector<std::pair<graph::id32, topo_dist_t> > order;
		struct visit { // features and key-frame->feature edges of one key-frame
			static void features(const rba_engine_t &self, graph::id32 kf, topo_dist_t d, std::vector<char> &lm_done, FEAT_VISITOR &fv, K2F_EDGE_VISITOR &ev) {
				const graph::topology &T = self.rba_state.topo;
				for (graph::id32 o = T.kf_obs_head[kf]; o != graph::NIL; o = T.obs_next_in_kf[o]) {
					const graph::id32 lm = T.obs_lm[o];
					if (!lm_done[lm]) { lm_done[lm] = 1; if (fv.visit_filter_feat(lm, d)) fv.visit_feat(lm, d); }
					const k2f_edge_t ed = self.rba_state.all_observations[o];
					if (ev.visit_filter_k2f(kf, &ed, d)) ev.visit_k2f(kf, &ed, d);
				}
			}
		};
		if (rely_on_prebuilt_spanning_trees) {
			order.push_back(std::make_pair(root, (topo_dist_t)0));
			const graph::st_entry *r = T.st.row(root); for (size_t i = 0; i < T.st.len(root); i++) order.push_back(std::make_pair(r[i].trg, (topo_dist_t)r[i].dist));
			for (size_t i = 0; i < order.size(); i++) {
				const graph::id32 kf = order[i].first; const topo_dist_t d = order[i].second; if (d > max_distance) continue;
				if (kf_visitor.visit_filter_kf(kf, d)) kf_visitor.visit_kf(kf, d);
				visit::features(*this, kf, d, lm_done, feat_visitor, k2f_edge_visitor);
				for (graph::id32 e = T.kf_adj_head[kf]; e != graph::NIL; e = T.next_adjacent(e, kf)) if (!edge_done[e]) { edge_done[e] = 1; const graph::id32 nk = T.other_end(e, kf);
					const k2k_edge_t *ed = &rba_state.k2k_edges[e]; if (k2k_edge_visitor.visit_filter_k2k(kf, nk, ed, d)) k2k_edge_visitor.visit_k2k(kf, nk, ed, d); }
			}
			return;
		}
		order.push_back(std::make_pair(root, (topo_dist_t)0)); kf_done[root] = 1;
		for (size_t i = 0; i < order.size(); i++) {
			const graph::id32 kf = order[i].first; const topo_dist_t d = order[i].second;
			kf_visitor.visit_kf(kf, d);
			visit::features(*this, kf, d, lm_done, feat_visitor, k2f_edge_visitor);
			if (d >= max_distance) continue;
			for (graph::id32 e = T.kf_adj_head[kf]; e != graph::NIL; e = T.next_adjacent(e, kf)) {
				const graph::id32 nk = T.other_end(e, kf);
				if (!kf_done[nk]) { kf_done[nk] = 1; if (kf_visitor.visit_filter_kf(nk, d)) order.push_back(std::make_pair(nk, d + 1)); }
				if (!edge_done[e]) { edge_done[e] = 1; const k2k_edge_t *ed = &rba_state.k2k_edges[e]; if (k2k_edge_visitor.visit_filter_k2k(kf, nk, ed, d)) k2k_edge_visitor.visit_k2k(kf, nk, ed, d);
					}
			}
		}
	}

	/** Poses of every key-frame within max_depth of root_id relative to it, composed down a breadth-first tree of the whole graph (impl/spantree_create_complete.h:18-126). */
	void create_complete_spanning_tree(const TKeyFrameID root_id, frameid2pose_map_t &span_tree, const size_t max_depth = std::numeric_limits<size_t>::max(), std::vector<bool> *aux_ws = NULL) const {
		(void)aux_ws; span_tree.clear();
		graph::topology &T = const_cast<graph::topology &>(rba_state.topo); const graph::id32 root = graph::topology::narrow(root_id); if (root >= T.n_keyframes()) return;
		std::vector<graph::id32> order, via; T.bfs_tree(root, max_depth, order, via); // BFS order = non-decreasing depth: parents come first
		for (size_t i = 0; i < order.size(); i++) {
			const graph::id32 kf = order[i]; pose_flag_t &mine = span_tree[kf];
			if (kf == root) { mine.pose = pose_t(); continue; }
			const k2k_edge_t &e = rba_state.k2k_edges[via[i]]; const pose_t &parent = span_tree[T.bfs_parent(kf)].pose;
			if (e.to == kf) mine.pose.composeFrom(parent, -e.inv_pose); else mine.pose.composeFrom(parent, e.inv_pose); // edge parent->me stores my pose inverted
		}
	}
	struct ExportGraphSLAM_Params { TKeyFrameID root_kf_id; ExportGraphSLAM_Params() : root_kf_id(0) {} };
	/** The map as a global pose graph (impl/get_global_graphslam_problem.h:17-48): node poses from the complete spanning tree of root_kf_id, one constraint
	 *  (to -> from, inv_pose) per kf2kf edge. POSE_GRAPH needs clear(), a map-like `nodes` and insertEdgeAtEnd(from, to, pose) (mrpt::graphs::CNetworkOfPoses). */
	template <class POSE_GRAPH> void get_global_graphslam_problem(POSE_GRAPH &global_graph, const ExportGraphSLAM_Params &params = ExportGraphSLAM_Params()) const {
		global_graph.clear(); if (rba_state.keyframes.empty()) return;
		frameid2pose_map_t tree; create_complete_spanning_tree(params.root_kf_id, tree);
		for (typename frameid2pose_map_t::const_iterator it = tree.begin(); it != tree.end(); ++it) global_graph.nodes[it->first] = it->second.pose;
		for (size_t e = 0; e < rba_state.k2k_edges.size(); e++) global_graph.insertEdgeAtEnd(rba_state.k2k_edges[e].to, rba_state.k2k_edges[e].from, rba_state.k2k_edges[e].inv_pose);
	}

	/** The map as a graphviz digraph (impl/export_dot.h:15-84): key-frames as boxes, kf2kf edges in bold, and -- if asked -- every landmark as a triangle hanging
	 *  from its base key-frame (grey = fixed relative position, white = unknown) with dotted observation arcs. Returns false if the file cannot be written. */
	bool save_graph_as_dot(const std::string &targetFileName, const bool all_landmarks = false) const {
		std::ofstream f(targetFileName.c_str()); if (!f.is_open()) return false;
		f << "digraph G {\n";
		const size_t nKF = rba_state.keyframes.size();
		if (nKF) {
			f << "/* KEYFRAMES */\nnode [shape=box,style=filled];\n";
			for (size_t id = 0; id < nKF; id++) f << id << "; ";
			f << "\n/* KEYFRAME->KEYFRAME edges */\nedge [style=bold];\n";
			for (size_t e = 0; e < rba_state.k2k_edges.size(); e++) f << rba_state.k2k_edges[e].from << "->" << rba_state.k2k_edges[e].to << ";\n";
			if (all_landmarks) {
				const struct { const TRelativeLandmarkPosMap *lms; const char *header; } groups[2] = {
					{&rba_state.known_lms,
						"/* LANDMARKS with known relative position, and its base keyframe */\nnode [shape=triangle,style=filled,fillcolor=gray80];\nedge [style=bold,color=black];\n"},
					{&rba_state.unknown_lms, "/* LANDMARKS with unknown relative position */\nnode [shape=triangle,style=filled,fillcolor=white];\nedge [style=solid,color=gray20];\n"}};
				for (int g = 0; g < 2; g++) {
					f << groups[g].header;
					for (typename TRelativeLandmarkPosMap::const_iterator it = groups[g].lms->begin(); it != groups[g].lms->end(); ++it) f << it->second.id_frame_base << " -> L" << it->first << "; ";
					f << "\n";
				}
				f << "/* OBSERVATIONS */\nedge [style=dotted,color=black];\n";
				for (size_t o = 0; o < rba_state.obs_table.size(); o++) f << rba_state.obs_table[o].kf_id << " -> L" << rba_state.obs_table[o].obs.feat_id << ";\n";
				f << "\n";
			}
		}
		f << "\n}\n";
		return f.good();
	}
	/** Only the key-frames with two or more kf2kf edges (the sub-map centres and loop closures) as an undirected graph (impl/export_dot.h:87-126); with
	 *  set_node_coordinates each node carries its position in the frame of key-frame 0 (complete spanning tree) as a graphviz "pos" attribute. */
	bool save_graph_top_structure_as_dot(const std::string &targetFileName, const bool set_node_coordinates) const {
		std::ofstream f(targetFileName.c_str()); if (!f.is_open()) return false;
		f << "graph G {\n";
		const size_t nKF = rba_state.keyframes.size();
		if (nKF) {
			std::vector<size_t> degree(nKF, 0);
			for (size_t e = 0; e < rba_state.k2k_edges.size(); e++) { degree[rba_state.k2k_edges[e].from]++; degree[rba_state.k2k_edges[e].to]++; }
			frameid2pose_map_t tree; if (set_node_coordinates) create_complete_spanning_tree(0, tree);
			f << "/* KEYFRAMES */\nnode [shape=box,style=filled];\n";
			for (size_t id = 0; id < nKF; id++) if (degree[id] >= 2) {
				f << id;
				if (set_node_coordinates) { const typename frameid2pose_map_t::const_iterator it = tree.find((TKeyFrameID)id); if (it != tree.end()) f << " [pos=\"" << it->second.pose.x() << "," <<
					it->second.pose.y() << "!\"]"; }
				f << "; ";
			}
			f << "\n/* KEYFRAME->KEYFRAME edges */\nedge [style=bold];\n";
			for (size_t e = 0; e < rba_state.k2k_edges.size(); e++) if (degree[rba_state.k2k_edges[e].from] >= 2 && degree[rba_state.k2k_edges[e].to] >= 2) f << rba_state.k2k_edges[e].from << "--" <<
				rba_state.k2k_edges[e].to << ";\n";
		}
		f << "\n}\n";
		return f.good();
	}

	/** Rendering options of build_opengl_representation() (RbaEngine.h:244-264 of the reference). */
	struct TOpenGLRepresentationOptions {
		size_t span_tree_max_depth; bool draw_unknown_feats, draw_unknown_feats_ellipses; double draw_unknown_feats_ellipses_quantiles; bool show_unknown_feats_ids, draw_kf_hierarchical;
			double draw_kf_hierarchical_height;
		TOpenGLRepresentationOptions() : span_tree_max_depth(std::numeric_limits<size_t>::max()), draw_unknown_feats(true), draw_unknown_feats_ellipses(true),
			draw_unknown_feats_ellipses_quantiles(1), show_unknown_feats_ids(true), draw_kf_hierarchical(false), draw_kf_hierarchical_height(10.0) {}
	};
	/** The geometry of the 3D view of the reference (impl/export_opengl.h:24-224): every key-frame within span_tree_max_depth of root_keyframe as a corner at its
	 *  pose in the root's frame, a line per kf2kf edge between drawn key-frames, and the landmarks as points (relative position composed with their base key-frame).
	 *  SCENE_PTR is mrpt::opengl::CSetOfObjectsPtr; this layer only emits primitives through insert_corner / insert_line / insert_point / insert_text (the
	 *  mrpt_lite stand-in records them, a build against the real MRPT maps them onto stock_objects::CornerXYZSimple, CSetOfLines and CPointCloud). */
	template <class SCENE_PTR> void build_opengl_representation(const TKeyFrameID root_keyframe, const TOpenGLRepresentationOptions &options, SCENE_PTR out_scene,
		SCENE_PTR out_root_tree = SCENE_PTR()) const {
		if (out_scene) {
			out_scene->clear();
			if (rba_state.keyframes.empty()) return;
			frameid2pose_map_t tree; create_complete_spanning_tree(root_keyframe, tree, options.span_tree_max_depth);
			std::vector<size_t> degree(rba_state.keyframes.size(), 0);
			for (size_t e = 0; e < rba_state.k2k_edges.size(); e++) { degree[rba_state.k2k_edges[e].from]++; degree[rba_state.k2k_edges[e].to]++; }
			auto lifted = [&](TKeyFrameID id, const pose_t &p) { mrpt::poses::CPose3D q(p); if (options.draw_kf_hierarchical && degree[id] >= 2) q.m_t[2] += options.draw_kf_hierarchical_height;
				return q; };
			for (typename frameid2pose_map_t::const_iterator it = tree.begin(); it != tree.end(); ++it) { out_scene->insert_corner(lifted(it->first, it->second.pose),
				it->first == root_keyframe ? 1.0 : 0.25); out_scene->insert_text(lifted(it->first, it->second.pose), mrpt::format("%u", (unsigned)it->first)); }
			for (size_t e = 0; e < rba_state.k2k_edges.size(); e++) {
				const typename frameid2pose_map_t::const_iterator a = tree.find(rba_state.k2k_edges[e].from), b = tree.find(rba_state.k2k_edges[e].to);
				if (a != tree.end() && b != tree.end()) out_scene->insert_line(lifted(a->first, a->second.pose), lifted(b->first, b->second.pose));
			}
			for (int g = 0; g < (options.draw_unknown_feats ? 2 : 1); g++) {
				const TRelativeLandmarkPosMap &lms = g ? rba_state.unknown_lms : rba_state.known_lms;
				for (typename TRelativeLandmarkPosMap::const_iterator it = lms.begin(); it != lms.end(); ++it) {
					const typename frameid2pose_map_t::const_iterator base = tree.find(it->second.id_frame_base); if (base == tree.end()) continue;
					double l[3] = {0, 0, 0}, gl[3]; for (size_t k = 0; k < LM_DIMS && k < 3; k++) l[k] = it->second.pos[k];
					mrpt::poses::CPose3D(base->second.pose).composePoint(l[0], l[1], l[2], gl[0], gl[1], gl[2]);
					out_scene->insert_point(gl[0], gl[1], gl[2], g != 0);
					if (g && options.show_unknown_feats_ids) out_scene->insert_text(mrpt::poses::CPose3D(gl[0], gl[1], gl[2], 0, 0, 0), mrpt::format("%u", (unsigned)it->first));
				}
			}
		}
		if (out_root_tree) { // schematic view of the root's symbolic spanning tree: one line per (key-frame, next key-frame towards the root), laid out by distance
			out_root_tree->clear();
			std::map<TKeyFrameID, std::pair<double, double> > where; where[root_keyframe] = std::make_pair(0.0, 0.0); std::map<topo_dist_t, size_t> used;
			const graph::st_table &st = rba_state.topo.st; const graph::st_entry *row = st.row((graph::id32)root_keyframe); const size_t n_row = st.len((graph::id32)root_keyframe);
			for (size_t i = 0; i < n_row; i++) where[row[i].trg] = std::make_pair((double)(used[row[i].dist]++), -(double)row[i].dist);
			for (size_t i = 0; i < n_row; i++) {
				const graph::st_entry *back = st.find(row[i].trg, (graph::id32)root_keyframe); if (!back) continue;
				const k2k_edge_t &ed = rba_state.k2k_edges[back->next]; const TKeyFrameID nxt = ed.from == (TKeyFrameID)row[i].trg ? ed.to : ed.from;
				const std::pair<double, double> a = where[row[i].trg], b = where[nxt];
				out_root_tree->insert_line(mrpt::poses::CPose3D(a.first, a.second, 0, 0, 0, 0), mrpt::poses::CPose3D(b.first, b.second, 0, 0, 0, 0));
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

	// =================================================================================== MI355X-specific additions
	/** Select the numeric back-end. Default (created lazily): srba::make_hip_backend(device) -- the GPU. */
	void set_numeric_backend(const std::shared_ptr<numeric_backend> &b) { m_backend = b; }
	void set_hip_device(int device) { m_hip_device = device; }
	/** Called with every capsule right before it is optimised (pre-optimisation values); stage = 1 for define_new_keyframe's single-edge optimisations, 0 otherwise. */
	std::function<void(const srba_hip_params &, CapsuleData &, int stage)> on_capsule;

	/** Squared error of ALL observations of the map with the current estimate (impl/eval_overall_error.h:15-137). The host lists, per distinct
	 *  (smaller id, larger id) pair of observer / base key-frames, the breadth-first path from the smaller id (the tree of spantree_create_complete.h,
	 *  searched only until every wanted key-frame is reached); the numeric back-end composes the poses and sums the residuals. */
	double eval_overall_squared_error() const {
		internal::profiler_scope ps(m_profiler, "eval_overall_squared_error");
		graph::topology &T = const_cast<graph::topology &>(rba_state.topo);
		const size_t nObs = T.n_observations(); if (!nObs) return 0;
		std::vector<std::pair<graph::id32, graph::id32> > want; // (root = smaller id, target)
		for (size_t i = 0; i < nObs; i++) { const graph::id32 a = T.obs_kf[i], b = T.lm_base[T.obs_lm[i]]; if (a != b) want.push_back(std::make_pair(std::min(a, b), std::max(a, b))); }
		std::sort(want.begin(), want.end()); want.erase(std::unique(want.begin(), want.end()), want.end());
		std::vector<int32_t> pair_path_off(1, 0), path_edge; std::vector<graph::id32> edges;
		for (size_t g = 0; g < want.size();) { // one search per root; shortest_path() re-walks the same first-found tree for each target
			size_t h = g; while (h < want.size() && want[h].first == want[g].first) h++;
			for (size_t k = g; k < h; k++) {
				if (!T.shortest_path(want[k].first, want[k].second, &edges, NULL)) throw std::runtime_error("eval_overall_squared_error: an observation relates two key-frames that are not connected");
				graph::id32 cur = want[k].first;
				for (size_t q = 0; q < edges.size(); q++) { const graph::id32 nxt = T.other_end(edges[q], cur); path_edge.push_back((int32_t)((edges[q] << 1) | (T.edge_to[edges[q]] == nxt ? 1 : 0)));
					cur = nxt; } // stepping parent->child along the edge direction composes the inverse of inv_pose
				pair_path_off.push_back((int32_t)path_edge.size());
			}
			g = h;
		}
		const size_t L = LM_DIMS, O = OBS_DIMS, PD = pose_t::storage_doubles();
		std::vector<int32_t> obs_pose(nObs), obs_lm(nObs); std::vector<double> obs_z(nObs * O), lm_pos; std::vector<int32_t> lm_slot(T.lm_base.size(), -1);
		for (size_t i = 0; i < nObs; i++) {
			const graph::id32 a = T.obs_kf[i], lm = T.obs_lm[i], b = T.lm_base[lm];
			if (a == b) obs_pose[i] = -1;
			else { const size_t p = (size_t)(std::lower_bound(want.begin(), want.end(), std::make_pair(std::min(a, b), std::max(a, b))) - want.begin());
				obs_pose[i] = (int32_t)(a < b ? 2 * p : 2 * p + 1); } // observer is the root: pose of the target; else its inverse
			if (lm_slot[lm] < 0) { lm_slot[lm] = (int32_t)(lm_pos.size() / L); for (size_t k = 0; k < L; k++) lm_pos.push_back(rba_state.lm_table[lm].pos[k]); }
			obs_lm[i] = lm_slot[lm];
			for (size_t k = 0; k < O; k++) obs_z[i * O + k] = rba_state.obs_table[i].obs_arr[k];
		}
		std::vector<double> edge_pose(rba_state.k2k_edges.size() * PD);
		for (size_t e = 0; e < rba_state.k2k_edges.size(); e++) rba_state.k2k_edges[e].inv_pose.storeTo(&edge_pose[e * PD]);
		srba_overall_problem q; std::memset(&q, 0, sizeof(q));
		q.n_edges = (int32_t)rba_state.k2k_edges.size(); q.n_pairs = (int32_t)pair_path_off.size() - 1; q.n_path = (int32_t)path_edge.size(); q.n_obs = (int32_t)nObs;
			q.n_lms = (int32_t)(lm_pos.size() / L);
		q.edge_pose = edge_pose.data(); q.pair_path_off = pair_path_off.data(); q.path_edge = path_edge.data(); q.obs_pose = obs_pose.data(); q.obs_lm = obs_lm.data(); q.obs_z = obs_z.data();
			q.lm_pos = lm_pos.data();
		srba_hip_params hp; fill_hip_params(hp);
		if (!m_backend) m_backend = make_hip_backend(m_hip_device);
		return m_backend->eval_overall(hp, q);
	}

	/** The parameter block of the numeric back-end: everything the reference's hot loops read from RbaEngine::parameters. */
	void fill_hip_params(srba_hip_params &hp) const {
		std::memset(&hp, 0, sizeof(hp));
		static_assert(family_pose_dims(device_family<kf2kf_pose_t, landmark_t, obs_t>::value) == (int)REL_POSE_DIMS, "this <key-frame pose, landmark, observation> combination has no device kernels");
		hp.family = device_family<kf2kf_pose_t, landmark_t, obs_t>::value; hp.solver = RBA_OPTIONS::solver_t::solver_id;
		hp.std_noise_observations = 1.0; for (int i = 0; i < 9; i++) hp.sensor_pose_se3[3 + i] = (i % 4 == 0) ? 1.0 : 0.0; hp.right_cam_pose[3] = 1.0;
		RBA_OPTIONS::obs_noise_matrix_t::fill_params(hp, parameters.obs_noise);
		RBA_OPTIONS::sensor_pose_on_robot_t::fill_params(hp, parameters.sensor_pose);
		sensor_model_t::fill_params(hp, parameters.sensor);
		hp.max_iters = (int)parameters.srba.max_iters; hp.use_robust_kernel = parameters.srba.use_robust_kernel ? 1 : 0; hp.kernel_param = parameters.srba.kernel_param;
		hp.max_error_per_obs_to_stop = parameters.srba.max_error_per_obs_to_stop; hp.max_rho = parameters.srba.max_rho; hp.max_lambda = parameters.srba.max_lambda;
		hp.min_error_reduction_ratio_to_relinearize = parameters.srba.min_error_reduction_ratio_to_relinearize; hp.cov_recovery = (parameters.srba.cov_recovery == crpLandmarksApprox) ? 1 : 0;
		hp.extensions = parameters.srba.schur_keeps_gradient ? SRBA_EXT_SCHUR_KEEPS_GRADIENT : 0;
	}

	/** One observation of key-frame observing_kf_id: stores it, creates the landmark on first sight (fixed position, caller's initial value, or the inverse
	 *  sensor model moved to the robot frame) and extends the symbolic Jacobians. Returns the observation index. */
	size_t add_observation(const TKeyFrameID observing_kf_id, const typename observation_traits_t::observation_t &new_obs, const array_landmark_t *fixed_relative_position = NULL,
		const array_landmark_t *unknown_relative_position_init_val = NULL) {
		ASSERT_(!(fixed_relative_position != NULL && unknown_relative_position_init_val != NULL));
		const graph::id32 kf = graph::topology::narrow(observing_kf_id), lm = graph::topology::narrow(new_obs.feat_id);
		const graph::topology::obs_result r = rba_state.topo.add_observation(kf, lm, fixed_relative_position != NULL);
		typename rba_problem_state_t::kf_observation_t rec; rec.obs = new_obs; rec.kf_id = observing_kf_id; new_obs.obs_data.getAsArray(rec.obs_arr);
		rba_state.obs_table.push_back(rec);
		if (r.first_seen) {
			if (lm >= rba_state.lm_table.size()) rba_state.lm_table.resize((size_t)lm + 1);
			TRelativeLandmarkPos &p = rba_state.lm_table[lm]; p.id_frame_base = observing_kf_id;
			if (r.fixed) { p.pos = *fixed_relative_position; rba_state.known_lms.add(new_obs.feat_id); }
			else {
				if (unknown_relative_position_init_val) p.pos = *unknown_relative_position_init_val;
				else { sensor_model_t::inverse_sensor_model(p.pos, new_obs.obs_data, parameters.sensor); RBA_OPTIONS::sensor_pose_on_robot_t::template sensor2robot_point<landmark_t>(p.pos,
					parameters.sensor_pose); }
				rba_state.unknown_lms.add(new_obs.feat_id);
			}
		}
		return r.obs;
	}

protected:
	int m_verbose_level;

	/** Edges for the new key-frame (the policy decides which) and, for each, the best initial relative pose at hand (impl/determine_kf2kf_edges_to_create.h:36-255):
	 *  (1) the pose the previous key-frame had relative to the same "from" key-frame, if that one was touched in the previous step; else
	 *  (2) a sensor-specific alignment of the landmarks both ends observe -- for an edge between two older area centres through the observer / base key-frames
	 *      the policy named, chained with what is already known about those two. */
	void determine_kf2kf_edges_to_create(const TKeyFrameID new_kf_id, const new_kf_observations_t &obs, std::vector<TNewEdgeInfo> &created) {
		created.clear();
		graph::topology &T = rba_state.topo;
		if (T.n_keyframes() == 1) return; // the very first key-frame has nothing to link to
		edge_creation_policy.template eval<traits_t, rba_engine_t>(new_kf_id, obs, created, *this, parameters.ecp);
		for (size_t i = 0; i < created.size(); i++) {
			TNewEdgeInfo &info = created[i]; if (info.has_approx_init_val) continue;
			k2k_edge_t &ed = rba_state.k2k_edges[info.id];
			const bool touches_new_kf = (ed.to == new_kf_id || ed.from == new_kf_id), points_to_new_kf = (ed.to == new_kf_id);
			if (touches_new_kf && std::binary_search(T.last_touched_kfs.begin(), T.last_touched_kfs.end(), (graph::id32)ed.from)) {
				if (const pose_t *prev = get_kf_relative_pose(new_kf_id - 1, ed.from)) { ed.inv_pose = points_to_new_kf ? -(*prev) : *prev; info.has_approx_init_val = true; continue; }
			}
			pose_t align; // pose of the later key-frame with respect to the earlier one, from matched observations
			bool ok = touches_new_kf ? align_by_common_landmarks(obs, true, new_kf_id, points_to_new_kf ? ed.from : ed.to, align)
			                         : align_by_common_landmarks(obs, false, ed.from, ed.to, align);
			const bool have_lc_hint = info.loopclosure_observer_kf != SRBA_INVALID_KEYFRAMEID && info.loopclosure_base_kf != SRBA_INVALID_KEYFRAMEID;
			const bool direct = ok; // the alignment relates the two ends of the edge themselves (the reference nevertheless sends it through the observer / base chain below when the edge does not
				// touch the new key-frame)
			if (!ok && have_lc_hint) ok = align_by_common_landmarks(obs, info.loopclosure_observer_kf == new_kf_id, info.loopclosure_observer_kf, info.loopclosure_base_kf, align);
			if (!ok) { if (m_verbose_level >= 2) std::cout << "[determine_kf2kf_edges_to_create] Could not provide initial value to relative pose " << ed.from << "<=>" << ed.to << "\n"; continue; }
			// the alignment relates SENSOR frames: move it to the robot frames
			const mrpt::poses::CPose3D S = RBA_OPTIONS::sensor_pose_on_robot_t::sensor_pose_as_3d(parameters.sensor_pose);
			align = pose_t((S + mrpt::poses::CPose3D(align)) + (-S));
			info.has_approx_init_val = true;
			if (touches_new_kf) { ed.inv_pose = points_to_new_kf ? -align : align; continue; }
			if (direct && parameters.srba.consistent_loop_closure_init) { ed.inv_pose = align; continue; } // (extension) align = pose of `from` w.r.t. `to` = inv_pose
			// edge between two older key-frames: (base wrt remote end) (+) align (+) (-)(observer wrt local end), the "local" end being the one the observer is known from
			const pose_t I; const TKeyFrameID ob = info.loopclosure_observer_kf, bs = info.loopclosure_base_kf;
			const pose_t *ob_to = (ob == ed.to) ? &I : get_kf_relative_pose(ob, ed.to), *bs_to = (bs == ed.to) ? &I : get_kf_relative_pose(bs, ed.to);
			const pose_t *ob_from = (ob == ed.from) ? &I : get_kf_relative_pose(ob, ed.from), *bs_from = (bs == ed.from) ? &I : get_kf_relative_pose(bs, ed.from);
			const bool local_is_to = (ob_to || bs_from) || !(ob_from || bs_to);
			const pose_t *ob_local = local_is_to ? ob_to : ob_from, *bs_remote = local_is_to ? bs_from : bs_to;
			pose_t ob_guess; // (extension) the observer is the key-frame being inserted: its pose in its own area is the initial value just given to the edge that links it to the local end
			if (!ob_local && parameters.srba.consistent_loop_closure_init && ob == new_kf_id) {
				const TKeyFrameID local = local_is_to ? ed.to : ed.from;
				for (size_t j = 0; j < created.size(); j++) { const k2k_edge_t &e2 = rba_state.k2k_edges[created[j].id]; if (!created[j].has_approx_init_val || j == i) continue;
					if (e2.from == local && e2.to == ob) { ob_guess = -e2.inv_pose; ob_local = &ob_guess; break; }      // inv_pose = pose of `from` (local) w.r.t. `to` (observer)
					if (e2.to == local && e2.from == ob) { ob_guess = e2.inv_pose; ob_local = &ob_guess; break; } }
			}
			const pose_t local_wrt_remote = ((bs_remote ? *bs_remote : I) + align) + (-(ob_local ? *ob_local : I));
			if (parameters.srba.consistent_loop_closure_init) ed.inv_pose = local_is_to ? -local_wrt_remote : local_wrt_remote; // pose of `from` w.r.t. `to`
			else ed.inv_pose = points_to_new_kf ? -local_wrt_remote : local_wrt_remote;
		}
		T.last_touched_kfs.clear();
		for (size_t i = 0; i < created.size(); i++) { T.last_touched_kfs.push_back((graph::id32)rba_state.k2k_edges[created[i].id].from);
			T.last_touched_kfs.push_back((graph::id32)rba_state.k2k_edges[created[i].id].to); }
		std::sort(T.last_touched_kfs.begin(), T.last_touched_kfs.end()); T.last_touched_kfs.erase(std::unique(T.last_touched_kfs.begin(), T.last_touched_kfs.end()), T.last_touched_kfs.end());
	}
	/** Relative pose of key-frame `later` with respect to `earlier` from the landmarks both observe (landmark_matcher<OBS>). `later` is either the key-frame being
	 *  inserted (its observations are still only in `obs`) or a stored one. Pairs are listed in the order `earlier` observed them; if `later` lists a landmark
	 *  twice its last entry counts. */
	bool align_by_common_landmarks(const new_kf_observations_t &obs, const bool later_is_new_kf, const TKeyFrameID later, const TKeyFrameID earlier, pose_t &out) {
		const graph::topology &T = rba_state.topo;
		size_t need = T.lm_base.size(); if (later_is_new_kf) for (size_t i = 0; i < obs.size(); i++) need = std::max(need, (size_t)obs[i].obs.feat_id + 1);
		if (m_lm_mark.size() < need) { m_lm_mark.resize(need, 0); m_lm_where.resize(need, 0); }
		if (++m_mark_epoch == 0) { std::fill(m_lm_mark.begin(), m_lm_mark.end(), 0u); m_mark_epoch = 1; }
		if (later_is_new_kf) for (size_t i = 0; i < obs.size(); i++) { m_lm_mark[obs[i].obs.feat_id] = m_mark_epoch; m_lm_where[obs[i].obs.feat_id] = (uint32_t)i; }
		else for (graph::id32 o = T.kf_obs_head[later]; o != graph::NIL; o = T.obs_next_in_kf[o]) { m_lm_mark[T.obs_lm[o]] = m_mark_epoch; m_lm_where[T.obs_lm[o]] = o; }
		std::vector<typename obs_t::obs_data_t> seen_later, seen_earlier;
		for (graph::id32 o = T.kf_obs_head[earlier]; o != graph::NIL; o = T.obs_next_in_kf[o]) {
			const graph::id32 lm = T.obs_lm[o]; if (lm >= m_lm_mark.size() || m_lm_mark[lm] != m_mark_epoch) continue;
			seen_earlier.push_back(rba_state.obs_table[o].obs.obs_data);
			seen_later.push_back(later_is_new_kf ? obs[m_lm_where[lm]].obs.obs_data : rba_state.obs_table[m_lm_where[lm]].obs.obs_data);
		}
		return observations::landmark_matcher<obs_t>::find_relative_pose(seen_later, seen_earlier, parameters.sensor, out);
	}

	/** The optimiser entry (reference impl/optimize_edges.h:44-793): flatten the call into a capsule, hand it to the numeric back-end, write the results back. */
	void optimize_edges(const std::vector<size_t> &run_k2k_edges_in, const std::vector<size_t> &run_feat_ids_in, TOptimizeExtraOutputInfo &out_info,
		const std::vector<size_t> &in_observation_indices_to_optimize = std::vector<size_t>()) {
		internal::profiler_scope ps(m_profiler, "opt");
		const int stage = m_capsule_stage; m_capsule_stage = 0;
		out_info.clear();
		if (parameters.srba.numeric_jacobians) throw std::runtime_error("RbaEngine: parameters.srba.numeric_jacobians (debug path of the reference) is not available with the GPU back-end");
		const int P = REL_POSE_DIMS, L = LM_DIMS;
		graph::topology &T = rba_state.topo;
		CapsuleData &cd = m_cd; graph::capsule_index &ix = m_ix;
		if (!build_capsule(run_k2k_edges_in, run_feat_ids_in, in_observation_indices_to_optimize, cd, ix)) return;
		const size_t nK = cd.n_unk_edges, nF = cd.n_unk_lms, nObs = ix.obs_rows.size();
		out_info.num_kf_optimized = ix.n_kfs_touched; out_info.num_lm_optimized = ix.n_lms_touched;

		srba_hip_params hp; fill_hip_params(hp);
		if (on_capsule) on_capsule(hp, cd, stage);
		srba_problem_capsule cap = cd.view();
		srba_lm_result res; std::memset(&res, 0, sizeof(res)); res.lambda_last_trial = std::numeric_limits<double>::quiet_NaN(); // (a back-end that does not know the field leaves the NaN: see
			// return_hessian below)
		if (!m_backend) m_backend = make_hip_backend(m_hip_device);
		m_backend->set_profiler(&m_profiler);
		{ internal::profiler_scope p2(m_profiler, "opt.backend"); m_backend->run(hp, cap, res); }
		if (res.status == 1) throw std::logic_error("optimize_edges: OBS_DIMS*nObs < number of unknown scalars (reference ASSERT_ABOVEEQ_, optimize_edges.h:355)");

		write_back(cd, ix, res, out_info, true);
		// The Hessian never leaves the device unless it is asked for (one more download per call): extra_results.hessian as the reference's solvers return it
		// (lev-marq_solvers.h:204-208, :586-590: the system matrix of the last solve, H + lambda I, reduced by the Schur complement for the Schur solvers) and
		// HAp_condition_number (optimize_edges.h:753-766: ratio of the extreme singular values of the dense HAp, both triangles)
		if (parameters.srba.compute_condition_number || parameters.srba.return_hessian) {
			std::vector<double> hap, hf, hapf;
			if (!m_backend->read_blocks(3, hap)) throw std::runtime_error(std::string("RbaEngine: numeric back-end '") + m_backend->name() +
				"' cannot return the Hessian (compute_condition_number / return_hessian)");
			const size_t nA = (size_t)P * nK; std::vector<double> dA(nA * nA, 0.0);
			for (size_t b = 0; b < cd.hap_i.size(); b++) { const size_t i = cd.hap_i[b], j = cd.hap_j[b]; // upper blocks (i <= j); diagonal blocks hold both triangles
				for (int r = 0; r < P; r++) for (int q = 0; q < P; q++) { const double v = hap[b * P * P + r * P + q]; dA[(P * i + r) * nA + P * j + q] = v;
					if (i != j) dA[(P * j + q) * nA + P * i + r] = v; } }
			if (parameters.srba.compute_condition_number) out_info.HAp_condition_number = internal::symmetric_condition_number(dA, nA);
			// lambda of the last trial: the trailing field srba_lm_result::lambda_last_trial (ABI: added in round 4 -- sizeof(srba_lm_result) grew by 8 bytes). A numeric back-end built
			// against the older record leaves it as initialised before the run (NaN): fall back on the trace while it covers the run, else no Hessian rather than one with a wrong lambda
			double lam_last = res.lambda_last_trial;
			if (!(lam_last > 0.0) && res.num_trials > 0 && res.num_trials <= SRBA_TRACE_LEN) lam_last = res.trace_lambda[res.num_trials - 1];
			if (parameters.srba.return_hessian && res.num_trials > 0 && lam_last > 0.0) { // (no trial: no solve, the reference's solver object holds no system either; hessian_valid stays false)
				const double lam = lam_last; const bool full = !RBA_OPTIONS::solver_t::USE_SCHUR && nF > 0;
				const size_t n = full ? nA + (size_t)L * nF : nA; std::vector<double> &H = out_info.extra_results.hessian; H.assign(n * n, 0.0);
				for (size_t r = 0; r < nA; r++) for (size_t q = 0; q < nA; q++) H[r * n + q] = dA[r * nA + q];
				if (full && m_backend->read_blocks(4, hf) && m_backend->read_blocks(5, hapf)) {
					for (size_t b = 0; b < cd.hf_i.size(); b++) { const size_t i = cd.hf_i[b], j = cd.hf_j[b];
						for (int r = 0; r < L; r++) for (int q = 0; q < L; q++) { const double v = hf[b * L * L + r * L + q]; H[(nA + L * i + r) * n + nA + L * j + q] = v;
							if (i != j) H[(nA + L * j + q) * n + nA + L * i + r] = v; } }
					for (size_t b = 0; b < cd.hapf_i.size(); b++) { const size_t i = cd.hapf_i[b], j = cd.hapf_j[b];
						for (int r = 0; r < P; r++) for (int q = 0; q < L; q++) { const double v = hapf[b * P * L + r * L + q]; H[(P * i + r) * n + nA + L * j + q] = v;
							H[(nA + L * j + q) * n + P * i + r] = v; } }
				}
				for (size_t k = 0; k < n; k++) H[k * n + k] += lam;
				out_info.extra_results.hessian_valid = true;
			}
		}
		if (parameters.srba.compute_sparsity_stats) { // occupancy of the block matrices: non-zero blocks / blocks of the bounding rectangle (reference optimize_edges.h:315-323 over [EXT]
			// MatrixBlockSparseCols::getSparsityStats)
			out_info.sparsity_dh_dAp_nnz = T.jp.size(); out_info.sparsity_dh_dAp_max_size = T.n_edges() * T.n_observations();
			size_t ndf = 0, ncol = 0; for (size_t l = 0; l < T.lm_df_count.size(); l++) if (T.lm_base[l] != graph::NIL && !T.lm_known[l]) { ndf += T.lm_df_count[l]; ncol++; }
			out_info.sparsity_dh_df_nnz = ndf; out_info.sparsity_dh_df_max_size = ncol * T.n_observations();
			out_info.sparsity_HAp_nnz = cd.hap_i.size(); out_info.sparsity_HAp_max_size = nK * nK; out_info.sparsity_Hf_nnz = cd.hf_i.size(); out_info.sparsity_Hf_max_size = nF * nF;
			out_info.sparsity_HApf_nnz = cd.hapf_i.size(); out_info.sparsity_HApf_max_size = nK * nF;
		}
		if (parameters.srba.feedback_user_iteration) { // served after the fact from the back-end's trial trace: the initial point, then every accepted step (optimize_edges.h:417-418,:595-596)
			(*parameters.srba.feedback_user_iteration)(0, res.total_sqr_error_init, std::sqrt(res.total_sqr_error_init / std::max<size_t>(1, nObs)));
			unsigned int it = 0;
			for (int t = 0; t < res.num_trials && t < SRBA_TRACE_LEN; t++) if (res.trace_rho[t] > 0) (*parameters.srba.feedback_user_iteration)(it++, res.trace_chi2[t],
				std::sqrt(res.trace_chi2[t] / std::max<size_t>(1, nObs)));
		}
		if (m_verbose_level >= 1) std::cout << "[OPT] Final RMSE=" << res.obs_rmse << " #iters=" << res.num_iters << "\n";
	}

	/** The integer tables and the typed payload of one optimize_edges() call from the map as it stands (no arithmetic); false: nothing to optimise */
	bool build_capsule(const std::vector<size_t> &run_k2k_edges_in, const std::vector<size_t> &run_feat_ids_in, const std::vector<size_t> &obs_subset, CapsuleData &cd, graph::capsule_index &ix) {
		const int P = REL_POSE_DIMS, L = LM_DIMS, O = OBS_DIMS, PD = (int)pose_t::storage_doubles();
		internal::profiler_scope p2(m_profiler, "opt.capsule");
		if (!m_builder.build(rba_state.topo, run_k2k_edges_in, run_feat_ids_in, obs_subset, parameters.srba.refresh_all_read_poses, RBA_OPTIONS::solver_t::USE_SCHUR, P, L, O, PD, cd, ix,
			parameters.srba.restore_spanning_tree_twins)) return false;
		// typed payload
		cd.edge_pose.resize(ix.edge_ids.size() * PD); for (size_t i = 0; i < ix.edge_ids.size(); i++) rba_state.k2k_edges[ix.edge_ids[i]].inv_pose.storeTo(&cd.edge_pose[i * PD]);
		cd.ulm_pos.resize(ix.unk_lms.size() * L); for (size_t i = 0; i < ix.unk_lms.size(); i++) for (int k = 0; k < L; k++) cd.ulm_pos[i * L + k] = rba_state.lm_table[ix.unk_lms[i]].pos[k];
		cd.klm_pos.resize(ix.const_lms.size() * L); for (size_t i = 0; i < ix.const_lms.size(); i++) for (int k = 0; k < L; k++) cd.klm_pos[i * L + k] = rba_state.lm_table[ix.const_lms[i]].pos[k];
		cd.obs_z.resize(ix.obs_rows.size() * O); for (size_t i = 0; i < ix.obs_rows.size(); i++) for (int k = 0; k < O; k++) cd.obs_z[i * O + k] = rba_state.obs_table[ix.obs_rows[i]].obs_arr[k];
		return true;
	}
	/** The reference optimises in place (optimize_edges.h:526,538,727-751): the unknowns, every refreshed spanning-tree pose and the landmark information matrices go back into the map */
	void write_back(const CapsuleData &cd, const graph::capsule_index &ix, const srba_lm_result &res, TOptimizeExtraOutputInfo &out_info, bool replace_inf_matrices) {
		const int P = REL_POSE_DIMS, L = LM_DIMS, PD = (int)pose_t::storage_doubles(); graph::topology &T = rba_state.topo;
		const size_t nK = cd.n_unk_edges, nF = cd.n_unk_lms, nObs = ix.obs_rows.size();
		for (size_t i = 0; i < nK; i++) rba_state.k2k_edges[ix.edge_ids[i]].inv_pose.loadFrom(&cd.edge_pose[i * PD]);
		for (size_t i = 0; i < nF; i++) for (int k = 0; k < L; k++) rba_state.lm_table[ix.unk_lms[i]].pos[k] = cd.ulm_pos[i * L + k];
		for (size_t p = 0; p < ix.pairs.size(); p++) {
			const int32_t s0 = T.ensure_num(ix.pairs[p].first, ix.pairs[p].second), s1 = T.ensure_num(ix.pairs[p].second, ix.pairs[p].first);
			pose_flag_t &a = rba_state.num_at(s0); a.pose.loadFrom(&cd.pose[(2 * p) * PD]); a.updated = true;
			pose_flag_t &b = rba_state.num_at(s1); b.pose.loadFrom(&cd.pose[(2 * p + 1) * PD]); b.updated = true;
		}
		if (replace_inf_matrices) rba_state.unknown_lms_inf_matrices.clear();
		if (parameters.srba.cov_recovery == crpLandmarksApprox)
			for (size_t i = 0; i < nF; i++) if (cd.ulm_inf_valid[i]) { rba_state.unknown_lms_inf_matrices.push_back(std::make_pair((TLandmarkID)ix.unk_lms[i],
				typename rba_problem_state_t::lm_inf_matrix_t())); for (int k = 0; k < L * L; k++) rba_state.unknown_lms_inf_matrices.back().second.m[k] = cd.ulm_inf[i * L * L + k]; }
		out_info.num_kf_optimized = ix.n_kfs_touched; out_info.num_lm_optimized = ix.n_lms_touched;
		out_info.num_observations = nObs; out_info.num_jacobians = res.num_jacobians; out_info.num_kf2kf_edges_optimized = nK; out_info.num_kf2lm_edges_optimized = nF;
		out_info.num_total_scalar_optimized = P * nK + L * nF; out_info.num_span_tree_numeric_updates = res.num_span_tree_numeric_updates;
		out_info.total_sqr_error_init = res.total_sqr_error_init; out_info.total_sqr_error_final = res.total_sqr_error_final; out_info.obs_rmse = res.obs_rmse; out_info.lm = res;
		out_info.optimized_k2k_edge_indices.assign(ix.edge_ids.begin(), ix.edge_ids.begin() + nK); out_info.optimized_landmark_indices.assign(ix.unk_lms.begin(), ix.unk_lms.end());
	}

public:
	// ---------------------------------------------------------------------------------------------------------------------------------------------------------------
	// Map sweeps (NO counterpart in the reference, which optimises one local area per call: SURVEY 8e "new mode", north_star "sub-maps shard across the GPUs ... RCCL only for
	// shared-edge reduction"). Re-optimising the local areas of MANY roots of one map is a sequence of optimize_local_area() calls; two of them commute when neither writes what
	// the other touches: a window WRITES its unknown kf2kf edges and landmarks and READS every edge on the spanning-tree paths of its observations and every landmark it observes
	// (window disjointness follows from impl/bfs_visitor.h:105-176 and ecps/local_areas_fixed_size.h:51-55: roots four sub-maps apart share nothing). plan_local_area_sweep()
	// deals the roots to ROUNDS of mutually independent windows (first round that fits, roots in the given order); a round is ONE batch for the numeric back-end
	// (optimize_local_areas_batch: one upload, one fused launch per size class) and -- with the roots of a round dealt to several processes, one GPU each -- one exchange of the
	// edges written in it (srba_amd/multi.py: an all-reduce over RCCL). Running the rounds in order, the windows of a round in any order, IS a sequential schedule of
	// optimize_local_area() calls: that schedule on the CPU engine is what parity is defined against (tests/test_sweep.py).
	/** The capsules of many local areas at once: the windows are selected one after the other (the selection stamps scratch of the topology), their tables are built side by side -- the
	 *  builder only READS the topology and the map, one builder per thread (SRBA_ENGINE_THREADS, default min(16, cores)). ok[i] = 0: nothing to optimise at roots[i]. */
	void build_capsules_parallel(const std::vector<TKeyFrameID> &roots, const unsigned int win_size, const TOptimizeLocalAreaParams &params, std::vector<CapsuleData> &cds,
		std::vector<graph::capsule_index> &ixs, std::vector<uint8_t> &ok, bool payload) {
		const size_t n = roots.size(); cds.assign(n, CapsuleData()); ixs.assign(n, graph::capsule_index()); ok.assign(n, 0);
		const bool use_prebuilt_st = (win_size <= parameters.srba.max_tree_depth);
		const graph::window_params wp = {params.optimize_k2k_edges, params.optimize_landmarks, params.dont_optimize_landmarks_seen_less_than_n_times, params.max_visitable_kf_id};
		std::vector<std::vector<size_t> > se(n), sl(n);
		for (size_t i = 0; i < n; i++) rba_state.topo.select_local_area(graph::topology::narrow(roots[i]), win_size, use_prebuilt_st, wp, se[i], sl[i]);
		int nt = (int)std::min<size_t>(16, std::max(1u, std::thread::hardware_concurrency())); if (const char *e = std::getenv("SRBA_ENGINE_THREADS")) nt = std::max(1, std::atoi(e));
		nt = (int)std::min<size_t>((size_t)nt, std::max<size_t>(1, n / 4));
		const int P = REL_POSE_DIMS, L = LM_DIMS, O = OBS_DIMS, PD = (int)pose_t::storage_doubles(); const std::vector<size_t> no_subset;
		std::vector<std::string> err(nt);
		auto work = [&](int t) {
			graph::capsule_builder builder;
			try { for (size_t i = (size_t)t; i < n; i += (size_t)nt) {
				if (se[i].empty() && sl[i].empty()) continue;
				CapsuleData &cd = cds[i]; graph::capsule_index &ix = ixs[i];
				if (!builder.build(rba_state.topo, se[i], sl[i], no_subset, parameters.srba.refresh_all_read_poses, RBA_OPTIONS::solver_t::USE_SCHUR, P, L, O, PD, cd, ix,
					parameters.srba.restore_spanning_tree_twins)) continue;
				if (payload) {
					cd.edge_pose.resize(ix.edge_ids.size() * PD); for (size_t k = 0; k < ix.edge_ids.size(); k++) rba_state.k2k_edges[ix.edge_ids[k]].inv_pose.storeTo(&cd.edge_pose[k * PD]);
					cd.ulm_pos.resize(ix.unk_lms.size() * L); for (size_t k = 0; k < ix.unk_lms.size(); k++) for (int q = 0; q < L; q++) cd.ulm_pos[k * L + q] = rba_state.lm_table[ix.unk_lms[k]].pos[q];
					cd.klm_pos.resize(ix.const_lms.size() * L); for (size_t k = 0; k < ix.const_lms.size(); k++) for (int q = 0; q < L; q++) cd.klm_pos[k * L + q] = rba_state.lm_table[ix.const_lms[k]].pos[q];
					cd.obs_z.resize(ix.obs_rows.size() * O); for (size_t k = 0; k < ix.obs_rows.size(); k++) for (int q = 0; q < O; q++) cd.obs_z[k * O + q] = rba_state.obs_table[ix.obs_rows[k]].obs_arr[q];
				}
				ok[i] = 1; } }
			catch (const std::exception &e) { err[t] = e.what(); }
		};
		if (nt <= 1) work(0);
		else { std::vector<std::thread> th; for (int t = 0; t < nt; t++) th.emplace_back(work, t); for (auto &x : th) x.join(); }
		for (int t = 0; t < nt; t++) if (!err[t].empty()) throw std::runtime_error(err[t]);
	}
	struct TSweepPlan {
		std::vector<int32_t> round_of;      //!< per root: its round, -1 = nothing to optimise there
		std::vector<int64_t> touch_off;     //!< per root: [touch_off[i], touch_off[i + 1]) of `touch`
		std::vector<uint32_t> touch;        //!< kf2kf edge id | 0x80000000 if the window writes it (unknown), else it only reads it
		std::vector<int64_t> touch_lm_off;  //!< per root: [touch_lm_off[i], touch_lm_off[i + 1]) of `touch_lm`
		std::vector<uint32_t> touch_lm;     //!< landmark id | 0x80000000 if the window writes its position (unknown landmark of the window), else it only reads it
		int32_t n_rounds; bool has_unknown_landmarks; //!< (some window optimises landmarks: the exchange of a sharded sweep then carries landmark positions too)
	};
	void plan_local_area_sweep(const std::vector<TKeyFrameID> &roots, const unsigned int win_size, TSweepPlan &plan, const TOptimizeLocalAreaParams &params = TOptimizeLocalAreaParams()) {
		const size_t n = roots.size(), nE = rba_state.k2k_edges.size(), nL = rba_state.lm_table.size();
		plan.round_of.assign(n, -1); plan.touch_off.assign(n + 1, 0); plan.touch.clear(); plan.touch_lm_off.assign(n + 1, 0); plan.touch_lm.clear(); plan.n_rounds = 0; plan.has_unknown_landmarks = false;
		std::vector<std::vector<uint8_t> > mark; // per round, per edge then per landmark: 1 read, 2 written
		const size_t CH = 2048; std::vector<CapsuleData> cds; std::vector<graph::capsule_index> ixs; std::vector<uint8_t> ok; // (the tables of a chunk of roots at a time: a capsule is ~100 KB)
		for (size_t c0 = 0; c0 < n; c0 += CH) {
			const std::vector<TKeyFrameID> part(roots.begin() + c0, roots.begin() + std::min(n, c0 + CH));
			build_capsules_parallel(part, win_size, params, cds, ixs, ok, false);
			for (size_t j = 0; j < part.size(); j++) { const size_t i = c0 + j;
				plan.touch_off[i + 1] = plan.touch_off[i]; plan.touch_lm_off[i + 1] = plan.touch_lm_off[i];
				if (!ok[j]) continue;
				const CapsuleData &cd = cds[j]; const graph::capsule_index &ix = ixs[j];
				const size_t nK = cd.n_unk_edges, nF = cd.n_unk_lms; if (nF) plan.has_unknown_landmarks = true;
				for (size_t k = 0; k < ix.edge_ids.size(); k++) plan.touch.push_back((uint32_t)ix.edge_ids[k] | (k < nK ? 0x80000000u : 0u));
				plan.touch_off[i + 1] = (int64_t)plan.touch.size();
				for (size_t k = 0; k < nF; k++) plan.touch_lm.push_back((uint32_t)ix.unk_lms[k] | 0x80000000u);
				for (size_t k = 0; k < ix.const_lms.size(); k++) plan.touch_lm.push_back((uint32_t)ix.const_lms[k]);
				plan.touch_lm_off[i + 1] = (int64_t)plan.touch_lm.size();
				int r = 0;
				for (;; r++) { // first round in which this window neither writes what a member touches nor touches what a member writes
					if (r == (int)mark.size()) { mark.push_back(std::vector<uint8_t>(nE + nL, 0)); break; }
					const std::vector<uint8_t> &m = mark[r]; bool clash = false;
					for (size_t k = 0; k < ix.edge_ids.size() && !clash; k++) clash = k < nK ? m[ix.edge_ids[k]] != 0 : m[ix.edge_ids[k]] == 2;
					for (size_t k = 0; k < nF && !clash; k++) clash = m[nE + ix.unk_lms[k]] != 0;
					for (size_t k = 0; k < ix.const_lms.size() && !clash; k++) clash = m[nE + ix.const_lms[k]] == 2;
					if (!clash) break;
				}
				std::vector<uint8_t> &m = mark[r];
				for (size_t k = 0; k < ix.edge_ids.size(); k++) m[ix.edge_ids[k]] = std::max<uint8_t>(m[ix.edge_ids[k]], k < nK ? 2 : 1);
				for (size_t k = 0; k < nF; k++) m[nE + ix.unk_lms[k]] = 2;
				for (size_t k = 0; k < ix.const_lms.size(); k++) m[nE + ix.const_lms[k]] = std::max<uint8_t>(m[nE + ix.const_lms[k]], 1);
				plan.round_of[i] = r;
			}
		}
		plan.n_rounds = (int32_t)mark.size();
	}
	/** optimize_local_area() of every root as ONE batch of the numeric back-end. The windows must be mutually independent (one round of plan_local_area_sweep): checked, std::logic_error
	 *  otherwise. out[i] as optimize_local_area() would fill it (roots with nothing to optimise: cleared). */
	void optimize_local_areas_batch(const std::vector<TKeyFrameID> &roots, const unsigned int win_size, std::vector<TOptimizeExtraOutputInfo> &out,
		const TOptimizeLocalAreaParams &params = TOptimizeLocalAreaParams()) {
		internal::profiler_scope ps(m_profiler, "optimize_local_areas_batch");
		if (parameters.srba.numeric_jacobians) throw std::runtime_error("RbaEngine: parameters.srba.numeric_jacobians (debug path of the reference) is not available with the GPU back-end");
		const size_t n = roots.size(), nE = rba_state.k2k_edges.size(), nL = rba_state.lm_table.size(); out.resize(n); for (size_t i = 0; i < n; i++) out[i].clear();
		std::vector<CapsuleData> cds; std::vector<graph::capsule_index> ixs; std::vector<uint8_t> ok; std::vector<size_t> who; std::vector<uint8_t> m(nE + nL, 0);
		{ internal::profiler_scope p2(m_profiler, "opt.capsule"); build_capsules_parallel(roots, win_size, params, cds, ixs, ok, true); }
		for (size_t i = 0; i < n; i++) { if (!ok[i]) continue;
			const CapsuleData &cd = cds[i]; const graph::capsule_index &ix = ixs[i]; const size_t nK = cd.n_unk_edges, nF = cd.n_unk_lms; bool clash = false;
			for (size_t k = 0; k < ix.edge_ids.size() && !clash; k++) clash = k < nK ? m[ix.edge_ids[k]] != 0 : m[ix.edge_ids[k]] == 2;
			for (size_t k = 0; k < nF && !clash; k++) clash = m[nE + ix.unk_lms[k]] != 0;
			for (size_t k = 0; k < ix.const_lms.size() && !clash; k++) clash = m[nE + ix.const_lms[k]] == 2;
			if (clash) throw std::logic_error("optimize_local_areas_batch: the local areas of the batch are not independent (use the rounds of plan_local_area_sweep)");
			for (size_t k = 0; k < ix.edge_ids.size(); k++) m[ix.edge_ids[k]] = std::max<uint8_t>(m[ix.edge_ids[k]], k < nK ? 2 : 1);
			for (size_t k = 0; k < nF; k++) m[nE + ix.unk_lms[k]] = 2;
			for (size_t k = 0; k < ix.const_lms.size(); k++) m[nE + ix.const_lms[k]] = std::max<uint8_t>(m[nE + ix.const_lms[k]], 1);
			who.push_back(i);
		}
		if (who.empty()) return;
		srba_hip_params hp; fill_hip_params(hp);
		std::vector<srba_problem_capsule> caps(who.size()); std::vector<srba_lm_result> res(who.size());
		for (size_t q = 0; q < who.size(); q++) { if (on_capsule) on_capsule(hp, cds[who[q]], 0); caps[q] = cds[who[q]].view(); std::memset(&res[q], 0, sizeof(res[q]));
			res[q].lambda_last_trial = std::numeric_limits<double>::quiet_NaN(); }
		if (!m_backend) m_backend = make_hip_backend(m_hip_device);
		m_backend->set_profiler(&m_profiler);
		{ internal::profiler_scope p2(m_profiler, "opt.backend"); m_backend->run_batch(hp, caps.data(), (int)caps.size(), res.data()); }
		rba_state.unknown_lms_inf_matrices.clear();
		for (size_t q = 0; q < who.size(); q++) {
			if (res[q].status == 1) throw std::logic_error("optimize_edges: OBS_DIMS*nObs < number of unknown scalars (reference ASSERT_ABOVEEQ_, optimize_edges.h:355)");
			write_back(cds[who[q]], ixs[who[q]], res[q], out[who[q]], false);
		}
	}
	/** kf2kf edge values by id (the exchange step of a sharded sweep reads and sets them in bulk) */
	void get_k2k_edge_poses(const size_t *ids, size_t n, double *out /* n x storage_doubles */) const { const size_t PD = pose_t::storage_doubles(); for (size_t i = 0; i < n; i++)
		rba_state.k2k_edges[ids[i]].inv_pose.storeTo(out + i * PD); }
	void get_lm_positions(const size_t *ids, size_t n, double *out /* n x LM_DIMS */) const { for (size_t i = 0; i < n; i++) for (size_t k = 0; k < LM_DIMS; k++) out[i * LM_DIMS + k] =
		rba_state.lm_table[ids[i]].pos[k]; }
	void set_lm_positions(const size_t *ids, size_t n, const double *in) { for (size_t i = 0; i < n; i++) for (size_t k = 0; k < LM_DIMS; k++) rba_state.lm_table[ids[i]].pos[k] = in[i * LM_DIMS + k]; }
	size_t lm_table_size() const { return rba_state.lm_table.size(); }
	void set_k2k_edge_poses(const size_t *ids, size_t n, const double *in) { const size_t PD = pose_t::storage_doubles(); for (size_t i = 0; i < n; i++) rba_state.k2k_edges[ids[i]].inv_pose.loadFrom(
		in + i * PD); }

private:
	rba_problem_state_t rba_state;
	mutable mrpt::utils::CTimeLogger m_profiler;
	mutable std::shared_ptr<numeric_backend> m_backend;
	int m_hip_device;
	// scratch kept between calls
	graph::capsule_builder m_builder; CapsuleData m_cd; graph::capsule_index m_ix; int m_capsule_stage = 0;
	std::vector<size_t> m_sel_edges, m_sel_lms;
	std::vector<uint32_t> m_lm_mark, m_lm_where; uint32_t m_mark_epoch = 0;
};

} // namespace srba
