/*
 * capsule_builder.h -- flattens ONE optimize_edges() call into the integer tables of a srba_problem_capsule (include/srba_hip.h).
 *
 * This is the host part of the reference's optimize_edges() before any arithmetic happens (impl/optimize_edges.h:71-313):
 *   S1  drop unknowns whose Jacobian column is empty                                   :71-119
 *   S3  involved observations, duplicated once per optimised edge on their path,
 *       and the observation -> residual-row map in which the LAST duplicate wins       :171-234   
 *   S4  key-frames whose numeric spanning trees are refreshed                          impl/jacobians.h:1020-1076
 *   +   the tables the reference reaches through pointers: spanning-tree pairs with their edge paths (spantree_update_numeric.h:111-127),
 *       Jacobian block tables in sweep order (jacobians.h:1094-1114), list_of_required_num_poses (:225-230,:900-901),
 *       Hessian / Schur plan (CapsuleData::build_plan).
 * It works on graph::topology only (integers); the typed front-end copies poses / coordinates / observation vectors in and out using the
 * index lists of capsule_index. All look-ups are epoch-stamped arrays kept between calls: no per-call node allocation.
 */
#pragma once
#include "capsule.h"
#include "graph_topology.h"
#include <iostream>

namespace srba {
namespace graph {

/** global ids behind the local slots of a capsule: what the typed layer needs to move numbers in and out */
struct capsule_index {
	std::vector<id32> edge_ids;    //!< local edge slot -> kf2kf edge id (unknowns first)
	std::vector<id32> unk_lms;     //!< unknown landmark slot -> feature id
	std::vector<id32> const_lms;   //!< constant landmark slot -> feature id
	std::vector<id32> obs_rows;    //!< residual row -> global observation index (with duplicates)
	std::vector<std::pair<id32, id32> > pairs; //!< spanning-tree pair -> (root, target), root > target
	size_t n_kfs_touched, n_lms_touched;
};

class capsule_builder {
public:
	capsule_builder() : m_tag(0) {}
	/** false: nothing left to optimise after filtering */
	bool build(topology &T, const std::vector<size_t> &edges_in, const std::vector<size_t> &lms_in, const std::vector<size_t> &obs_subset,
	           bool refresh_all_read_poses, bool with_schur, int P, int L, int O, int PD, CapsuleData &cd, capsule_index &ix, bool restore_twins = false) {
		next_tag(T);
		ix.edge_ids.clear(); ix.unk_lms.clear(); ix.const_lms.clear(); ix.obs_rows.clear(); ix.pairs.clear();
		// ---- S1
		m_kfs.clear();
		for (size_t i = 0; i < edges_in.size(); i++) {
			const id32 e = topology::narrow(edges_in[i]);
			if (e >= T.n_edges()) throw std::out_of_range("optimize_edges: unknown kf2kf edge id");
			if (T.edge_jp_count[e]) { m_edge_slot[e] = (int32_t)ix.edge_ids.size(); m_edge_tag[e] = m_tag; ix.edge_ids.push_back(e); m_kfs.push_back(T.edge_from[e]); m_kfs.push_back(T.edge_to[e]); }
			else std::cerr << "[RbaEngine::optimize_edges] *Warning*: Skipping optimization of k2k edge #" << e << " (" << T.edge_from[e] << "->" << T.edge_to[e] <<
				") since no observation depends on it.\n";
		}
		std::sort(m_kfs.begin(), m_kfs.end()); ix.n_kfs_touched = (size_t)(std::unique(m_kfs.begin(), m_kfs.end()) - m_kfs.begin());
		for (size_t i = 0; i < lms_in.size(); i++) {
			const id32 l = topology::narrow(lms_in[i]);
			if (l >= T.lm_base.size() || T.lm_base[l] == NIL) throw std::invalid_argument("Trying to optimize an unknown feature ID");
			if (T.lm_known[l]) throw std::invalid_argument("Trying to optimize a feature with fixed (known) value");
			if (T.lm_df_count[l]) { m_lm_ref[l] = (int32_t)ix.unk_lms.size(); m_lm_tag[l] = m_tag; ix.unk_lms.push_back(l); }
			else std::cerr << "[RbaEngine::optimize_edges] *Warning*: Skipping optimization of k2f edge #" << l << " since no observation depends on it.\n";
		}
		ix.n_lms_touched = ix.unk_lms.size();
		const int nK = (int)ix.edge_ids.size(), nF = (int)ix.unk_lms.size();
		if (!nK && !nF) { std::cerr << "[RbaEngine::optimize_edges] *Warning*: Skipping optimization since no observation depends on any of the given variables.\n"; return false; }
		reset(cd); cd.P = P; cd.L = L; cd.O = O; cd.PD = PD; cd.n_unk_edges = nK; cd.n_unk_lms = nF;
		for (int i = 0; i < nK; i++) cd.unk_edge_ids.push_back(ix.edge_ids[i]);
		for (int i = 0; i < nF; i++) cd.unk_lm_ids.push_back(ix.unk_lms[i]);

		// ---- S3: residual rows
		if (obs_subset.empty()) {
			for (int i = 0; i < nK; i++) for (id32 b = T.edge_jp_head[ix.edge_ids[i]]; b != NIL; b = T.jp[b].next) { const id32 o = T.jp[b].obs; m_obs_row[o] = (int32_t)ix.obs_rows.size();
				m_obs_tag[o] = m_tag; ix.obs_rows.push_back(o); }
			for (int i = 0; i < nF; i++) for (id32 o = T.lm_df_head[ix.unk_lms[i]]; o != NIL; o = T.obs_next_in_lm[o]) if (m_obs_tag[o] != m_tag) { m_obs_row[o] = (int32_t)ix.obs_rows.size();
				m_obs_tag[o] = m_tag; ix.obs_rows.push_back(o); }
		} else for (size_t i = 0; i < obs_subset.size(); i++) { const id32 o = topology::narrow(obs_subset[i]);
			if (o >= T.n_observations()) throw std::out_of_range("optimize_edges: unknown observation index"); m_obs_row[o] = (int32_t)ix.obs_rows.size(); m_obs_tag[o] = m_tag;
			ix.obs_rows.push_back(o); }
		const size_t nObs = ix.obs_rows.size();

		// ---- S4: roots of the numeric spanning trees to refresh, ascending
		m_roots.clear();
		for (int i = 0; i < nK; i++) for (id32 b = T.edge_jp_head[ix.edge_ids[i]]; b != NIL; b = T.jp[b].next) { const id32 o = T.jp[b].obs; m_roots.push_back(T.jp[b].kf_d);
			m_roots.push_back(T.obs_kf[o]); m_roots.push_back(T.lm_base[T.obs_lm[o]]); }
		for (int i = 0; i < nF; i++) for (id32 o = T.lm_df_head[ix.unk_lms[i]]; o != NIL; o = T.obs_next_in_lm[o]) { m_roots.push_back(T.lm_base[T.obs_lm[o]]); m_roots.push_back(T.obs_kf[o]); }
		std::sort(m_roots.begin(), m_roots.end()); m_roots.erase(std::unique(m_roots.begin(), m_roots.end()), m_roots.end());

		// ---- spanning-tree pair table: every stored path of every root (= the entries of its row towards smaller ids), local edge table
		cd.pair_path_off.push_back(0);
		for (size_t q = 0; q < m_roots.size(); q++) {
			const id32 root = m_roots[q]; m_root_base[root] = (int32_t)ix.pairs.size(); m_root_tag[root] = m_tag;
			const st_entry *row = T.st.row(root); const size_t len = T.st.len(root);
			for (size_t k = 0; k < len && row[k].trg < root; k++) {
				ix.pairs.push_back(std::make_pair(root, row[k].trg));
				id32 cur = root;
				for (uint32_t h = 0; h < row[k].path_len; h++) {
					const id32 e = T.path_pool[row[k].path + h];
					if (m_edge_tag[e] != m_tag) { m_edge_tag[e] = m_tag; m_edge_slot[e] = (int32_t)ix.edge_ids.size(); ix.edge_ids.push_back(e); }
					const int inv = (T.edge_to[e] == cur) ? 0 : 1; // walking against the edge direction composes inv_pose itself, along it its inverse (spantree_update_numeric.h:47-65)
					cur = T.other_end(e, cur);
					cd.path_edge.push_back((m_edge_slot[e] << 1) | inv);
				}
				cd.pair_path_off.push_back((int32_t)cd.path_edge.size());
			}
		}
		const size_t nPairs = ix.pairs.size();
		cd.pair_needed.assign(nPairs, 0); cd.pose_required.assign(2 * nPairs, 0);

		// ---- residual rows: pose, landmark reference, validity slot
		int n_valid = 0;
		cd.obs_pose.resize(nObs); cd.obs_lm.resize(nObs); cd.obs_valid.resize(nObs);
		for (size_t i = 0; i < nObs; i++) {
			const id32 o = ix.obs_rows[i], okf = T.obs_kf[o], lm = T.obs_lm[o], base = T.lm_base[lm];
			cd.obs_pose[i] = (okf == base) ? -1 : pose_index(T, okf, base);
			if (refresh_all_read_poses && cd.obs_pose[i] >= 0) cd.pose_required[cd.obs_pose[i]] = 1;
			cd.obs_lm[i] = lm_reference(lm, ix);
			if (m_valid_tag[o] != m_tag) { m_valid_tag[o] = m_tag; m_valid_slot[o] = n_valid++; }
			cd.obs_valid[i] = m_valid_slot[o];
		}
		cd.n_valid = n_valid;

		// ---- Jacobian block tables in the reference's sweep order + the poses they read
		m_bp_row.clear(); m_bf_row.clear();
		cd.colp_off.push_back(0);
		for (int i = 0; i < nK; i++) {
			for (id32 b = T.edge_jp_head[ix.edge_ids[i]]; b != NIL; b = T.jp[b].next) {
				const jp_block &s = T.jp[b]; const id32 o = s.obs, lm = T.obs_lm[o], base = T.lm_base[lm];
				if (m_obs_tag[o] != m_tag) throw std::logic_error("optimize_edges: a Jacobian block refers to an observation outside the given subset");
				const int32_t A = s.has_A ? pose_index(T, T.obs_kf[o], s.kf_d) : -1, D = pose_index(T, s.kf_d, base);
				cd.bp_col.push_back(i); cd.bp_res.push_back(m_obs_row[o]); cd.bp_A.push_back(A); cd.bp_D.push_back(D); cd.bp_lm.push_back(lm_reference(lm, ix)); cd.bp_normal.push_back(s.normal_dir);
				m_bp_row.push_back(o);
				if (A >= 0) cd.pose_required[A] = 1;
				cd.pose_required[D] = 1;
			}
			cd.colp_off.push_back((int32_t)cd.bp_col.size());
		}
		cd.colf_off.push_back(0);
		for (int i = 0; i < nF; i++) {
			for (id32 o = T.lm_df_head[ix.unk_lms[i]]; o != NIL; o = T.obs_next_in_lm[o]) {
				if (m_obs_tag[o] != m_tag) throw std::logic_error("optimize_edges: a Jacobian block refers to an observation outside the given subset");
				const int32_t pi = T.obs_first_of_unknown[o] ? -1 : pose_index(T, T.obs_kf[o], T.lm_base[T.obs_lm[o]]);
				cd.bf_col.push_back(i); cd.bf_res.push_back(m_obs_row[o]); cd.bf_pose.push_back(pi); m_bf_row.push_back(o);
				if (pi >= 0) cd.pose_required[pi] = 1;
			}
			cd.colf_off.push_back((int32_t)cd.bf_col.size());
		}
		for (size_t p = 0; p < nPairs; p++) cd.pair_needed[p] = (cd.pose_required[2 * p] || cd.pose_required[2 * p + 1]) ? 1 : 0;
		if (restore_twins) for (size_t p = 0; p < nPairs; p++) if (cd.pair_needed[p]) cd.pose_required[2 * p] = cd.pose_required[2 * p + 1] = 1;
			// (extension) back up / restore both poses of every refreshed pair
		cd.pair_kfs.resize(nPairs); for (size_t p = 0; p < nPairs; p++) cd.pair_kfs[p] = std::make_pair((uint64_t)ix.pairs[p].first, (uint64_t)ix.pairs[p].second);
		cd.build_plan(m_bp_row, m_bf_row, with_schur);
		return true;
	}

private:
	void next_tag(const topology &T) {
		if (++m_tag == 0) { std::fill(m_edge_tag.begin(), m_edge_tag.end(), 0u); std::fill(m_lm_tag.begin(), m_lm_tag.end(), 0u); std::fill(m_obs_tag.begin(), m_obs_tag.end(), 0u);
			std::fill(m_valid_tag.begin(), m_valid_tag.end(), 0u); std::fill(m_root_tag.begin(), m_root_tag.end(), 0u); m_tag = 1; }
		grow(m_edge_tag, T.n_edges()); grow(m_edge_slot, T.n_edges()); grow(m_lm_tag, T.lm_base.size()); grow(m_lm_ref, T.lm_base.size());
		grow(m_obs_tag, T.n_observations()); grow(m_obs_row, T.n_observations()); grow(m_valid_tag, T.n_observations()); grow(m_valid_slot, T.n_observations());
		grow(m_root_tag, T.n_keyframes()); grow(m_root_base, T.n_keyframes());
	}
	template <class V> static void grow(V &v, size_t n) { if (v.size() < n) v.resize(n + n / 2 + 16, 0); }
	/** index of num[src][trg] in the capsule's pose table: pair (root > target) at 2p, its inverse at 2p+1 */
	int32_t pose_index(const topology &T, id32 src, id32 trg) const {
		const id32 hi = src > trg ? src : trg, lo = src > trg ? trg : src;
		if (m_root_tag[hi] == m_tag) {
			const uint32_t pos = T.st.lower(hi, lo);
			if (pos < T.st.len(hi) && T.st.row(hi)[pos].trg == lo) return 2 * (m_root_base[hi] + (int32_t)pos) + (src > trg ? 0 : 1);
		}
		throw std::logic_error("optimize_edges: numeric spanning-tree entry not available (graph deeper than max_tree_depth?)");
	}
	/** >= 0: unknown landmark slot; < 0: constant landmark -(1+slot), slots in order of first use */
	int32_t lm_reference(id32 lm, capsule_index &ix) {
		if (m_lm_tag[lm] != m_tag) { m_lm_tag[lm] = m_tag; m_lm_ref[lm] = -1 - (int32_t)ix.const_lms.size(); ix.const_lms.push_back(lm); }
		return m_lm_ref[lm];
	}
	static void reset(CapsuleData &cd) { const CapsuleData fresh; cd = fresh; }

	uint32_t m_tag;
	std::vector<uint32_t> m_edge_tag, m_lm_tag, m_obs_tag, m_valid_tag, m_root_tag;
	std::vector<int32_t> m_edge_slot, m_lm_ref, m_obs_row, m_valid_slot, m_root_base;
	std::vector<id32> m_kfs, m_roots; std::vector<uint64_t> m_bp_row, m_bf_row;
};

} // namespace graph
} // namespace srba
