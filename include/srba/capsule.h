/*
 * capsule.h -- owning storage for one srba_problem_capsule (include/srba_hip.h) plus the host-side "symbolic"
 * planning that the reference performs inside optimize_edges():
 *   - sparse_hessian_build_symbolic()   include/srba/impl/sparse_hessian_build_symbolic.h:22-237
 *   - SchurComplement<> constructor     include/srba/impl/schur.h:25-159
 * The reference finds common observations with O(nK^2)+O(nK*nF) sorted-merge passes over std::map columns
 * (SURVEY App. B-5); here the same block / term lists (same content, same order) are produced by one pass over the
 * observation rows followed by a sort, which stays linear in the number of Jacobian blocks.
 */
#pragma once
#include "../srba_hip.h"
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace srba {

struct CapsuleData {
	// sizes that are not implied by vector lengths
	int n_unk_edges = 0, n_unk_lms = 0, n_valid = 0;
	std::vector<double> edge_pose, ulm_pos, klm_pos, pose, obs_z, ulm_inf;
	std::vector<int32_t> pair_path_off, path_edge, obs_pose, obs_lm, obs_valid;
	std::vector<uint8_t> pair_needed, pose_required, bp_normal, ulm_inf_valid;
	std::vector<int32_t> bp_col, bp_res, bp_A, bp_D, bp_lm, colp_off, bf_col, bf_res, bf_pose, colf_off;
	std::vector<int32_t> hap_i, hap_j, hap_term_off, hap_t1, hap_t2, hf_i, hf_j, hf_term_off, hf_t1, hf_t2;
	std::vector<int32_t> hapf_i, hapf_j, hapf_term_off, hapf_t1, hapf_t2, hap_diag, hf_diag;
	std::vector<int32_t> sch_term_off, sch_b1, sch_b2, sch_lm, lm_hapf_off, lm_hapf_idx;
	int PD = 3, L = 3, O = 3, P = 3;
	// bookkeeping for the host engine (not part of the ABI): global ids behind the local slots
	std::vector<uint64_t> unk_edge_ids, unk_lm_ids; std::vector<std::pair<uint64_t, uint64_t> > pair_kfs;

	/** Build HAp/Hf/HApf block + term lists from the Jacobian block tables, and the Schur plan if requested.
	 *  bp_row/bf_row: global observation index of each block (ascending inside each column). */
	void build_plan(const std::vector<uint64_t> &bp_row, const std::vector<uint64_t> &bf_row, bool with_schur) {
		const int nK = n_unk_edges, nF = n_unk_lms;
		typedef std::pair<int, int> key_t; // (col j, row i) so that std::map order == reference iteration order getCol(j)[i]
		typedef std::vector<std::pair<int32_t, int32_t> > terms_t;
		std::map<key_t, terms_t> HAp, Hf; std::map<key_t, terms_t> HApf; // HApf keyed (row i = edge, col j = lm): stored by rows
		// group blocks by observation row, ascending row; inside a row ascending column slot
		std::map<uint64_t, std::vector<std::pair<int, int> > > rows_p; // row -> (col slot, block idx)
		for (size_t b = 0; b < bp_col.size(); b++) rows_p[bp_row[b]].push_back(std::make_pair(bp_col[b], (int)b));
		std::map<uint64_t, std::pair<int, int> > rows_f; // row -> (lm slot, block idx): one landmark per observation
		for (size_t b = 0; b < bf_col.size(); b++) rows_f[bf_row[b]] = std::make_pair(bf_col[b], (int)b);
		for (int i = 0; i < nK; i++) HAp[key_t(i, i)]; // diagonal blocks always exist (columns are non-empty after filtering)
		for (int i = 0; i < nF; i++) Hf[key_t(i, i)];
		for (std::map<uint64_t, std::vector<std::pair<int, int> > >::iterator it = rows_p.begin(); it != rows_p.end(); ++it) {
			std::vector<std::pair<int, int> > &v = it->second; std::sort(v.begin(), v.end());
			for (size_t a = 0; a < v.size(); a++) for (size_t b = a; b < v.size(); b++) HAp[key_t(v[b].first, v[a].first)].push_back(std::make_pair(v[a].second, v[b].second));
			std::map<uint64_t, std::pair<int, int> >::const_iterator f = rows_f.find(it->first);
			if (f != rows_f.end()) for (size_t a = 0; a < v.size(); a++) HApf[key_t(v[a].first, f->second.first)].push_back(std::make_pair(v[a].second, f->second.second));
		}
		for (std::map<uint64_t, std::pair<int, int> >::iterator it = rows_f.begin(); it != rows_f.end(); ++it) Hf[key_t(it->second.first, it->second.first)].push_back(std::make_pair(it->second.second, it->second.second));
		// HApf flatten (ordered by (i,j))
		hapf_i.clear(); hapf_j.clear(); hapf_term_off.assign(1, 0); hapf_t1.clear(); hapf_t2.clear();
		std::map<key_t, int> hapf_index;
		for (std::map<key_t, terms_t>::iterator it = HApf.begin(); it != HApf.end(); ++it) {
			hapf_index[it->first] = (int)hapf_i.size(); hapf_i.push_back(it->first.first); hapf_j.push_back(it->first.second);
			for (size_t t = 0; t < it->second.size(); t++) { hapf_t1.push_back(it->second[t].first); hapf_t2.push_back(it->second[t].second); }
			hapf_term_off.push_back((int)hapf_t1.size());
		}
		// lm -> HApf blocks in ascending edge slot
		lm_hapf_off.assign(nF + 1, 0); lm_hapf_idx.assign(hapf_i.size(), 0);
		for (size_t b = 0; b < hapf_j.size(); b++) lm_hapf_off[hapf_j[b] + 1]++;
		for (int l = 0; l < nF; l++) lm_hapf_off[l + 1] += lm_hapf_off[l];
		{ std::vector<int> cur(lm_hapf_off.begin(), lm_hapf_off.end() - 1); for (size_t b = 0; b < hapf_j.size(); b++) lm_hapf_idx[cur[hapf_j[b]]++] = (int)b; }
		// Schur plan: for every landmark, every pair of edges that see it (schur.h:56-157). Creates fill-in HAp blocks.
		std::map<key_t, std::vector<int32_t> > sch; // HAp key -> flat triplets (b1,b2,lm)
		const bool schur = with_schur && nF > 0 && nK > 0;
		if (schur) {
			for (int l = 0; l < nF; l++) // ascending lm => ascending inside every block's term list (set_intersection order)
				for (int a = lm_hapf_off[l]; a < lm_hapf_off[l + 1]; a++) for (int b = a; b < lm_hapf_off[l + 1]; b++) {
					const int ba = lm_hapf_idx[a], bb = lm_hapf_idx[b]; // hapf_i[ba] <= hapf_i[bb]
					const key_t k(hapf_i[bb], hapf_i[ba]);
					HAp[k]; // fill-in block if absent (schur.h:142-149)
					std::vector<int32_t> &v = sch[k]; v.push_back(ba); v.push_back(bb); v.push_back(l);
				}
		}
		// HAp flatten, ordered by (col j, row i)
		hap_i.clear(); hap_j.clear(); hap_term_off.assign(1, 0); hap_t1.clear(); hap_t2.clear(); hap_diag.assign(nK, -1);
		sch_term_off.clear(); sch_b1.clear(); sch_b2.clear(); sch_lm.clear(); if (schur) sch_term_off.push_back(0);
		for (std::map<key_t, terms_t>::iterator it = HAp.begin(); it != HAp.end(); ++it) {
			if (it->first.first == it->first.second) hap_diag[it->first.first] = (int)hap_i.size();
			hap_j.push_back(it->first.first); hap_i.push_back(it->first.second);
			for (size_t t = 0; t < it->second.size(); t++) { hap_t1.push_back(it->second[t].first); hap_t2.push_back(it->second[t].second); }
			hap_term_off.push_back((int)hap_t1.size());
			if (schur) {
				std::map<key_t, std::vector<int32_t> >::iterator s = sch.find(it->first);
				if (s != sch.end()) for (size_t t = 0; t + 2 < s->second.size(); t += 3) { sch_b1.push_back(s->second[t]); sch_b2.push_back(s->second[t + 1]); sch_lm.push_back(s->second[t + 2]); }
				sch_term_off.push_back((int)sch_b1.size());
			}
		}
		hf_i.clear(); hf_j.clear(); hf_term_off.assign(1, 0); hf_t1.clear(); hf_t2.clear(); hf_diag.assign(nF, -1);
		for (std::map<key_t, terms_t>::iterator it = Hf.begin(); it != Hf.end(); ++it) {
			if (it->first.first == it->first.second) hf_diag[it->first.first] = (int)hf_i.size();
			hf_j.push_back(it->first.first); hf_i.push_back(it->first.second);
			for (size_t t = 0; t < it->second.size(); t++) { hf_t1.push_back(it->second[t].first); hf_t2.push_back(it->second[t].second); }
			hf_term_off.push_back((int)hf_t1.size());
		}
	}

	template <class T> static T *ptr(std::vector<T> &v) { return v.empty() ? (T *)0 : &v[0]; }
	/** Plain-C view over the vectors (valid while this object is alive and unmodified). */
	srba_problem_capsule view() {
		srba_problem_capsule c; std::memset(&c, 0, sizeof(c));
		c.n_edges = (int)(edge_pose.size() / PD); c.n_unk_edges = n_unk_edges; c.n_unk_lms = n_unk_lms; c.n_known_lms = (int)(klm_pos.size() / L);
		c.n_pairs = (int)pair_needed.size(); c.n_path = (int)path_edge.size(); c.n_obs = (int)obs_pose.size(); c.n_valid = n_valid;
		c.n_bp = (int)bp_col.size(); c.n_bf = (int)bf_col.size();
		c.n_hap = (int)hap_i.size(); c.n_hap_terms = (int)hap_t1.size(); c.n_hf = (int)hf_i.size(); c.n_hf_terms = (int)hf_t1.size();
		c.n_hapf = (int)hapf_i.size(); c.n_hapf_terms = (int)hapf_t1.size(); c.n_sch_terms = (int)sch_b1.size();
		pose.resize((size_t)2 * c.n_pairs * PD); ulm_inf.resize((size_t)n_unk_lms * L * L); ulm_inf_valid.resize(n_unk_lms);
		c.edge_pose = ptr(edge_pose); c.ulm_pos = ptr(ulm_pos); c.klm_pos = ptr(klm_pos);
		c.pair_path_off = ptr(pair_path_off); c.path_edge = ptr(path_edge); c.pair_needed = ptr(pair_needed); c.pose_required = ptr(pose_required); c.pose = ptr(pose);
		c.obs_pose = ptr(obs_pose); c.obs_lm = ptr(obs_lm); c.obs_valid = ptr(obs_valid); c.obs_z = ptr(obs_z);
		c.bp_col = ptr(bp_col); c.bp_res = ptr(bp_res); c.bp_A = ptr(bp_A); c.bp_D = ptr(bp_D); c.bp_lm = ptr(bp_lm); c.bp_normal = ptr(bp_normal); c.colp_off = ptr(colp_off);
		c.bf_col = ptr(bf_col); c.bf_res = ptr(bf_res); c.bf_pose = ptr(bf_pose); c.colf_off = ptr(colf_off);
		c.hap_i = ptr(hap_i); c.hap_j = ptr(hap_j); c.hap_term_off = ptr(hap_term_off); c.hap_t1 = ptr(hap_t1); c.hap_t2 = ptr(hap_t2);
		c.hf_i = ptr(hf_i); c.hf_j = ptr(hf_j); c.hf_term_off = ptr(hf_term_off); c.hf_t1 = ptr(hf_t1); c.hf_t2 = ptr(hf_t2);
		c.hapf_i = ptr(hapf_i); c.hapf_j = ptr(hapf_j); c.hapf_term_off = ptr(hapf_term_off); c.hapf_t1 = ptr(hapf_t1); c.hapf_t2 = ptr(hapf_t2);
		c.hap_diag = ptr(hap_diag); c.hf_diag = ptr(hf_diag);
		c.sch_term_off = ptr(sch_term_off); c.sch_b1 = ptr(sch_b1); c.sch_b2 = ptr(sch_b2); c.sch_lm = ptr(sch_lm);
		c.lm_hapf_off = ptr(lm_hapf_off); c.lm_hapf_idx = ptr(lm_hapf_idx);
		c.ulm_inf = ptr(ulm_inf); c.ulm_inf_valid = ptr(ulm_inf_valid);
		return c;
	}

	// ------------------------------------------------------------------ binary (de)serialisation: golden fixtures
	template <class T> static void wv(FILE *f, const std::vector<T> &v) { const uint64_t n = v.size(); fwrite(&n, 8, 1, f); if (n) fwrite(&v[0], sizeof(T), n, f); }
	template <class T> static void rv(FILE *f, std::vector<T> &v) { uint64_t n = 0; if (fread(&n, 8, 1, f) != 1) throw std::runtime_error("capsule: short read"); v.resize(n); if (n && fread(&v[0], sizeof(T), n, f) != n) throw std::runtime_error("capsule: short read"); }
	template <class FN> void for_all_vectors(FN &fn) {
		fn(edge_pose); fn(ulm_pos); fn(klm_pos); fn(obs_z);
		fn(pair_path_off); fn(path_edge); fn(obs_pose); fn(obs_lm); fn(obs_valid); fn(pair_needed); fn(pose_required); fn(bp_normal);
		fn(bp_col); fn(bp_res); fn(bp_A); fn(bp_D); fn(bp_lm); fn(colp_off); fn(bf_col); fn(bf_res); fn(bf_pose); fn(colf_off);
		fn(hap_i); fn(hap_j); fn(hap_term_off); fn(hap_t1); fn(hap_t2); fn(hf_i); fn(hf_j); fn(hf_term_off); fn(hf_t1); fn(hf_t2);
		fn(hapf_i); fn(hapf_j); fn(hapf_term_off); fn(hapf_t1); fn(hapf_t2); fn(hap_diag); fn(hf_diag);
		fn(sch_term_off); fn(sch_b1); fn(sch_b2); fn(sch_lm); fn(lm_hapf_off); fn(lm_hapf_idx);
	}
	struct Writer { FILE *f; template <class T> void operator()(std::vector<T> &v) { wv(f, v); } };
	struct Reader { FILE *f; template <class T> void operator()(std::vector<T> &v) { rv(f, v); } };
	void write(FILE *f) { int32_t h[7] = {n_unk_edges, n_unk_lms, n_valid, PD, L, O, P}; fwrite(h, 4, 7, f); Writer w = {f}; for_all_vectors(w); }
	void read(FILE *f) { int32_t h[7]; if (fread(h, 4, 7, f) != 7) throw std::runtime_error("capsule: short read"); n_unk_edges = h[0]; n_unk_lms = h[1]; n_valid = h[2]; PD = h[3]; L = h[4]; O = h[5]; P = h[6]; Reader r = {f}; for_all_vectors(r); }
};

} // namespace srba
