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

	/** Hessian block + term lists (sparse_hessian_build_symbolic.h:22-237) and, if requested, the Schur plan (schur.h:25-159) from the Jacobian
	 *  block tables. bp_row/bf_row: global observation index of each block (ascending inside each column).
	 *  One pass over the observation rows emits every (block, term) pair tagged with its block key; a stable sort by key then yields the
	 *  reference's orders: blocks by (column j, row i) like getCol(j)[i] (HApf: by (edge i, landmark j), it is stored by rows), terms inside a
	 *  block by ascending observation. Linear in the number of emitted terms (+ the sorts); no node-based containers. */
	void build_plan(const std::vector<uint64_t> &bp_row, const std::vector<uint64_t> &bf_row, bool with_schur) {
		const int nK = n_unk_edges, nF = n_unk_lms;
		struct term { uint64_t key; int32_t a, b; };
		struct by_key { bool operator()(const term &x, const term &y) const { return x.key < y.key; } };
		struct rowrec { uint64_t row; int32_t col, blk; };
		struct by_row_col { bool operator()(const rowrec &x, const rowrec &y) const { return x.row != y.row ? x.row < y.row : x.col < y.col; } };
		// dh_dAp blocks grouped by observation row (ascending), ascending column slot inside a row
		std::vector<rowrec> rp(bp_col.size());
		for (size_t b = 0; b < bp_col.size(); b++) { rp[b].row = bp_row[b]; rp[b].col = bp_col[b]; rp[b].blk = (int32_t)b; }
		std::sort(rp.begin(), rp.end(), by_row_col());
		// dh_df blocks by observation row: an observation sees exactly one landmark
		std::vector<rowrec> rf(bf_col.size());
		for (size_t b = 0; b < bf_col.size(); b++) { rf[b].row = bf_row[b]; rf[b].col = bf_col[b]; rf[b].blk = (int32_t)b; }
		std::sort(rf.begin(), rf.end(), by_row_col());
		std::vector<term> tp, tpf; tp.reserve(2 * rp.size()); tpf.reserve(rp.size());
		size_t f = 0;
		for (size_t g = 0; g < rp.size();) {
			size_t h = g; while (h < rp.size() && rp[h].row == rp[g].row) h++;
			for (size_t a = g; a < h; a++) for (size_t b = a; b < h; b++) { term t = {(uint64_t)rp[b].col * (uint64_t)nK + (uint64_t)rp[a].col, rp[a].blk, rp[b].blk}; tp.push_back(t); }
			while (f < rf.size() && rf[f].row < rp[g].row) f++;
			if (f < rf.size() && rf[f].row == rp[g].row) for (size_t a = g; a < h; a++) { term t = {(uint64_t)rp[a].col * (uint64_t)nF + (uint64_t)rf[f].col, rp[a].blk, rf[f].blk}; tpf.push_back(t); }
			g = h;
		}
		std::stable_sort(tp.begin(), tp.end(), by_key()); std::stable_sort(tpf.begin(), tpf.end(), by_key());
		// ---- HApf: blocks in (edge, landmark) order
		hapf_i.clear(); hapf_j.clear(); hapf_term_off.assign(1, 0); hapf_t1.clear(); hapf_t2.clear();
		for (size_t t = 0; t < tpf.size(); t++) {
			if (t == 0 || tpf[t].key != tpf[t - 1].key) { if (t) hapf_term_off.push_back((int32_t)hapf_t1.size()); hapf_i.push_back((int32_t)(tpf[t].key / (uint64_t)nF));
				hapf_j.push_back((int32_t)(tpf[t].key % (uint64_t)nF)); }
			hapf_t1.push_back(tpf[t].a); hapf_t2.push_back(tpf[t].b);
		}
		if (!tpf.empty()) hapf_term_off.push_back((int32_t)hapf_t1.size());
		// landmark -> its HApf blocks by ascending edge slot (counting sort; the block list is already ascending in the edge slot)
		lm_hapf_off.assign(nF + 1, 0); lm_hapf_idx.assign(hapf_i.size(), 0);
		for (size_t b = 0; b < hapf_j.size(); b++) lm_hapf_off[hapf_j[b] + 1]++;
		for (int l = 0; l < nF; l++) lm_hapf_off[l + 1] += lm_hapf_off[l];
		{ std::vector<int32_t> cur(lm_hapf_off.begin(), lm_hapf_off.end() - 1); for (size_t b = 0; b < hapf_j.size(); b++) lm_hapf_idx[cur[hapf_j[b]]++] = (int32_t)b; }
		// ---- Schur plan: per landmark every pair of edges that see it (schur.h:56-157); pairs without a J^t J block create fill-in HAp blocks (:142-149)
		const bool schur = with_schur && nF > 0 && nK > 0;
		std::vector<term> ts; std::vector<int32_t> ts_lm;
		if (schur) {
			struct sterm { uint64_t key; int32_t a, b, l; };
			struct by_skey { bool operator()(const sterm &x, const sterm &y) const { return x.key < y.key; } };
			std::vector<sterm> raw;
			for (int l = 0; l < nF; l++) // ascending landmark => ascending inside every block's list
				for (int a = lm_hapf_off[l]; a < lm_hapf_off[l + 1]; a++) for (int b = a; b < lm_hapf_off[l + 1]; b++) {
					const int32_t ba = lm_hapf_idx[a], bb = lm_hapf_idx[b]; // hapf_i[ba] <= hapf_i[bb]
					sterm t = {(uint64_t)hapf_i[bb] * (uint64_t)nK + (uint64_t)hapf_i[ba], ba, bb, l}; raw.push_back(t);
				}
			std::stable_sort(raw.begin(), raw.end(), by_skey());
			ts.resize(raw.size()); ts_lm.resize(raw.size());
			for (size_t t = 0; t < raw.size(); t++) { term x = {raw[t].key, raw[t].a, raw[t].b}; ts[t] = x; ts_lm[t] = raw[t].l; }
		}
		// ---- HAp: union of the J^t J keys, the diagonal (always present: columns are non-empty after filtering) and the Schur fill-in keys
		std::vector<uint64_t> keys; keys.reserve(tp.size() / 2 + nK + ts.size() / 2);
		for (int i = 0; i < nK; i++) keys.push_back((uint64_t)i * (uint64_t)nK + (uint64_t)i);
		for (size_t t = 0; t < tp.size(); t++) if (t == 0 || tp[t].key != tp[t - 1].key) keys.push_back(tp[t].key);
		for (size_t t = 0; t < ts.size(); t++) if (t == 0 || ts[t].key != ts[t - 1].key) keys.push_back(ts[t].key);
		std::sort(keys.begin(), keys.end()); keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
		hap_i.clear(); hap_j.clear(); hap_term_off.assign(1, 0); hap_t1.clear(); hap_t2.clear(); hap_diag.assign(nK, -1);
		sch_term_off.clear(); sch_b1.clear(); sch_b2.clear(); sch_lm.clear(); if (schur) sch_term_off.push_back(0);
		size_t it = 0, is = 0;
		for (size_t k = 0; k < keys.size(); k++) {
			const int32_t j = (int32_t)(keys[k] / (uint64_t)nK), i = (int32_t)(keys[k] % (uint64_t)nK);
			if (i == j) hap_diag[i] = (int32_t)hap_i.size();
			hap_j.push_back(j); hap_i.push_back(i);
			for (; it < tp.size() && tp[it].key == keys[k]; it++) { hap_t1.push_back(tp[it].a); hap_t2.push_back(tp[it].b); }
			hap_term_off.push_back((int32_t)hap_t1.size());
			if (schur) { for (; is < ts.size() && ts[is].key == keys[k]; is++) { sch_b1.push_back(ts[is].a); sch_b2.push_back(ts[is].b); sch_lm.push_back(ts_lm[is]); }
				sch_term_off.push_back((int32_t)sch_b1.size()); }
		}
		// ---- Hf: block-diagonal (one landmark per observation); the terms of landmark l are its own dh_df blocks in column order
		hf_i.clear(); hf_j.clear(); hf_term_off.assign(1, 0); hf_t1.clear(); hf_t2.clear(); hf_diag.assign(nF, -1);
		for (int l = 0; l < nF; l++) {
			hf_diag[l] = l; hf_i.push_back(l); hf_j.push_back(l);
			for (int32_t b = colf_off[l]; b < colf_off[l + 1]; b++) { hf_t1.push_back(b); hf_t2.push_back(b); }
			hf_term_off.push_back((int32_t)hf_t1.size());
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
	template <class T> static void rv(FILE *f, std::vector<T> &v) { uint64_t n = 0; if (fread(&n, 8, 1, f) != 1) throw std::runtime_error("capsule: short read"); v.resize(n); if (n && fread(&v[0],
		sizeof(T), n, f) != n) throw std::runtime_error("capsule: short read"); }
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
	void read(FILE *f) { int32_t h[7]; if (fread(h, 4, 7, f) != 7) throw std::runtime_error("capsule: short read"); n_unk_edges = h[0]; n_unk_lms = h[1]; n_valid = h[2]; PD = h[3]; L = h[4];
		O = h[5]; P = h[6]; Reader r = {f}; for_all_vectors(r); }
};

} // namespace srba
