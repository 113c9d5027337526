/*
 * graph_topology.h -- the integer ("symbolic") half of the SRBA host layer on flat, index-linked containers.
 *
 * Everything in the reference that is pure bookkeeping -- key-frames, kf2kf edges and their adjacency, observations, landmarks'
 * base key-frames, the depth-limited symbolic spanning trees and the symbolic Jacobian structure -- lives here in ONE non-template
 * class over 32-bit indices.  The typed payload (poses, landmark coordinates, observation vectors) stays in srba::RbaEngine<>.
 * The reference keeps the same information in a web of std::map / std::deque nodes holding raw pointers into each other
 * (srba_types.h:548-785); here
 *   - adjacency lists, per-key-frame observation lists and the Jacobian columns are intrusive singly-linked chains threaded
 *     through flat record arrays (O(1) append, insertion-order traversal, no allocation per item),
 *   - the two spanning-tree tables next_edge[][] / all_edges[][] and the index of the numeric pose num[][] are one sorted row of
 *     st_entry per source key-frame inside a single arena (binary search, contiguous scans),
 *   - "visited" sets are epoch-stamped arrays.
 * Behaviour that the numeric layer can observe is kept EXACTLY (north_star: "bit-exact on spanning-tree indices"):
 *   spanning-tree update rule        impl/spantree_update_symbolic.h:19-211  (strict '<' improvement, snapshot of the two node sets,
 *                                                                              "next" read through the live tables)
 *   stored shortest paths            impl/spantree_update_symbolic.h:227-300 (plain BFS, neighbours in edge-creation order, first found wins)
 *   symbolic Jacobian blocks         impl/add-observations.h:139-257
 *   local-area selection             impl/bfs_visitor.h:105-176 + RbaEngine.h:591-617
 * tests/test_graph_golden.py checks every table against dumps of the round-1 (std::map based) implementation.
 */
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <utility>
#include <vector>

namespace srba {
namespace graph {

typedef uint32_t id32;
static const id32 NIL = 0xffffffffu;

/** One (source -> target) entry of the symbolic spanning trees. Both directions of a pair are stored; the edge path only in the row of the larger id. */
struct st_entry {
	id32 trg;          //!< target key-frame (rows are sorted by this)
	id32 next;         //!< first hop from the source towards trg           (reference: TSpanTreeEntry::next)
	uint32_t dist;     //!< topological distance                            (reference: TSpanTreeEntry::distance)
	id32 path;         //!< offset into topology::path_pool of the edge ids source -> trg, NIL when source < trg
	uint16_t path_len, path_cap;
	int32_t num;       //!< slot of the numeric pose "trg as seen from source" in the typed layer's pool, -1 = none yet
};

/** Sorted rows of st_entry, one per key-frame, inside one arena. A row that outgrows its slot moves to the end of the arena. */
class st_table {
	struct row_ref { uint32_t off, len, cap; };
	std::vector<row_ref> m_rows; std::vector<st_entry> m_pool;
public:
	void clear() { m_rows.clear(); m_pool.clear(); }
	void ensure_rows(size_t n) { if (m_rows.size() < n) { row_ref z = {0, 0, 0}; m_rows.resize(n, z); } }
	size_t rows() const { return m_rows.size(); }
	size_t len(id32 src) const { return src < m_rows.size() ? m_rows[src].len : 0; }
	const st_entry *row(id32 src) const { return src < m_rows.size() ? &m_pool[0] + m_rows[src].off : (const st_entry *)0; }
	st_entry *row(id32 src) { return src < m_rows.size() ? &m_pool[0] + m_rows[src].off : (st_entry *)0; }
	/** position of trg in the row of src (lower bound) */
	uint32_t lower(id32 src, id32 trg) const {
		const row_ref &r = m_rows[src]; const st_entry *e = &m_pool[0] + r.off; uint32_t lo = 0, hi = r.len;
		while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (e[mid].trg < trg) lo = mid + 1; else hi = mid; }
		return lo;
	}
	st_entry *find(id32 src, id32 trg) {
		if (src >= m_rows.size() || !m_rows[src].len) return 0;
		const uint32_t p = lower(src, trg); st_entry *e = &m_pool[0] + m_rows[src].off;
		return (p < m_rows[src].len && e[p].trg == trg) ? e + p : (st_entry *)0;
	}
	const st_entry *find(id32 src, id32 trg) const { return const_cast<st_table *>(this)->find(src, trg); }
	/** new entry at its sorted place (must not exist); every pointer into the table is invalid afterwards */
	st_entry &insert(id32 src, id32 trg) {
		ensure_rows((size_t)src + 1);
		if (m_rows[src].len == m_rows[src].cap) { // relocate to the end of the arena with twice the room
			const uint32_t ncap = m_rows[src].cap ? 2 * m_rows[src].cap : 8, noff = (uint32_t)m_pool.size();
			m_pool.resize(m_pool.size() + ncap);
			if (m_rows[src].len) std::memcpy(&m_pool[noff], &m_pool[m_rows[src].off], sizeof(st_entry) * m_rows[src].len);
			m_rows[src].off = noff; m_rows[src].cap = ncap;
		}
		row_ref &r = m_rows[src]; const uint32_t p = r.len ? lower(src, trg) : 0; st_entry *e = &m_pool[0] + r.off;
		if (p < r.len) std::memmove(e + p + 1, e + p, sizeof(st_entry) * (r.len - p));
		r.len++;
		st_entry z = {trg, NIL, 0, NIL, 0, 0, -1}; e[p] = z; return e[p];
	}
};

/** Symbolic record of one dh_dAp block (reference TJacobianSymbolicInfo_dh_dAp, srba_types.h:204-240) */
struct jp_block { id32 obs, kf_d, next; uint8_t normal_dir, has_A; };

struct window_params { bool optimize_k2k_edges, optimize_landmarks; size_t min_times_seen; uint64_t max_visitable_kf; };

class topology {
public:
	// ---- key-frames
	std::vector<id32> kf_adj_head, kf_adj_tail, kf_obs_head, kf_obs_tail; std::vector<uint32_t> kf_degree;
	// ---- kf2kf edges (id = creation order = Jacobian column)
	std::vector<id32> edge_from, edge_to, edge_next_at_from, edge_next_at_to, edge_jp_head, edge_jp_tail; std::vector<uint32_t> edge_jp_count;
	// ---- observations (id = global observation index)
	std::vector<id32> obs_kf, obs_lm, obs_next_in_kf, obs_next_in_lm; std::vector<uint8_t> obs_known, obs_first_of_unknown;
	// ---- landmarks, indexed by feature id
	std::vector<id32> lm_base, lm_df_head, lm_df_tail; std::vector<uint8_t> lm_known; std::vector<uint32_t> lm_df_count;
	// ---- symbolic Jacobian blocks of the kf2kf columns
	std::vector<jp_block> jp;
	// ---- symbolic spanning trees
	st_table st; std::vector<id32> path_pool; int32_t num_slots;
	struct loose_num { id32 src, trg; int32_t slot; }; std::vector<loose_num> loose; //!< numeric entries of pairs without a symbolic entry (not expected)
	std::vector<id32> last_touched_kfs; //!< sorted: end points of the edges created with the previous key-frame

	topology() : num_slots(0), m_epoch(0), m_sel_epoch(0) {}
	void clear() { *this = topology(); }
	size_t n_keyframes() const { return kf_adj_head.size(); }
	size_t n_edges() const { return edge_from.size(); }
	size_t n_observations() const { return obs_kf.size(); }
	static id32 narrow(uint64_t v) { if (v >= (uint64_t)NIL) throw std::out_of_range("srba: identifier does not fit the 32-bit index space of the graph layer"); return (id32)v; }

	id32 new_keyframe() {
		kf_adj_head.push_back(NIL); kf_adj_tail.push_back(NIL); kf_obs_head.push_back(NIL); kf_obs_tail.push_back(NIL); kf_degree.push_back(0);
		m_stamp.push_back(0); m_via_edge.push_back(NIL); m_via_prev.push_back(NIL); m_depth.push_back(0);
		return (id32)(kf_adj_head.size() - 1);
	}
	/** O(1) edge allocation + adjacency of both ends (reference impl/alloc_kf2kf_edge.h:17-61) */
	id32 new_edge(id32 from, id32 to) {
		if (from == to || from >= n_keyframes() || to >= n_keyframes()) throw std::invalid_argument("srba: kf2kf edge between a key-frame and itself, or to an unknown key-frame");
		const id32 e = (id32)edge_from.size();
		edge_from.push_back(from); edge_to.push_back(to); edge_next_at_from.push_back(NIL); edge_next_at_to.push_back(NIL);
		edge_jp_head.push_back(NIL); edge_jp_tail.push_back(NIL); edge_jp_count.push_back(0);
		attach(from, e); attach(to, e);
		return e;
	}
	id32 other_end(id32 e, id32 kf) const { return edge_from[e] == kf ? edge_to[e] : edge_from[e]; }
	id32 next_adjacent(id32 e, id32 kf) const { return edge_from[e] == kf ? edge_next_at_from[e] : edge_next_at_to[e]; }
	bool connected(id32 a, id32 b) const { for (id32 e = kf_adj_head[a]; e != NIL; e = next_adjacent(e, a)) if (other_end(e, a) == b) return true; return false; }

	// ------------------------------------------------------------------------------------------------ spanning trees
	uint32_t st_distance(id32 a, id32 b) const { const st_entry *e = st.find(a, b); return e ? e->dist : 0xffffffffu; }

	/** Update of the symbolic spanning trees after the edge (n -- other) has been added to the graph; n is the key-frame the caller
	 *  names as "new" (impl/spantree_update_symbolic.h:19-211). */
	void st_attach_edge(id32 n, id32 other, uint32_t max_depth) {
		if (max_depth < 1) throw std::invalid_argument("srba: max_tree_depth must be >= 1");
		if (max_depth > 65535) throw std::invalid_argument("srba: max_tree_depth must be <= 65535 (path lengths of the spanning-tree rows are 16-bit)");
		st.ensure_rows(n_keyframes());
		// the two node sets are fixed before anything is modified: `near` = nodes closer than max_depth to `other` (+ itself), `mine` = tree of n (+ itself)
		m_near.clear(); m_mine.clear();
		{ const st_entry *r = st.row(other); for (size_t i = 0, L = st.len(other); i < L; i++) if (r[i].dist < max_depth) m_near.push_back(std::make_pair(r[i].trg, r[i].dist)); }
		m_near.push_back(std::make_pair(other, 0u));
		{ const st_entry *r = st.row(n); for (size_t i = 0, L = st.len(n); i < L; i++) m_mine.push_back(std::make_pair(r[i].trg, r[i].dist)); }
		m_mine.push_back(std::make_pair(n, 0u));
		m_touched.clear();
		for (size_t a = 0; a < m_mine.size(); a++) {
			const id32 r = m_mine[a].first; const uint32_t d_r = m_mine[a].second;
			for (size_t b = 0; b < m_near.size(); b++) {
				const id32 s = m_near[b].first; if (s == r) continue;
				const uint32_t d_new = d_r + m_near[b].second + 1;
				st_entry *rs = st.find(r, s);
				if (rs ? !(d_new < rs->dist) : !(d_new <= max_depth)) continue; // only strictly shorter routes replace a known one
				// first hops: from r towards n (or straight over the new edge when r is n), from s towards `other` (or over the new edge)
				const id32 hop_r = (r == n) ? other : st.find(r, n)->next;
				if (!rs) rs = &st.insert(r, s);
				rs->dist = d_new; rs->next = hop_r;
				const id32 hop_s = (s == other) ? n : st.find(s, other)->next;
				st_entry *sr = st.find(s, r); if (!sr) sr = &st.insert(s, r);
				sr->dist = d_new; sr->next = hop_s;
				m_touched.push_back(r > s ? std::make_pair(r, s) : std::make_pair(s, r));
			}
		}
		// stored paths of every pair whose entry changed: one breadth-first search per distinct larger end point serves all its targets
		std::sort(m_touched.begin(), m_touched.end()); m_touched.erase(std::unique(m_touched.begin(), m_touched.end()), m_touched.end());
		for (size_t i = 0; i < m_touched.size();) {
			size_t j = i; uint32_t limit = 0;
			while (j < m_touched.size() && m_touched[j].first == m_touched[i].first) { limit = std::max(limit, st.find(m_touched[j].first, m_touched[j].second)->dist); j++; }
			const id32 root = m_touched[i].first;
			bfs_from(root, limit, &m_touched[i], j - i);
			for (size_t k = i; k < j; k++) store_path(root, m_touched[k].second);
			i = j;
		}
	}
	/** Edge ids of the first-found shortest chain a -> b over the whole graph (impl/spantree_update_symbolic.h:227-300). False if not connected. */
	bool shortest_path(id32 a, id32 b, std::vector<id32> *edges_out, std::vector<id32> *kfs_out) {
		if (edges_out) edges_out->clear(); if (kfs_out) kfs_out->clear();
		if (a == b) return true;
		std::pair<id32, id32> want(a, b);
		bfs_from(a, 0xffffffffu, &want, 1);
		if (m_stamp[b] != m_epoch) return false;
		const uint32_t L = m_depth[b];
		if (edges_out) edges_out->resize(L); if (kfs_out) kfs_out->resize(L);
		uint32_t p = L; for (id32 k = b; k != a; k = m_via_prev[k]) { --p; if (edges_out) (*edges_out)[p] = m_via_edge[k]; if (kfs_out) (*kfs_out)[p] = k; }
		return true;
	}
	/** Unbounded-width BFS tree from root down to max_depth (impl/spantree_create_complete.h:18-70): visit order, parent and connecting edge of every reached key-frame. */
	void bfs_tree(id32 root, size_t max_depth, std::vector<id32> &order, std::vector<id32> &parent_edge) {
		bfs_from(root, max_depth > 0xfffffffeu ? 0xffffffffu : (uint32_t)max_depth, 0, 0);
		order = m_queue; parent_edge.resize(order.size());
		for (size_t i = 0; i < order.size(); i++) parent_edge[i] = order[i] == root ? NIL : m_via_edge[order[i]];
	}
	id32 bfs_parent(id32 kf) const { return m_via_prev[kf]; }
	uint32_t bfs_depth(id32 kf) const { return m_depth[kf]; }

	/** slot of the numeric pose num[src][trg], created on demand (the reference creates the map node when it takes its address, add-observations.h:199-204) */
	int32_t ensure_num(id32 src, id32 trg) {
		if (src == trg) return -1;
		if (st_entry *e = st.find(src, trg)) { if (e->num < 0) e->num = num_slots++; return e->num; }
		for (size_t i = 0; i < loose.size(); i++) if (loose[i].src == src && loose[i].trg == trg) return loose[i].slot;
		loose_num l = {src, trg, num_slots++}; loose.push_back(l); return l.slot;
	}
	int32_t find_num(id32 src, id32 trg) const {
		if (const st_entry *e = st.find(src, trg)) return e->num;
		for (size_t i = 0; i < loose.size(); i++) if (loose[i].src == src && loose[i].trg == trg) return loose[i].slot;
		return -1;
	}

	// ------------------------------------------------------------------------------------------------ observations
	struct obs_result { id32 obs; bool first_seen, fixed, ignored_by_graph; };
	/** Registers one observation of landmark lm from key-frame kf and appends its symbolic Jacobian blocks: one dh_dAp block per edge of the
	 *  stored path observer -> base, walked from the observer (impl/add-observations.h:139-215), and the dh_df block of an unknown landmark (:220-257). */
	obs_result add_observation(id32 kf, id32 lm, bool fixed_position_given) {
		if (lm >= lm_base.size()) { lm_base.resize((size_t)lm + 1, NIL); lm_df_head.resize((size_t)lm + 1, NIL); lm_df_tail.resize((size_t)lm + 1, NIL); lm_known.resize((size_t)lm + 1, 1);
			lm_df_count.resize((size_t)lm + 1, 0); }
		obs_result R; R.first_seen = (lm_base[lm] == NIL); R.fixed = fixed_position_given || (!R.first_seen && lm_known[lm]); R.ignored_by_graph = false;
		if (R.first_seen) { lm_base[lm] = kf; lm_known[lm] = R.fixed ? 1 : 0; }
		const id32 o = R.obs = (id32)obs_kf.size(), base = lm_base[lm];
		obs_kf.push_back(kf); obs_lm.push_back(lm); obs_known.push_back(R.fixed ? 1 : 0); obs_first_of_unknown.push_back((R.first_seen && !R.fixed) ? 1 : 0);
		obs_next_in_kf.push_back(NIL); obs_next_in_lm.push_back(NIL);
		if (kf_obs_tail[kf] == NIL) kf_obs_head[kf] = o; else obs_next_in_kf[kf_obs_tail[kf]] = o;
		kf_obs_tail[kf] = o;
		if (base != kf) {
			const id32 hi = std::max(kf, base), lo = std::min(kf, base);
			const st_entry *e = st.find(hi, lo);
			if (e && e->path != NIL) {
				const uint32_t L = e->path_len, off = e->path; const bool from_far_end = (hi != kf);
				id32 cur = kf;
				for (uint32_t j = 0; j < L; j++) {
					const id32 ed = path_pool[off + (from_far_end ? L - 1 - j : j)];
					jp_block b; b.obs = o; b.kf_d = cur; b.next = NIL; b.normal_dir = (edge_to[ed] == cur) ? 1 : 0; b.has_A = (cur != kf) ? 1 : 0;
					const id32 bi = (id32)jp.size(); jp.push_back(b);
					if (edge_jp_tail[ed] == NIL) edge_jp_head[ed] = bi; else jp[edge_jp_tail[ed]].next = bi;
					edge_jp_tail[ed] = bi; edge_jp_count[ed]++;
					ensure_num(cur, base); if (b.has_A) ensure_num(kf, cur);
					cur = b.normal_dir ? edge_from[ed] : edge_to[ed];
				}
			} else R.ignored_by_graph = true;
		}
		if (!R.fixed && !R.ignored_by_graph) {
			if (lm_df_tail[lm] == NIL) lm_df_head[lm] = o; else obs_next_in_lm[lm_df_tail[lm]] = o;
			lm_df_tail[lm] = o; lm_df_count[lm]++;
			if (!obs_first_of_unknown[o]) ensure_num(kf, base);
		}
		return R;
	}

	// ------------------------------------------------------------------------------------------------ local area
	/** Unknowns of optimize_local_area(root, win): kf2kf edge ids and landmark ids in the order the reference's visitor collects them
	 *  (impl/bfs_visitor.h:105-176 with prebuilt trees / :35-104 without; RbaEngine.h:591-617): root first, then key-frames by ascending id. */
	void select_local_area(id32 root, uint32_t win, bool prebuilt_trees, const window_params &wp, std::vector<size_t> &edges_out, std::vector<size_t> &lms_out) {
		edges_out.clear(); lms_out.clear();
		if (root >= n_keyframes()) return;
		new_epoch(); // key-frame stamps (plain BFS only)
		m_edge_seen.resize(n_edges(), 0); m_lm_seen.resize(lm_base.size(), 0); m_lm_count.resize(lm_base.size(), 0);
		const uint32_t tag = ++m_sel_epoch;
		m_queue.clear();
		if (prebuilt_trees) {
			if (root >= st.rows()) return;
			m_queue.push_back(root);
			const st_entry *r = st.row(root); for (size_t i = 0, L = st.len(root); i < L; i++) if (r[i].dist <= win) m_queue.push_back(r[i].trg);
			for (size_t i = 0; i < m_queue.size(); i++) {
				const id32 k = m_queue[i];
				collect_landmarks(k, tag, wp, lms_out);
				for (id32 e = kf_adj_head[k]; e != NIL; e = next_adjacent(e, k)) // every edge touching a window key-frame, once, in creation order
					if (m_edge_seen[e] != tag) { m_edge_seen[e] = tag; if (wp.optimize_k2k_edges) edges_out.push_back(e); }
			}
		} else { // window wider than the prebuilt trees: plain breadth-first search, edges collected while expanding
			m_queue.push_back(root); m_stamp[root] = m_epoch; m_depth[root] = 0;
			for (size_t i = 0; i < m_queue.size(); i++) {
				const id32 k = m_queue[i];
				collect_landmarks(k, tag, wp, lms_out);
				if (m_depth[k] >= win) continue;
				for (id32 e = kf_adj_head[k]; e != NIL; e = next_adjacent(e, k)) {
					const id32 v = other_end(e, k);
					if (m_stamp[v] != m_epoch) { m_stamp[v] = m_epoch; m_depth[v] = m_depth[k] + 1; if ((uint64_t)v <= wp.max_visitable_kf) m_queue.push_back(v); }
					if (m_edge_seen[e] != tag) { m_edge_seen[e] = tag; if (wp.optimize_k2k_edges) edges_out.push_back(e); }
				}
			}
		}
	}

	// scratch shared with the capsule builder (epoch-stamped lookups)
	void new_epoch() { if (++m_epoch == 0) { std::fill(m_stamp.begin(), m_stamp.end(), 0u); m_epoch = 1; } }

private:
	void attach(id32 kf, id32 e) {
		if (kf_adj_tail[kf] == NIL) kf_adj_head[kf] = e;
		else { const id32 t = kf_adj_tail[kf]; if (edge_from[t] == kf) edge_next_at_from[t] = e; else edge_next_at_to[t] = e; }
		kf_adj_tail[kf] = e; kf_degree[kf]++;
	}
	/** Breadth-first search from root, neighbours in edge-creation order, first discovery fixes parent and edge. Stops expanding at depth `limit`
	 *  and returns early once every wanted target (second members of want[0..n_want)) has been discovered. */
	void bfs_from(id32 root, uint32_t limit, const std::pair<id32, id32> *want, size_t n_want) {
		new_epoch(); m_queue.clear();
		m_queue.push_back(root); m_stamp[root] = m_epoch; m_depth[root] = 0; m_via_prev[root] = NIL; m_via_edge[root] = NIL;
		size_t missing = n_want;
		for (size_t i = 0; i < n_want; i++) if (want[i].second == root) missing--;
		for (size_t h = 0; h < m_queue.size() && (!n_want || missing); h++) {
			const id32 u = m_queue[h]; if (m_depth[u] >= limit) continue;
			for (id32 e = kf_adj_head[u]; e != NIL; e = next_adjacent(e, u)) {
				const id32 v = other_end(e, u); if (m_stamp[v] == m_epoch) continue;
				m_stamp[v] = m_epoch; m_depth[v] = m_depth[u] + 1; m_via_prev[v] = u; m_via_edge[v] = e; m_queue.push_back(v);
				if (n_want) for (size_t i = 0; i < n_want; i++) if (want[i].second == v) { missing--; break; }
			}
		}
	}
	void store_path(id32 hi, id32 lo) {
		st_entry *e = st.find(hi, lo);
		if (m_stamp[lo] != m_epoch) throw std::logic_error("srba: spanning-tree pair without a connecting path");
		const uint32_t L = m_depth[lo];
		if (e->path == NIL || L > e->path_cap) { e->path = (id32)path_pool.size(); e->path_cap = (uint16_t)std::max<uint32_t>(L, e->dist); path_pool.resize(path_pool.size() + e->path_cap, NIL); }
		e->path_len = (uint16_t)L;
		uint32_t p = L; for (id32 k = lo; k != hi; k = m_via_prev[k]) path_pool[e->path + --p] = m_via_edge[k];
	}
	/** key-frame -> feature edges of k in observation order: a landmark becomes an unknown when its count inside the window reaches the threshold (RbaEngine.h:607-617) */
	void collect_landmarks(id32 k, uint32_t tag, const window_params &wp, std::vector<size_t> &lms_out) {
		if (!wp.optimize_landmarks) return;
		for (id32 o = kf_obs_head[k]; o != NIL; o = obs_next_in_kf[o]) {
			if (obs_known[o]) continue;
			const id32 lm = obs_lm[o];
			if (m_lm_seen[lm] != tag) { m_lm_seen[lm] = tag; m_lm_count[lm] = 0; }
			if (++m_lm_count[lm] == wp.min_times_seen) lms_out.push_back(lm);
		}
	}

	uint32_t m_epoch; std::vector<uint32_t> m_stamp, m_depth; std::vector<id32> m_via_edge, m_via_prev, m_queue;
	std::vector<std::pair<id32, uint32_t> > m_near, m_mine; std::vector<std::pair<id32, id32> > m_touched;
	std::vector<uint32_t> m_edge_seen, m_lm_seen, m_lm_count; uint32_t m_sel_epoch;
};

} // namespace graph
} // namespace srba
