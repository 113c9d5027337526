/*
 * srba_ctx.hpp -- host-side declarations shared by the translation units of libsrba_hip.so: the context behind `srba_hip_ctx *` (include/srba_hip.h), the launch-plan records,
 * the family dispatch (with_family) and the entry points of the multi-workgroup path (srba_big.hip) that srba_hip.hip calls. Not part of the C ABI.
 */
#pragma once
#include <atomic>
#include <string>
#include <type_traits>
#include <vector>
#include "srba_assemble.hpp"

extern thread_local std::string g_last_error; // (defined in srba_hip.hip: srba_hip_last_error(nullptr))

using srbadev::Batch; using srbadev::DevParams; using srbadev::ProbDesc;

namespace srbahost { // (a named namespace: BigLane appears in signatures that cross translation units)

struct FamDims { int P, L, O, PD; int PDX() const { return PD == 3 ? 5 : PD; } }; // PDX: device pose stride (SE2: [x y phi cos sin])
const FamDims kDims[SRBA_NUM_FAMILIES] = {{3, 3, 3, 3}, {3, 2, 2, 3}, {3, 2, 2, 3}, {6, 3, 4, 12}, {6, 3, 2, 12}, {6, 3, 3, 12}, {6, 3, 3, 12}, {6, 6, 6, 12}, {3, 3, 4, 3}};
// every model family the kernels are instantiated for
#ifdef SRBA_ONLY_RELPOSE2D /* experiment builds (tools/quick_build.sh): only the headline family is instantiated, the unit compiles in a fraction of the time; other families are rejected at run time */
#define SRBA_ALL_FAMILIES(X) X(SRBA_SE2_RELPOSE2D)
#elif defined(SRBA_ONLY_FAMILY) /* ... or any one family: -DSRBA_ONLY_FAMILY=SRBA_SE3_STEREO */
#define SRBA_ALL_FAMILIES(X) X(SRBA_ONLY_FAMILY)
#else
#define SRBA_ALL_FAMILIES(X) X(SRBA_SE2_RELPOSE2D) X(SRBA_SE2_RB2D) X(SRBA_SE2_CART2D) X(SRBA_SE3_STEREO) X(SRBA_SE3_MONO) X(SRBA_SE3_CART3D) X(SRBA_SE3_RB3D) X(SRBA_SE3_RELPOSE3D) X(SRBA_SE2_STEREO)
#endif

struct Arena { // layout builder: 256-byte aligned sub-allocations inside one buffer
	size_t size = 0;
	size_t add(size_t bytes) { const size_t off = (size + 255) & ~size_t(255); size = off + bytes; return off; }
};

#define HIPCHK(ctx, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { (ctx)->fail(std::string(#call) + ": " + hipGetErrorString(e_)); return -1; } } while (0)
#define LNCHK(lane, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { (lane)->error = std::string(#call) + ": " + hipGetErrorString(e_); return -1; } } while (0)
static const int kBigLanes = 16;
struct BigLane { int id = 0, slots = 1 /* windows its buffers have room for (srbadev::Gang) */; hipStream_t stream = nullptr; double *d_part = nullptr, *d_scal = nullptr;
	int *d_iscal = nullptr /* behind the scalars in the same allocation: one copy reads both back */; void *h_fetch = nullptr /* page-locked landing buffer of that copy */; hipEvent_t e0 = nullptr,
	e1 = nullptr; double chol_ms = 0, chol_flops = 0, t_flops = 0; long long chol_count = 0, chol_seqs = 0, t_seqs = 0 /* t_*: the sequences that were timed */, seq_no = 0; bool timed = false;
	int chol_nmax = 0; std::string error; };

} // namespace srbahost
using namespace srbahost;

#define SRBA_NLDS 19        /* LDS size classes (6 KB ... 152 KB per wavefront) of the one-wavefront kernels */
#define SRBA_CLS_WG128 19   /* landmark windows on a workgroup of two wavefronts (k_lm_wg<FAM, 128>), four workgroups per CU: at most 40 KB of LDS each */
#define SRBA_CLS_WG256 20   /* ... of four wavefronts, two per CU: at most 80 KB */
#define SRBA_CLS_WG512 21   /* ... of eight wavefronts, one per CU: the windows whose U_Ap blocks need up to 159 KB of LDS */
#define SRBA_NCLS 23        /* + the last class: systems factored by the multi-workgroup path (srba_big.hpp) */

struct LaunchJob { int queue, cls, first, count; double cost; int grid; int delay_us = 0; int lean = 0 /* k_lm_run_lean: three wavefronts per SIMD */,
	two = 0 /* k_lm_run2: two wavefronts per capsule */; };
using srbadev::kBigPart;
static const int kMaxJobs = 1024;

struct srba_hip_ctx {
	int device = 0; srba_hip_params params; DevParams dp; FamDims dm;
	hipStream_t stream = nullptr; hipEvent_t ev0 = nullptr, ev1 = nullptr;
	static constexpr int kRing = 64; hipEvent_t ring0[kRing] = {nullptr}, ring1[kRing] = {nullptr}; long long n_launches = 0; // event pairs of the last launches (srba_hip_kernel_ms_history)
	hipStream_t cls_stream[SRBA_NCLS] = {nullptr}; hipEvent_t ev_fork = nullptr, cls_done[SRBA_NCLS] = {nullptr}; // size classes run concurrently
	std::string error;
	// batch
	int n_prob = 0; std::vector<ProbDesc> desc; Batch B; srba_batch_stats stats;
	char *d_in = nullptr; size_t cap_in = 0; char *d_wk = nullptr; size_t cap_wk = 0; int *d_next = nullptr; Batch *d_batch = nullptr; bool batch_copied = false;
		// d_batch: device copy of B (the workgroup kernels read the batch's pointers from it)
	double *d_part = nullptr; double *d_scal = nullptr; int *d_iscal = nullptr; // big path: partial sums [3][kBigPart], scalars, int flags (ninv, not-pd)
	std::vector<int> big_ld; // per capsule: leading dimension of its dense system when it runs on the big path, else 0
	double big_t_flops = 0; long long big_t_seqs = 0; int big_time_every = 8; // timing events around every n-th factorisation sequence of a lane (big_timed_cholesky)
	double big_chol_ms = 0, big_chol_flops = 0; long long big_chol_count = 0, big_chol_seqs = 0 /* launch sequences: one factors all windows of a gang */; int big_chol_nmax = 0;
		// Cholesky time / flops of the big path since the last upload (sum over the lanes)
	BigLane lanes[kBigLanes]; int n_lanes_ready = 0; // lane 0 = the context stream and buffers; the others are created on first use
	bool big_gang = true, big_persistent = false, big_fused_step = true; int big_lanes_max = kBigLanes, big_gang_slots = srbadev::kGang,
		big_fresh = 1 /* SRBA_HIP_BIG_FRESH: see big_run_class */,
		big_gangs = 4 /* SRBA_HIP_BIG_GANGS: gangs side by side (big_run_class); measured 1: 8 100, 2: 9 000, 3: 9 100, 4: 9 350, 5: 9 340, 6: 5 200 LM iterations/s on cfg4 */,
		sch_xcd = 1 /* SRBA_HIP_SCHUR_XCD: a window's workgroups of kb_schur_reduce_wave on one XCD */, sch_sort = 1 /* SRBA_HIP_SCHUR_SORT: its blocks longest first (0: block order) */,
		sch_wave = 1 /* SRBA_HIP_SCHUR_WAVE: kb_schur_reduce_wave (a wavefront per U_Ap block) on the multi-workgroup class; 0: a workgroup per block */,
		gang_from_nb = 0 /* landmark windows with this many block rows
		or more take the gang instead of one wavefront (0: off) */; // big path: windows of a batch in lock-step on one stream (gang) or one host thread + stream per window
	int upload_threads = 1; bool dense_left = true; int hbm_from_kb = 48; bool dense_blocks_ok = true; // mid-size nearly-full systems use the dense block layout in LDS
	bool lin_terms = true, lm_terms = true; // term-parallel U_Ap accumulation in LDS: srba_hip_linearize / the fused LM kernel
	// fused normal equations of the relative-pose SE2 family (srba_assemble.hpp): capsules packed into bins (workgroups) by the LDS image they need, one launch; the rest take k_linearize
	bool od_on = true; // K4 over distinct observations (Worker::residuals_distinct)
	bool asm_on = true, asm_ready = false, asm_flags_set = false, jp_stale = false; size_t off_valid = 0, off_bp_ok = 0; long long n_valid_total = 0, n_bp_total = 0;
		int asm_max_kb = 160, asm_wpw = srbadev::ASM_MAX_WPW, asm_bin_bytes = srbadev::ASM_DEFAULT_BIN_KB * 1024; srbadev::AsmTables asm_tab = {nullptr, nullptr}; const int *asm_list = nullptr;
	int asm_bins = 0, asm_rest = 0; // bins of the fused launch; capsules left to the unfused kernel (asm_list holds their indices)
	srbadev::FlatMap flat; bool flat_ready = false, use_flat = true; // pair -> capsule map of the flat spanning-tree kernel (srba_flat.hpp), filled on first use after an upload
	int big_min_sys = 480;   // systems with more scalar unknowns than this skip the block-sparse symbolic analysis and go dense (big path)
	struct Staging { char *p = nullptr; bool pinned = false; char *get() const { return p; } void release() { if (p) { if (pinned) hipHostFree(p); else delete[] p; } p = nullptr; } } h_in;
		size_t h_in_cap = 0; size_t h_off_order = 0; // host staging of the input arena (kept: the launch order is read back from it); page-locked while it is small (the per-key-frame use: the copy
		// to the device then needs no wait)
	static constexpr size_t kPinnedMax = (size_t)8 << 20; hipEvent_t ev_h2d = nullptr; bool h2d_pending = false, defer_upload_sync = false; // optimize_capsule: the upload is not waited for;
		// the next writer of the staging buffer waits for this event
	char *h_out = nullptr; size_t h_out_cap = 0; // page-locked landing area of srba_hip_optimize_capsule (result record | unknowns .. spanning-tree poses)
	std::vector<int> delay_us; int class_prio = 0; // experiment knobs: per plan job delay before its launch (overrides the staggered start); stream priorities by class size
	int stagger_ns = 300, stagger_max_us = 5000; // staggered start of the class launches: see plan_launches
	// lambda-ladder speculation for a batch of ONE capsule (k_lm_spec): spec_w replicas of the work arena, spec_stride bytes apart; d_spec = flags | outcomes | increments (SpecCtl)
	long long spec_launches = 0; bool spec_on = true, spec_ready = false; int spec_w = 12; size_t spec_stride = 0; char *d_spec = nullptr; static constexpr int kSpecMaxW = 32, kSpecMaxN = 768;
		static constexpr size_t kSpecBackupOff = 256 + 8 * (2 * kSpecMaxW * 4) + 8 * (2 * (size_t)kSpecMaxW * kSpecMaxN), kSpecBytes = kSpecBackupOff + 8 * 5 * (size_t)kSpecMaxN;
		bool spec_suppress = false, spec_test_drop = false; long long spec_fallbacks = 0;
	bool wg_hs = true; /* SRBA_HIP_WG_HS=0: U_Ap blocks of the workgroup windows in memory (the first version of the path) instead of in LDS */
	bool wg_on = true; int wg_from_sys = 24, wg256_from_sys = 96; // SE3 landmark windows with a Schur-reduced system of at least wg_from_sys scalars run on a workgroup (k_lm_wg: 128 threads,
		// 256 from wg256_from_sys); SRBA_HIP_WG=0 / SRBA_HIP_WG_FROM / SRBA_HIP_WG256_FROM
	bool two_on = true; int two_from_kb = 20, two_min_count = 128; // k_lm_run2 (two wavefronts per capsule) for the relative-pose SE2 classes whose LDS image is at least this big
	bool lean_on = true; int lean_min_count = 512; // k_lm_run_lean for the size classes of which at least nine wavefronts fit the LDS of a CU (relative-pose SE2,
		// classes with at least this many capsules)
	int max_lds_kb = 1 << 20, min_chunk = 384, max_parts_per_queue = 2; int class_streams = 64 /* sched 3: the class launches are dealt round-robin to at most this many streams */, n_queues = 16,
		sched = 3, n_streams_used = 1, n_cu = 256, waves_per_cu = 8, lds_per_cu = 160 * 1024; std::vector<LaunchJob> plan; size_t lds_pad = 0; double last_ms = 0; int cls_first[SRBA_NCLS] = {0},
		cls_count[SRBA_NCLS] = {0}; size_t cls_lds[SRBA_NCLS] = {0};
	// offsets needed for downloads (bytes inside the wk arena)
	size_t off_edge = 0, off_ulm = 0, off_pose = 0, off_inf = 0, off_infv = 0, off_res = 0;
	size_t off_dbg[10] = {0}; int64_t len_dbg[10] = {0};
	size_t in_off_edge0 = 0, in_off_ulm0 = 0; long long tot_edge = 0, tot_ulm = 0;
	size_t off_phase = 0; bool phase_timing = false; long long n_pose_total = 0; std::vector<int> cls_of;
	void fail(const std::string &m) { error = m; g_last_error = m; }
};

// ---- launch helpers
// family id -> template argument: f(std::integral_constant<int, FAM>()) for the family of the context
template <class F> static bool with_family(int family, F &&f) {
	switch (family) {
#define X(FAM) case FAM: f(std::integral_constant<int, FAM>()); return true;
		SRBA_ALL_FAMILIES(X)
#undef X
	}
	return false;
}
// ---- what the translation units of the library must agree on: the records they share by header only (srba_hip_ctx, Batch, ProbDesc, DevParams) and the families they instantiate.
// Every unit defines its own signature from what IT sees; srba_hip_create compares them (a unit compiled with other SRBA_* macros would read the records at wrong offsets or find
// no kernel for a family, silently).
#define SRBA_FAM_BIT(FAM) | (1ull << (FAM))
static inline unsigned long long layout_signature() { return ((unsigned long long)sizeof(srba_hip_ctx) * 1000003ull + sizeof(srbadev::Batch) * 10007ull + sizeof(srbadev::ProbDesc) * 101ull + sizeof(
	srbadev::DevParams)) ^ ((0ull SRBA_ALL_FAMILIES(SRBA_FAM_BIT)) << 40); }
unsigned long long big_layout_signature(); // srba_big.hip
// ---- the multi-workgroup path for large capsules (srba_big.hip)
int big_prepare_lanes(srba_hip_ctx *c, int n);
void big_collect_lane_stats(srba_hip_ctx *c);
int big_solve(srba_hip_ctx *c, BigLane *ln, int p, double lambda, bool *pos_def); // the stepwise entry point (srba_hip_solve)
int big_run_class(srba_hip_ctx *c, const int32_t *caps, int count);               // all capsules of the big class of a batch: the LM loop driven from the host
