/*
 * srba_flat.hpp -- K1 of the batch-wide ("stepwise") API as a FLAT grid: one thread per spanning-tree pair of the whole batch, whatever capsule it belongs to.
 *
 * The one-wavefront-per-capsule form (k_spantree in srba_hip.hip) keeps ~12 of 64 lanes live (path lengths differ, capsules have a ragged number of pairs) and
 * every lane stores its two poses as its own partial lines: 143 M write requests to L2 for 1.8 GB of poses, 89 % of the wave cycles waiting
 * (profiles/r02_sq_summary.json). Here 64 consecutive pairs of the concatenated pose array go to one wavefront (all lanes live), the capsule of a pair comes from a
 * per-pair map filled on the device once per upload, consecutive workgroups stay on one XCD, and a wavefront writes its 64 pose pairs through LDS as one contiguous
 * span of full lines: 1.18 ms -> 0.60 ms on the benchmark batch (28 % -> 55 % of the HBM peak by algorithmic bytes).
 * The same treatment of the linearisation phases was measured and NOT kept: flat Jacobian / Hessian / gradient launches took 1.56 ms and a
 * one-workgroup-per-capsule form with the Jacobians staged in LDS 2.13 ms against 1.34 ms for the one-wavefront-per-capsule kernel.
 */
#pragma once

namespace srbadev {

struct FlatMap { int *pair; long long n_pair; };
#ifndef SRBA_FLAT_DECLS_ONLY /* (srba_big.hip needs the record only: it is a member of the context) */

// Workgroups are dealt round-robin to the 8 XCDs (each with its own L2): consecutive LOGICAL workgroups -- which share a capsule's poses, Jacobians and term lists --
// are mapped onto workgroup ids of the same XCD, so that a capsule's data is fetched into one L2 instead of up to eight.
__device__ __forceinline__ long long flat_block() {
	const unsigned n = gridDim.x, b = blockIdx.x, per = n >> 3;
	if (b >= (per << 3)) return b;               // the ragged tail keeps its place
	return (long long)(b & 7) * per + (b >> 3);
}
#define FLAT_GID() (flat_block() * blockDim.x + threadIdx.x)
#define FLAT_STRIDE() ((long long)gridDim.x * blockDim.x)
// A wavefront that produced one fixed-size record per lane for 64 CONSECUTIVE items writes them through LDS: the records of consecutive items are contiguous in
// memory, so the wavefront stores the whole span 16 bytes per lane and instruction (full 64-byte lines) instead of one partial line per lane and instruction.
template <int N> __device__ __forceinline__ void wave_store_records(double *lds /* 64 * N doubles of this wavefront */, double *dst_first /* record of lane 0 */, const double (&rec)[N], bool live,
	int n_live) {
	const int lane = threadIdx.x & 63;
	if (live) {
#pragma unroll
		for (int k = 0; k < N; k++) lds[lane * N + k] = rec[k];
	}
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
	const int total = n_live * N; // doubles to write (the live lanes are the first n_live)
	for (int k = 2 * lane; k < total; k += 128) {
		if (k + 1 < total) { f64x2u v; v.x = lds[k]; v.y = lds[k + 1]; *(f64x2u *)(dst_first + k) = v; } else dst_first[k] = lds[k];
	}
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// one workgroup per capsule: pair -> capsule map
__global__ void __launch_bounds__(256) kf_fill_maps(const Batch B, const FlatMap M) {
	const int p = blockIdx.x; const ProbDesc &d = B.desc[p];
	for (int i = threadIdx.x; i < d.n_pairs; i += 256) M.pair[d.o_pair + i] = p;
}

// K1: pose of the base key-frame seen from the observer (and its inverse) for every pair of the batch
template <int FAM> __global__ void __launch_bounds__(256) kf_spantree(const Batch B, const DevParams prm, const FlatMap M, int only_needed) {
	typedef Worker<FAM> W; typedef typename W::PO PO; typedef typename W::pose_t pose_t; constexpr int PD = W::PD;
	__shared__ double stage[4][64 * 2 * PD];
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	for (long long g0 = FLAT_GID() - lane; g0 < M.n_pair; g0 += FLAT_STRIDE()) { // g0: first pair of this wavefront's span
		const long long g = g0 + lane; const bool live = g < M.n_pair;
		pose_t acc = PO::ident();
		if (live && !(only_needed && !B.pair_needed[g])) {
			const ProbDesc &d = B.desc[M.pair[g]]; const int pr = (int)(g - d.o_pair);
			const int kb = B.pair_path_off[d.o_ppoff + pr], ke = B.pair_path_off[d.o_ppoff + pr + 1];
			for (int k = kb; k < ke; k++) {
				const int pe = B.path_edge[d.o_path + k]; const pose_t ed = PO::ld(B.edge + (d.o_edge + (pe >> 1)) * PD);
				acc = (pe & 1) ? comp(acc, inv(ed)) : comp(acc, ed);
			}
		}
		if (only_needed) { // sparse refresh: only the pairs in use are rewritten
			if (live && B.pair_needed[g]) { PO::st(B.pose + g * 2 * PD, acc); PO::st(B.pose + (g * 2 + 1) * PD, inv(acc)); }
		} else {
			double rec[2 * PD]; PO::st(rec, acc); PO::st(rec + PD, inv(acc));
			const long long left = M.n_pair - g0; wave_store_records<2 * PD>(stage[w], B.pose + g0 * 2 * PD, rec, live, (int)(left < 64 ? left : 64));
		}
	}
}
#endif // SRBA_FLAT_DECLS_ONLY
} // namespace srbadev
