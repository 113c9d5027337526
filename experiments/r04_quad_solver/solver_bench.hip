// Micro-benchmark (round 4): the block-sparse LL^t + substitution of the fused LM kernel in isolation.
//   A: one system per wavefront (lane = 3*block + row, 21 groups) -- sp_factor_fsub_rows / sp_bsub_rows as shipped
//   B: four systems per wavefront, one per 16-lane row (5 groups of 3 lanes), lock-step
// Each repetition re-assembles the LDS image from a pristine copy in global memory (H + lambda I) and factors + substitutes.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 solver_bench.hip -o solver_bench
#include "../../srba_amd/csrc/srba_device.hpp"
#include <cstdio>
#include <vector>
#include <cstdlib>
#include <cmath>
using namespace srbadev;
extern __shared__ double lds[];

struct Sym { int nb, nnz, nitems; std::vector<int> col_off, row, item, rptr, rent; };
static Sym make_sym(int nb, int bw, int clique) {
	Sym s; s.nb = nb; s.col_off.push_back(0);
	std::vector<std::vector<int>> cols(nb);
	for (int k = 0; k < nb; k++) { int last = (k >= nb - clique) ? nb - 1 : std::min(nb - 1, k + bw); for (int r = k + 1; r <= last; r++) cols[k].push_back(r); s.col_off.push_back(s.col_off.back() + (int)cols[k].size()); for (int r : cols[k]) s.row.push_back(r); }
	s.nnz = (int)s.row.size();
	auto find = [&](int r, int c) { for (int i = s.col_off[c]; i < s.col_off[c + 1]; i++) if (s.row[i] == r) return i; return -1; };
	for (int k = 0; k < nb; k++) { int cn = (int)cols[k].size(); for (int a = 0; a < cn; a++) for (int b = 0; b <= a; b++) { int ra = cols[k][a], rb = cols[k][b]; int u = (ra == rb) ? ra : nb + find(ra, rb); if (ra != rb && find(ra, rb) < 0) { std::printf("fill!\n"); std::exit(1); } s.item.push_back((u << 18) | (a << 9) | b); } }
	s.nitems = (int)s.item.size();
	s.rptr.assign(nb + 1, 0); std::vector<std::vector<int>> rows(nb);
	for (int c = 0; c < nb; c++) for (int i = s.col_off[c]; i < s.col_off[c + 1]; i++) rows[s.row[i]].push_back((c << 14) | i);
	for (int a = 0; a < nb; a++) { s.rptr[a + 1] = s.rptr[a] + (int)rows[a].size(); for (int e : rows[a]) s.rent.push_back(e); }
	return s;
}

__device__ __forceinline__ long long now() { return __builtin_readcyclecounter(); }

// ---------------- A: as shipped
__global__ void __launch_bounds__(64) kA(int nb, int nnz, int nitems, const int *sym, const double *H, double *out, long long *cyc, int reps) {
	SparseSys S; S.nb = nb; S.nnzoff = nnz; S.dense = 0; S.diag = lds; S.off = lds + 9 * nb; S.rhs = S.off + 9 * nnz;
	int *ip = (int *)(S.rhs + 3 * nb); const int nint = 2 * (nb + 1) + 2 * nnz + nitems;
	for (int k = threadIdx.x; k < nint; k += 64) ip[k] = sym[k];
	S.col_off = ip; S.rptr = ip + nb + 1; S.row = ip + 2 * nb + 2; S.rent = S.row + nnz; S.item = S.rent + nnz; S.perm = nullptr;
	const int nd = 9 * nb + 9 * nnz + 3 * nb; const double *Hs = H + (size_t)blockIdx.x * nd;
	__syncthreads();
	long long t0 = now(), tf = 0; bool ok = true;
	for (int r = 0; r < reps; r++) {
		for (int k = threadIdx.x; k < nd; k += 64) lds[k] = Hs[k];
		__syncthreads();
		long long a = now();
		ok = sp_factor_fsub_rows(S) && ok;
		sp_bsub_rows(S);
		__syncthreads();
		tf += now() - a;
	}
	long long t1 = now();
	if (threadIdx.x < 3 * nb) out[(size_t)blockIdx.x * 3 * nb + threadIdx.x] = ok ? S.rhs[threadIdx.x] : NAN;
	if (3 * nb > 64 && threadIdx.x + 64 < 3 * nb) out[(size_t)blockIdx.x * 3 * nb + threadIdx.x + 64] = S.rhs[threadIdx.x + 64];
	if (threadIdx.x == 0) { cyc[2 * blockIdx.x] = t1 - t0; cyc[2 * blockIdx.x + 1] = tf; }
}

// ---------------- B: four systems per wavefront
struct QSys { int nb; double *diag, *off, *rhs; const int *col_off, *row, *item, *rptr, *rent; };
__device__ __forceinline__ int rows_max(int v) {
	const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16), c = __builtin_amdgcn_readlane(v, 32), e = __builtin_amdgcn_readlane(v, 48); /* (halves: lanes 16 / 48 repeat lanes 0 / 32) */
	return max(max(a, b), max(c, e));
}
template <int G> __device__ __forceinline__ bool q_factor(const QSys &S, bool live) {
	constexpr int NG = G / 3; const int rl = threadIdx.x & (G - 1); const int grp = (rl * 171) >> 9, sub = rl - 3 * grp; const bool worker = rl < 3 * NG;
	const int nbmax = rows_max(live ? S.nb : 0);
	bool ok = live;
	int cb = S.col_off[0], ce = S.col_off[1], ib = 0;
	for (int k = 0; k < nbmax; k++) {
		const bool on = ok && k < S.nb;
		const int kk = on ? k : 0;
		const int cn = on ? ce - cb : 0, nitems = cn * (cn + 1) / 2;
		double *D = S.diag + 9 * kk;
		const double a00 = D[0], a10 = D[3], a11 = D[4], a20 = D[6], a21 = D[7], a22 = D[8];
		const double b0 = S.rhs[3 * kk], b1 = S.rhs[3 * kk + 1], b2 = S.rhs[3 * kk + 2];
		const int ce_n = (on && k + 2 <= S.nb) ? S.col_off[k + 2] : ce;
		Chol3 c;
		const bool pd = chol3v(a00, a10, a11, a20, a21, a22, c);
		if (on && !pd) ok = false;
		const bool go = on && pd;
		const double y0 = b0 * c.r0, y1 = (b1 - c.l10 * y0) * c.r1, y2 = (b2 - c.l20 * y0 - c.l21 * y1) * c.r2;
		if (go && worker) for (int p = grp; p < cn; p += NG) {
			double *Ax = S.off + 9 * (cb + p) + 3 * sub; double *rx = S.rhs + 3 * S.row[cb + p] + sub;
			const double x0 = Ax[0] * c.r0, x1 = (Ax[1] - x0 * c.l10) * c.r1, x2 = (Ax[2] - x0 * c.l20 - x1 * c.l21) * c.r2;
			const double rv = *rx;
			Ax[0] = x0; Ax[1] = x1; Ax[2] = x2; *rx = rv - (x0 * y0 + x1 * y1 + x2 * y2);
		}
		if (go && rl == G - 1) {
			D[0] = c.l00; D[3] = c.l10; D[4] = c.l11; D[6] = c.l20; D[7] = c.l21; D[8] = c.l22; D[1] = c.r0; D[2] = c.r1; D[5] = c.r2;
			S.rhs[3 * kk] = y0; S.rhs[3 * kk + 1] = y1; S.rhs[3 * kk + 2] = y2;
		}
		solver_sync();
		if (go && worker) for (int t = grp; t < nitems; t += NG) {
			const unsigned w = (unsigned)S.item[ib + t];
			const double *La = S.off + 9 * (cb + ((w >> 9) & 511)) + 3 * sub, *Lb = S.off + 9 * (cb + (w & 511)); double *T = S.diag + 9 * (w >> 18) + 3 * sub;
			const double la0 = La[0], la1 = La[1], la2 = La[2];
			double lb[9];
#pragma unroll
			for (int q = 0; q < 9; q++) lb[q] = Lb[q];
			const double t0 = T[0], t1 = T[1], t2 = T[2];
			T[0] = t0 - (la0 * lb[0] + la1 * lb[1] + la2 * lb[2]);
			T[1] = t1 - (la0 * lb[3] + la1 * lb[4] + la2 * lb[5]);
			T[2] = t2 - (la0 * lb[6] + la1 * lb[7] + la2 * lb[8]);
		}
		solver_sync();
		if (on) { cb = ce; ce = ce_n; ib += nitems; }
	}
	return ok;
}
template <int G> __device__ __forceinline__ void q_bsub(const QSys &S, bool live) {
	constexpr int NG = G / 3; const int rl = threadIdx.x & (G - 1); const int grp = (rl * 171) >> 9, sub = rl - 3 * grp; const bool worker = rl < 3 * NG;
	const int nbmax = rows_max(live ? S.nb : 0);
	for (int a0 = nbmax - 1; a0 >= 0; a0--) {
		const bool on = live && a0 < S.nb; const int a = on ? a0 : 0;
		const int rb = S.rptr[a], re = on ? S.rptr[a + 1] : rb;
		const double *D = S.diag + 9 * a;
		const double r0 = S.rhs[3 * a], r1 = S.rhs[3 * a + 1], r2 = S.rhs[3 * a + 2], d5 = D[5], d7 = D[7], d2 = D[2], d3 = D[3], d6 = D[6], d1 = D[1];
		const double x2 = r2 * d5, x1 = (r1 - d7 * x2) * d2, x0 = (r0 - d3 * x1 - d6 * x2) * d1;
		if (worker) for (int j = rb + grp; j < re; j += NG) {
			const unsigned wx = (unsigned)S.rent[j]; const double *Lx = S.off + 9 * (wx & 0x3fff) + sub; double *yx = S.rhs + 3 * (wx >> 14) + sub;
			*yx -= Lx[0] * x0 + Lx[3] * x1 + Lx[6] * x2;
		}
		if (on && rl == G - 1) { S.rhs[3 * a] = x0; S.rhs[3 * a + 1] = x1; S.rhs[3 * a + 2] = x2; }
		solver_sync();
	}
}
template <int G, bool SHARED> __global__ void __launch_bounds__(64) kB(int nb, int nnz, int nitems, const int *sym, const double *H, double *out, long long *cyc, int reps, int rowdoubles) {
	constexpr int R = 64 / G; const int rw = threadIdx.x / G, rl = threadIdx.x & (G - 1);
	const int ndd = 9 * nb + 9 * nnz + 3 * nb;
	double *img = lds + rw * (SHARED ? ndd : rowdoubles);
	QSys S; S.nb = nb; S.diag = img; S.off = img + 9 * nb; S.rhs = S.off + 9 * nnz;
	int *ip = SHARED ? (int *)(lds + R * ndd) : (int *)(S.rhs + 3 * nb); const int nint = 2 * (nb + 1) + 2 * nnz + nitems;
	for (int k = rl; k < nint; k += G) ip[k] = sym[k];
	S.col_off = ip; S.rptr = ip + nb + 1; S.row = ip + 2 * nb + 2; S.rent = S.row + nnz; S.item = S.rent + nnz;
	const int nd = 9 * nb + 9 * nnz + 3 * nb; const double *Hs = H + ((size_t)blockIdx.x * R + rw) * nd;
	__syncthreads();
	long long t0 = now(), tf = 0; bool ok = true;
	for (int r = 0; r < reps; r++) {
		for (int k = rl; k < nd; k += G) img[k] = Hs[k];
		__syncthreads();
		long long a = now();
		ok = q_factor<G>(S, true) && ok;
		q_bsub<G>(S, true);
		__syncthreads();
		tf += now() - a;
	}
	long long t1 = now();
	for (int k = rl; k < 3 * nb; k += G) out[((size_t)blockIdx.x * R + rw) * 3 * nb + k] = ok ? S.rhs[k] : NAN;
	if (threadIdx.x == 0) { cyc[2 * blockIdx.x] = t1 - t0; cyc[2 * blockIdx.x + 1] = tf; }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); std::exit(1); } } while (0)
int main(int argc, char **argv) {
	const int nb = argc > 1 ? atoi(argv[1]) : 27, bw = argc > 2 ? atoi(argv[2]) : 3, clique = argc > 3 ? atoi(argv[3]) : 8, reps = 50;
	Sym s = make_sym(nb, bw, clique);
	std::printf("nb %d nnz %d items %d\n", s.nb, s.nnz, s.nitems);
	std::vector<int> sym; for (int v : s.col_off) sym.push_back(v); for (int v : s.rptr) sym.push_back(v); for (int v : s.row) sym.push_back(v); for (int v : s.rent) sym.push_back(v); for (int v : s.item) sym.push_back(v);
	const int nd = 9 * nb + 9 * s.nnz + 3 * nb; const int nsys_max = 256 * 16 * 4;
	std::vector<double> H((size_t)nsys_max * nd);
	srand(1);
	for (int q = 0; q < nsys_max; q++) { double *h = &H[(size_t)q * nd];
		for (int k = 0; k < nd; k++) h[k] = 0.2 * ((rand() % 2001) / 1000.0 - 1.0);
		for (int k = 0; k < nb; k++) { double *D = h + 9 * k; D[0] = 30 + D[0]; D[4] = 30 + D[4]; D[8] = 30 + D[8]; } }
	int *dsym; double *dH, *dout; long long *dcyc;
	CK(hipMalloc(&dsym, sym.size() * 4)); CK(hipMemcpy(dsym, sym.data(), sym.size() * 4, hipMemcpyHostToDevice));
	CK(hipMalloc(&dH, H.size() * 8)); CK(hipMemcpy(dH, H.data(), H.size() * 8, hipMemcpyHostToDevice));
	CK(hipMalloc(&dout, (size_t)nsys_max * 3 * nb * 8)); CK(hipMalloc(&dcyc, 2 * 8 * nsys_max));
	const int ldsA = nd * 8 + (int)sym.size() * 4 + 64, rowd = (ldsA + 7) / 8, ldsB = rowd * 8 * 4;
	CK(hipFuncSetAttribute((const void *)kA, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); CK(hipFuncSetAttribute((const void *)kB<16, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); CK(hipFuncSetAttribute((const void *)kB<16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); CK(hipFuncSetAttribute((const void *)kB<32, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); CK(hipFuncSetAttribute((const void *)kB<32, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
	std::printf("LDS per system %d B (A), per wave %d B (B)\n", ldsA, ldsB);
	std::vector<double> outA((size_t)nsys_max * 3 * nb), outB(outA.size());
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	const int ldsQS = (4 * nd * 8 + (int)sym.size() * 4 + 64), ldsH = rowd * 8 * 2, ldsHS = (2 * nd * 8 + (int)sym.size() * 4 + 64);
	const char *vname[] = {"A  1/wave      ", "B  4/wave      ", "BS 4/wave sh.sym", "C  2/wave      ", "CS 2/wave sh.sym"}; const int vper[] = {1, 4, 4, 2, 2}; const int vlds[] = {ldsA, ldsB, ldsQS, ldsH, ldsHS};
	for (int wpc : {0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14}) {
		for (int var = 0; var < 5; var++) {
			if (wpc > 0 && (size_t)vlds[var] * wpc > 160 * 1024) continue;
			const int waves = wpc == 0 ? 1 : 256 * wpc; const int nsys = waves * vper[var];
			if (nsys > nsys_max) continue;
			std::vector<long long> cyc(2 * waves);
			for (int it = 0; it < 2; it++) {
				CK(hipEventRecord(e0));
				if (var == 0) hipLaunchKernelGGL(kA, dim3(waves), dim3(64), ldsA, 0, nb, s.nnz, s.nitems, dsym, dH, dout, dcyc, reps);
				else if (var == 1) hipLaunchKernelGGL((kB<16, false>), dim3(waves), dim3(64), ldsB, 0, nb, s.nnz, s.nitems, dsym, dH, dout, dcyc, reps, rowd);
				else if (var == 2) hipLaunchKernelGGL((kB<16, true>), dim3(waves), dim3(64), ldsQS, 0, nb, s.nnz, s.nitems, dsym, dH, dout, dcyc, reps, rowd);
				else if (var == 3) hipLaunchKernelGGL((kB<32, false>), dim3(waves), dim3(64), ldsH, 0, nb, s.nnz, s.nitems, dsym, dH, dout, dcyc, reps, rowd);
				else hipLaunchKernelGGL((kB<32, true>), dim3(waves), dim3(64), ldsHS, 0, nb, s.nnz, s.nitems, dsym, dH, dout, dcyc, reps, rowd);
				CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
			}
			float ms; CK(hipEventElapsedTime(&ms, e0, e1));
			CK(hipMemcpy(cyc.data(), dcyc, 16 * waves, hipMemcpyDeviceToHost));
			if (wpc == 1) CK(hipMemcpy(var ? outB.data() : outA.data(), dout, (size_t)256 * 3 * nb * 8, hipMemcpyDeviceToHost));
			if (wpc == 1 && var) { size_t bad = 0; for (size_t i = 0; i < (size_t)256 * 3 * nb; i++) if (outA[i] != outB[i]) bad++; std::printf("   variant %d vs A: %zu scalars differ\n", var, bad); }
			double c0 = 0, c1 = 0; for (int i = 0; i < waves; i++) { c0 += cyc[2 * i]; c1 += cyc[2 * i + 1]; }
			std::printf("%s waves/CU %2d: %6d waves %6d systems  kernel %.3f ms  -> %.2f us per solve-rep per wave, factor+bsub cycles/rep %.0f (all %.0f); systems/us %.2f\n", vname[var], wpc, waves, nsys, ms,
			            1e3 * ms / reps, c1 / waves / reps, c0 / waves / reps, nsys * (double)reps / (1e3 * ms));
		}
	}
	// equality of the solutions (same systems 0..N)
	return 0;
}
