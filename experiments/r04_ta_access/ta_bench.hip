// What does the vector-memory front end (TA / L1) charge for the access shapes of the LM kernel when the data is cache resident? Every workgroup (one wavefront) works on its OWN
// 64 KB region again and again (L2 / L1 resident: DRAM is not in the picture), 8 wavefronts per CU. Reported: bytes per cycle per CU and cycles per wave-instruction.
//   own72    : lane i reads record i of 72 bytes (4 x 16 + 8 B) -- "a lane owns a record", records consecutive (the K6 / assemble / K2 shape)
//   own40    : the same with 40-byte records (poses)
//   own24    : 24-byte records (residual rows, observations)
//   flat16   : the same spans read flat: lane i reads the 16-byte piece i, i + 64, ... (fully coalesced)
//   gather40 : lane i reads the 40-byte record perm[i] (a scattered pose gather inside the 64 KB region)
//   flat8 / flat4 : coalesced 8- and 4-byte loads
// and the store forms st_own72 / st_own40 / st_own24 / st_flat16.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double f64x2u __attribute__((ext_vector_type(2), aligned(8)));
template <int N> __device__ __forceinline__ void ldn(double *dst, const double *src) {
#pragma unroll
	for (int k = 0; k + 1 < N; k += 2) { const f64x2u v = *(const f64x2u *)(src + k); dst[k] = v.x; dst[k + 1] = v.y; }
	if (N & 1) dst[N - 1] = src[N - 1];
}
template <int N> __device__ __forceinline__ void stn(double *dst, const double *src) {
#pragma unroll
	for (int k = 0; k + 1 < N; k += 2) { f64x2u v; v.x = src[k]; v.y = src[k + 1]; *(f64x2u *)(dst + k) = v; }
	if (N & 1) dst[N - 1] = src[N - 1];
}
constexpr int REGION = 8192; // doubles = 64 KB per wavefront
template <int N, bool GATHER> __global__ void __launch_bounds__(64) k_own(const double *buf, const int *perm, double *sink, int reps, long long *cyc) {
	const double *p = buf + (size_t)blockIdx.x * REGION; const int nrec = REGION / N; double a = 0; const long long t0 = __builtin_readcyclecounter();
	for (int r = 0; r < reps; r++) for (int i = threadIdx.x; i < nrec; i += 64) { double v[N]; ldn<N>(v, p + (size_t)(GATHER ? perm[i] % nrec : i) * N);
#pragma unroll
		for (int k = 0; k < N; k++) a += v[k]; }
	if (a == 1.2345) *sink = a; if (threadIdx.x == 0) cyc[blockIdx.x] = __builtin_readcyclecounter() - t0;
}
template <int W> __global__ void __launch_bounds__(64) k_flat(const double *buf, double *sink, int reps, long long *cyc) { // W bytes per lane and load
	const char *p = (const char *)(buf + (size_t)blockIdx.x * REGION); double a = 0; const long long t0 = __builtin_readcyclecounter();
	for (int r = 0; r < reps; r++) for (int i = threadIdx.x * W; i < REGION * 8; i += 64 * W) {
		if constexpr (W == 16) { const f64x2u v = *(const f64x2u *)(p + i); a += v.x + v.y; } else if constexpr (W == 8) a += *(const double *)(p + i); else a += (double)*(const int *)(p + i); }
	if (a == 1.2345) *sink = a; if (threadIdx.x == 0) cyc[blockIdx.x] = __builtin_readcyclecounter() - t0;
}
template <int N> __global__ void __launch_bounds__(64) k_st_own(double *buf, int reps, long long *cyc) {
	double *p = buf + (size_t)blockIdx.x * REGION; const int nrec = REGION / N; const long long t0 = __builtin_readcyclecounter();
	for (int r = 0; r < reps; r++) for (int i = threadIdx.x; i < nrec; i += 64) { double v[N];
#pragma unroll
		for (int k = 0; k < N; k++) v[k] = (double)(i + k + r); stn<N>(p + (size_t)i * N, v); }
	if (threadIdx.x == 0) cyc[blockIdx.x] = __builtin_readcyclecounter() - t0;
}
__global__ void __launch_bounds__(64) k_st_flat(double *buf, int reps, long long *cyc) {
	double *p = buf + (size_t)blockIdx.x * REGION; const long long t0 = __builtin_readcyclecounter();
	for (int r = 0; r < reps; r++) for (int i = threadIdx.x * 2; i < REGION; i += 128) { f64x2u v; v.x = (double)(i + r); v.y = 1.0; *(f64x2u *)(p + i) = v; }
	if (threadIdx.x == 0) cyc[blockIdx.x] = __builtin_readcyclecounter() - t0;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); std::exit(1); } } while (0)
int main() {
	const int waves = 256 * 8, reps = 64; double *buf, *sink; int *perm; long long *cyc;
	CK(hipMalloc(&buf, (size_t)waves * REGION * 8)); CK(hipMemset(buf, 0, (size_t)waves * REGION * 8)); CK(hipMalloc(&sink, 8)); CK(hipMalloc(&cyc, 8 * waves));
	std::vector<int> hp(REGION); for (int i = 0; i < REGION; i++) hp[i] = (int)(((unsigned)i * 2654435761u) >> 7) & 0xffff; CK(hipMalloc(&perm, 4 * REGION)); CK(hipMemcpy(perm, hp.data(), 4 * REGION, hipMemcpyHostToDevice));
	std::vector<long long> h(waves);
	auto report = [&](const char *name, double bytes_per_wave_rep, double instr_per_wave_rep) { CK(hipDeviceSynchronize()); CK(hipMemcpy(h.data(), cyc, 8 * waves, hipMemcpyDeviceToHost)); double c = 0; for (auto x : h) c += (double)x; c /= waves;
		std::printf("%-10s %8.0f cycles per pass of 64 KB-ish per wave (8 waves per CU): %6.1f B/cycle/CU, %6.1f cycles per wave-instruction per CU-share (x8 waves)\n", name, c / reps, 8.0 * bytes_per_wave_rep * reps / c, c / reps / instr_per_wave_rep / 8.0); };
	for (int it = 0; it < 2; it++) {
		const bool p = it == 1;
		hipLaunchKernelGGL((k_own<9, false>), dim3(waves), dim3(64), 0, 0, buf, perm, sink, reps, cyc); if (p) report("own72", (REGION / 9) * 72.0, (REGION / 9) / 64.0 * 5); else CK(hipDeviceSynchronize());
		hipLaunchKernelGGL((k_own<5, false>), dim3(waves), dim3(64), 0, 0, buf, perm, sink, reps, cyc); if (p) report("own40", (REGION / 5) * 40.0, (REGION / 5) / 64.0 * 3); else CK(hipDeviceSynchronize());
		hipLaunchKernelGGL((k_own<3, false>), dim3(waves), dim3(64), 0, 0, buf, perm, sink, reps, cyc); if (p) report("own24", (REGION / 3) * 24.0, (REGION / 3) / 64.0 * 2); else CK(hipDeviceSynchronize());
		hipLaunchKernelGGL((k_own<5, true>), dim3(waves), dim3(64), 0, 0, buf, perm, sink, reps, cyc); if (p) report("gather40", (REGION / 5) * 40.0, (REGION / 5) / 64.0 * 4); else CK(hipDeviceSynchronize());
		hipLaunchKernelGGL((k_flat<16>), dim3(waves), dim3(64), 0, 0, buf, sink, reps, cyc); if (p) report("flat16", REGION * 8.0, REGION * 8.0 / 1024); else CK(hipDeviceSynchronize());
		hipLaunchKernelGGL((k_flat<8>), dim3(waves), dim3(64), 0, 0, buf, sink, reps, cyc); if (p) report("flat8", REGION * 8.0, REGION * 8.0 / 512); else CK(hipDeviceSynchronize());
		hipLaunchKernelGGL((k_flat<4>), dim3(waves), dim3(64), 0, 0, buf, sink, reps, cyc); if (p) report("flat4", REGION * 8.0, REGION * 8.0 / 256); else CK(hipDeviceSynchronize());
		hipLaunchKernelGGL((k_st_own<9>), dim3(waves), dim3(64), 0, 0, buf, reps, cyc); if (p) report("st_own72", (REGION / 9) * 72.0, (REGION / 9) / 64.0 * 5); else CK(hipDeviceSynchronize());
		hipLaunchKernelGGL((k_st_own<5>), dim3(waves), dim3(64), 0, 0, buf, reps, cyc); if (p) report("st_own40", (REGION / 5) * 40.0, (REGION / 5) / 64.0 * 3); else CK(hipDeviceSynchronize());
		hipLaunchKernelGGL((k_st_own<3>), dim3(waves), dim3(64), 0, 0, buf, reps, cyc); if (p) report("st_own24", (REGION / 3) * 24.0, (REGION / 3) / 64.0 * 2); else CK(hipDeviceSynchronize());
		hipLaunchKernelGGL(k_st_flat, dim3(waves), dim3(64), 0, 0, buf, reps, cyc); if (p) report("st_flat16", REGION * 8.0, REGION * 8.0 / 1024); else CK(hipDeviceSynchronize());
	}
	return 0;
}
