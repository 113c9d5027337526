// latency probes (one wavefront): dependent FP64 chains, rsqrt, LDS round trips
#include <hip/hip_runtime.h>
#include <cstdio>
extern __shared__ double lds[];
__device__ __forceinline__ long long now() { return __builtin_readcyclecounter(); }
__global__ void __launch_bounds__(64) k(double *out, long long *cyc, int n, double seed, int lanes) {
	if ((int)threadIdx.x >= lanes) return;
	double x = seed + threadIdx.x * 1e-9, y = 1.0000001;
	long long t0, t1; int s = 0;
	// 1: dependent fma chain
	t0 = now();
	for (int i = 0; i < n; i++) { x = __builtin_fma(x, y, 1e-9); x = __builtin_fma(x, y, 1e-9); x = __builtin_fma(x, y, 1e-9); x = __builtin_fma(x, y, 1e-9); }
	t1 = now(); if (threadIdx.x == 0) cyc[s] = t1 - t0; s++;
	// 2: dependent mul chain
	t0 = now();
	for (int i = 0; i < n; i++) { x = x * y; x = x * y; x = x * y; x = x * y; }
	t1 = now(); if (threadIdx.x == 0) cyc[s] = t1 - t0; s++;
	// 3: 4 independent fma chains
	{ double a = x, b = x + 1, c = x + 2, d = x + 3; t0 = now();
	for (int i = 0; i < n; i++) { a = __builtin_fma(a, y, 1e-9); b = __builtin_fma(b, y, 1e-9); c = __builtin_fma(c, y, 1e-9); d = __builtin_fma(d, y, 1e-9); }
	t1 = now(); x = a + b + c + d; if (threadIdx.x == 0) cyc[s] = t1 - t0; s++; }
	// 4: dependent rsqrt (library: v_rsq_f64 + refinement)
	x = fabs(x) + 2.0; t0 = now();
	for (int i = 0; i < n; i++) { x = rsqrt(x) + 1.5; x = rsqrt(x) + 1.5; x = rsqrt(x) + 1.5; x = rsqrt(x) + 1.5; }
	t1 = now(); if (threadIdx.x == 0) cyc[s] = t1 - t0; s++;
	// 5: dependent raw v_rsq_f64
	t0 = now();
	for (int i = 0; i < n; i++) { x = __builtin_amdgcn_rsq(x) + 1.5; x = __builtin_amdgcn_rsq(x) + 1.5; x = __builtin_amdgcn_rsq(x) + 1.5; x = __builtin_amdgcn_rsq(x) + 1.5; }
	t1 = now(); if (threadIdx.x == 0) cyc[s] = t1 - t0; s++;
	// 6: LDS pointer chase (ds_read_b32 dependent)
	int *il = (int *)lds; for (int i = threadIdx.x; i < 1024; i += lanes) il[i] = (i * 37 + 11) & 1023; __syncthreads();
	int p = threadIdx.x; t0 = now();
	for (int i = 0; i < n; i++) { p = il[p]; p = il[p]; p = il[p]; p = il[p]; }
	t1 = now(); if (threadIdx.x == 0) cyc[s] = t1 - t0; s++;
	// 7: LDS ds_read_b64 -> fma -> ds_write_b64 -> (next read of another lane's value)
	lds[512 + threadIdx.x] = x; __syncthreads(); t0 = now();
	for (int i = 0; i < n; i++) {
#pragma unroll
		for (int u = 0; u < 4; u++) { double v = lds[512 + ((threadIdx.x + 1) & (lanes - 1))]; v = __builtin_fma(v, y, 1e-9); __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); lds[512 + threadIdx.x] = v; __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
	}
	t1 = now(); if (threadIdx.x == 0) cyc[s] = t1 - t0; s++;
	// 8: global pointer chase through out[] (L2 hits after first pass)
	out[threadIdx.x] = x + p;
}
int main() {
	double *o; long long *c; hipMalloc(&o, 4096); hipMalloc(&c, 256);
	const char *names[] = {"dep fma_f64", "dep mul_f64", "4 indep fma_f64 (per op)", "dep rsqrt() lib", "dep v_rsq_f64+add", "dep ds_read_b32", "ds_read_b64->fma->ds_write_b64"};
	for (int lanes : {64, 16}) {
		const int n = 256; long long h[8];
		for (int it = 0; it < 2; it++) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 16384, 0, o, c, n, 1.0, lanes); hipDeviceSynchronize(); }
		hipMemcpy(h, c, 64, hipMemcpyDeviceToHost);
		for (int i = 0; i < 7; i++) std::printf("lanes %2d  %-34s %.1f cycles per step\n", lanes, names[i], (double)h[i] / (4.0 * n) / (i == 2 ? 1 : 1));
	}
	return 0;
}
