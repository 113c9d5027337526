// What does a ds_add_f64 (no return) wavefront instruction cost on gfx950, by address pattern? The question behind the observation-major normal-equations kernel
// (srba_assemble.hpp, round 6): a lane per observation adds its 3 x 3 products into the Hessian image of its capsule with LDS atomics -- ~160 wavefront instructions
// per capsule, lanes of one instruction landing on 10..30 distinct blocks.
// Patterns: every lane its own double (stride 1), stride 9 (a block per lane), G lanes per address (same-address conflicts) for G = 2..64, and the stride-9 pattern
// with G lanes per block. Also: is the sum reproducible run to run (one wavefront per image)?
// build: hipcc --offload-arch=gfx950 -O3 lds_atomic_rate.hip -o lds_atomic_rate ; run: ./lds_atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
extern __shared__ double lds[];
// every wavefront owns `img` doubles of LDS; lane l adds to img[(l / G) * stride + (k % 9)] for k = 0 .. n-1
template <int WAVES> __global__ void __launch_bounds__(64 * WAVES) k_add(int n, int G, int stride, int img, long long *ticks, double *out) {
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6; double *my = lds + w * img;
	for (int k = lane; k < img; k += 64) my[k] = 0; __syncthreads();
	double *p = my + (lane / G) * stride; const double v = 1.0 + 1e-9 * lane;
	const long long t0 = wall_clock64();
	for (int k = 0; k < n; k += 9) {
#pragma unroll
		for (int e = 0; e < 9; e++) __hip_atomic_fetch_add(p + e, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
	}
	asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
	const long long t1 = wall_clock64();
	__syncthreads();
	if (lane == 0) ticks[blockIdx.x * WAVES + w] = t1 - t0;
	if (blockIdx.x == 0 && w == 0) for (int k = lane; k < img; k += 64) out[k] = my[k];
}
template <int WAVES> static void run(const char *what, int G, int stride) {
	const int n = 9 * 200, img = 64 * 9 + 16, grid = 256; long long *ticks; double *out; (void)hipMalloc(&ticks, 8 * grid * WAVES); (void)hipMalloc(&out, 8 * img);
	double h0[64 * 9 + 16], h1[64 * 9 + 16]; long long ht[256 * 4 * 16];
	for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k_add<WAVES>, dim3(grid), dim3(64 * WAVES), (size_t)8 * img * WAVES, 0, n, G, stride, img, ticks, out); (void)hipDeviceSynchronize();
		(void)hipMemcpy(rep ? h1 : h0, out, 8 * img, hipMemcpyDeviceToHost); }
	(void)hipMemcpy(ht, ticks, 8 * grid * WAVES, hipMemcpyDeviceToHost); double mean = 0; for (int i = 0; i < grid * WAVES; i++) mean += (double)ht[i]; mean /= grid * WAVES;
	// wall_clock64 ticks at 100 MHz: 10 ns each
	printf("%-44s %2d wavefronts/workgroup: %7.1f ns per wavefront instruction (as seen by one wavefront); reproducible %s (err %s)\n", what, WAVES, 10.0 * mean / n,
		std::memcmp(h0, h1, sizeof(h0)) == 0 ? "yes" : "NO", hipGetErrorString(hipGetLastError()));
	(void)hipFree(ticks); (void)hipFree(out);
}
int main() {
	run<1>("stride 1, a double per lane", 1, 1); run<4>("stride 1, a double per lane", 1, 1); run<8>("stride 1, a double per lane", 1, 1);
	run<1>("stride 9, a block per lane", 1, 9); run<4>("stride 9, a block per lane", 1, 9); run<8>("stride 9, a block per lane", 1, 9);
	for (int G : {2, 4, 8, 16, 64}) { char s[64]; std::snprintf(s, sizeof(s), "stride 9, %d lanes per block", G); run<1>(s, G, 9); run<8>(s, G, 9); }
	return 0;
}
