// Does a wave64 FP64 instruction cost less when only some of its 16-lane quarters have live lanes? The block-sparse LDL^t of the headline kernel issues ~80 FP64 instructions per column
// with 12 of 64 lanes live on average (profiles/r06_sq_summary.json) and is bound by VALU issue on SIMDs that hold two or three wavefronts: if the hardware skipped dead quarters,
// packing the live lanes into the low quarter would pay. A chain-free stream of v_fma_f64 (8 independent accumulators) under EXEC = the low N lanes.
// build: hipcc --offload-arch=gfx950 -O3 valu_mask_rate.hip -o valu_mask_rate ; run: ./valu_mask_rate
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(64) k_fma(int live, int iters, long long *ticks, double *sink) {
	const int lane = threadIdx.x; double a[8]; for (int k = 0; k < 8; k++) a[k] = 1.0 + 1e-3 * (lane + k); const double x = 1.0000001, y = 1e-9;
	long long t0 = 0, t1 = 0;
	if (lane < live) {
		t0 = wall_clock64();
		for (int i = 0; i < iters; i++) {
#pragma unroll
			for (int r = 0; r < 8; r++) {
#pragma unroll
				for (int k = 0; k < 8; k++) a[k] = fma(a[k], x, y);
			}
		}
		t1 = wall_clock64();
	}
	double s = 0; for (int k = 0; k < 8; k++) s += a[k];
	if (lane == 0) { ticks[blockIdx.x] = t1 - t0; sink[blockIdx.x] = s; }
}
int main() {
	long long *ticks; double *sink; const int grid = 256 * 8; (void)hipMalloc(&ticks, 8 * grid); (void)hipMalloc(&sink, 8 * grid); long long h[256 * 8];
	for (int live : {64, 48, 32, 16, 8, 1}) {
		const int iters = 2000; for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k_fma, dim3(grid), dim3(64), 0, 0, live, iters, ticks, sink); (void)hipDeviceSynchronize(); }
		(void)hipMemcpy(h, ticks, 8 * grid, hipMemcpyDeviceToHost); double m = 0; for (int i = 0; i < grid; i++) m += (double)h[i]; m /= grid;
		// 100 MHz ticks: 10 ns each; 64 instructions per iteration; eight wavefronts per CU = two per SIMD
		printf("live lanes %2d: %.2f ns per v_fma_f64 as seen by one of two wavefronts of a SIMD (%s)\n", live, 10.0 * m / (iters * 64.0), hipGetErrorString(hipGetLastError()));
	}
	return 0;
}
