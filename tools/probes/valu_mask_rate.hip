// Does a wave64 FP64 instruction cost less when only some of its 16-lane quarters have live lanes? The block-sparse LDL^t of the headline kernel issues ~80 FP64 instructions per column
// with 12 of 64 lanes live on average (profiles/r06_sq_summary.json) and is bound by VALU issue on SIMDs that hold two or three wavefronts: if the hardware skipped dead quarters,
// packing the live lanes into the low quarter would pay. A chain-free stream of v_fma_f64 (8 independent accumulators) under EXEC = the low N lanes.
// build: hipcc --offload-arch=gfx950 -O3 valu_mask_rate.hip -o valu_mask_rate ; run: ./valu_mask_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <class T> __global__ void __launch_bounds__(64) k_fma(int live, int iters, long long *ticks, double *sink, int spread) {
	const int lane = threadIdx.x; T a[8]; for (int k = 0; k < 8; k++) a[k] = (T)(1 + (lane + k)); const T x = (T)3, y = (T)1;
	long long t0 = 0, t1 = 0, c0 = 0, c1 = 0;
	if (spread ? ((lane % spread) == 0 && lane / spread < live) : lane < live) {
		t0 = wall_clock64(); c0 = clock64();
		for (int i = 0; i < iters; i++) {
#pragma unroll
			for (int r = 0; r < 8; r++) {
#pragma unroll
				for (int k = 0; k < 8; k++) { if constexpr (sizeof(T) == 8 || !__is_integral(T)) a[k] = a[k] * x + y; else a[k] = a[k] * x + y; }
			}
		}
		t1 = wall_clock64(); c1 = clock64();
	}
	double s = 0; for (int k = 0; k < 8; k++) s += (double)a[k];
	if (lane == 0) { ticks[blockIdx.x] = t1 - t0; ticks[gridDim.x + blockIdx.x] = c1 - c0; sink[blockIdx.x] = s; }
}
int main() {
	long long *ticks; double *sink; const int grid = 256 * 8; (void)hipMalloc(&ticks, 16 * grid); (void)hipMalloc(&sink, 8 * grid); long long h[256 * 16];
	for (int ty = 0; ty < 3; ty++) for (int spread : {0}) for (int live : {64, 16, 9, 8, 1}) {
		const int iters = 2000; for (int rep = 0; rep < 2; rep++) { if (ty == 0) hipLaunchKernelGGL(k_fma<double>, dim3(grid), dim3(64), 0, 0, live, iters, ticks, sink, spread); else if (ty == 1) hipLaunchKernelGGL(k_fma<float>, dim3(grid), dim3(64), 0, 0, live, iters, ticks, sink, spread); else hipLaunchKernelGGL(k_fma<int>, dim3(grid), dim3(64), 0, 0, live, iters, ticks, sink, spread); (void)hipDeviceSynchronize(); }
		(void)hipMemcpy(h, ticks, 16 * grid, hipMemcpyDeviceToHost); double m = 0, mc = 0; for (int i = 0; i < grid; i++) { m += (double)h[i]; mc += (double)h[grid + i]; } m /= grid; mc /= grid;
		// 100 MHz ticks: 10 ns each; 64 instructions per iteration; eight wavefronts per CU = two per SIMD
		printf("%s spread %d live lanes %2d: %.2f ns per multiply-add as seen by one of two wavefronts of a SIMD = %.1f shader-clock ticks (clock64): %.0f MHz (%s)\n", ty == 0 ? "f64" : ty == 1 ? "f32" : "i32", spread, live, 10.0 * m / (iters * 64.0), mc / (iters * 64.0), mc / (10.0 * m) * 1e3, hipGetErrorString(hipGetLastError()));
	}
	return 0;
}
