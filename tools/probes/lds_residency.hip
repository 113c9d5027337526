// How many workgroups of (threads, dynamic LDS bytes, VGPR budget) does a CU of gfx950 hold at once? Each workgroup waits 50 us; wall time / 50 us = number of rounds = grid / (CUs x resident).
// build: hipcc --offload-arch=gfx950 -O3 lds_residency.hip -o lds_residency ; run: ./lds_residency
#include <hip/hip_runtime.h>
#include <cstdio>
extern __shared__ double lds[];
template <int WAVES_PER_EU, int THREADS = 128> __global__ void __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(WAVES_PER_EU, WAVES_PER_EU))) k_wait(long long ticks, double *sink) {
	lds[threadIdx.x & 127] = 1.0; __syncthreads();
	const long long t0 = wall_clock64(); while (wall_clock64() - t0 < ticks) { }
	if (lds[(threadIdx.x + 1) & 127] == 2.0) sink[0] = 1.0;
}
template <int W, int THREADS = 128> static void run(int kb, int bytes_extra) {
	const size_t lds_bytes = (size_t)kb * 1024 + bytes_extra; const int grid = 256 * 48; double *sink; (void)hipMalloc(&sink, 8);
	(void)hipFuncSetAttribute((const void *)k_wait<W, THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
	hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
	for (int rep = 0; rep < 2; rep++) { (void)hipEventRecord(a); hipLaunchKernelGGL((k_wait<W, THREADS>), dim3(grid), dim3(THREADS), lds_bytes, 0, 5000 /* 100 MHz ticks = 50 us */, sink); (void)hipEventRecord(b); (void)hipEventSynchronize(b); }
	float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
	printf("waves_per_eu %d, %d threads, LDS %6zu B: %.3f ms -> %.1f rounds of 50 us -> %.2f workgroups resident per CU (err %s)\n", W, THREADS, lds_bytes, ms, ms / 0.05, grid / 256.0 / (ms / 0.05), hipGetErrorString(hipGetLastError()));
}
int main() { for (int kb : {20, 26, 32, 36, 39}) run<2>(kb, 0); run<2>(40, 0); run<2>(40, -256); run<2>(40, -1280); run<2>(53, 0); run<2>(80, 0); run<3>(20, 0); run<3>(26, 0); run<3>(13, 0);
	for (int kb : {50, 51, 52, 53}) run<3, 256>(kb, 0); run<3, 256>(53, -768); run<3, 256>(53, -512); run<3, 256>(53, -256); run<4, 256>(40, 0); run<4, 256>(39, 0); return 0; } // (round 6: bins of four wavefronts)
