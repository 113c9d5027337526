// N host threads, each with its own stream, enqueue chains of small dependent kernels (a 20 us wait, one 256-thread workgroup x 8) with a copy-back + stream synchronisation every
// 40 launches -- the launch pattern of a lock-step gang (srba_hip.hip big_gang_run). Prints the mean kernel-to-kernel period per stream for N = 1 .. 8: does it stay flat?
// build: hipcc --offload-arch=gfx950 -O3 -pthread stream_chains.hip -o stream_chains
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
__global__ void k_wait(long long ticks, double *sink) { const long long t0 = wall_clock64(); while (wall_clock64() - t0 < ticks) { } if (ticks < 0) sink[0] = 1; }
static void worker(int launches, int sync_every, double *out_us) {
	hipStream_t s; (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking); double *d; (void)hipMalloc(&d, 64); double *h; (void)hipHostMalloc(&h, 64, hipHostMallocDefault);
	for (int i = 0; i < 20; i++) hipLaunchKernelGGL(k_wait, dim3(8), dim3(256), 0, s, 100, d);
	(void)hipStreamSynchronize(s);
	const auto t0 = std::chrono::steady_clock::now();
	for (int i = 0; i < launches; i++) { hipLaunchKernelGGL(k_wait, dim3(8), dim3(256), 0, s, 2000 /* 20 us */, d);
		if ((i + 1) % sync_every == 0) { (void)hipMemcpyAsync(h, d, 64, hipMemcpyDeviceToHost, s); (void)hipStreamSynchronize(s); } }
	(void)hipStreamSynchronize(s);
	*out_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / launches;
	(void)hipFree(d); (void)hipHostFree(h); (void)hipStreamDestroy(s);
}
int main() {
	for (int n : {1, 2, 3, 4, 6, 8}) { std::vector<double> us(n, 0); std::vector<std::thread> th;
		for (int i = 0; i < n; i++) th.emplace_back([&, i] { (void)hipSetDevice(0); worker(2000, 40, &us[i]); });
		for (auto &t : th) t.join();
		printf("%d streams: period per launch (20 us kernels)", n); for (double u : us) printf(" %.1f", u); printf(" us\n"); }
	return 0;
}
