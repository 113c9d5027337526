// One wavefront that samples (constant 100 MHz wall clock, shader-clock counter) pairs every few microseconds while other kernels run: the ratio of the two counters' increments is the
// shader clock the chip actually ran at (tools/diag_clocks.py). hipcc --offload-arch=gfx950 -O3 -shared -fPIC clock_probe.hip -o libclock_probe.so
#include <hip/hip_runtime.h>
__global__ void k_probe(unsigned long long *out, int n, volatile int *stop, int sleep_iters) {
	if (threadIdx.x != 0) return;
	int i = 0;
	for (; i < n && !*stop; i++) {
		out[2 * i] = wall_clock64(); out[2 * i + 1] = __builtin_readcyclecounter();
		for (int k = 0; k < sleep_iters; k++) __builtin_amdgcn_s_sleep(127);
	}
	out[2 * n] = (unsigned long long)i;
}
extern "C" {
static hipStream_t g_s = nullptr; static unsigned long long *g_buf = nullptr; static int *g_stop = nullptr; static int g_n = 0;
int probe_start(int n, int sleep_iters) {
	if (!g_s && hipStreamCreateWithFlags(&g_s, hipStreamNonBlocking) != hipSuccess) return -1;
	if (g_buf) hipFree(g_buf); if (hipMalloc(&g_buf, 16 * (size_t)n + 16) != hipSuccess) return -1; g_n = n;
	if (!g_stop && hipHostMalloc(&g_stop, 4, hipHostMallocMapped) != hipSuccess) return -1; *g_stop = 0;
	int *dstop = nullptr; if (hipHostGetDevicePointer((void **)&dstop, g_stop, 0) != hipSuccess) return -1;
	hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, g_s, g_buf, n, (volatile int *)dstop, sleep_iters);
	return hipGetLastError() == hipSuccess ? 0 : -1;
}
int probe_stop(unsigned long long *host_out /* 2*n + 1 */) {
	*g_stop = 1; if (hipStreamSynchronize(g_s) != hipSuccess) return -1;
	return hipMemcpy(host_out, g_buf, 16 * (size_t)g_n + 8, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
}
