// Counter calibration (round 4): kernels that move a KNOWN number of bytes in the access shapes of the fused LM kernel, each over a buffer four times the
// 256 MB Infinity Cache, to be run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (tools/calib/run_calib.sh). Build: hipcc --offload-arch=gfx950 -O3 counter_calib.hip
//   rd_stream16   : coalesced 16 B per lane streaming read                          (the guide's x2 case)
//   rd_gather40   : one 40-byte pose record per lane, records visited in a scattered order, 16+16+8 B at 8-byte alignment (K1 / K4 / K2 pose gathers)
//   rd_gather72   : one 72-byte block per lane, scattered, 4 x 16 + 8 B                (K6 / K5 Jacobian-block gathers)
//   rd_gather8    : one 8-byte scalar per lane, scattered
//   wr_stream16   : coalesced 16 B per lane streaming write
//   wr_rec72      : every lane stores its own 72-byte record, consecutive lanes consecutive records (K2 block stores)
//   wr_rec40x2    : every lane stores two consecutive 40-byte poses (K1: pose and its inverse)
//   wr_scatter40  : one 40-byte record per lane at a scattered position
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
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
__device__ __forceinline__ size_t scatter(size_t i, size_t n_pow2) { return (i * 0x9E3779B97F4A7C15ull + 12345) & (n_pow2 - 1); } // odd multiplier: a permutation of [0, n)
__global__ void rd_stream16(const double *p, size_t n16, double *sink) { double a = 0; for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) { const f64x2u v = *(const f64x2u *)(p + 2 * i); a += v.x + v.y; } if (a == 1.2345) *sink = a; }
template <int N> __global__ void rd_gather(const double *p, size_t nrec, double *sink) { double a = 0; for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nrec; i += (size_t)gridDim.x * blockDim.x) { double v[N]; ldn<N>(v, p + scatter(i, nrec) * N); for (int k = 0; k < N; k++) a += v[k]; } if (a == 1.2345) *sink = a; }
__global__ void wr_stream16(double *p, size_t n16) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) { f64x2u v; v.x = (double)i; v.y = 1.0; *(f64x2u *)(p + 2 * i) = v; } }
template <int N, int R> __global__ void wr_rec(double *p, size_t nrec) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nrec / R; i += (size_t)gridDim.x * blockDim.x) { double v[N]; for (int k = 0; k < N; k++) v[k] = (double)(i + k); for (int r = 0; r < R; r++) stn<N>(p + (i * R + r) * N, v); } }
template <int N> __global__ void wr_scatter(double *p, size_t nrec) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nrec; i += (size_t)gridDim.x * blockDim.x) { double v[N]; for (int k = 0; k < N; k++) v[k] = (double)(i + k); stn<N>(p + scatter(i, nrec) * N, v); } }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); std::exit(1); } } while (0)
int main() {
	const size_t bytes = 1ull << 30; double *buf, *sink; CK(hipMalloc(&buf, bytes + 4096)); CK(hipMalloc(&sink, 8)); CK(hipMemset(buf, 0, bytes));
	const int grid = 256 * 16, block = 256; hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	auto pow2_below = [](size_t x) { size_t p = 1; while (2 * p <= x) p *= 2; return p; };
	const size_t n40 = pow2_below(bytes / 40), n72 = pow2_below(bytes / 72), n8 = pow2_below(bytes / 8);
	struct { const char *name; double bytes; } rows[8];
	for (int rep = 0; rep < 2; rep++) { int r = 0; float ms;
#define RUN(NAME, BYTES, ...) CK(hipEventRecord(e0)); __VA_ARGS__; CK(hipEventRecord(e1)); CK(hipDeviceSynchronize()); CK(hipEventElapsedTime(&ms, e0, e1)); rows[r].name = NAME; rows[r].bytes = (double)(BYTES); if (rep) std::printf("CALIB %-14s bytes %.0f  ms %.3f  GB/s %.1f\n", NAME, (double)(BYTES), ms, (BYTES) / ms / 1e6); r++;
		RUN("rd_stream16", bytes, hipLaunchKernelGGL(rd_stream16, dim3(grid), dim3(block), 0, 0, buf, bytes / 16, sink));
		RUN("rd_gather40", n40 * 40, hipLaunchKernelGGL(rd_gather<5>, dim3(grid), dim3(block), 0, 0, buf, n40, sink));
		RUN("rd_gather72", n72 * 72, hipLaunchKernelGGL(rd_gather<9>, dim3(grid), dim3(block), 0, 0, buf, n72, sink));
		RUN("rd_gather8", n8 * 8, hipLaunchKernelGGL(rd_gather<1>, dim3(grid), dim3(block), 0, 0, buf, n8, sink));
		RUN("wr_stream16", bytes, hipLaunchKernelGGL(wr_stream16, dim3(grid), dim3(block), 0, 0, buf, bytes / 16));
		RUN("wr_rec72", (bytes / 72) * 72, hipLaunchKernelGGL((wr_rec<9, 1>), dim3(grid), dim3(block), 0, 0, buf, bytes / 72));
		RUN("wr_rec40x2", (bytes / 80) * 80, hipLaunchKernelGGL((wr_rec<5, 2>), dim3(grid), dim3(block), 0, 0, buf, (bytes / 80) * 2));
		RUN("wr_scatter40", n40 * 40, hipLaunchKernelGGL(wr_scatter<5>, dim3(grid), dim3(block), 0, 0, buf, n40));
	}
	return 0;
}
