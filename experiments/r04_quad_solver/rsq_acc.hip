// accuracy of the raw v_rsq_f64 against 1/sqrt in long double (host) and against the library rsqrt
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const double *x, double *raw, double *lib, double *nr1, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) { const double v = x[i]; const double y = __builtin_amdgcn_rsq(v); raw[i] = y; lib[i] = rsqrt(v);
	const double e = __builtin_fma(-v * y, y, 1.0); nr1[i] = __builtin_fma(y * e, 0.5, y); } }
int main() { const int n = 1 << 20; std::vector<double> x(n); srand(3); for (int i = 0; i < n; i++) x[i] = std::exp(((rand() % 2000001) / 1000000.0 - 1.0) * 60.0) * (1.0 + (rand() % 1000) * 1e-3);
	double *dx, *dr, *dl, *dn; hipMalloc(&dx, 8 * n); hipMalloc(&dr, 8 * n); hipMalloc(&dl, 8 * n); hipMalloc(&dn, 8 * n); hipMemcpy(dx, x.data(), 8 * n, hipMemcpyHostToDevice);
	hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dr, dl, dn, n); std::vector<double> r(n), l(n), q(n); hipMemcpy(r.data(), dr, 8 * n, hipMemcpyDeviceToHost); hipMemcpy(l.data(), dl, 8 * n, hipMemcpyDeviceToHost); hipMemcpy(q.data(), dn, 8 * n, hipMemcpyDeviceToHost);
	double er = 0, el = 0, eq = 0; for (int i = 0; i < n; i++) { const long double t = 1.0L / sqrtl((long double)x[i]); er = fmax(er, (double)fabsl((r[i] - t) / t)); el = fmax(el, (double)fabsl((l[i] - t) / t)); eq = fmax(eq, (double)fabsl((q[i] - t) / t)); }
	std::printf("max relative error: raw v_rsq_f64 %.3e | library rsqrt %.3e | raw + one Newton step (2 fma + 2 mul) %.3e   (eps = 1.1e-16)\n", er, el, eq); return 0; }
