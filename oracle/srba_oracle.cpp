/*
 * srba_oracle.cpp -- CPU ORACLE for the SRBA local-optimisation hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under srba_amd/ or include/ may include, link or call this file; it is
 * used by tests/, __graft_entry__.smoke() and the `cpu_baseline` leg of bench.py, as the checker / CPU comparator.
 *
 * What it is: a plain-C++17 restatement (no Eigen/MRPT/CSparse, none of which exist in this image) of the
 * reference's numeric optimiser RbaEngine<>::optimize_edges() (include/srba/impl/optimize_edges.h:44-793) and
 * everything it calls, operating on the flat "problem capsule" of include/srba_hip.h:
 *   K1  spanning-tree numeric refresh     impl/spantree_update_numeric.h:19-81
 *   K2  dh_dAp blocks                     impl/jacobians.h:207-354, :364-494 (SE3 points), :501-641 (SE2 points), :645-744 (SE2 rel. poses)
 *   K3  dh_df blocks                      impl/jacobians.h:888-1013
 *   K4  residuals / chi2 / pseudo-Huber   impl/reprojection_residuals.h:16-81, RbaEngine.h:810-813
 *   K5  minus gradient                    impl/compute_minus_gradient.h:20-91
 *   K6  Hessian blocks                    impl/sparse_hessian_update_numeric.h:22-60, srba_options_noise.h:44-71,102-131
 *   K7,K8,K10 Schur complement            impl/schur.h:180-311
 *   K9  solvers                           impl/lev-marq_solvers.h:80-187 (sparse), :279-381 (Schur+sparse), :474-568 (Schur+dense LLT)
 *   K11,K12 update/backup/restore + LM control flow   impl/optimize_edges.h:361-696
 *   sensor models                         models/sensors.h (RelativePoses2D :770-834, RangeBearing2D :661-736, Cartesian2D :421-515,
 *                                         Stereo :175-315, Monocular :50-141, Cartesian3D :347-415)
 *   sensor-pose policy                    srba_options_sensor_pose.h:32-135
 * It reproduces the behaviours listed in SURVEY.md Appendix B (duplicated observations in chi2, 1/sigma scaling,
 * in-place Schur gradient mutation across lambda retries, partial in-loop spanning-tree refresh, stale/zeroed
 * blocks of invalid Jacobian rows).
 *
 * PARITY UNPINNED against a compiled reference: MRPT >= 1.3.0, Eigen3 and CSparse are un-vendored third-party
 * dependencies (README.md:13) and absent here, so no reference binary can be built (oracle/_ref does not exist).
 * Third-party arithmetic is restated from the published algorithms:
 *   - mrpt::poses::CPose2D / CPose3D compose, inverse, composePoint; mrpt::math::wrapToPi; SE_traits<3>::pseudo_exp
 *     (Rodrigues) -- MRPT 1.x;
 *   - Eigen::LLT (upper, fails on pivot <= 0), Eigen::FullPivLU::isInvertible()/inverse() (threshold eps*n*|maxpivot|) -- Eigen 3.2;
 *   - CSparse cs_etree/cs_ereach/cs_chol/cs_lsolve/cs_ltsolve (T. Davis, "Direct Methods for Sparse Linear Systems", CSparse 2.x/3.x,
 *     as wrapped by mrpt::math::CSparseMatrix::CholeskyDecomp).  cs_amd (order=1) is replaced by an exact minimum-degree ordering of
 *     the block graph: a fill-reducing permutation changes rounding only.
 * The oracle is pinned instead by the reference's own tests (tests/ of this repo restate them): SchurTests (1e-10),
 * MiniProblems.* with their literal inputs and tolerances, plus finite-difference Jacobian checks in numpy.
 */
#include "../include/srba_hip.h"

#include <atomic>
#include <chrono>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

namespace {

static std::atomic<long long> g_symbolic_ns(0); // time spent in the per-call symbolic Cholesky analysis (sparse_setup), summed over threads
static const bool g_refresh_all = (getenv("SRBA_ORACLE_REFRESH_ALL") != nullptr); // diagnostic: refresh every ST pose in every trial (study of App. B-12)
static const bool g_exact_relpose_jacobian = (getenv("SRBA_ORACLE_EXACT_JAC") != nullptr); // diagnostic switch, see jacobian_dh_dp

// ------------------------------------------------------------------------------------------------
// MRPT angle wrapping  [EXT mrpt/math/wrap2pi.h]
// ------------------------------------------------------------------------------------------------
inline double wrapTo2Pi(double a) { const bool neg = a < 0; a = std::fmod(a, 2.0 * M_PI); if (neg) a += 2.0 * M_PI; return a; }
inline double wrapToPi(double a) { return wrapTo2Pi(a + M_PI) - M_PI; }

// ------------------------------------------------------------------------------------------------
// Poses
// ------------------------------------------------------------------------------------------------
struct Pose2 {
	double x = 0, y = 0, phi = 0;
	static constexpr int PD = 3;
	void load(const double *p) { x = p[0]; y = p[1]; phi = p[2]; }
	void store(double *p) const { p[0] = x; p[1] = y; p[2] = phi; }
};
// A (+) B  [EXT CPose2D::composeFrom]
inline Pose2 compose(const Pose2 &A, const Pose2 &B) {
	const double c = std::cos(A.phi), s = std::sin(A.phi);
	Pose2 r; r.x = A.x + B.x * c - B.y * s; r.y = A.y + B.x * s + B.y * c; r.phi = wrapToPi(A.phi + B.phi); return r;
}
// (-)P  [EXT CPose2D::inverse]
inline Pose2 inverse(const Pose2 &P) {
	const double c = std::cos(P.phi), s = std::sin(P.phi);
	Pose2 r; r.x = -P.x * c - P.y * s; r.y = P.x * s - P.y * c; r.phi = -P.phi; return r;
}
// A (-) B = (-)B (+) A   [EXT CPose2D::inverseComposeFrom]
inline Pose2 inv_compose(const Pose2 &A, const Pose2 &B) {
	const double c = std::cos(B.phi), s = std::sin(B.phi);
	Pose2 r; r.x = (A.x - B.x) * c + (A.y - B.y) * s; r.y = -(A.x - B.x) * s + (A.y - B.y) * c; r.phi = wrapToPi(A.phi - B.phi); return r;
}
inline void compose_point(const Pose2 &P, double lx, double ly, double &gx, double &gy) {
	const double c = std::cos(P.phi), s = std::sin(P.phi);
	gx = P.x + lx * c - ly * s; gy = P.y + lx * s + ly * c;
}

struct Pose3 {
	double t[3] = {0, 0, 0};
	double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}; // row-major
	static constexpr int PD = 12;
	void load(const double *p) { for (int i = 0; i < 3; i++) t[i] = p[i]; for (int i = 0; i < 9; i++) R[i] = p[3 + i]; }
	void store(double *p) const { for (int i = 0; i < 3; i++) p[i] = t[i]; for (int i = 0; i < 9; i++) p[3 + i] = R[i]; }
};
inline Pose3 compose(const Pose3 &A, const Pose3 &B) { // [EXT CPose3D::composeFrom]
	Pose3 r;
	for (int i = 0; i < 3; i++)
		for (int j = 0; j < 3; j++) r.R[3 * i + j] = A.R[3 * i + 0] * B.R[0 + j] + A.R[3 * i + 1] * B.R[3 + j] + A.R[3 * i + 2] * B.R[6 + j];
	for (int i = 0; i < 3; i++) r.t[i] = A.t[i] + A.R[3 * i + 0] * B.t[0] + A.R[3 * i + 1] * B.t[1] + A.R[3 * i + 2] * B.t[2];
	return r;
}
inline Pose3 inverse(const Pose3 &P) { // [EXT CPose3D::inverse -> homogeneousMatrixInverse]
	Pose3 r;
	for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.R[3 * i + j] = P.R[3 * j + i];
	for (int i = 0; i < 3; i++) r.t[i] = -(P.R[0 + i] * P.t[0] + P.R[3 + i] * P.t[1] + P.R[6 + i] * P.t[2]);
	return r;
}
inline Pose3 inv_compose(const Pose3 &A, const Pose3 &B) { return compose(inverse(B), A); } // A (-) B
inline void compose_point(const Pose3 &P, const double l[3], double g[3]) {
	for (int i = 0; i < 3; i++) g[i] = P.t[i] + P.R[3 * i + 0] * l[0] + P.R[3 * i + 1] * l[1] + P.R[3 * i + 2] * l[2];
}
inline void inv_compose_point(const Pose3 &P, const double g[3], double l[3]) { // l = (-)P (+) g
	const double d[3] = {g[0] - P.t[0], g[1] - P.t[1], g[2] - P.t[2]};
	for (int i = 0; i < 3; i++) l[i] = P.R[0 + i] * d[0] + P.R[3 + i] * d[1] + P.R[6 + i] * d[2];
}
// SE_traits<3>::pseudo_exp: t = v[0:3], R = exp_so3(v[3:6])  [EXT]
inline Pose3 pseudo_exp3(const double v[6]) {
	Pose3 r; r.t[0] = v[0]; r.t[1] = v[1]; r.t[2] = v[2];
	const double wx = v[3], wy = v[4], wz = v[5];
	const double th2 = wx * wx + wy * wy + wz * wz, th = std::sqrt(th2);
	double a, b; // R = I + a*W + b*W^2
	if (th < 1e-8) { a = 1.0 - th2 / 6.0; b = 0.5 - th2 / 24.0; }
	else { a = std::sin(th) / th; b = (1.0 - std::cos(th)) / th2; }
	const double W[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
	double W2[9];
	for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) W2[3 * i + j] = W[3 * i] * W[j] + W[3 * i + 1] * W[3 + j] + W[3 * i + 2] * W[6 + j];
	for (int i = 0; i < 9; i++) r.R[i] = ((i % 4 == 0) ? 1.0 : 0.0) + a * W[i] + b * W2[i];
	return r;
}
// [EXT] CPose3D(x,y,z,yaw,pitch,roll): R = Rz(yaw) Ry(pitch) Rx(roll)
inline Pose3 pose3_from_ypr(const double v[6]) {
	Pose3 r; r.t[0] = v[0]; r.t[1] = v[1]; r.t[2] = v[2];
	const double cy = std::cos(v[3]), sy = std::sin(v[3]), cp = std::cos(v[4]), sp = std::sin(v[4]), cr = std::cos(v[5]), sr = std::sin(v[5]);
	const double R[9] = {cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr, sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr, -sp, cp * sr, cp * cr};
	for (int k = 0; k < 9; k++) r.R[k] = R[k];
	return r;
}
// [EXT] SE_traits<3>::pseudo_ln: (t, ln(R)); ln(R) = theta / (2 sin theta) * vee(R - R^t), theta = acos((tr R - 1)/2)  (include/mrpt_lite.h CPose3D::ln_rotation)
inline void pseudo_ln3(const Pose3 &P, double out[6]) {
	for (int i = 0; i < 3; i++) out[i] = P.t[i];
	double c = 0.5 * (P.R[0] + P.R[4] + P.R[8] - 1.0); c = std::max(-1.0, std::min(1.0, c));
	const double th = std::acos(c), f = (th < 1e-8) ? 0.5 : th / (2.0 * std::sin(th));
	out[3] = f * (P.R[7] - P.R[5]); out[4] = f * (P.R[2] - P.R[6]); out[5] = f * (P.R[3] - P.R[1]);
}
// [EXT] CPose3D::ln_rot_jacob: d ln(R) / d vec(R), 3x9, vec(R) = stacked COLUMNS (Blanco, "A tutorial on SE(3) transformation parameterizations and on-manifold
// optimization", section 10.3.2): the derivative of the formula above taking the nine entries as independent; small-angle branch at (tr R - 1)/2 > 0.99999.
inline void ln_rot_jacob(const double R[9], double M[27]) {
	const double d = 0.5 * (R[0] + R[4] + R[8] - 1.0); double a[3] = {0, 0, 0}, b = 0.5;
	if (!(d > 0.99999)) {
		const double th = std::acos(d), sq = std::sqrt(1.0 - d * d), k = (d * th - sq) / (4.0 * sq * sq * sq);
		b = th / (2.0 * sq); a[0] = k * (R[7] - R[5]); a[1] = k * (R[2] - R[6]); a[2] = k * (R[3] - R[1]);
	}
	const double m[27] = {a[0], 0, 0, 0, a[0], b, 0, -b, a[0],   a[1], 0, -b, 0, a[1], 0, b, 0, a[1],   a[2], b, 0, -b, a[2], 0, 0, 0, a[2]};
	for (int k = 0; k < 27; k++) M[k] = m[k];
}
// [EXT] CPose3D(CPose2D): rotation about z, z = 0
inline Pose3 pose3_from_pose2(const Pose2 &p) {
	Pose3 r; r.t[0] = p.x; r.t[1] = p.y; r.t[2] = 0; const double c = std::cos(p.phi), s = std::sin(p.phi);
	const double R[9] = {c, -s, 0, s, c, 0, 0, 0, 1}; for (int k = 0; k < 9; k++) r.R[k] = R[k]; return r;
}
inline Pose2 pseudo_exp2(const double v[3]) { Pose2 r; r.x = v[0]; r.y = v[1]; r.phi = v[2]; return r; } // SE_traits<2>: identity map [EXT]

// quaternion (r,x,y,z) -> rotation matrix [EXT CQuaternion::rotationMatrixNoResize]
inline void quat_to_R(const double q[4], double R[9]) {
	const double r = q[0], x = q[1], y = q[2], z = q[3];
	R[0] = r * r + x * x - y * y - z * z; R[1] = 2 * (x * y - r * z);         R[2] = 2 * (z * x + r * y);
	R[3] = 2 * (x * y + r * z);         R[4] = r * r - x * x + y * y - z * z; R[5] = 2 * (y * z - r * x);
	R[6] = 2 * (z * x - r * y);         R[7] = 2 * (y * z + r * x);         R[8] = r * r - x * x - y * y + z * z;
}

// ------------------------------------------------------------------------------------------------
// small dense helpers (row-major)
// ------------------------------------------------------------------------------------------------
template <int M, int K, int N> inline void mm(const double *A, const double *B, double *C) { // C(MxN) = A(MxK) B(KxN)
	for (int i = 0; i < M; i++) for (int j = 0; j < N; j++) { double s = 0; for (int k = 0; k < K; k++) s += A[i * K + k] * B[k * N + j]; C[i * N + j] = s; }
}
template <int M, int K, int N> inline void mtm_acc(const double *A, const double *B, double *C) { // C(KxN)... C += A^t(KxM)... A is MxK
	for (int i = 0; i < K; i++) for (int j = 0; j < N; j++) { double s = 0; for (int k = 0; k < M; k++) s += A[k * K + i] * B[k * N + j]; C[i * N + j] += s; }
}

// Eigen::FullPivLU<N x N>: invertibility test + inverse [EXT Eigen 3.2 FullPivLU.h: threshold = eps * diagonalSize, rank counts |pivot| > |maxpivot|*threshold]
template <int N> bool fullpivlu_inverse(const double *A, double *Ainv) {
	double lu[N * N]; int rowp[N], colp[N];
	for (int i = 0; i < N * N; i++) lu[i] = A[i];
	for (int i = 0; i < N; i++) { rowp[i] = i; colp[i] = i; }
	double maxpivot = 0; int nonzero = N; double piv[N];
	for (int k = 0; k < N; k++) {
		int br = k, bc = k; double best = -1;
		for (int c = k; c < N; c++) for (int r = k; r < N; r++) { const double v = std::fabs(lu[r * N + c]); if (v > best) { best = v; br = r; bc = c; } }
		if (best == 0.0) { nonzero = k; for (int i = k; i < N; i++) piv[i] = 0; break; }
		if (best > maxpivot) maxpivot = best;
		if (br != k) { for (int c = 0; c < N; c++) std::swap(lu[k * N + c], lu[br * N + c]); std::swap(rowp[k], rowp[br]); }
		if (bc != k) { for (int r = 0; r < N; r++) std::swap(lu[r * N + k], lu[r * N + bc]); std::swap(colp[k], colp[bc]); }
		piv[k] = lu[k * N + k];
		for (int r = k + 1; r < N; r++) lu[r * N + k] /= lu[k * N + k];
		for (int r = k + 1; r < N; r++) for (int c = k + 1; c < N; c++) lu[r * N + c] -= lu[r * N + k] * lu[k * N + c];
	}
	(void)nonzero;
	const double thr = std::numeric_limits<double>::epsilon() * N * std::fabs(maxpivot);
	int rank = 0;
	for (int k = 0; k < N; k++) if (std::fabs(piv[k]) > thr) rank++;
	if (rank != N) return false;
	// inverse: solve P A Q = L U  ->  A^-1 = Q U^-1 L^-1 P
	for (int col = 0; col < N; col++) {
		double b[N]; for (int r = 0; r < N; r++) b[r] = (rowp[r] == col) ? 1.0 : 0.0;
		for (int r = 0; r < N; r++) for (int c = 0; c < r; c++) b[r] -= lu[r * N + c] * b[c];
		for (int r = N - 1; r >= 0; r--) { for (int c = r + 1; c < N; c++) b[r] -= lu[r * N + c] * b[c]; b[r] /= lu[r * N + r]; }
		for (int r = 0; r < N; r++) Ainv[colp[r] * N + col] = b[r];
	}
	return true;
}

// ------------------------------------------------------------------------------------------------
// Dense Cholesky, upper storage A = U^t U  [EXT Eigen::LLT<Upper>: fails when a pivot is <= 0]
// ------------------------------------------------------------------------------------------------
bool dense_llt_upper(std::vector<double> &A, int n) { // in place; uses upper triangle (row-major A[i*n+j], i<=j)
	for (int k = 0; k < n; k++) {
		double x = A[k * n + k];
		for (int i = 0; i < k; i++) x -= A[i * n + k] * A[i * n + k];
		if (!(x > 0.0)) return false;
		const double d = std::sqrt(x); A[k * n + k] = d;
		for (int j = k + 1; j < n; j++) {
			double s = A[k * n + j];
			for (int i = 0; i < k; i++) s -= A[i * n + k] * A[i * n + j];
			A[k * n + j] = s / d;
		}
	}
	return true;
}
void dense_llt_solve(const std::vector<double> &U, int n, const double *b, double *x) {
	std::vector<double> y(n);
	for (int i = 0; i < n; i++) { double s = b[i]; for (int k = 0; k < i; k++) s -= U[k * n + i] * y[k]; y[i] = s / U[i * n + i]; } // U^t y = b
	for (int i = n - 1; i >= 0; i--) { double s = y[i]; for (int k = i + 1; k < n; k++) s -= U[i * n + k] * x[k]; x[i] = s / U[i * n + i]; }
}

// ------------------------------------------------------------------------------------------------
// Sparse Cholesky on an upper-triangular CSC matrix: restatement of CSparse cs_etree / cs_ereach / cs_chol
// (up-looking), with symbolic analysis done once and reused (CholeskyDecomp::update) [EXT].
// ------------------------------------------------------------------------------------------------
struct SparseChol {
	int n = 0;
	std::vector<int> perm, pinv;       // fill-reducing permutation (new -> old) and its inverse
	std::vector<int> parent, Lp, Li;   // etree and pattern of L (CSC, diagonal first in each column)
	std::vector<double> Lx;
	bool symbolic_done = false;

	// C = P A P^t (upper), given as CSC Cp/Ci/Cx
	static void etree(int n, const std::vector<int> &Cp, const std::vector<int> &Ci, std::vector<int> &parent) {
		parent.assign(n, -1); std::vector<int> anc(n, -1);
		for (int k = 0; k < n; k++)
			for (int p = Cp[k]; p < Cp[k + 1]; p++)
				for (int i = Ci[p]; i != -1 && i < k;) { const int inext = anc[i]; anc[i] = k; if (inext == -1) parent[i] = k; i = inext; }
	}
	static int ereach(int k, const std::vector<int> &Cp, const std::vector<int> &Ci, const std::vector<int> &parent, std::vector<int> &s, std::vector<int> &w, int n) {
		int top = n; w[k] = k; // mark with stamp k (w initialised to -1, stamps increase)
		for (int p = Cp[k]; p < Cp[k + 1]; p++) {
			int i = Ci[p]; if (i > k) continue;
			int len = 0;
			for (; w[i] != k; i = parent[i]) { s[len++] = i; w[i] = k; }
			while (len > 0) s[--top] = s[--len];
		}
		return top;
	}
	// Returns false if not positive definite.
	bool factor(const std::vector<int> &Cp, const std::vector<int> &Ci, const std::vector<double> &Cx) {
		std::vector<int> s(n), w(n, -1), c(n);
		if (!symbolic_done) {
			etree(n, Cp, Ci, parent);
			std::vector<int> cnt(n, 1);
			for (int k = 0; k < n; k++) { const int top = ereach(k, Cp, Ci, parent, s, w, n); for (int t = top; t < n; t++) cnt[s[t]]++; }
			Lp.assign(n + 1, 0); for (int i = 0; i < n; i++) Lp[i + 1] = Lp[i] + cnt[i];
			Li.assign(Lp[n], 0); Lx.assign(Lp[n], 0.0); symbolic_done = true;
			std::fill(w.begin(), w.end(), -1);
		}
		std::vector<double> x(n, 0.0);
		for (int k = 0; k < n; k++) c[k] = Lp[k];
		for (int k = 0; k < n; k++) {
			const int top = ereach(k, Cp, Ci, parent, s, w, n);
			x[k] = 0;
			for (int p = Cp[k]; p < Cp[k + 1]; p++) if (Ci[p] <= k) x[Ci[p]] = Cx[p];
			double d = x[k]; x[k] = 0;
			for (int t = top; t < n; t++) {
				const int i = s[t];
				const double lki = x[i] / Lx[Lp[i]]; x[i] = 0;
				for (int p = Lp[i] + 1; p < c[i]; p++) x[Li[p]] -= Lx[p] * lki;
				d -= lki * lki;
				const int p = c[i]++; Li[p] = k; Lx[p] = lki;
			}
			if (d <= 0) return false; // cs_chol: "not pos def" -> CExceptionNotDefPos
			const int p = c[k]++; Li[p] = k; Lx[p] = std::sqrt(d);
		}
		return true;
	}
	void solve(const double *b, double *xout) const { // backsub: x = P^t L^-t L^-1 P b
		std::vector<double> y(n);
		for (int k = 0; k < n; k++) y[k] = b[perm[k]];
		for (int j = 0; j < n; j++) { y[j] /= Lx[Lp[j]]; for (int p = Lp[j] + 1; p < Lp[j + 1]; p++) y[Li[p]] -= Lx[p] * y[j]; }
		for (int j = n - 1; j >= 0; j--) { for (int p = Lp[j] + 1; p < Lp[j + 1]; p++) y[j] -= Lx[p] * y[Li[p]]; y[j] /= Lx[Lp[j]]; }
		for (int k = 0; k < n; k++) xout[perm[k]] = y[k];
	}
};

// Exact minimum-degree ordering of a block graph (stand-in for cs_amd, see header).
std::vector<int> min_degree_order(int nb, const std::vector<std::pair<int, int>> &edges) {
	std::vector<std::vector<char>> adj(nb, std::vector<char>(nb, 0));
	for (auto &e : edges) if (e.first != e.second) { adj[e.first][e.second] = 1; adj[e.second][e.first] = 1; }
	std::vector<char> gone(nb, 0); std::vector<int> order; order.reserve(nb);
	for (int step = 0; step < nb; step++) {
		int best = -1, bestdeg = 1 << 30;
		for (int v = 0; v < nb; v++) if (!gone[v]) { int d = 0; for (int u = 0; u < nb; u++) if (!gone[u] && adj[v][u]) d++; if (d < bestdeg) { bestdeg = d; best = v; } }
		gone[best] = 1; order.push_back(best);
		std::vector<int> nbrs; for (int u = 0; u < nb; u++) if (!gone[u] && adj[best][u]) nbrs.push_back(u);
		for (size_t a = 0; a < nbrs.size(); a++) for (size_t b = a + 1; b < nbrs.size(); b++) { adj[nbrs[a]][nbrs[b]] = 1; adj[nbrs[b]][nbrs[a]] = 1; }
	}
	return order;
}

// ------------------------------------------------------------------------------------------------
// Family traits
// ------------------------------------------------------------------------------------------------
template <int FAM> struct Fam;
template <> struct Fam<SRBA_SE2_RELPOSE2D> { using pose_t = Pose2; static constexpr int P = 3, L = 3, O = 3; static constexpr bool relpose = true; };
template <> struct Fam<SRBA_SE2_RB2D>      { using pose_t = Pose2; static constexpr int P = 3, L = 2, O = 2; static constexpr bool relpose = false; };
template <> struct Fam<SRBA_SE2_CART2D>    { using pose_t = Pose2; static constexpr int P = 3, L = 2, O = 2; static constexpr bool relpose = false; };
template <> struct Fam<SRBA_SE3_STEREO>    { using pose_t = Pose3; static constexpr int P = 6, L = 3, O = 4; static constexpr bool relpose = false; };
template <> struct Fam<SRBA_SE3_MONO>      { using pose_t = Pose3; static constexpr int P = 6, L = 3, O = 2; static constexpr bool relpose = false; };
template <> struct Fam<SRBA_SE3_CART3D>    { using pose_t = Pose3; static constexpr int P = 6, L = 3, O = 3; static constexpr bool relpose = false; };
template <> struct Fam<SRBA_SE3_RB3D>      { using pose_t = Pose3; static constexpr int P = 6, L = 3, O = 3; static constexpr bool relpose = false; };
template <> struct Fam<SRBA_SE3_RELPOSE3D> { using pose_t = Pose3; static constexpr int P = 6, L = 6, O = 6; static constexpr bool relpose = true; };
template <> struct Fam<SRBA_SE2_STEREO>    { using pose_t = Pose2; static constexpr int P = 3, L = 3, O = 4; static constexpr bool relpose = false; }; // SE(2) key-frames + 3D points
// every family the oracle is instantiated for
#define ORACLE_ALL_FAMILIES(X) X(SRBA_SE2_RELPOSE2D) X(SRBA_SE2_RB2D) X(SRBA_SE2_CART2D) X(SRBA_SE3_STEREO) X(SRBA_SE3_MONO) X(SRBA_SE3_CART3D) X(SRBA_SE3_RB3D) X(SRBA_SE3_RELPOSE3D) X(SRBA_SE2_STEREO)

// ------------------------------------------------------------------------------------------------
// The optimiser
// ------------------------------------------------------------------------------------------------
template <int FAM>
struct Problem {
	using F = Fam<FAM>; using pose_t = typename F::pose_t;
	static constexpr int P = F::P, L = F::L, O = F::O, PD = pose_t::PD;
	static constexpr bool SE3 = (PD == 12);

	const srba_hip_params &prm;
	srba_problem_capsule &c;
	const int nK, nF, n; // unknown edges, unknown lms, scalars

	// state
	std::vector<pose_t> edge;            // local edge table
	std::vector<double> ulm;             // unknown lm positions
	std::vector<pose_t> pose;            // ST poses (2 per pair)
	std::vector<char> valid;             // validity flags
	std::vector<double> Jp, Jf;          // Jacobian blocks
	std::vector<double> HAp, Hf, HApf;   // Hessian blocks (numeric)
	std::vector<double> HAp_orig;        // Schur: latched original HAp (schur.h:38,167)
	std::vector<double> resid, new_resid;
	std::vector<double> grad, delta;
	std::vector<double> Hf_inv; std::vector<char> Hf_invertible;
	std::vector<double> YW;              // Hpi_lk * inv(Hf_lk) of the diagonal blocks' terms, reused for the gradient (schur.h:104-111)
	std::vector<int> ywt_of_term;        // for a Schur term of a diagonal block: index into YW ; else -1
	pose_t sensor_pose; double RS[9];    // sensor pose on robot
	Pose3 sensor_pose3;                  // the same as an SE(3) pose: what <SE2, Euclidean3D, StereoCamera> composes with (srba_options_sensor_pose.h:101-113 on CPose2D key-frame poses)
	static constexpr bool SE2_3D = (FAM == SRBA_SE2_STEREO);
	double R2L_R[9], R2L_t[3];           // stereo: (-)rightCameraPose

	bool use_schur, dense_chol;
	SparseChol sp; std::vector<int> Cp, Ci; std::vector<double> Cx; std::vector<int> c_slot_of; // sparse system pattern
	int n_sys = 0;

	Problem(const srba_hip_params &p, srba_problem_capsule &cap)
		: prm(p), c(cap), nK(cap.n_unk_edges), nF(cap.n_unk_lms), n(P * cap.n_unk_edges + L * cap.n_unk_lms) {
		edge.resize(c.n_edges); for (int i = 0; i < c.n_edges; i++) edge[i].load(c.edge_pose + (size_t)i * PD);
		ulm.assign(c.ulm_pos, c.ulm_pos + (size_t)nF * L);
		pose.resize(2 * (size_t)c.n_pairs);
		valid.assign(std::max(1, c.n_valid), 1);
		Jp.assign((size_t)c.n_bp * O * P, 0.0); Jf.assign((size_t)c.n_bf * O * L, 0.0);
		HAp.assign((size_t)c.n_hap * P * P, 0.0); Hf.assign((size_t)c.n_hf * L * L, 0.0); HApf.assign((size_t)c.n_hapf * P * L, 0.0);
		resid.assign((size_t)c.n_obs * O, 0.0); grad.assign(n, 0.0); delta.assign(n, 0.0);
		use_schur = (prm.solver != SRBA_SOLVER_NO_SCHUR_SPARSE_CHOL);
		dense_chol = (prm.solver == SRBA_SOLVER_SCHUR_DENSE_CHOL);
		if constexpr (SE3 || SE2_3D) {
			if (prm.sensor_pose == SRBA_SENSOR_POSE_SE3) { sensor_pose3.load(prm.sensor_pose_se3); if constexpr (SE3) sensor_pose.load(prm.sensor_pose_se3); }
			for (int i = 0; i < 9; i++) RS[i] = sensor_pose3.R[i];
			double Rq[9]; quat_to_R(prm.right_cam_pose + 3, Rq); // R2L = (-)rightCameraPose  (sensors.h:193)
			for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R2L_R[3 * i + j] = Rq[3 * j + i];
			for (int i = 0; i < 3; i++) R2L_t[i] = -(Rq[0 + i] * prm.right_cam_pose[0] + Rq[3 + i] * prm.right_cam_pose[1] + Rq[6 + i] * prm.right_cam_pose[2]);
		}
	}

	const double *lm_ptr(int ref) const { return ref >= 0 ? &ulm[(size_t)ref * L] : c.klm_pos + (size_t)(-1 - ref) * L; }

	// ---------------- K1: spantree_update_numeric.h:19-81 ----------------
	int update_spantree(bool only_needed) {
		int cnt = 0;
		for (int p = 0; p < c.n_pairs; p++) {
			cnt++; // the reference counts it->second.size() regardless of skipping (:80)
			if (only_needed && !c.pair_needed[p]) continue; // skip_marked_as_uptodate && both updated (:34-35)
			pose_t acc;
			for (int k = c.pair_path_off[p]; k < c.pair_path_off[p + 1]; k++) {
				const int e = c.path_edge[k] >> 1, inv = c.path_edge[k] & 1;
				acc = inv ? compose(acc, inverse(edge[e])) : compose(acc, edge[e]); // (:51,:59)
			}
			pose[2 * p] = acc; pose[2 * p + 1] = inverse(acc); // (:68-73)
		}
		return cnt;
	}

	// ---------------- sensor models (models/sensors.h) ----------------
	// residual = z - h(pose (+) lm), pose = base wrt SENSOR
	template <class SENSOR_POSE>
	void observe_error(double *r, const double *z, const SENSOR_POSE &base_wrt_sensor, const double *lm) const {
		if constexpr (FAM == SRBA_SE3_RELPOSE3D) { // sensors.h:873-879: h = P(z) (-) pose ; err = pseudo_ln(h)
			const Pose3 h = inv_compose(pose3_from_ypr(z), base_wrt_sensor); pseudo_ln3(h, r);
		} else if constexpr (FAM == SRBA_SE2_STEREO) { // the stereo model of sensors.h:175-211 on the SE(3) pose "base wrt sensor"
			double l[3]; compose_point(base_wrt_sensor, lm, l);
			r[0] = z[0] - (prm.cam_left[2] + prm.cam_left[0] * l[0] / l[2]); r[1] = z[1] - (prm.cam_left[3] + prm.cam_left[1] * l[1] / l[2]);
			double rr[3]; for (int i = 0; i < 3; i++) rr[i] = R2L_t[i] + R2L_R[3 * i] * l[0] + R2L_R[3 * i + 1] * l[1] + R2L_R[3 * i + 2] * l[2];
			r[2] = z[2] - (prm.cam_right[2] + prm.cam_right[0] * rr[0] / rr[2]); r[3] = z[3] - (prm.cam_right[3] + prm.cam_right[1] * rr[1] / rr[2]);
		} else if constexpr (FAM == SRBA_SE2_RELPOSE2D) { // sensors.h:770-785
			Pose2 Z; Z.x = z[0]; Z.y = z[1]; Z.phi = z[2];
			const Pose2 h = inv_compose(Z, base_wrt_sensor); r[0] = h.x; r[1] = h.y; r[2] = h.phi;
		} else if constexpr (FAM == SRBA_SE2_RB2D) { // sensors.h:661-679
			double lx, ly; compose_point(base_wrt_sensor, lm[0], lm[1], lx, ly);
			r[0] = z[0] - std::hypot(lx, ly); r[1] = z[1] - std::atan2(ly, lx);
		} else if constexpr (FAM == SRBA_SE2_CART2D) { // sensors.h:447-463
			double lx, ly; compose_point(base_wrt_sensor, lm[0], lm[1], lx, ly); r[0] = z[0] - lx; r[1] = z[1] - ly;
		} else if constexpr (FAM == SRBA_SE3_CART3D) { // sensors.h:347-362
			double l[3]; compose_point(base_wrt_sensor, lm, l); for (int i = 0; i < 3; i++) r[i] = z[i] - l[i];
		} else if constexpr (FAM == SRBA_SE3_RB3D) { // sensors.h:545-566; [EXT] CPose3D::sphericalCoordinates: range, yaw = atan2(y,x), pitch = -asin(z/range)
			double l[3]; compose_point(base_wrt_sensor, lm, l);
			const double rg = std::sqrt(l[0] * l[0] + l[1] * l[1] + l[2] * l[2]);
			r[0] = z[0] - rg; r[1] = z[1] - std::atan2(l[1], l[0]); r[2] = z[2] - (-std::asin(l[2] / rg));
		} else if constexpr (FAM == SRBA_SE3_MONO) { // sensors.h:50-66
			double l[3]; compose_point(base_wrt_sensor, lm, l);
			r[0] = z[0] - (prm.cam_left[2] + prm.cam_left[0] * l[0] / l[2]); r[1] = z[1] - (prm.cam_left[3] + prm.cam_left[1] * l[1] / l[2]);
		} else if constexpr (FAM == SRBA_SE3_STEREO) { // sensors.h:175-211
			double l[3]; compose_point(base_wrt_sensor, lm, l);
			r[0] = z[0] - (prm.cam_left[2] + prm.cam_left[0] * l[0] / l[2]); r[1] = z[1] - (prm.cam_left[3] + prm.cam_left[1] * l[1] / l[2]);
			double rr[3]; for (int i = 0; i < 3; i++) rr[i] = R2L_t[i] + R2L_R[3 * i] * l[0] + R2L_R[3 * i + 1] * l[1] + R2L_R[3 * i + 2] * l[2];
			r[2] = z[2] - (prm.cam_right[2] + prm.cam_right[0] * rr[0] / rr[2]); r[3] = z[3] - (prm.cam_right[3] + prm.cam_right[1] * rr[1] / rr[2]);
		}
	}
	// dh_dx (O x L) at x = landmark wrt sensor; false => invalid
	bool eval_dh_dx(double *H, const double *x) const {
		if constexpr (FAM == SRBA_SE2_RELPOSE2D || FAM == SRBA_SE2_CART2D || FAM == SRBA_SE3_CART3D || FAM == SRBA_SE3_RELPOSE3D) { // identity (:804-814, :481-490, :380-389, :905-913)
			for (int i = 0; i < O * L; i++) H[i] = 0; for (int i = 0; i < O; i++) H[i * L + i] = 1; return true;
		} else if constexpr (FAM == SRBA_SE2_RB2D) { // sensors.h:698-715
			const double r = std::hypot(x[0], x[1]); if (r == 0) return false;
			const double ri = 1.0 / r, ri2 = ri * ri;
			H[0] = x[0] * ri; H[1] = x[1] * ri; H[2] = -x[1] * ri2; H[3] = x[0] * ri2; return true;
		} else if constexpr (FAM == SRBA_SE3_RB3D) { // sensors.h:588-608; [EXT] Jacobian of sphericalCoordinates wrt the point
			const double x2y2 = x[0] * x[0] + x[1] * x[1], r2 = x2y2 + x[2] * x[2], rg = std::sqrt(r2), rxy = std::sqrt(x2y2);
			H[0] = x[0] / rg; H[1] = x[1] / rg; H[2] = x[2] / rg;
			H[3] = -x[1] / x2y2; H[4] = x[0] / x2y2; H[5] = 0;
			H[6] = x[0] * x[2] / (r2 * rxy); H[7] = x[1] * x[2] / (r2 * rxy); H[8] = -rxy / r2; return true;
		} else if constexpr (FAM == SRBA_SE3_MONO) { // sensors.h:85-110
			if (x[2] <= 0) return false;
			const double zi = 1.0 / x[2], zi2 = zi * zi, fx = prm.cam_left[0], fy = prm.cam_left[1];
			H[0] = fx * zi; H[1] = 0; H[2] = -fx * x[0] * zi2; H[3] = 0; H[4] = fy * zi; H[5] = -fy * x[1] * zi2; return true;
		} else { // stereo sensors.h:230-280
			if (x[2] <= 0) return false;
			{ const double zi = 1.0 / x[2], zi2 = zi * zi, fx = prm.cam_left[0], fy = prm.cam_left[1];
			  H[0] = fx * zi; H[1] = 0; H[2] = -fx * x[0] * zi2; H[3] = 0; H[4] = fy * zi; H[5] = -fy * x[1] * zi2; }
			double xr[3]; for (int i = 0; i < 3; i++) xr[i] = R2L_t[i] + R2L_R[3 * i] * x[0] + R2L_R[3 * i + 1] * x[1] + R2L_R[3 * i + 2] * x[2];
			{ const double zi = 1.0 / xr[2], zi2 = zi * zi, fx = prm.cam_right[0], fy = prm.cam_right[1];
			  H[6] = fx * zi; H[7] = 0; H[8] = -fx * xr[0] * zi2; H[9] = 0; H[10] = fy * zi; H[11] = -fy * xr[1] * zi2; }
			return true;
		}
	}
	// pose wrt robot -> wrt sensor (srba_options_sensor_pose.h:56-59,110-113)
	pose_t pose_robot2sensor(const pose_t &p) const {
		if constexpr (SE3) { if (prm.sensor_pose == SRBA_SENSOR_POSE_SE3) return inv_compose(p, sensor_pose); }
		return p;
	}
	void point_robot2sensor(double *x) const { // (:63-66,:117-120)
		if constexpr (SE3 || SE2_3D) if (prm.sensor_pose == SRBA_SENSOR_POSE_SE3) { double l[3]; inv_compose_point(sensor_pose3, x, l); x[0] = l[0]; x[1] = l[1]; x[2] = l[2]; }
	}
	void dh_dx_rotate(double *H) const { // dh_dx = dh_dx * R_S^t (:124-128)
		if constexpr (SE3 || SE2_3D) if (prm.sensor_pose == SRBA_SENSOR_POSE_SE3) {
			double T[O * 3];
			for (int i = 0; i < O; i++) for (int j = 0; j < 3; j++) T[i * 3 + j] = H[i * 3 + 0] * RS[3 * j + 0] + H[i * 3 + 1] * RS[3 * j + 1] + H[i * 3 + 2] * RS[3 * j + 2];
			for (int i = 0; i < O * 3; i++) H[i] = T[i];
		}
	}

	// ---------------- K4: reprojection_residuals.h:16-81 ----------------
	double residuals(std::vector<double> &res) const {
		res.resize((size_t)c.n_obs * O);
		double total = 0;
		for (int i = 0; i < c.n_obs; i++) {
			pose_t bp; if (c.obs_pose[i] >= 0) bp = pose[c.obs_pose[i]]; // else identity (aux_null_pose :36-39)
			double *r = &res[(size_t)i * O];
			if constexpr (SE2_3D) { // pose_robot2sensor with a CPose2D key-frame pose gives an SE(3) pose (srba_options_sensor_pose.h:108-113)
				Pose3 bs = pose3_from_pose2(bp); if (prm.sensor_pose == SRBA_SENSOR_POSE_SE3) bs = inv_compose(bs, sensor_pose3);
				observe_error(r, c.obs_z + (size_t)i * O, bs, lm_ptr(c.obs_lm[i]));
			} else { const pose_t bs = pose_robot2sensor(bp); observe_error(r, c.obs_z + (size_t)i * O, bs, lm_ptr(c.obs_lm[i])); }
			double sum2 = 0; for (int k = 0; k < O; k++) sum2 += r[k] * r[k];
			if (prm.use_robust_kernel) { // :67-72, huber RbaEngine.h:810-813
				const double nrm = std::max(1e-11, std::sqrt(sum2));
				const double kp = prm.kernel_param, q = nrm / kp;
				const double hub = std::fabs(2 * kp * kp * (std::sqrt(1 + q * q) - 1));
				const double w = std::sqrt(hub) / nrm;
				for (int k = 0; k < O; k++) r[k] *= w;
				total += (w * w) * sum2;
			} else total += sum2;
		}
		return total;
	}

	// ---------------- K2: jacobians.h:207-354 + families ----------------
	void jacobian_dh_dp(int b) {
		const int vs = c.obs_valid[c.bp_res[b]];
		if (!valid[vs]) return; // :215-216 (keeps stale values)
		double *J = &Jp[(size_t)b * O * P];
		const bool hasA = c.bp_A[b] >= 0, inverse_edge = !c.bp_normal[b];
		const pose_t D = pose[c.bp_D[b]];
		pose_t A; if (hasA) A = pose[c.bp_A[b]];
		const pose_t i_wrt_l = hasA ? compose(A, D) : D; // :259-262
		const double *xji_i = lm_ptr(c.bp_lm[b]);
		double xl[L]; for (int k = 0; k < L; k++) xl[k] = xji_i[k];
		// xji_l = pose_i_wrt_l (+) xji_i (:269-270; not applicable to relative-pose landmarks, landmarks.h:112-116)
		if constexpr (!F::relpose) {
			if constexpr (SE3) { double g[3]; compose_point(i_wrt_l, xl, g); xl[0] = g[0]; xl[1] = g[1]; xl[2] = g[2]; }
			else { double gx, gy; compose_point(i_wrt_l, xl[0], xl[1], gx, gy); xl[0] = gx; xl[1] = gy; } // a 2D pose leaves z untouched (landmarks.h Euclidean3D::composePosePoint)
			point_robot2sensor(xl); // :318
		}
		double dh_dx[O * L];
		if (!eval_dh_dx(dh_dx, xl)) { valid[vs] = 0; for (int k = 0; k < O * P; k++) J[k] = 0; return; } // :321-327
		if constexpr (!F::relpose) dh_dx_rotate(dh_dx); // :330
		const pose_t &pe = edge[c.bp_col[b]]; // the edge's own inv_pose (unknown slot == local edge index)
		if constexpr (FAM == SRBA_SE2_RELPOSE2D) { // jacobians.h:645-744
			double Xd, Yd, PHIa; Pose2 ad;
			if (!inverse_edge) { Xd = D.x; Yd = D.y; PHIa = hasA ? A.phi : 0.0; ad = i_wrt_l; }
			else {
				const Pose2 Dp = compose(pe, D); const Pose2 pinv = inverse(pe);
				const Pose2 Ap = hasA ? compose(A, pinv) : pinv;
				Xd = Dp.x; Yd = Dp.y; PHIa = Ap.phi; ad = compose(Ap, Dp);
			}
			const double cad = std::cos(ad.phi), sad = std::sin(ad.phi), ca = std::cos(PHIa), sa = std::sin(PHIa);
			const double J0[9] = {cad, sad, 0, -sad, cad, 0, 0, 0, 1};
			const double J1[9] = {1, 0, -Xd * sa - Yd * ca, 0, 1, Xd * ca - Yd * sa, 0, 0, 1};
			const double J2[9] = {ca, -sa, 0, sa, ca, 0, 0, 0, 1};
			double T0[9], T1[9]; mm<3, 3, 3>(dh_dx, J0, T0); mm<3, 3, 3>(T0, J1, T1); mm<3, 3, 3>(T1, J2, J);
			if (inverse_edge) for (int k = 0; k < 9; k++) J[k] = -J[k];
			if (g_exact_relpose_jacobian) { // DIAGNOSTIC ONLY (SRBA_ORACLE_EXACT_JAC=1): J = -dr/deps by central differences, to study App. B-13
				const double *z = c.obs_z + (size_t)c.bp_res[b] * O; Pose2 Z; Z.x = z[0]; Z.y = z[1]; Z.phi = z[2];
				const Pose2 Drest = inverse_edge ? compose(pe, D) : D;
				for (int dd = 0; dd < 3; dd++) {
					double rr[2][3];
					for (int sgn = 0; sgn < 2; sgn++) {
						double e[3] = {0, 0, 0}; e[dd] = sgn ? 1e-6 : -1e-6;
						Pose2 T;
						if (!inverse_edge) { T = compose(pseudo_exp2(e), D); if (hasA) T = compose(A, T); }
						else { const Pose2 pe2 = compose(pseudo_exp2(e), pe); T = compose(inverse(pe2), Drest); if (hasA) T = compose(A, T); }
						const Pose2 h = inv_compose(Z, T); rr[sgn][0] = h.x; rr[sgn][1] = h.y; rr[sgn][2] = wrapToPi(h.phi);
					}
					for (int k = 0; k < 3; k++) J[k * 3 + dd] = -(rr[1][k] - rr[0][k]) / 2e-6;
				}
			}
		} else if constexpr (FAM == SRBA_SE3_RELPOSE3D) { // jacobians.h:748-873, verbatim: jacob = dLnRelPose_deps (6x12) * dAeD_de (12x6)
			Pose3 Dd, ad; double ROTA[9];
			if (!inverse_edge) { Dd = D; if (hasA) { ad = compose(A, Dd); for (int k = 0; k < 9; k++) ROTA[k] = A.R[k]; } else { ad = Dd; for (int k = 0; k < 9;
				k++) ROTA[k] = (k % 4 == 0) ? 1.0 : 0.0; } }
			else {
				const Pose3 Dp = compose(pe, D), pinv = inverse(pe); const Pose3 Ap = hasA ? compose(A, pinv) : pinv;
				for (int k = 0; k < 9; k++) ROTA[k] = Ap.R[k]; Dd = Dp; ad = compose(Ap, Dp);
			}
			double dLn[6 * 12]; for (int k = 0; k < 72; k++) dLn[k] = 0;
			for (int k = 0; k < 3; k++) dLn[k * 12 + 9 + k] = 1; // block<3,3>(0,9) = I
			{ double M[27]; ln_rot_jacob(ad.R, M); for (int r = 0; r < 3; r++) for (int q = 0; q < 9; q++) dLn[(3 + r) * 12 + q] = M[r * 9 + q]; } // block<3,9>(3,0)
			double dAeD[12 * 6]; for (int k = 0; k < 72; k++) dAeD[k] = 0;
			{ double RD[9]; mm<3, 3, 3>(ROTA, Dd.R, RD); for (int r = 0; r < 3; r++) for (int q = 0; q < 3; q++) dAeD[(9 + r) * 6 + q] = RD[3 * q + r]; } // block<3,3>(9,0) = (ROTA * R(D))^t (:842)
			for (int i = 0; i < 4; i++) { // blocks (i,2) = -ROTA * [column i of the homogeneous matrix of D]_x (:848-856)
				const double h0 = i < 3 ? Dd.R[i] : Dd.t[0], h1 = i < 3 ? Dd.R[3 + i] : Dd.t[1], h2 = i < 3 ? Dd.R[6 + i] : Dd.t[2];
				const double aux[9] = {0, -h2, h1, h2, 0, -h0, -h1, h0, 0}; double G[9]; mm<3, 3, 3>(ROTA, aux, G);
				for (int r = 0; r < 3; r++) for (int q = 0; q < 3; q++) dAeD[(3 * i + r) * 6 + 3 + q] = -G[3 * r + q];
			}
			mm<6, 12, 6>(dLn, dAeD, J);
			if (inverse_edge) for (int k = 0; k < 36; k++) J[k] = -J[k];
		} else if constexpr (FAM == SRBA_SE2_STEREO) { // SE2 + 3D points, jacobians.h:501-641 with POINT_DIMS = 3
			double Xd, Yd, PHIa; Pose2 AD;
			if (!inverse_edge) { Xd = D.x; Yd = D.y; PHIa = hasA ? A.phi : 0.0; AD = i_wrt_l; }
			else {
				const Pose2 Dp = compose(pe, D); const Pose2 pinv = inverse(pe);
				const Pose2 Ap = hasA ? compose(A, pinv) : pinv;
				AD = compose(Ap, Dp); Xd = Dp.x; Yd = Dp.y; PHIa = Ap.phi;
			}
			const double cad = std::cos(AD.phi), sad = std::sin(AD.phi);
			const double dPx[9] = {1, 0, -xji_i[0] * sad - xji_i[1] * cad, 0, 1, xji_i[0] * cad - xji_i[1] * sad, 0, 0, 1}; // (2,2) = 1 as in the reference (:546)
			const double ca = std::cos(PHIa), sa = std::sin(PHIa);
			const double dAD[9] = {ca, -sa, -sa * Xd - ca * Yd, sa, ca, ca * Xd - sa * Yd, 0, 0, 1};
			double T[O * 3]; mm<O, 3, 3>(dh_dx, dPx, T); mm<O, 3, 3>(T, dAD, J);
			if (inverse_edge) for (int k = 0; k < O * P; k++) J[k] = -J[k];
		} else if constexpr (!SE3) { // SE2 + 2D points, jacobians.h:501-634
			double Xd, Yd, PHIa; Pose2 AD;
			if (!inverse_edge) { Xd = D.x; Yd = D.y; PHIa = hasA ? A.phi : 0.0; AD = i_wrt_l; }
			else {
				const Pose2 Dp = compose(pe, D); const Pose2 pinv = inverse(pe);
				const Pose2 Ap = hasA ? compose(A, pinv) : pinv;
				AD = compose(Ap, Dp); Xd = Dp.x; Yd = Dp.y; PHIa = Ap.phi;
			}
			const double cad = std::cos(AD.phi), sad = std::sin(AD.phi);
			const double dPx[6] = {1, 0, -xji_i[0] * sad - xji_i[1] * cad, 0, 1, xji_i[0] * cad - xji_i[1] * sad};
			const double ca = std::cos(PHIa), sa = std::sin(PHIa);
			const double dAD[9] = {ca, -sa, -sa * Xd - ca * Yd, sa, ca, ca * Xd - sa * Yd, 0, 0, 1};
			double T[O * 3]; mm<O, 2, 3>(dh_dx, dPx, T); mm<O, 3, 3>(T, dAD, J);
			if (inverse_edge) for (int k = 0; k < O * P; k++) J[k] = -J[k];
		} else { // SE3 + 3D points, jacobians.h:364-494
			pose_t Dd = D; double RA[9]; bool haveRA = hasA;
			if (hasA) for (int k = 0; k < 9; k++) RA[k] = A.R[k];
			if (inverse_edge) {
				Dd = compose(pe, D); // D' (:438-439)
				// R(A') = R(A) * R(p)^t (:453,:460)
				double Rt[9]; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Rt[3 * i + j] = pe.R[3 * j + i];
				if (hasA) { double T[9]; mm<3, 3, 3>(A.R, Rt, T); for (int k = 0; k < 9; k++) RA[k] = T[k]; }
				else { for (int k = 0; k < 9; k++) RA[k] = Rt[k]; haveRA = true; }
			}
			double HR[O * 3];
			if (haveRA) mm<O, 3, 3>(dh_dx, RA, HR); else for (int k = 0; k < O * 3; k++) HR[k] = dh_dx[k];
			double v[3];
			for (int i = 0; i < 3; i++) v[i] = -Dd.t[i] - xji_i[0] * Dd.R[3 * i] - xji_i[1] * Dd.R[3 * i + 1] - xji_i[2] * Dd.R[3 * i + 2]; // :410-412
			const double aux[9] = {0, -v[2], v[1], v[2], 0, -v[0], -v[1], v[0], 0}; // :416-425
			double HRa[O * 3]; mm<O, 3, 3>(HR, aux, HRa);
			for (int i = 0; i < O; i++) for (int j = 0; j < 3; j++) { J[i * 6 + j] = HR[i * 3 + j]; J[i * 6 + 3 + j] = HRa[i * 3 + j]; }
			if (inverse_edge) for (int k = 0; k < O * P; k++) J[k] = -J[k];
		}
	}

	// ---------------- K3: jacobians.h:888-1013 ----------------
	void jacobian_dh_df(int b) {
		if constexpr (F::relpose) return; else {
			const int vs = c.obs_valid[c.bf_res[b]];
			if (!valid[vs]) return; // :894-895
			double *J = &Jf[(size_t)b * O * L];
			const bool hasP = c.bf_pose[b] >= 0;
			pose_t bp; if (hasP) bp = pose[c.bf_pose[b]];
			const double *xji_i = &ulm[(size_t)c.bf_col[b] * L];
			double xl[L]; for (int k = 0; k < L; k++) xl[k] = xji_i[k];
			if (hasP) {
				if constexpr (SE3) { double g[3]; compose_point(bp, xl, g); for (int k = 0; k < 3; k++) xl[k] = g[k]; }
				else { double gx, gy; compose_point(bp, xl[0], xl[1], gx, gy); xl[0] = gx; xl[1] = gy; }
			}
			point_robot2sensor(xl); // :968
			double dh_dx[O * L];
			if (!eval_dh_dx(dh_dx, xl)) { valid[vs] = 0; for (int k = 0; k < O * L; k++) J[k] = 0; return; } // :971-977
			dh_dx_rotate(dh_dx); // :980
			if (hasP) { // J = dh_dx * R(base<-obs) (:984-989)
				if constexpr (SE3) mm<O, 3, 3>(dh_dx, bp.R, J);
				else if constexpr (SE2_3D) { const double cc = std::cos(bp.phi), ss = std::sin(bp.phi); const double R[9] = {cc, -ss, 0, ss, cc, 0, 0, 0, 1}; mm<O, 3, 3>(dh_dx, R, J); }
					// [EXT] CPose2D::getRotationMatrix into a 3x3
				else { const double cc = std::cos(bp.phi), ss = std::sin(bp.phi); const double R[4] = {cc, -ss, ss, cc}; mm<O, 2, 2>(dh_dx, R, J); }
			} else for (int k = 0; k < O * L; k++) J[k] = dh_dx[k];
		}
	}
	int recompute_all_jacobians() { // jacobians.h:1083-1117 (sweep order = block order of the capsule)
		for (int b = 0; b < c.n_bp; b++) jacobian_dh_dp(b);
		for (int b = 0; b < c.n_bf; b++) jacobian_dh_df(b);
		return c.n_bp + c.n_bf;
	}

	// ---------------- K6: sparse_hessian_update_numeric.h:22-60 + noise policies ----------------
	template <int M1, int M2>
	int hessian_blocks(int nblk, const int *off, const int *t1, const int *t2, const double *J1, const double *J2,
	                   const int *res1, const int *res2, std::vector<double> &H) {
		int nInvalid = 0;
		for (int b = 0; b < nblk; b++) {
			double Hij[M1 * M2]; for (int k = 0; k < M1 * M2; k++) Hij[k] = 0;
			for (int t = off[b]; t < off[b + 1]; t++) {
				const double *A = J1 + (size_t)t1[t] * O * M1, *B = J2 + (size_t)t2[t] * O * M2;
				if (valid[c.obs_valid[res1[t1[t]]]] && valid[c.obs_valid[res2[t2[t]]]]) {
					if (prm.noise == SRBA_NOISE_CONSTANT_MATRIX) { // H += (J1^t Lambda) J2  (srba_options_noise.h:102-108)
						double JtL[M1 * O];
						for (int i = 0; i < M1; i++) for (int j = 0; j < O; j++) { double s = 0; for (int k = 0; k < O; k++) s += A[k * M1 + i] * prm.lambda[k * O + j]; JtL[i * O + j] = s; }
						for (int i = 0; i < M1; i++) for (int j = 0; j < M2; j++) { double s = 0; for (int k = 0; k < O; k++) s += JtL[i * O + k] * B[k * M2 + j]; Hij[i * M2 + j] += s; }
					} else mtm_acc<O, M1, M2>(A, B, Hij); // :44-49
				} else nInvalid++;
			}
			if (prm.noise == SRBA_NOISE_IDENTITY) { const double s = 1.0 / prm.std_noise_observations; for (int k = 0; k < M1 * M2; k++) Hij[k] *= s; } // scale_H :52-56 (sic)
			for (int k = 0; k < M1 * M2; k++) H[(size_t)b * M1 * M2 + k] = Hij[k];
		}
		return nInvalid;
	}
	int hessian_update_numeric() {
		int inv = 0;
		inv += hessian_blocks<P, P>(c.n_hap, c.hap_term_off, c.hap_t1, c.hap_t2, Jp.data(), Jp.data(), c.bp_res, c.bp_res, HAp);
		inv += hessian_blocks<L, L>(c.n_hf, c.hf_term_off, c.hf_t1, c.hf_t2, Jf.data(), Jf.data(), c.bf_res, c.bf_res, Hf);
		inv += hessian_blocks<P, L>(c.n_hapf, c.hapf_term_off, c.hapf_t1, c.hapf_t2, Jp.data(), Jf.data(), c.bp_res, c.bf_res, HApf);
		return inv;
	}

	// ---------------- K5: compute_minus_gradient.h:20-91 ----------------
	template <int M> void grad_cols(int ncols, const int *coff, const double *J, const int *res, double *g) {
		for (int i = 0; i < ncols; i++) {
			double acc[M]; for (int k = 0; k < M; k++) acc[k] = 0;
			for (int b = coff[i]; b < coff[i + 1]; b++) {
				const double *A = J + (size_t)b * O * M, *r = &resid[(size_t)res[b] * O];
				if (prm.noise == SRBA_NOISE_CONSTANT_MATRIX) { // g += (J^t Lambda) r (:117-123)
					for (int ii = 0; ii < M; ii++) { double s = 0; for (int j = 0; j < O; j++) { double jl = 0; for (int k = 0; k < O; k++) jl += A[k * M + ii] * prm.lambda[k * O + j];
						s += jl * r[j]; } acc[ii] += s; }
				} else for (int ii = 0; ii < M; ii++) { double s = 0; for (int k = 0; k < O; k++) s += A[k * M + ii] * r[k]; acc[ii] += s; }
			}
			if (prm.noise == SRBA_NOISE_IDENTITY) { const double s = 1.0 / prm.std_noise_observations; for (int k = 0; k < M; k++) acc[k] *= s; } // scale_Jtr :67-71
			for (int k = 0; k < M; k++) g[i * M + k] = acc[k];
		}
	}
	std::vector<double> grad_pristine; // SRBA_EXT_SCHUR_KEEPS_GRADIENT: what compute_minus_gradient produced (the Schur reduction mutates `grad`)
	void compute_minus_gradient() {
		grad_cols<P>(nK, c.colp_off, Jp.data(), c.bp_res, grad.data());
		if (nF) grad_cols<L>(nF, c.colf_off, Jf.data(), c.bf_res, grad.data() + (size_t)P * nK);
		if (prm.extensions & SRBA_EXT_SCHUR_KEEPS_GRADIENT) grad_pristine = grad;
	}

	// ---------------- Schur: schur.h ----------------
	bool schur_active() const { return use_schur && nF > 0 && nK > 0; } // schur.h:34,182
	void schur_ctor() {
		HAp_orig = HAp; // :38
		Hf_inv.assign((size_t)nF * L * L, 0.0); Hf_invertible.assign(nF, 0);
		ywt_of_term.assign(c.n_sch_terms, -1); int cnt = 0;
		for (int b = 0; b < c.n_hap; b++) if (c.hap_i[b] == c.hap_j[b]) for (int t = c.sch_term_off[b]; t < c.sch_term_off[b + 1]; t++) ywt_of_term[t] = cnt++;
		YW.assign((size_t)cnt * P * L, 0.0);
	}
	void schur_build_reduced(double lambda) { // :180-268
		HAp = HAp_orig; // :188
		for (int i = 0; i < nF; i++) { // :193-208
			double Hfi[L * L]; const double *src = &Hf[(size_t)c.hf_diag[i] * L * L];
			for (int k = 0; k < L * L; k++) Hfi[k] = src[k];
			for (int k = 0; k < L; k++) Hfi[k * L + k] += lambda;
			double inv[L * L];
			Hf_invertible[i] = fullpivlu_inverse<L>(Hfi, inv) ? 1 : 0;
			if (Hf_invertible[i]) for (int k = 0; k < L * L; k++) Hf_inv[(size_t)i * L * L + k] = inv[k];
		}
		for (int b = 0; b < c.n_hap; b++) { // :213-244, iteration order = (col, row) order of HAp blocks
			double *Hij = &HAp[(size_t)b * P * P];
			for (int t = c.sch_term_off[b]; t < c.sch_term_off[b + 1]; t++) {
				const int l = c.sch_lm[t]; if (!Hf_invertible[l]) continue;
				const double *W1 = &HApf[(size_t)c.sch_b1[t] * P * L], *W2 = &HApf[(size_t)c.sch_b2[t] * P * L];
				double Ytmp[P * L]; double *Y = ywt_of_term[t] >= 0 ? &YW[(size_t)ywt_of_term[t] * P * L] : Ytmp;
				mm<P, L, L>(W1, &Hf_inv[(size_t)l * L * L], Y); // :237
				for (int i = 0; i < P; i++) for (int j = 0; j < P; j++) { double s = 0; for (int k = 0; k < L; k++) s += Y[i * L + k] * W2[j * L + k]; Hij[i * P + j] -= s; } // :240
			}
		}
		double *gf = grad.data() + (size_t)P * nK;
		for (int i = 0; i < nK; i++) { // :248-265  (IN PLACE on minus_grad: App. B-3)
			const int b = c.hap_diag[i];
			for (int t = c.sch_term_off[b]; t < c.sch_term_off[b + 1]; t++) {
				const int l = c.sch_lm[t]; if (!Hf_invertible[l]) continue;
				const double *Y = &YW[(size_t)ywt_of_term[t] * P * L];
				for (int r = 0; r < P; r++) { double s = 0; for (int k = 0; k < L; k++) s += Y[r * L + k] * gf[l * L + k]; grad[i * P + r] -= s; }
			}
		}
	}
	void schur_solve_features() { // :271-311
		double *gf = grad.data() + (size_t)P * nK; double *df = delta.data() + (size_t)P * nK;
		for (int b = 0; b < c.n_hapf; b++) { // blocks ordered by (edge, lm) = the reference's nested loops
			const int i = c.hapf_i[b], l = c.hapf_j[b]; if (!Hf_invertible[l]) continue;
			const double *W = &HApf[(size_t)b * P * L];
			for (int k = 0; k < L; k++) { double s = 0; for (int r = 0; r < P; r++) s += W[r * L + k] * delta[i * P + r]; gf[l * L + k] -= s; } // :294
		}
		for (int l = 0; l < nF; l++) { if (!Hf_invertible[l]) continue;
			for (int r = 0; r < L; r++) { double s = 0; for (int k = 0; k < L; k++) s += Hf_inv[(size_t)l * L * L + r * L + k] * gf[l * L + k]; df[l * L + r] = s; } } // :308
	}

	// ---------------- K9: lev-marq_solvers.h ----------------
	void sparse_setup() { // symbolic part of CholeskyDecomp ctor (cs_schol), once per optimize_edges call (:164-166)
		struct stopwatch { std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now(); ~stopwatch() {
			g_symbolic_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); } } sw; // bench.py reports the CPU rate with and without this setup
		const bool full = !use_schur; // full system vs HAp only
		n_sys = full ? n : P * nK;
		const int nb = full ? nK + nF : nK;
		std::vector<std::pair<int, int>> be;
		for (int b = 0; b < c.n_hap; b++) be.push_back({c.hap_i[b], c.hap_j[b]});
		if (full) { for (int b = 0; b < c.n_hapf; b++) be.push_back({c.hapf_i[b], nK + c.hapf_j[b]}); for (int b = 0; b < c.n_hf; b++) be.push_back({nK + c.hf_i[b], nK + c.hf_j[b]}); }
		const std::vector<int> bo = min_degree_order(nb, be);
		auto bstart = [&](int blk) { return blk < nK ? P * blk : P * nK + L * (blk - nK); };
		auto bsize = [&](int blk) { return blk < nK ? P : L; };
		sp = SparseChol(); sp.n = n_sys; sp.perm.clear();
		for (int v : bo) for (int k = 0; k < bsize(v); k++) sp.perm.push_back(bstart(v) + k);
		sp.pinv.assign(n_sys, 0); for (int k = 0; k < n_sys; k++) sp.pinv[sp.perm[k]] = k;
		// pattern of C = P A P^t upper: collect (row,col) scalar entries
		std::vector<std::vector<int>> cols(n_sys);
		auto add_block = [&](int r0, int c0, int nr, int nc, bool diag) {
			for (int i = 0; i < nr; i++) for (int j = 0; j < nc; j++) {
				if (diag && i > j) continue; // diagonal blocks: upper part only (triplet of full block is symmetric; CSparse chol uses upper)
				int a = sp.pinv[r0 + i], b = sp.pinv[c0 + j]; if (a > b) std::swap(a, b); cols[b].push_back(a);
			}
		};
		for (int b = 0; b < c.n_hap; b++) add_block(P * c.hap_i[b], P * c.hap_j[b], P, P, c.hap_i[b] == c.hap_j[b]);
		if (full) {
			for (int b = 0; b < c.n_hapf; b++) add_block(P * c.hapf_i[b], P * nK + L * c.hapf_j[b], P, L, false);
			for (int b = 0; b < c.n_hf; b++) add_block(P * nK + L * c.hf_i[b], P * nK + L * c.hf_j[b], L, L, c.hf_i[b] == c.hf_j[b]);
		}
		Cp.assign(n_sys + 1, 0); Ci.clear();
		for (int j = 0; j < n_sys; j++) { std::sort(cols[j].begin(), cols[j].end()); cols[j].erase(std::unique(cols[j].begin(), cols[j].end()), cols[j].end());
			Cp[j + 1] = Cp[j] + (int)cols[j].size(); Ci.insert(Ci.end(), cols[j].begin(), cols[j].end()); }
		Cx.assign(Ci.size(), 0.0);
	}
	void sparse_fill(double lambda) { // SparseTripletFill + compressFromTriplet (:88-156 / :303-332)
		std::fill(Cx.begin(), Cx.end(), 0.0);
		auto put = [&](int r, int col, double v) {
			int a = sp.pinv[r], b = sp.pinv[col]; if (a > b) std::swap(a, b);
			const int *lo = &Ci[Cp[b]], *hi = &Ci[Cp[b + 1]]; const int *it = std::lower_bound(lo, hi, a); Cx[it - &Ci[0]] += v; // triplet duplicates are summed by cs_compress/cs_dupl
		};
		auto put_block = [&](int r0, int c0, int nr, int nc, const double *M, bool diag) {
			for (int i = 0; i < nr; i++) for (int j = 0; j < nc; j++) { if (diag && i > j) continue; put(r0 + i, c0 + j, M[i * nc + j] + ((diag && i == j) ? lambda : 0.0)); }
		};
		for (int b = 0; b < c.n_hap; b++) put_block(P * c.hap_i[b], P * c.hap_j[b], P, P, &HAp[(size_t)b * P * P], c.hap_i[b] == c.hap_j[b]);
		if (!use_schur) {
			for (int b = 0; b < c.n_hapf; b++) put_block(P * c.hapf_i[b], P * nK + L * c.hapf_j[b], P, L, &HApf[(size_t)b * P * L], false);
			for (int b = 0; b < c.n_hf; b++) put_block(P * nK + L * c.hf_i[b], P * nK + L * c.hf_j[b], L, L, &Hf[(size_t)b * L * L], c.hf_i[b] == c.hf_j[b]);
		}
	}
	std::vector<double> denseU;
	bool solve(double lambda) {
		if (schur_active() && (prm.extensions & SRBA_EXT_SCHUR_KEEPS_GRADIENT)) grad = grad_pristine; // (extension, default off: the reference keeps reducing the already reduced vector, App. B-3)
		if (schur_active()) schur_build_reduced(lambda); // :286 / :481
		if (use_schur && dense_chol) { // :489-551 (denseChol_is_uptodate is false at every call, see optimize_edges.h:656,689)
			const int m = P * nK; denseU.assign((size_t)m * m, 0.0);
			for (int b = 0; b < c.n_hap; b++) { const int i = c.hap_i[b], j = c.hap_j[b]; const double *M = &HAp[(size_t)b * P * P];
				for (int r = 0; r < P; r++) for (int q = 0; q < P; q++) denseU[(size_t)(P * i + r) * m + P * j + q] = M[r * P + q] + ((i == j && r == q) ? lambda : 0.0); }
			if (!dense_llt_upper(denseU, m)) return false; // :530-534
			std::fill(delta.begin() + m, delta.end(), 0.0); // :547
			dense_llt_solve(denseU, m, grad.data(), delta.data()); // :549
		} else {
			sparse_fill(lambda);
			if (!sp.factor(Cp, Ci, Cx)) return false; // CExceptionNotDefPos (:172-176 / :347-352)
			std::fill(delta.begin(), delta.end(), 0.0); // :360
			sp.solve(grad.data(), delta.data()); // :183 / :362
		}
		if (schur_active()) schur_solve_features(); // :372 / :559
		return true;
	}

	// ---------------- the driver: optimize_edges.h S5..S17 ----------------
	// Decision replay (test infrastructure of the test infrastructure: tests/test_gpu_parity.py, tools/soak_parity.py): instead of deciding not-PD / reject / accept itself the loop
	// takes the sequence another run took (the GPU's, read from its trial trace) and reports, per trial, what it would have decided: its own rho, the chi2 of the trial point and the chi2
	// it started from. Two runs that part at a rounding-floor decision can then be compared over their WHOLE length instead of over the common prefix (optimize_edges.h:579-656).
	struct Replay { const int32_t *dec; int n; double *rho, *chi2, *E; int32_t *flags; int diverged_at; }; // dec: 0 not PD, 1 rejected, 2 accepted; flags: 1 own factorisation PD,
		// 2 own rho sign differs from the decision, 4 forced not-PD although PD
	void run(srba_lm_result &out, Replay *rp = nullptr) {
		std::memset(&out, 0, sizeof(out));
		for (int k = 0; k < SRBA_TRACE_LEN; k++) { out.trace_chi2[k] = out.trace_lambda[k] = out.trace_rho[k] = std::numeric_limits<double>::quiet_NaN(); }
		out.lambda_last_trial = std::numeric_limits<double>::quiet_NaN();
		const int nObs = c.n_obs;
		out.num_observations = nObs;
		out.num_span_tree_numeric_updates = update_spantree(false); // S5 :256
		std::fill(valid.begin(), valid.end(), 1); // S6 :269-270
		out.num_jacobians = recompute_all_jacobians(); // S7 :276
		out.num_invalid_jacobs = hessian_update_numeric(); // S10 :329-331
		if ((long)O * nObs < (long)n) { out.status = 1; return; } // S11 :355 (ASSERT_ABOVEEQ_ throws in the reference)
		double nu = 2, lambda; // S12 :361-390
		{
			double mx = 0;
			for (int i = 0; i < nK; i++) { const double *H = &HAp[(size_t)c.hap_diag[i] * P * P]; double m = H[0]; for (int k = 1; k < P; k++) m = std::max(m, H[k * P + k]); mx = std::max(mx, m); }
			for (int i = 0; i < nF; i++) { const double *H = &Hf[(size_t)c.hf_diag[i] * L * L]; double m = H[0]; for (int k = 1; k < L; k++) m = std::max(m, H[k * L + k]); mx = std::max(mx, m); }
			lambda = 1e-3 * mx;
		}
		out.lambda_init = lambda;
		double total_err = residuals(resid); // S13 :398-404
		double RMSE = std::sqrt(total_err / nObs);
		out.total_sqr_error_init = total_err;
		compute_minus_gradient(); // S14 :425
		if (schur_active()) schur_ctor(); // S15 :431-436
		if (!(use_schur && dense_chol)) sparse_setup();
		const double MAX_LAMBDA = prm.max_lambda;
		std::vector<pose_t> old_edges(nK), old_poses; std::vector<double> old_ulm;
		std::vector<int> req; for (int i = 0; i < 2 * c.n_pairs; i++) if (c.pose_required[i] || g_refresh_all) req.push_back(i);
		old_poses.resize(req.size());
		int iter; bool stop = false; int trials = 0;
		double rho = 0;
		// one pass of the inner while (optimize_edges.h:471-692); forced < 0: the loop's own decisions (the reference), else the replayed one. Returns false when a replay cannot be followed.
		auto trial = [&](int forced) -> bool {
				const int tr = trials++;
				if (tr < SRBA_TRACE_LEN) out.trace_lambda[tr] = lambda;
				out.lambda_last_trial = lambda;
				const bool solved = solve(lambda);
				if (rp) { rp->flags[tr] = solved ? 1 : 0; rp->rho[tr] = rp->chi2[tr] = std::numeric_limits<double>::quiet_NaN(); rp->E[tr] = total_err; }
				if (forced > 0 && !solved) { rp->diverged_at = tr; trials--; return false; } // the other run solved a system this one calls not positive definite: nothing to follow
				if (!solved || forced == 0) { // :476-485
					if (solved) rp->flags[tr] |= 4;
					out.num_not_pd++;
					lambda *= nu; nu *= 2.; stop = (lambda > MAX_LAMBDA); if (stop) out.stop_reason |= 1 << SRBA_STOP_LAMBDA;
					return true;
				}
				for (int i = 0; i < nK; i++) old_edges[i] = edge[i]; // :491-495
				old_ulm = ulm; // :497-501
				for (int i = 0; i < nK; i++) { // :508-527  new = exp(delta) (+) old
					if constexpr (SE3) edge[i] = compose(pseudo_exp3(&delta[(size_t)i * P]), edge[i]);
					else edge[i] = compose(pseudo_exp2(&delta[(size_t)i * P]), edge[i]);
				}
				for (int i = 0; i < nF * L; i++) ulm[i] += delta[(size_t)P * nK + i]; // :534-539
				for (size_t i = 0; i < req.size(); i++) old_poses[i] = pose[req[i]]; // :550-557
				update_spantree(!g_refresh_all); // :562-565
				const double new_err = residuals(new_resid); // :573-577
				const double new_RMSE = std::sqrt(new_err / nObs);
				const double err_red = total_err > 0 ? (total_err - new_err) / total_err : 0; // :581
				double den = 0; for (int k = 0; k < n; k++) den += delta[k] * (lambda * delta[k] + grad[k]);
				rho = (total_err - new_err) / den; // :585
				if (tr < SRBA_TRACE_LEN) { out.trace_chi2[tr] = new_err; out.trace_rho[tr] = rho; }
				const bool own_accept = rho > 0, accept = forced < 0 ? own_accept : forced == 2;
				if (rp) { rp->rho[tr] = rho; rp->chi2[tr] = new_err; if (own_accept != accept) rp->flags[tr] |= 2; }
				if (accept) { // :587
					out.num_accepted++;
					// (a replayed acceptance of a step this run would have rejected is a step that does not reduce the error here: the run that accepted it saw a reduction below its rounding,
						// far below the relinearisation threshold)
					const bool relin = (forced >= 0 && !own_accept) ? false : (err_red < 0 || err_red > prm.min_error_reduction_ratio_to_relinearize); // :592
					resid.swap(new_resid); total_err = new_err; RMSE = new_RMSE; // :601-604
					if (relin) { // :606-629
						out.num_relinearized++;
						std::fill(valid.begin(), valid.end(), 1);
						recompute_all_jacobians();
						hessian_update_numeric();
						if (schur_active()) HAp_orig = HAp; // realize_relinearized -> realize_HAp_changed (schur.h:165-168)
					}
					compute_minus_gradient(); // :633
					double ninf = 0; for (int k = 0; k < n; k++) ninf = std::max(ninf, std::fabs(grad[k]));
					if (ninf <= 1e-15) { stop = true; out.stop_reason |= 1 << SRBA_STOP_GRADIENT; } // :636-641
					if (RMSE < prm.max_error_per_obs_to_stop) { stop = true; out.stop_reason |= 1 << SRBA_STOP_RMSE; } // :642-646
					if (rho > prm.max_rho) { stop = true; out.stop_reason |= 1 << SRBA_STOP_RHO; } // :647-651
					lambda *= 1.0 / 3.0; nu = 2.0; // :653-654
				} else { // :658-690
					for (size_t i = 0; i < req.size(); i++) pose[req[i]] = old_poses[i]; // :664-670
					for (int i = 0; i < nK; i++) edge[i] = old_edges[i]; // :673-676
					ulm = old_ulm; // :677-680
					lambda *= nu; nu *= 2.0; stop = (lambda > MAX_LAMBDA); if (stop) out.stop_reason |= 1 << SRBA_STOP_LAMBDA; // :685-687
				}
				return true;
		};
		if (rp) { // the other run's trial sequence, whatever this run's stop tests say
			rp->diverged_at = -1; iter = 0;
			for (int t = 0; t < rp->n; t++) if (!trial(rp->dec[t])) break;
		} else
		for (iter = 0; iter < prm.max_iters && !stop; iter++) { // :454
			rho = 0;
			if (lambda >= MAX_LAMBDA) { stop = true; out.stop_reason |= 1 << SRBA_STOP_LAMBDA; } // :460-464
			if (RMSE < prm.max_error_per_obs_to_stop) { stop = true; out.stop_reason |= 1 << SRBA_STOP_RMSE; } // :465-469
			while (rho <= 0 && !stop) trial(-1); // :471
		}
		if (!stop) out.stop_reason |= 1 << SRBA_STOP_MAX_ITERS;
		out.num_iters = iter; out.num_trials = trials;
		out.total_sqr_error_final = total_err; out.obs_rmse = RMSE; out.lambda_final = lambda;
		// write back (the reference optimises in place)
		for (int i = 0; i < nK; i++) edge[i].store(c.edge_pose + (size_t)i * PD);
		for (int i = 0; i < nF * L; i++) c.ulm_pos[i] = ulm[i];
		if (c.pose) for (size_t i = 0; i < pose.size(); i++) pose[i].store(c.pose + i * PD);
		if (c.ulm_inf && c.ulm_inf_valid) { // S17 :726-751 crpLandmarksApprox
			for (int i = 0; i < nF; i++) {
				const bool ok = prm.cov_recovery == 1 && (!schur_active() ? true : (bool)Hf_invertible[i]); // was_ith_feature_invertible (:736; no-Schur solver returns true :197-201)
				c.ulm_inf_valid[i] = ok ? 1 : 0;
				if (ok) for (int k = 0; k < L * L; k++) c.ulm_inf[(size_t)i * L * L + k] = Hf[(size_t)c.hf_diag[i] * L * L + k];
			}
		}
	}
};

struct ReplayIO { const int32_t *dec; int n; double *rho, *chi2, *E; int32_t *flags; int diverged_at; };
template <int FAM> void run_one(const srba_hip_params &p, srba_problem_capsule &c, srba_lm_result &r, ReplayIO *io = nullptr) {
	Problem<FAM> pr(p, c);
	if (!io) { pr.run(r); return; }
	typename Problem<FAM>::Replay rp = {io->dec, io->n, io->rho, io->chi2, io->E, io->flags, -1}; pr.run(r, &rp); io->diverged_at = rp.diverged_at;
}

void dispatch_run(const srba_hip_params &p, srba_problem_capsule &c, srba_lm_result &r, ReplayIO *io = nullptr) {
	switch (p.family) {
#define X(F) case F: run_one<F>(p, c, r, io); break;
		ORACLE_ALL_FAMILIES(X)
#undef X
		default: std::memset(&r, 0, sizeof(r)); r.status = -1;
	}
}

// Initial linearisation only (S5..S14) + optional single solve, for per-kernel parity tests.
template <int FAM>
void stage_one(const srba_hip_params &p, srba_problem_capsule &c, int do_solve, double lambda_in,
               double *residuals, double *Jp, double *Jf, double *HAp, double *Hf, double *HApf, double *grad, double *delta,
               double *poses, double *scalars /*[4]: chi2, lambda0, not_pd, n_invalid*/) {
	Problem<FAM> pr(p, c);
	pr.update_spantree(false);
	std::fill(pr.valid.begin(), pr.valid.end(), 1);
	pr.recompute_all_jacobians();
	const int ninv = pr.hessian_update_numeric();
	const double chi2 = pr.residuals(pr.resid);
	pr.compute_minus_gradient();
	double mx = 0;
	for (int i = 0; i < pr.nK; i++) { const double *H = &pr.HAp[(size_t)c.hap_diag[i] * pr.P * pr.P]; for (int k = 0; k < pr.P; k++) mx = std::max(mx, H[k * pr.P + k]); }
	for (int i = 0; i < pr.nF; i++) { const double *H = &pr.Hf[(size_t)c.hf_diag[i] * pr.L * pr.L]; for (int k = 0; k < pr.L; k++) mx = std::max(mx, H[k * pr.L + k]); }
	auto cp = [](double *dst, const std::vector<double> &src) { if (dst) std::copy(src.begin(), src.end(), dst); };
	cp(residuals, pr.resid); cp(Jp, pr.Jp); cp(Jf, pr.Jf); cp(Hf, pr.Hf); cp(HApf, pr.HApf);
	if (grad) cp(grad, pr.grad); // gradient BEFORE any Schur mutation
	if (poses) for (size_t i = 0; i < pr.pose.size(); i++) pr.pose[i].store(poses + i * pr.PD);
	double not_pd = 0;
	if (do_solve) {
		if (pr.schur_active()) pr.schur_ctor();
		if (!(pr.use_schur && pr.dense_chol)) pr.sparse_setup();
		not_pd = pr.solve(lambda_in) ? 0 : 1;
		cp(delta, pr.delta);
	}
	cp(HAp, pr.HAp); // after solve: the Schur-reduced HAp when Schur is active
	if (scalars) { scalars[0] = chi2; scalars[1] = 1e-3 * mx; scalars[2] = not_pd; scalars[3] = ninv; }
}

// The reference's SchurTests body (tests/schur_unittest.cpp:71-279) on GIVEN Jacobian blocks: numeric Hessians over the capsule's symbolic plan
// (sparse_hessian_update_numeric), then SchurComplement::numeric_build_reduced_system(lambda) with the given minus-gradient.
template <int FAM>
void schur_from_jacobians(const srba_hip_params &p, srba_problem_capsule &c, const double *Jp, const double *Jf, const double *grad_in, double lambda, double *HAp_out, double *Hf_out,
	double *HApf_out, double *grad_out) {
	Problem<FAM> pr(p, c);
	std::fill(pr.valid.begin(), pr.valid.end(), 1);
	std::copy(Jp, Jp + pr.Jp.size(), pr.Jp.begin()); std::copy(Jf, Jf + pr.Jf.size(), pr.Jf.begin());
	pr.hessian_update_numeric();
	if (Hf_out) std::copy(pr.Hf.begin(), pr.Hf.end(), Hf_out);
	if (HApf_out) std::copy(pr.HApf.begin(), pr.HApf.end(), HApf_out);
	std::copy(grad_in, grad_in + pr.grad.size(), pr.grad.begin());
	if (pr.schur_active()) { pr.schur_ctor(); pr.schur_build_reduced(lambda); }
	std::copy(pr.HAp.begin(), pr.HAp.end(), HAp_out); std::copy(pr.grad.begin(), pr.grad.end(), grad_out);
}

/* Whole-map squared error (impl/eval_overall_error.h:15-137) on the path lists prepared by the front-end: poses composed from the root of
 * every pair towards the leaf (impl/spantree_create_complete.h:96-124), then sum ||z - h||^2 without robust kernel (:116-129). */
template <int FAM> static double overall_error(const srba_hip_params &p, const srba_overall_problem &q) {
	srba_hip_params pp = p; pp.use_robust_kernel = 0;
	srba_problem_capsule cap; std::memset(&cap, 0, sizeof(cap));
	cap.n_edges = q.n_edges; cap.edge_pose = const_cast<double *>(q.edge_pose); cap.n_pairs = q.n_pairs; cap.n_path = q.n_path;
	cap.pair_path_off = const_cast<int32_t *>(q.pair_path_off); cap.path_edge = const_cast<int32_t *>(q.path_edge);
	cap.n_obs = q.n_obs; cap.obs_pose = const_cast<int32_t *>(q.obs_pose); cap.obs_z = const_cast<double *>(q.obs_z);
	std::vector<int32_t> lmref(q.n_obs); for (int i = 0; i < q.n_obs; i++) lmref[i] = -1 - q.obs_lm[i]; // landmarks enter as the capsule's known-landmark table
	cap.obs_lm = lmref.data(); cap.n_known_lms = q.n_lms; cap.klm_pos = const_cast<double *>(q.lm_pos);
	Problem<FAM> P(pp, cap); P.update_spantree(false);
	std::vector<double> res; return P.residuals(res);
}
} // namespace

extern "C" {

/* Runs optimize_edges S5..S17 on each capsule (in place). n_threads<=1: serial (the reference is single-threaded). */
int srba_oracle_lm_run(const srba_hip_params *params, srba_problem_capsule *caps, int n, srba_lm_result *results, int n_threads) {
	if (!params || !caps || n < 0) return -1;
	std::vector<srba_lm_result> tmp; if (!results) { tmp.resize(n); results = tmp.data(); }
	if (n_threads <= 1) { for (int i = 0; i < n; i++) dispatch_run(*params, caps[i], results[i]); return 0; }
	// dynamic work queue: capsules differ 10x in cost (loop-closure windows), static interleaving leaves threads idle at the end
	std::atomic<int> next(0); const int chunk = 4;
	std::vector<std::thread> th;
	for (int t = 0; t < n_threads; t++) th.emplace_back([&, params, caps, results, n]() { for (;;) { const int b = next.fetch_add(chunk); if (b >= n) break; for (int i = b; i < std::min(n,
		b + chunk); i++) dispatch_run(*params, caps[i], results[i]); } });
	for (auto &x : th) x.join();
	return 0;
}
/* Decision replay (see Problem::run): capsule i runs the n_decisions[i] trials of decisions[i * stride ...] (0 not PD, 1 rejected, 2 accepted) and reports, per trial, its own rho, the chi2 of the
 * trial point, the chi2 it started from and flags (1 own factorisation PD, 2 own rho sign differs from the decision, 4 not-PD forced on a PD system); diverged_at[i] = the trial at which the sequence
 * could not be followed (the other run solved a system this one calls not PD), else -1. results[i] is the usual result of the replayed run (final chi2 = after the last replayed trial). */
int srba_oracle_lm_run_replay(const srba_hip_params *params, srba_problem_capsule *caps, int n, const int32_t *decisions, const int32_t *n_decisions, int stride,
                              double *own_rho, double *own_chi2, double *own_E, int32_t *flags, int32_t *diverged_at, srba_lm_result *results, int n_threads) {
	if (!params || !caps || n < 0 || !decisions || !n_decisions || stride < 1 || !own_rho || !own_chi2 || !own_E || !flags || !diverged_at || !results) return -1;
	std::atomic<int> next(0);
	auto work = [&]() { for (;;) { const int i = next.fetch_add(1); if (i >= n) break;
		ReplayIO io = {decisions + (size_t)i * stride, std::min(n_decisions[i], stride), own_rho + (size_t)i * stride, own_chi2 + (size_t)i * stride, own_E + (size_t)i * stride,
			flags + (size_t)i * stride, -1};
		dispatch_run(*params, caps[i], results[i], &io); diverged_at[i] = io.diverged_at; } };
	if (n_threads <= 1) { work(); return 0; }
	std::vector<std::thread> th; for (int t = 0; t < n_threads; t++) th.emplace_back(work);
	for (auto &x : th) x.join();
	return 0;
}
/* Seconds spent so far in the symbolic Cholesky analysis (summed over threads); resets the counter. */
double srba_oracle_take_symbolic_seconds(void) { return 1e-9 * (double)g_symbolic_ns.exchange(0); }

/* Same signature as srba_backend_fn (srba_amd/csrc/engine_capi.h): lets tests drive the product front-end with the oracle as numeric back-end. */
int srba_oracle_run_one(const srba_hip_params *params, srba_problem_capsule *cap, srba_lm_result *result) {
	if (!params || !cap || !result) return -1;
	dispatch_run(*params, *cap, *result);
	return result->status < 0 ? -1 : 0;
}

int srba_oracle_eval_overall(const srba_hip_params *p, const srba_overall_problem *q, double *out) {
	if (!p || !q || !out) return -1;
	switch (p->family) {
#define X(F) case F: *out = overall_error<F>(*p, *q); break;
		ORACLE_ALL_FAMILIES(X)
#undef X
		default: return -1;
	}
	return 0;
}

int srba_oracle_schur_from_jacobians(const srba_hip_params *p, srba_problem_capsule *c, const double *Jp, const double *Jf, const double *grad_in, double lambda,
                                     double *HAp_out, double *Hf_out, double *HApf_out, double *grad_out) {
	switch (p->family) {
#define CASE(F) case F: schur_from_jacobians<F>(*p, *c, Jp, Jf, grad_in, lambda, HAp_out, Hf_out, HApf_out, grad_out); return 0;
		ORACLE_ALL_FAMILIES(CASE)
#undef CASE
	}
	return -1;
}

int srba_oracle_stage(const srba_hip_params *p, srba_problem_capsule *c, int do_solve, double lambda,
                      double *residuals, double *Jp, double *Jf, double *HAp, double *Hf, double *HApf, double *grad, double *delta,
                      double *poses, double *scalars) {
	switch (p->family) {
#define CASE(F) case F: stage_one<F>(*p, *c, do_solve, lambda, residuals, Jp, Jf, HAp, Hf, HApf, grad, delta, poses, scalars); return 0;
		ORACLE_ALL_FAMILIES(CASE)
#undef CASE
	}
	return -1;
}

} // extern "C"
