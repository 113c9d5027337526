/*
 * mrpt_lite.h -- the handful of MRPT 1.x value types that appear in SRBA's public API (poses, points, camera
 * calibration, fixed arrays/matrices), re-implemented dependency-free so that user code written against
 * <srba.h> of MRPT/srba compiles against this repo unchanged where neither MRPT nor Eigen exist.
 *
 * Semantics follow MRPT 1.x: CPose2D/CPose3D composition "A (+) B", unary minus = inverse,
 * binary "A - B" = (-)B (+) A, CPose3D(x,y,z,yaw,pitch,roll) with R = Rz(yaw) Ry(pitch) Rx(roll),
 * SE_traits<N>::pseudo_exp / pseudo_ln, tfest::se2_l2 / se3_l2 (closed-form least-squares alignment; Horn 1987).
 * Reference call sites: include/srba/impl/jacobians.h:261,387-396; impl/spantree_update_numeric.h:51,59,72;
 * impl/optimize_edges.h:517-521; impl/determine_kf2kf_edges_to_create.h:54-55,193-197,251;
 * models/observations_*.h (landmark_matcher<>::find_relative_pose).
 *
 * There is no switch to the real library: these headers always use the value types below (a build against an installed MRPT 1.x would have to convert at the
 * boundary -- CPose3D::getHomogeneousMatrix <-> storeTo / loadFrom -- and is not provided).
 */
#pragma once
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <deque>
#include <fstream>
#include <iostream>
#include <list>
#include <map>
#include <random>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

#define SRBA_STR2(x) #x
#define SRBA_STR(x) SRBA_STR2(x)
#ifndef ASSERT_
// (MRPT's assertion macros are complete statements: code written for it uses them with and without a trailing ';')
#define ASSERT_(c) { if (!(c)) throw std::logic_error(std::string("Assert failed: " #c " at " __FILE__ ":" SRBA_STR(__LINE__))); }
#define ASSERTMSG_(c, msg) { if (!(c)) throw std::logic_error(std::string(msg)); }
#define ASSERTDEB_(c) ((void)0)
#define MRPT_UNUSED_PARAM(x) (void)(x)
#define ASSERT_EQUAL_(a, b) ASSERT_((a) == (b))
#define ASSERT_ABOVE_(a, b) ASSERT_((a) > (b))
#define ASSERT_BELOW_(a, b) ASSERT_((a) < (b))
#define ASSERT_ABOVEEQ_(a, b) ASSERT_((a) >= (b))
#define ASSERT_BELOWEQ_(a, b) ASSERT_((a) <= (b))
#define THROW_EXCEPTION(msg) { throw std::logic_error(std::string(msg)); }
#define MRPT_TODO(x)
#define MRPT_START
#define MRPT_END
#endif
#ifndef MRPT_HAS_WXWIDGETS
#define MRPT_HAS_WXWIDGETS 0
#endif
#ifndef MRPT_HAS_CXX11
#define MRPT_HAS_CXX11 1
#endif

namespace mrpt {
namespace utils {
class CConfigFileBase;
template <class T> inline T square(const T x) { return x * x; }
inline double DEG2RAD(const double x) { return x * M_PI / 180.0; }
inline double RAD2DEG(const double x) { return x * 180.0 / M_PI; }
struct TPixelCoordf { float x, y; TPixelCoordf() : x(0), y(0) {} TPixelCoordf(float x_, float y_) : x(x_), y(y_) {} };

/** Pinhole intrinsics, the subset SRBA reads (models/sensors.h:59-60,98-99). */
struct TCamera {
	double m_fx, m_fy, m_cx, m_cy; unsigned ncols, nrows;
	struct dist_t { double v[5]; dist_t() { setZero(); } void setZero() { for (int i = 0; i < 5; i++) v[i] = 0; } double &operator[](int i) { return v[i]; } const double &operator[](int i) const {
		return v[i]; } } dist; // [k1 k2 t1 t2 k3]; SRBA's sensor models ignore distortion
	TCamera() : m_fx(1), m_fy(1), m_cx(0), m_cy(0), ncols(640), nrows(480) {}
	double fx() const { return m_fx; } double fy() const { return m_fy; } double cx() const { return m_cx; } double cy() const { return m_cy; }
	void fx(double v) { m_fx = v; } void fy(double v) { m_fy = v; } void cx(double v) { m_cx = v; } void cy(double v) { m_cy = v; }
	void setIntrinsicParamsFromValues(double fx_, double fy_, double cx_, double cy_) { m_fx = fx_; m_fy = fy_; m_cx = cx_; m_cy = cy_; }
	/** keys resolution = [W H], cx, cy, fx, fy, dist = [k1 k2 t1 t2 k3] of the given section (the layout of the reference's dataset .cfg files) */
	void loadFromConfigFile(const std::string &section, const CConfigFileBase &cfg);
};
} // namespace utils

namespace math {
inline double wrapTo2Pi(double a) { const bool neg = a < 0; a = std::fmod(a, 2.0 * M_PI); if (neg) a += 2.0 * M_PI; return a; }
inline double wrapToPi(double a) { return wrapTo2Pi(a + M_PI) - M_PI; }
struct TPoint2D { double x, y; TPoint2D() : x(0), y(0) {} TPoint2D(double x_, double y_) : x(x_), y(y_) {} };
struct TPoint3D { double x, y, z; TPoint3D() : x(0), y(0), z(0) {} TPoint3D(double x_, double y_, double z_) : x(x_), y(y_), z(z_) {} };
struct TPose2D { double x, y, phi; TPose2D() : x(0), y(0), phi(0) {} };

/** Fixed-length array of doubles (mrpt::math::CArrayDouble<N>). */
template <std::size_t N> struct CArrayDouble {
	double v[N];
	CArrayDouble() { for (std::size_t i = 0; i < N; i++) v[i] = 0; }
	explicit CArrayDouble(const double *p) { for (std::size_t i = 0; i < N; i++) v[i] = p[i]; }
	double &operator[](std::size_t i) { return v[i]; } const double &operator[](std::size_t i) const { return v[i]; }
	void setZero() { for (std::size_t i = 0; i < N; i++) v[i] = 0; } void zeros() { setZero(); }
	static std::size_t size() { return N; }
	const double *data() const { return v; } double *data() { return v; }
};
/** Fixed-size row-major matrix: stands in for Eigen::Matrix<double,R,C> in parameter structs
 * (e.g. parameters.obs_noise.lambda, srba_options_noise.h:91-97). */
template <std::size_t R, std::size_t C> struct CMatrixFixed {
	double m[R * C];
	CMatrixFixed() { setZero(); }
	/** from any matrix expression with operator()(r, c) (an Eigen matrix in user code written for the reference) */
	template <class M, class = decltype(std::declval<const M &>()(0, 0))> CMatrixFixed(const M &o) { for (std::size_t r = 0; r < R; r++) for (std::size_t c = 0; c < C; c++) m[r * C + c] = o(r, c); }
	double &operator()(std::size_t r, std::size_t c) { return m[r * C + c]; } const double &operator()(std::size_t r, std::size_t c) const { return m[r * C + c]; }
	double &coeffRef(std::size_t r, std::size_t c) { return m[r * C + c]; } double coeff(std::size_t r, std::size_t c) const { return m[r * C + c]; }
	void setZero() { for (std::size_t i = 0; i < R * C; i++) m[i] = 0; }
	void setIdentity() { setZero(); for (std::size_t i = 0; i < (R < C ? R : C); i++) m[i * C + i] = 1; }
	static CMatrixFixed Identity() { CMatrixFixed r; r.setIdentity(); return r; }
	static std::size_t rows() { return R; } static std::size_t cols() { return C; }
};
typedef CMatrixFixed<3, 3> CMatrixDouble33;
typedef CMatrixFixed<4, 4> CMatrixDouble44;
/** Run-time sized row-major matrix (mrpt::math::CMatrixDouble / CMatrixD), the subset the reference's tutorials and apps use. */
struct CMatrixDouble {
	std::size_t nr, nc; std::vector<double> m;
	CMatrixDouble(std::size_t r = 0, std::size_t c = 0) : nr(r), nc(c), m(r * c, 0.0) {}
	void setSize(std::size_t r, std::size_t c) { nr = r; nc = c; m.assign(r * c, 0.0); }
	double &operator()(std::size_t r, std::size_t c) { return m[r * nc + c]; } const double &operator()(std::size_t r, std::size_t c) const { return m[r * nc + c]; }
	double &coeffRef(std::size_t r, std::size_t c) { return m[r * nc + c]; } double coeff(std::size_t r, std::size_t c) const { return m[r * nc + c]; }
	std::size_t size() const { return nr * nc; } std::size_t rows() const { return nr; } std::size_t cols() const { return nc; } std::size_t getRowCount() const { return nr; }
		std::size_t getColCount() const { return nc; }
	void loadFromTextFile(const std::string &file); void saveToTextFile(const std::string &file) const;
};
typedef CMatrixDouble CMatrixD;

struct CQuaternionDouble {
	double q[4]; // r,x,y,z
	CQuaternionDouble() { q[0] = 1; q[1] = q[2] = q[3] = 0; }
	CQuaternionDouble(double r, double x, double y, double z) { q[0] = r; q[1] = x; q[2] = y; q[3] = z; }
	double r() const { return q[0]; } double x() const { return q[1]; } double y() const { return q[2]; } double z() const { return q[3]; }
	void rotationMatrix(double R[9]) const {
		const double r = q[0], x = q[1], y = q[2], z = q[3];
		R[0] = r * r + x * x - y * y - z * z; R[1] = 2 * (x * y - r * z); R[2] = 2 * (z * x + r * y);
		R[3] = 2 * (x * y + r * z); R[4] = r * r - x * x + y * y - z * z; R[5] = 2 * (y * z - r * x);
		R[6] = 2 * (z * x - r * y); R[7] = 2 * (y * z + r * x); R[8] = r * r - x * x - y * y + z * z;
	}
};
} // namespace math

namespace poses {
enum TConstructorFlags_Poses { UNINITIALIZED_POSE = 0 };
class CPose3D;

/** 2D pose (x,y,phi). */
class CPose2D {
public:
	enum { rotation_dimensions = 2 };
	double m_x, m_y, m_phi;
	CPose2D() : m_x(0), m_y(0), m_phi(0) {}
	CPose2D(double x, double y, double phi) : m_x(x), m_y(y), m_phi(phi) {}
	explicit CPose2D(TConstructorFlags_Poses) {}
	explicit CPose2D(const math::TPose2D &p) : m_x(p.x), m_y(p.y), m_phi(p.phi) {}
	explicit CPose2D(const CPose3D &p);
	double x() const { return m_x; } double y() const { return m_y; } double phi() const { return m_phi; }
	void x(double v) { m_x = v; } void y(double v) { m_y = v; } void phi(double v) { m_phi = v; }
	/** this = A (+) B (safe if this==A or this==B) */
	void composeFrom(const CPose2D &A, const CPose2D &B) {
		const double c = std::cos(A.m_phi), s = std::sin(A.m_phi);
		const double nx = A.m_x + B.m_x * c - B.m_y * s, ny = A.m_y + B.m_x * s + B.m_y * c;
		m_phi = math::wrapToPi(A.m_phi + B.m_phi); m_x = nx; m_y = ny;
	}
	/** this = A (-) B */
	void inverseComposeFrom(const CPose2D &A, const CPose2D &B) {
		const double c = std::cos(B.m_phi), s = std::sin(B.m_phi);
		const double nx = (A.m_x - B.m_x) * c + (A.m_y - B.m_y) * s, ny = -(A.m_x - B.m_x) * s + (A.m_y - B.m_y) * c;
		m_phi = math::wrapToPi(A.m_phi - B.m_phi); m_x = nx; m_y = ny;
	}
	void inverse() { const double c = std::cos(m_phi), s = std::sin(m_phi); const double nx = -m_x * c - m_y * s, ny = m_x * s - m_y * c; m_x = nx; m_y = ny; m_phi = -m_phi; }
	void composePoint(double lx, double ly, double &gx, double &gy) const { const double c = std::cos(m_phi), s = std::sin(m_phi); gx = m_x + lx * c - ly * s; gy = m_y + lx * s + ly * c; }
	void composePoint(double lx, double ly, double lz, double &gx, double &gy, double &gz) const { composePoint(lx, ly, gx, gy); gz = lz; }
	void inverseComposePoint(double gx, double gy, double &lx, double &ly) const { const double c = std::cos(m_phi), s = std::sin(m_phi); lx = (gx - m_x) * c + (gy - m_y) * s;
		ly = -(gx - m_x) * s + (gy - m_y) * c; }
	CPose2D operator+(const CPose2D &b) const { CPose2D r; r.composeFrom(*this, b); return r; }
	CPose2D operator-(const CPose2D &b) const { CPose2D r; r.inverseComposeFrom(*this, b); return r; }
	void getAsVector(double v[3]) const { v[0] = m_x; v[1] = m_y; v[2] = m_phi; }
	static std::size_t storage_doubles() { return 3; }
	void storeTo(double *p) const { p[0] = m_x; p[1] = m_y; p[2] = m_phi; }
	void loadFrom(const double *p) { m_x = p[0]; m_y = p[1]; m_phi = p[2]; }
};
inline CPose2D operator-(const CPose2D &p) { CPose2D r(p); r.inverse(); return r; }
inline std::ostream &operator<<(std::ostream &o, const CPose2D &p) { return o << "(" << p.x() << "," << p.y() << "," << utils::RAD2DEG(p.phi()) << "deg)"; }

class CPose3DQuat;
/** 3D pose: translation + 3x3 rotation (row-major). */
class CPose3D {
public:
	enum { rotation_dimensions = 3 };
	double m_t[3]; double m_R[9];
	CPose3D() { setIdentity(); }
	explicit CPose3D(TConstructorFlags_Poses) {}
	CPose3D(double x, double y, double z, double yaw = 0, double pitch = 0, double roll = 0) { setFromValues(x, y, z, yaw, pitch, roll); }
	explicit CPose3D(const CPose2D &p) { setFromValues(p.x(), p.y(), 0, p.phi(), 0, 0); }
	explicit CPose3D(const CPose3DQuat &q);
	void setIdentity() { m_t[0] = m_t[1] = m_t[2] = 0; for (int i = 0; i < 9; i++) m_R[i] = (i % 4 == 0) ? 1.0 : 0.0; }
	void setFromValues(double x, double y, double z, double yaw = 0, double pitch = 0, double roll = 0) {
		m_t[0] = x; m_t[1] = y; m_t[2] = z;
		const double cy = std::cos(yaw), sy = std::sin(yaw), cp = std::cos(pitch), sp = std::sin(pitch), cr = std::cos(roll), sr = std::sin(roll);
		m_R[0] = cy * cp; m_R[1] = cy * sp * sr - sy * cr; m_R[2] = cy * sp * cr + sy * sr;
		m_R[3] = sy * cp; m_R[4] = sy * sp * sr + cy * cr; m_R[5] = sy * sp * cr - cy * sr;
		m_R[6] = -sp;     m_R[7] = cp * sr;                m_R[8] = cp * cr;
	}
	double x() const { return m_t[0]; } double y() const { return m_t[1]; } double z() const { return m_t[2]; }
	void getYawPitchRoll(double &yaw, double &pitch, double &roll) const {
		pitch = std::atan2(-m_R[6], std::hypot(m_R[0], m_R[3]));
		if (std::fabs(std::fabs(pitch) - M_PI / 2) < 1e-10) { roll = 0; yaw = (pitch > 0) ? std::atan2(m_R[5], m_R[2]) : std::atan2(-m_R[5], -m_R[2]); }
		else { roll = std::atan2(m_R[7], m_R[8]); yaw = std::atan2(m_R[3], m_R[0]); }
	}
	double yaw() const { double y, p, r; getYawPitchRoll(y, p, r); return y; }
	double pitch() const { double y, p, r; getYawPitchRoll(y, p, r); return p; }
	double roll() const { double y, p, r; getYawPitchRoll(y, p, r); return r; }
	void getAsVector(double v[6]) const { v[0] = m_t[0]; v[1] = m_t[1]; v[2] = m_t[2]; getYawPitchRoll(v[3], v[4], v[5]); }
	math::CMatrixDouble33 getRotationMatrix() const { math::CMatrixDouble33 R; for (int i = 0; i < 9; i++) R.m[i] = m_R[i]; return R; }
	void setRotationMatrix(const math::CMatrixDouble33 &R) { for (int i = 0; i < 9; i++) m_R[i] = R.m[i]; }
	math::CMatrixDouble44 getHomogeneousMatrixVal() const {
		math::CMatrixDouble44 M; for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) M(i, j) = m_R[3 * i + j]; M(i, 3) = m_t[i]; } M(3, 3) = 1; return M;
	}
	void composeFrom(const CPose3D &A, const CPose3D &B) {
		double R[9], t[3];
		for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R[3 * i + j] = A.m_R[3 * i] * B.m_R[j] + A.m_R[3 * i + 1] * B.m_R[3 + j] + A.m_R[3 * i + 2] * B.m_R[6 + j];
		for (int i = 0; i < 3; i++) t[i] = A.m_t[i] + A.m_R[3 * i] * B.m_t[0] + A.m_R[3 * i + 1] * B.m_t[1] + A.m_R[3 * i + 2] * B.m_t[2];
		for (int i = 0; i < 9; i++) m_R[i] = R[i]; for (int i = 0; i < 3; i++) m_t[i] = t[i];
	}
	void inverse() {
		double R[9], t[3];
		for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R[3 * i + j] = m_R[3 * j + i];
		for (int i = 0; i < 3; i++) t[i] = -(m_R[i] * m_t[0] + m_R[3 + i] * m_t[1] + m_R[6 + i] * m_t[2]);
		for (int i = 0; i < 9; i++) m_R[i] = R[i]; for (int i = 0; i < 3; i++) m_t[i] = t[i];
	}
	void inverseComposeFrom(const CPose3D &A, const CPose3D &B) { CPose3D Bi(B); Bi.inverse(); composeFrom(Bi, A); }
	void composePoint(double lx, double ly, double lz, double &gx, double &gy, double &gz) const {
		const double x = m_t[0] + m_R[0] * lx + m_R[1] * ly + m_R[2] * lz, y = m_t[1] + m_R[3] * lx + m_R[4] * ly + m_R[5] * lz, z = m_t[2] + m_R[6] * lx + m_R[7] * ly + m_R[8] * lz;
		gx = x; gy = y; gz = z;
	}
	void inverseComposePoint(double gx, double gy, double gz, double &lx, double &ly, double &lz) const {
		const double dx = gx - m_t[0], dy = gy - m_t[1], dz = gz - m_t[2];
		const double x = m_R[0] * dx + m_R[3] * dy + m_R[6] * dz, y = m_R[1] * dx + m_R[4] * dy + m_R[7] * dz, z = m_R[2] * dx + m_R[5] * dy + m_R[8] * dz;
		lx = x; ly = y; lz = z;
	}
	CPose3D operator+(const CPose3D &b) const { CPose3D r(UNINITIALIZED_POSE); r.composeFrom(*this, b); return r; }
	CPose3D operator-(const CPose3D &b) const { CPose3D r(UNINITIALIZED_POSE); r.inverseComposeFrom(*this, b); return r; }
	static std::size_t storage_doubles() { return 12; }
	void storeTo(double *p) const { for (int i = 0; i < 3; i++) p[i] = m_t[i]; for (int i = 0; i < 9; i++) p[3 + i] = m_R[i]; }
	void loadFrom(const double *p) { for (int i = 0; i < 3; i++) m_t[i] = p[i]; for (int i = 0; i < 9; i++) m_R[i] = p[3 + i]; }
	/** exp map of so(3) (Rodrigues) */
	static math::CMatrixDouble33 exp_rotation(const double w[3]) {
		const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], th = std::sqrt(th2);
		double a, b; if (th < 1e-8) { a = 1 - th2 / 6; b = 0.5 - th2 / 24; } else { a = std::sin(th) / th; b = (1 - std::cos(th)) / th2; }
		const double W[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
		math::CMatrixDouble33 R;
		for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { const double w2 = W[3 * i] * W[j] + W[3 * i + 1] * W[3 + j] + W[3 * i + 2] * W[6 + j]; R(i,
			j) = (i == j ? 1.0 : 0.0) + a * W[3 * i + j] + b * w2; }
		return R;
	}
	/** log map of SO(3) */
	void ln_rotation(double w[3]) const {
		const double tr = m_R[0] + m_R[4] + m_R[8]; double c = 0.5 * (tr - 1); c = std::max(-1.0, std::min(1.0, c));
		const double th = std::acos(c); const double v[3] = {m_R[7] - m_R[5], m_R[2] - m_R[6], m_R[3] - m_R[1]};
		const double f = (th < 1e-8) ? 0.5 : th / (2 * std::sin(th));
		for (int i = 0; i < 3; i++) w[i] = f * v[i];
	}
};
inline CPose3D operator-(const CPose3D &p) { CPose3D r(p); r.inverse(); return r; }
inline std::ostream &operator<<(std::ostream &o, const CPose3D &p) { double y, pi, r; p.getYawPitchRoll(y, pi, r); return o << "(" << p.x() << "," << p.y() << "," << p.z() << "," << utils::RAD2DEG(y)
	<< "deg," << utils::RAD2DEG(pi) << "deg," << utils::RAD2DEG(r) << "deg)"; }
inline CPose2D::CPose2D(const CPose3D &p) : m_x(p.x()), m_y(p.y()), m_phi(p.yaw()) {}

/** 3D pose with quaternion (x y z qr qx qy qz). */
class CPose3DQuat {
public:
	double m_t[3]; math::CQuaternionDouble m_q;
	CPose3DQuat() { m_t[0] = m_t[1] = m_t[2] = 0; }
	CPose3DQuat(double x, double y, double z, const math::CQuaternionDouble &q) : m_q(q) { m_t[0] = x; m_t[1] = y; m_t[2] = z; }
	double x() const { return m_t[0]; } double y() const { return m_t[1]; } double z() const { return m_t[2]; }
	const math::CQuaternionDouble &quat() const { return m_q; }
	explicit CPose3DQuat(const CPose3D &p);
	/** "[x y z qr qx qy qz]" */
	void fromString(const std::string &s) { std::string t(s); for (char &ch : t) if (ch == '[' || ch == ']' || ch == ',') ch = ' '; std::istringstream is(t); double v[7] = {0, 0, 0, 1, 0, 0, 0};
		for (int i = 0; i < 7 && (is >> v[i]); i++) {} *this = CPose3DQuat(v[0], v[1], v[2], math::CQuaternionDouble(v[3], v[4], v[5], v[6])); }
	std::string asString() const { std::ostringstream o; o << "[" << m_t[0] << " " << m_t[1] << " " << m_t[2] << " " << m_q.r() << " " << m_q.x() << " " << m_q.y() << " " << m_q.z() << "]";
		return o.str(); }
};
inline CPose3D::CPose3D(const CPose3DQuat &q) { m_t[0] = q.m_t[0]; m_t[1] = q.m_t[1]; m_t[2] = q.m_t[2]; q.m_q.rotationMatrix(m_R); }

/** quaternion of a rotation matrix (row-major), w >= 0 */
inline CPose3DQuat::CPose3DQuat(const CPose3D &p) {
	m_t[0] = p.m_t[0]; m_t[1] = p.m_t[1]; m_t[2] = p.m_t[2]; const double *R = p.m_R; const double tr = R[0] + R[4] + R[8]; double q[4];
	if (tr > 0) { const double s = 2 * std::sqrt(tr + 1); q[0] = 0.25 * s; q[1] = (R[7] - R[5]) / s; q[2] = (R[2] - R[6]) / s; q[3] = (R[3] - R[1]) / s; }
	else if (R[0] > R[4] && R[0] > R[8]) { const double s = 2 * std::sqrt(1 + R[0] - R[4] - R[8]); q[0] = (R[7] - R[5]) / s; q[1] = 0.25 * s; q[2] = (R[1] + R[3]) / s; q[3] = (R[2] + R[6]) / s; }
	else if (R[4] > R[8]) { const double s = 2 * std::sqrt(1 + R[4] - R[0] - R[8]); q[0] = (R[2] - R[6]) / s; q[1] = (R[1] + R[3]) / s; q[2] = 0.25 * s; q[3] = (R[5] + R[7]) / s; }
	else { const double s = 2 * std::sqrt(1 + R[8] - R[0] - R[4]); q[0] = (R[3] - R[1]) / s; q[1] = (R[2] + R[6]) / s; q[2] = (R[5] + R[7]) / s; q[3] = 0.25 * s; }
	if (q[0] < 0) for (int i = 0; i < 4; i++) q[i] = -q[i];
	m_q = math::CQuaternionDouble(q[0], q[1], q[2], q[3]);
}
/** A - B = (-)B (+) A: the pose of A as seen from B */
inline CPose3DQuat operator-(const CPose3DQuat &a, const CPose3DQuat &b) { return CPose3DQuat(CPose3D(a) - CPose3D(b)); }
inline std::ostream &operator<<(std::ostream &o, const CPose3DQuat &p) { return o << p.asString(); }

template <std::size_t DOF> struct SE_traits;
template <> struct SE_traits<3> {
	enum { VECTOR_SIZE = 6 }; typedef math::CArrayDouble<6> array_t; typedef CPose3D pose_t;
	static void pseudo_exp(const array_t &x, CPose3D &P) { P.m_t[0] = x[0]; P.m_t[1] = x[1]; P.m_t[2] = x[2]; P.setRotationMatrix(CPose3D::exp_rotation(&x.v[3])); }
	static void pseudo_ln(const CPose3D &P, array_t &x) { x[0] = P.m_t[0]; x[1] = P.m_t[1]; x[2] = P.m_t[2]; P.ln_rotation(&x.v[3]); }
};
template <> struct SE_traits<2> {
	enum { VECTOR_SIZE = 3 }; typedef math::CArrayDouble<3> array_t; typedef CPose2D pose_t;
	static void pseudo_exp(const array_t &x, CPose2D &P) { P.x(x[0]); P.y(x[1]); P.phi(x[2]); }
	static void pseudo_ln(const CPose2D &P, array_t &x) { x[0] = P.x(); x[1] = P.y(); x[2] = P.phi(); }
};
} // namespace poses

namespace utils {
/** Stereo rig calibration (mrpt::utils::TStereoCamera). */
struct TStereoCamera {
	TCamera leftCamera, rightCamera; poses::CPose3DQuat rightCameraPose;
	/** sections <section>_LEFT, <section>_RIGHT (TCamera keys) and <section>_LEFT2RIGHT_POSE with pose_quaternion = [x y z qr qx qy qz] */
	void loadFromConfigFile(const std::string &section, const CConfigFileBase &cfg);
};
struct TMatchingPair {
	unsigned this_idx, other_idx; double this_x, this_y, this_z, other_x, other_y, other_z;
	TMatchingPair(unsigned ti, unsigned oi, double tx, double ty, double tz, double ox, double oy, double oz) : this_idx(ti), other_idx(oi), this_x(tx), this_y(ty), this_z(tz), other_x(ox),
		other_y(oy), other_z(oz) {}
};
typedef std::vector<TMatchingPair> TMatchingPairList;
} // namespace utils

namespace tfest {
/** Least-squares SE(2) alignment: this = pose (+) other  [EXT mrpt::tfest::se2_l2]. */
inline bool se2_l2(const utils::TMatchingPairList &in, math::TPose2D &out) {
	const std::size_t N = in.size(); if (N < 2) return false;
	double mxa = 0, mya = 0, mxb = 0, myb = 0;
	for (const auto &m : in) { mxa += m.this_x; mya += m.this_y; mxb += m.other_x; myb += m.other_y; }
	mxa /= N; mya /= N; mxb /= N; myb /= N;
	double Ax = 0, Ay = 0;
	for (const auto &m : in) {
		const double xa = m.this_x - mxa, ya = m.this_y - mya, xb = m.other_x - mxb, yb = m.other_y - myb;
		Ax += xa * xb + ya * yb; Ay += xb * ya - xa * yb;
	}
	out.phi = (Ax != 0 || Ay != 0) ? std::atan2(Ay, Ax) : 0.0;
	const double c = std::cos(out.phi), s = std::sin(out.phi);
	out.x = mxa - mxb * c + myb * s; out.y = mya - mxb * s - myb * c;
	return true;
}
namespace detail {
/** Jacobi eigen-decomposition of a symmetric 4x4 matrix; returns eigenvector of the largest eigenvalue. */
inline void largest_eigvec4(double A[16], double v[4]) {
	double V[16]; for (int i = 0; i < 16; i++) V[i] = (i % 5 == 0) ? 1.0 : 0.0;
	for (int sweep = 0; sweep < 60; sweep++) {
		double off = 0; for (int i = 0; i < 4; i++) for (int j = i + 1; j < 4; j++) off += A[4 * i + j] * A[4 * i + j];
		if (off < 1e-30) break;
		for (int p = 0; p < 4; p++) for (int q = p + 1; q < 4; q++) {
			if (std::fabs(A[4 * p + q]) < 1e-300) continue;
			const double th = (A[4 * q + q] - A[4 * p + p]) / (2 * A[4 * p + q]);
			const double t = (th >= 0 ? 1.0 : -1.0) / (std::fabs(th) + std::sqrt(th * th + 1)), c = 1 / std::sqrt(t * t + 1), s = t * c;
			for (int k = 0; k < 4; k++) { const double akp = A[4 * k + p], akq = A[4 * k + q]; A[4 * k + p] = c * akp - s * akq; A[4 * k + q] = s * akp + c * akq; }
			for (int k = 0; k < 4; k++) { const double apk = A[4 * p + k], aqk = A[4 * q + k]; A[4 * p + k] = c * apk - s * aqk; A[4 * q + k] = s * apk + c * aqk; }
			for (int k = 0; k < 4; k++) { const double vkp = V[4 * k + p], vkq = V[4 * k + q]; V[4 * k + p] = c * vkp - s * vkq; V[4 * k + q] = s * vkp + c * vkq; }
		}
	}
	int best = 0; for (int i = 1; i < 4; i++) if (A[5 * i] > A[5 * best]) best = i;
	for (int k = 0; k < 4; k++) v[k] = V[4 * k + best];
}
} // namespace detail
/** Least-squares SE(3) alignment (Horn's quaternion method), unit scale: this = pose (+) other [EXT mrpt::tfest::se3_l2]. */
inline bool se3_l2(const utils::TMatchingPairList &in, poses::CPose3DQuat &out, double &out_scale, bool forceScaleToUnity = true) {
	(void)forceScaleToUnity;
	const std::size_t N = in.size(); if (N < 3) return false;
	double ct[3] = {0, 0, 0}, co[3] = {0, 0, 0};
	for (const auto &m : in) { ct[0] += m.this_x; ct[1] += m.this_y; ct[2] += m.this_z; co[0] += m.other_x; co[1] += m.other_y; co[2] += m.other_z; }
	for (int i = 0; i < 3; i++) { ct[i] /= N; co[i] /= N; }
	double S[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}; // S = sum other' * this'^t
	for (const auto &m : in) {
		const double o[3] = {m.other_x - co[0], m.other_y - co[1], m.other_z - co[2]}, t[3] = {m.this_x - ct[0], m.this_y - ct[1], m.this_z - ct[2]};
		for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) S[3 * i + j] += o[i] * t[j];
	}
	double Nm[16] = {
		S[0] + S[4] + S[8], S[5] - S[7], S[6] - S[2], S[1] - S[3],
		S[5] - S[7], S[0] - S[4] - S[8], S[1] + S[3], S[6] + S[2],
		S[6] - S[2], S[1] + S[3], -S[0] + S[4] - S[8], S[5] + S[7],
		S[1] - S[3], S[6] + S[2], S[5] + S[7], -S[0] - S[4] + S[8]};
	double q[4]; detail::largest_eigvec4(Nm, q);
	const double nq = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]); for (int i = 0; i < 4; i++) q[i] /= nq;
	if (q[0] < 0) for (int i = 0; i < 4; i++) q[i] = -q[i];
	math::CQuaternionDouble Q(q[0], q[1], q[2], q[3]); double R[9]; Q.rotationMatrix(R);
	out_scale = 1.0;
	const double tx = ct[0] - (R[0] * co[0] + R[1] * co[1] + R[2] * co[2]), ty = ct[1] - (R[3] * co[0] + R[4] * co[1] + R[5] * co[2]), tz = ct[2] - (R[6] * co[0] + R[7] * co[1] + R[8] * co[2]);
	out = poses::CPose3DQuat(tx, ty, tz, Q);
	return true;
}
} // namespace tfest

namespace utils {
/** Wall-clock section profiler with the interface SRBA uses (mrpt::utils::CTimeLogger: enter/leave/enable). */
class CTimeLogger {
public:
	explicit CTimeLogger(bool enabled = true) : m_enabled(enabled) {}
	void enable(bool e = true) { m_enabled = e; } void disable() { m_enabled = false; }
	void enter(const char *name); double leave(const char *name);
	void registerUserMeasure(const char *name, double v) { if (m_enabled) { auto &d = m_data[name]; d.n++; d.total += v; } }
	struct TCallData { std::size_t n = 0; double total = 0, t0 = 0; };
	const std::map<std::string, TCallData> &getStats() const { return m_data; }
	struct TCallStats { double min_t, max_t, mean_t, total_t; std::size_t n_calls; TCallStats() : min_t(0), max_t(0), mean_t(0), total_t(0), n_calls(0) {} };
	/** per-section summary with the reference's field names (min / max are not tracked here: both report the mean) */
	void getStats(std::map<std::string, TCallStats> &out) const { out.clear(); for (std::map<std::string, TCallData>::const_iterator it = m_data.begin(); it != m_data.end(); ++it) {
		TCallStats &s = out[it->first]; s.n_calls = it->second.n; s.total_t = it->second.total; s.mean_t = it->second.n ? it->second.total / it->second.n : 0; s.min_t = s.max_t = s.mean_t; } }
	void clear(bool = false) { m_data.clear(); }
	void dumpAllStats(std::size_t = 0) const { for (std::map<std::string, TCallData>::const_iterator it = m_data.begin(); it != m_data.end();
		++it) std::cout << it->first << ": calls " << it->second.n << " total " << it->second.total << " s\n"; }
	double getMeanTime(const std::string &name) const { auto it = m_data.find(name); return (it == m_data.end() || !it->second.n) ? 0 : it->second.total / it->second.n; }
private:
	bool m_enabled; std::map<std::string, TCallData> m_data;
};
} // namespace utils
} // namespace mrpt

#include <chrono>
inline void mrpt::utils::CTimeLogger::enter(const char *name) {
	if (!m_enabled) return;
	m_data[name].t0 = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
inline double mrpt::utils::CTimeLogger::leave(const char *name) {
	if (!m_enabled) return 0;
	auto &d = m_data[name];
	const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() - d.t0;
	d.n++; d.total += dt; return dt;
}

inline void mrpt::math::CMatrixDouble::loadFromTextFile(const std::string &file) {
	std::ifstream f(file.c_str()); if (!f) throw std::runtime_error("CMatrixDouble::loadFromTextFile: cannot open " + file);
	std::vector<std::vector<double> > rowsv; std::string line;
	while (std::getline(f, line)) { if (line.empty() || line[0] == '%' || line[0] == '#') continue; std::istringstream is(line); std::vector<double> r; double v; while (is >> v) r.push_back(v);
		if (!r.empty()) rowsv.push_back(r); }
	setSize(rowsv.size(), rowsv.empty() ? 0 : rowsv[0].size());
	for (std::size_t r = 0; r < nr; r++) for (std::size_t c = 0; c < nc && c < rowsv[r].size(); c++) m[r * nc + c] = rowsv[r][c];
}
inline void mrpt::math::CMatrixDouble::saveToTextFile(const std::string &file) const {
	std::ofstream f(file.c_str()); f.precision(17); for (std::size_t r = 0; r < nr; r++) { for (std::size_t c = 0; c < nc; c++) f << (c ? " " : "") << m[r * nc + c]; f << "\n"; }
}

namespace mrpt {
namespace utils {
/** INI-style configuration source/sink (mrpt::utils::CConfigFileBase): sections of name = value lines, '#' / ';' / '//' comments. */
class CConfigFileBase {
public:
	virtual ~CConfigFileBase() {}
	template <class T> T read(const std::string &section, const std::string &name, const T &def, bool fail_if_missing = false) const {
		const std::string *v = find(section, name);
		if (!v) { if (fail_if_missing) throw std::runtime_error("config value not found: [" + section + "] " + name); return def; }
		return parse(*v, def);
	}
	double read_double(const std::string &s, const std::string &n, double d, bool f = false) const { return read<double>(s, n, d, f); }
	int read_int(const std::string &s, const std::string &n, int d, bool f = false) const { return read<int>(s, n, d, f); }
	uint64_t read_uint64_t(const std::string &s, const std::string &n, uint64_t d, bool f = false) const { return read<uint64_t>(s, n, d, f); }
	bool read_bool(const std::string &s, const std::string &n, bool d, bool f = false) const { return read<bool>(s, n, d, f); }
	std::string read_string(const std::string &s, const std::string &n, const std::string &d, bool f = false) const { return read<std::string>(s, n, d, f); }
	bool sectionExists(const std::string &section) const { return m_data.count(section) != 0; }
	template <class T> void write(const std::string &section, const std::string &name, const T &value, int = -1, int = -1, const std::string &comment = std::string()) {
		std::ostringstream o; o.precision(17); o << value; m_data[section][name] = o.str(); if (!comment.empty()) m_comments[section + "\n" + name] = comment; m_dirty = true;
	}
	void write(const std::string &section, const std::string &name, const bool &value, int = -1, int = -1, const std::string &comment = std::string()) { write<std::string>(section, name,
		value ? "true" : "false", -1, -1, comment); }
protected:
	const std::string *find(const std::string &section, const std::string &name) const {
		const std::map<std::string, std::map<std::string, std::string> >::const_iterator s = m_data.find(section); if (s == m_data.end()) return NULL;
		const std::map<std::string, std::string>::const_iterator v = s->second.find(name); return v == s->second.end() ? NULL : &v->second;
	}
	template <class T> static T parse(const std::string &v, const T &) { std::istringstream is(v); T out = T(); is >> out; return out; }
	static bool parse(const std::string &v, const bool &) { return !v.empty() && (v[0] == '1' || v[0] == 't' || v[0] == 'T' || v[0] == 'y' || v[0] == 'Y'); }
	static std::string parse(const std::string &v, const std::string &) { return v; }
	std::map<std::string, std::map<std::string, std::string> > m_data; std::map<std::string, std::string> m_comments; bool m_dirty = false;
};
/** file-backed variant: parsed on construction, written back by writeNow() or on destruction if modified */
class CConfigFile : public CConfigFileBase {
public:
	explicit CConfigFile(const std::string &file) : m_file(file) {
		std::ifstream f(file.c_str()); std::string line, section;
		while (std::getline(f, line)) {
			const size_t c = line.find_first_of("#;"); if (c != std::string::npos) line.erase(c); const size_t c2 = line.find("//"); if (c2 != std::string::npos) line.erase(c2);
			const size_t a = line.find_first_not_of(" \t\r"); if (a == std::string::npos) continue; const size_t b = line.find_last_not_of(" \t\r"); line = line.substr(a, b - a + 1);
			if (line[0] == '[') { const size_t e = line.find(']'); section = line.substr(1, e == std::string::npos ? std::string::npos : e - 1); continue; }
			const size_t eq = line.find('='); if (eq == std::string::npos) continue;
			std::string k = line.substr(0, eq), v = line.substr(eq + 1);
			k.erase(k.find_last_not_of(" \t") + 1); const size_t vs = v.find_first_not_of(" \t"); v = vs == std::string::npos ? std::string() : v.substr(vs);
			m_data[section][k] = v;
		}
	}
	~CConfigFile() { if (m_dirty) writeNow(); }
	void writeNow() {
		std::ofstream f(m_file.c_str());
		for (std::map<std::string, std::map<std::string, std::string> >::const_iterator s = m_data.begin(); s != m_data.end(); ++s) {
			f << "[" << s->first << "]\n";
			for (std::map<std::string, std::string>::const_iterator v = s->second.begin(); v != s->second.end(); ++v) {
				f << v->first << " = " << v->second; const std::map<std::string, std::string>::const_iterator c = m_comments.find(s->first + "\n" + v->first);
					if (c != m_comments.end()) f << "   // " << c->second; f << "\n";
			}
			f << "\n";
		}
		m_dirty = false;
	}
private:
	std::string m_file;
};
/** in-memory variant */
class CConfigFileMemory : public CConfigFileBase {
public:
	std::string getContent() const { std::ostringstream o; for (std::map<std::string, std::map<std::string, std::string> >::const_iterator s = m_data.begin(); s != m_data.end(); ++s) {
		o << "[" << s->first << "]\n"; for (std::map<std::string, std::string>::const_iterator v = s->second.begin(); v != s->second.end(); ++v) o << v->first << " = " << v->second << "\n"; }
		return o.str(); }
};
/** parameter blocks that read / write themselves from a configuration section (mrpt::utils::CLoadableOptions) */
class CLoadableOptions {
public:
	virtual ~CLoadableOptions() {}
	virtual void loadFromConfigFile(const CConfigFileBase &source, const std::string &section) = 0;
	virtual void saveToConfigFile(CConfigFileBase &, const std::string &) const {}
	void loadFromConfigFileName(const std::string &config_file, const std::string &section) { CConfigFile f(config_file); loadFromConfigFile(f, section); }
	void saveToConfigFileName(const std::string &config_file, const std::string &section) const { CConfigFile f(config_file); saveToConfigFile(f, section); f.writeNow(); }
	virtual void dumpToConsole() const { CConfigFileMemory m; saveToConfigFile(m, ""); std::cout << m.getContent(); }
};
} // namespace utils
namespace utils {
namespace detail { inline std::vector<double> numbers_of(const std::string &s) { std::string t(s); for (char &ch : t) if (ch == '[' || ch == ']' || ch == ',' || ch == ';') ch = ' ';
	std::istringstream is(t); std::vector<double> v; double x; while (is >> x) v.push_back(x); return v; } }
inline void TCamera::loadFromConfigFile(const std::string &section, const CConfigFileBase &cfg) {
	const std::vector<double> res = detail::numbers_of(cfg.read<std::string>(section, "resolution", "", true));
	if (res.size() != 2) throw std::runtime_error("[" + section + "] resolution: expected [W H]");
	ncols = (unsigned)res[0]; nrows = (unsigned)res[1];
	m_fx = cfg.read<double>(section, "fx", 0, true); m_fy = cfg.read<double>(section, "fy", 0, true); m_cx = cfg.read<double>(section, "cx", 0, true); m_cy = cfg.read<double>(section, "cy", 0, true);
	const std::vector<double> d = detail::numbers_of(cfg.read<std::string>(section, "dist", "")); dist.setZero(); for (std::size_t i = 0; i < d.size() && i < 5; i++) dist[(int)i] = d[i];
}
inline void TStereoCamera::loadFromConfigFile(const std::string &section, const CConfigFileBase &cfg) {
	leftCamera.loadFromConfigFile(section + "_LEFT", cfg); rightCamera.loadFromConfigFile(section + "_RIGHT", cfg);
	rightCameraPose.fromString(cfg.read<std::string>(section + "_LEFT2RIGHT_POSE", "pose_quaternion", "", true));
}
} // namespace utils
#define MRPT_LOAD_CONFIG_VAR(var, type, source, section) var = (source).template read<type>((section), #var, static_cast<type>(var));
/** printf-style formatting into a std::string (mrpt::format) */
inline std::string format(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
inline std::string format(const char *fmt, ...) { char buf[2048]; va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap); return std::string(buf); }
template <class T1, class T2 = void> struct aligned_containers { typedef std::vector<T1> vector_t; typedef std::deque<T1> deque_t; typedef std::map<T1, T2> map_t; };
template <class T1> struct aligned_containers<T1, void> { typedef std::vector<T1> vector_t; typedef std::deque<T1> deque_t; };
namespace random {
/** mrpt::random::CRandomGenerator, the draws SRBA's callers use (Gaussian / uniform), on std::mt19937_64 */
class CRandomGenerator {
public:
	CRandomGenerator() : m_gen(5489u) {}
	void randomize(const uint32_t seed) { m_gen.seed(seed); } void randomize() { m_gen.seed(std::random_device()()); }
	double drawGaussian1D_normalized() { return m_norm(m_gen); }
	double drawGaussian1D(const double mean, const double std) { return mean + std * m_norm(m_gen); }
	double drawUniform(const double a, const double b) { return a + (b - a) * std::generate_canonical<double, 53>(m_gen); }
	uint32_t drawUniform32bit() { return (uint32_t)m_gen(); }
private:
	std::mt19937_64 m_gen; std::normal_distribution<double> m_norm;
};
inline CRandomGenerator &getRandomGenerator() { static CRandomGenerator g; return g; }
static CRandomGenerator &randomGenerator = getRandomGenerator();
} // namespace random
} // namespace mrpt

/** User code written for the reference spells fixed-size matrices as Eigen::Matrix<double,R,C> (e.g. parameters.obs_noise.lambda); without Eigen
 *  the same spelling maps to mrpt::math::CMatrixFixed. Skipped when the real Eigen has been included first. */
#if !defined(EIGEN_CORE_H) && !defined(EIGEN_CORE_MODULE_H) && !defined(SRBA_NO_EIGEN_ALIAS)
namespace Eigen {
template <class T, int R, int C> using Matrix = mrpt::math::CMatrixFixed<(std::size_t)R, (std::size_t)C>;
typedef Matrix<double, 2, 2> Matrix2d; typedef Matrix<double, 3, 3> Matrix3d; typedef Matrix<double, 4, 4> Matrix4d;
} // namespace Eigen
#endif
