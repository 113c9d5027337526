/*
 * models.h -- pose / landmark / observation parameterisations of the host front-end.
 *
 * Same tag types and data fields as the reference's include/srba/models/{kf2kf_poses.h:26-38, landmarks.h:25-151,
 * observations_*.h, sensors.h}.  The numeric sensor models (observe_error, eval_jacob_dh_dx) run on the GPU
 * (srba_amd/csrc); the host keeps only what the graph layer needs:
 *   - inverse_sensor_model   (first-seen landmark initialisation; sensors.h:121-141,291-315,396-407,726-736,825-834)
 *   - landmark_matcher<OBS>  (initial relative pose of new kf2kf edges; observations_*.h)
 *   - the mapping of the template triple to the device "family" id and of TObservationParams to srba_hip_params.
 */
#pragma once
#include "../srba_hip.h"
#include "srba_types.h"

namespace srba {
enum landmark_jacob_family_t { jacob_point_landmark, jacob_relpose_landmark };

namespace kf2kf_poses {
struct SE3 { static const size_t REL_POSE_DIMS = 6; typedef mrpt::poses::CPose3D pose_t; typedef mrpt::poses::SE_traits<3> se_traits_t; };
struct SE2 { static const size_t REL_POSE_DIMS = 3; typedef mrpt::poses::CPose2D pose_t; typedef mrpt::poses::SE_traits<2> se_traits_t; };
} // namespace kf2kf_poses

namespace landmarks {
struct Euclidean3D {
	static const size_t LM_DIMS = 3; static const landmark_jacob_family_t jacob_family = jacob_point_landmark;
	template <class POSE, class VECTOR> static void composePosePoint(VECTOR &pt, const POSE &pose) { pose.composePoint(pt[0], pt[1], pt[2], pt[0], pt[1], pt[2]); }
};
struct Euclidean2D {
	static const size_t LM_DIMS = 2; static const landmark_jacob_family_t jacob_family = jacob_point_landmark;
	template <class POSE, class VECTOR> static void composePosePoint(VECTOR &pt, const POSE &pose) { double lx, ly, lz; pose.composePoint(pt[0], pt[1], 0, lx, ly, lz); pt[0] = lx; pt[1] = ly; }
};
struct RelativePoses2D {
	static const size_t LM_DIMS = 3; static const landmark_jacob_family_t jacob_family = jacob_relpose_landmark;
	template <class POSE, class VECTOR> static void composePosePoint(VECTOR &, const POSE &) {}
};
struct RelativePoses3D { // SE(3) relative poses as "landmarks" (graph-SLAM in 3D): x y z yaw pitch roll
	static const size_t LM_DIMS = 6; static const landmark_jacob_family_t jacob_family = jacob_relpose_landmark;
	template <class POSE, class VECTOR> static void composePosePoint(VECTOR &, const POSE &) {}
};
} // namespace landmarks

namespace observations {
template <class OBS> struct landmark_matcher;

struct MonocularCamera {
	static const size_t OBS_DIMS = 2;
	struct obs_data_t { mrpt::utils::TPixelCoordf px; template <class A> void getAsArray(A &o) const { o[0] = px.x; o[1] = px.y; } };
	struct TObservationParams { mrpt::utils::TCamera camera_calib; };
};
struct StereoCamera {
	static const size_t OBS_DIMS = 4;
	struct obs_data_t { mrpt::utils::TPixelCoordf left_px, right_px; template <class A> void getAsArray(A &o) const { o[0] = left_px.x; o[1] = left_px.y; o[2] = right_px.x; o[3] = right_px.y; } };
	struct TObservationParams { mrpt::utils::TStereoCamera camera_calib; };
};
struct Cartesian_3D {
	static const size_t OBS_DIMS = 3;
	struct obs_data_t { mrpt::math::TPoint3D pt; template <class A> void getAsArray(A &o) const { o[0] = pt.x; o[1] = pt.y; o[2] = pt.z; } };
	struct TObservationParams {};
};
struct RangeBearing_3D {
	static const size_t OBS_DIMS = 3;
	struct obs_data_t { double range, yaw, pitch; obs_data_t() : range(0), yaw(0), pitch(0) {} template <class A> void getAsArray(A &o) const { o[0] = range; o[1] = yaw; o[2] = pitch; } };
	struct TObservationParams {};
};
struct Cartesian_2D {
	static const size_t OBS_DIMS = 2;
	struct obs_data_t { mrpt::math::TPoint2D pt; template <class A> void getAsArray(A &o) const { o[0] = pt.x; o[1] = pt.y; } };
	struct TObservationParams {};
};
struct RangeBearing_2D {
	static const size_t OBS_DIMS = 2;
	struct obs_data_t { double range, yaw; obs_data_t() : range(0), yaw(0) {} template <class A> void getAsArray(A &o) const { o[0] = range; o[1] = yaw; } };
	struct TObservationParams {};
};
struct RelativePoses_2D {
	static const size_t OBS_DIMS = 3;
	struct obs_data_t { double x, y, yaw; obs_data_t() : x(0), y(0), yaw(0) {} template <class A> void getAsArray(A &o) const { o[0] = x; o[1] = y; o[2] = yaw; } };
	struct TObservationParams {};
};

struct RelativePoses_3D {
	static const size_t OBS_DIMS = 6;
	struct obs_data_t { double x, y, z, yaw, pitch, roll; obs_data_t() : x(0), y(0), z(0), yaw(0), pitch(0), roll(0) {} template <class A> void getAsArray(A &o) const { o[0] = x; o[1] = y; o[2] = z;
		o[3] = yaw; o[4] = pitch; o[5] = roll; } };
	struct TObservationParams {};
};

namespace detail {
template <class POSE> bool pose_from_matches(const mrpt::utils::TMatchingPairList &matches, POSE &out) {
	if (POSE::rotation_dimensions == 2) { mrpt::math::TPose2D f; if (!mrpt::tfest::se2_l2(matches, f)) return false; out = POSE(mrpt::poses::CPose2D(f)); }
	else { mrpt::poses::CPose3DQuat f; double s; if (!mrpt::tfest::se3_l2(matches, f, s)) return false; out = POSE(mrpt::poses::CPose3D(f)); }
	return true;
}
} // namespace detail

/** observations_RelativePoses_2D.h:46-71: use the observation of one KF made from the other (its own is exactly 0) */
template <> struct landmark_matcher<RelativePoses_2D> {
	template <class POSE> static bool find_relative_pose(const std::vector<RelativePoses_2D::obs_data_t> &new_kf_obs, const std::vector<RelativePoses_2D::obs_data_t> &old_kf_obs,
		const RelativePoses_2D::TObservationParams &, POSE &pose_new_kf_wrt_old_kf) {
		for (size_t i = 0; i < new_kf_obs.size(); i++) {
			const RelativePoses_2D::obs_data_t &kf0 = new_kf_obs[i], &kf1 = old_kf_obs[i];
			if ((kf0.x != 0 || kf0.y != 0 || kf0.yaw != 0) && (kf1.x != 0 || kf1.y != 0 || kf1.yaw != 0)) continue;
			const mrpt::poses::CPose2D new_obs(kf0.x, kf0.y, kf0.yaw), old_obs(kf1.x, kf1.y, kf1.yaw);
			pose_new_kf_wrt_old_kf = POSE(old_obs - new_obs);
			return true;
		}
		return false;
	}
};
/** observations_RelativePoses_3D.h:47-72: as in 2D, one of the two key-frames observes itself at exactly the null pose */
template <> struct landmark_matcher<RelativePoses_3D> {
	template <class POSE> static bool find_relative_pose(const std::vector<RelativePoses_3D::obs_data_t> &new_kf_obs, const std::vector<RelativePoses_3D::obs_data_t> &old_kf_obs,
		const RelativePoses_3D::TObservationParams &, POSE &pose_new_kf_wrt_old_kf) {
		struct is_null { static bool of(const RelativePoses_3D::obs_data_t &o) { return o.x == 0 && o.y == 0 && o.z == 0 && o.yaw == 0 && o.pitch == 0 && o.roll == 0; } };
		for (size_t i = 0; i < new_kf_obs.size(); i++) {
			const RelativePoses_3D::obs_data_t &n = new_kf_obs[i], &o = old_kf_obs[i];
			if (!is_null::of(n) && !is_null::of(o)) continue;
			pose_new_kf_wrt_old_kf = POSE(mrpt::poses::CPose3D(o.x, o.y, o.z, o.yaw, o.pitch, o.roll) - mrpt::poses::CPose3D(n.x, n.y, n.z, n.yaw, n.pitch, n.roll));
			return true;
		}
		return false;
	}
};
/** observations_RangeBearing_2D.h:46-82 */
template <> struct landmark_matcher<RangeBearing_2D> {
	template <class POSE> static bool find_relative_pose(const std::vector<RangeBearing_2D::obs_data_t> &n, const std::vector<RangeBearing_2D::obs_data_t> &o,
		const RangeBearing_2D::TObservationParams &, POSE &out) {
		mrpt::utils::TMatchingPairList m;
		for (size_t i = 0; i < n.size(); i++) m.push_back(mrpt::utils::TMatchingPair(i, i, o[i].range * std::cos(o[i].yaw), o[i].range * std::sin(o[i].yaw), 0, n[i].range * std::cos(n[i].yaw),
			n[i].range * std::sin(n[i].yaw), 0));
		return detail::pose_from_matches(m, out);
	}
};
template <> struct landmark_matcher<Cartesian_2D> {
	template <class POSE> static bool find_relative_pose(const std::vector<Cartesian_2D::obs_data_t> &n, const std::vector<Cartesian_2D::obs_data_t> &o, const Cartesian_2D::TObservationParams &,
		POSE &out) {
		mrpt::utils::TMatchingPairList m;
		for (size_t i = 0; i < n.size(); i++) m.push_back(mrpt::utils::TMatchingPair(i, i, o[i].pt.x, o[i].pt.y, 0, n[i].pt.x, n[i].pt.y, 0));
		return detail::pose_from_matches(m, out);
	}
};
/** observations_RangeBearing_3D.h:47-86 */
template <> struct landmark_matcher<RangeBearing_3D> {
	template <class POSE> static bool find_relative_pose(const std::vector<RangeBearing_3D::obs_data_t> &n, const std::vector<RangeBearing_3D::obs_data_t> &o,
		const RangeBearing_3D::TObservationParams &, POSE &out) {
		mrpt::utils::TMatchingPairList m;
		for (size_t i = 0; i < n.size(); i++) m.push_back(mrpt::utils::TMatchingPair(i, i, o[i].range * std::cos(o[i].yaw) * std::cos(o[i].pitch),
			o[i].range * std::sin(o[i].yaw) * std::cos(o[i].pitch), -o[i].range * std::sin(o[i].pitch),
			n[i].range * std::cos(n[i].yaw) * std::cos(n[i].pitch), n[i].range * std::sin(n[i].yaw) * std::cos(n[i].pitch), -n[i].range * std::sin(n[i].pitch)));
		return detail::pose_from_matches(m, out);
	}
};
/** observations_Cartesian_3D.h:45-79 */
template <> struct landmark_matcher<Cartesian_3D> {
	template <class POSE> static bool find_relative_pose(const std::vector<Cartesian_3D::obs_data_t> &n, const std::vector<Cartesian_3D::obs_data_t> &o, const Cartesian_3D::TObservationParams &,
		POSE &out) {
		mrpt::utils::TMatchingPairList m;
		for (size_t i = 0; i < n.size(); i++) m.push_back(mrpt::utils::TMatchingPair(i, i, o[i].pt.x, o[i].pt.y, o[i].pt.z, n[i].pt.x, n[i].pt.y, n[i].pt.z));
		return detail::pose_from_matches(m, out);
	}
};
/** observations_StereoCamera.h:52-110: triangulate both sets, then least-squares alignment */
template <> struct landmark_matcher<StereoCamera> {
	template <class POSE> static bool find_relative_pose(const std::vector<StereoCamera::obs_data_t> &n, const std::vector<StereoCamera::obs_data_t> &o, const StereoCamera::TObservationParams &p,
		POSE &out) {
		const double cx = p.camera_calib.leftCamera.cx(), cy = p.camera_calib.leftCamera.cy(), b = p.camera_calib.rightCameraPose.x(), f = p.camera_calib.leftCamera.fx();
		mrpt::utils::TMatchingPairList m;
		for (size_t i = 0; i < n.size(); i++) {
			const double d_old = o[i].left_px.x - o[i].right_px.x; if (d_old <= .0) continue;
			const mrpt::math::TPoint3D po((o[i].left_px.x - cx) * b / d_old, (o[i].left_px.y - cy) * b / d_old, f * b / d_old);
			const double d_new = n[i].left_px.x - n[i].right_px.x; if (d_new <= .0) continue;
			const mrpt::math::TPoint3D pn((n[i].left_px.x - cx) * b / d_new, (n[i].left_px.y - cy) * b / d_new, f * b / d_new);
			m.push_back(mrpt::utils::TMatchingPair(i, i, po.x, po.y, po.z, pn.x, pn.y, pn.z));
		}
		return detail::pose_from_matches(m, out);
	}
};
/** observations_MonocularCamera.h:45-58: no metric relative pose from two monocular views */
template <> struct landmark_matcher<MonocularCamera> {
	template <class POSE> static bool find_relative_pose(const std::vector<MonocularCamera::obs_data_t> &, const std::vector<MonocularCamera::obs_data_t> &,
		const MonocularCamera::TObservationParams &, POSE &) { return false; }
};
} // namespace observations

// ---------------------------------------------------------------------------------------------
// sensor_model<LM,OBS>: host part (inverse model) + device family id + parameter marshalling
// ---------------------------------------------------------------------------------------------
template <> struct sensor_model<landmarks::Euclidean3D, observations::MonocularCamera> {
	static const int family = SRBA_SE3_MONO;
	template <class LM, class OBSD, class PRM> static void inverse_sensor_model(LM &out, const OBSD &obs, const PRM &p) { // sensors.h:121-141
		out[0] = (obs.px.x - p.camera_calib.cx()) / p.camera_calib.fx(); out[1] = (obs.px.y - p.camera_calib.cy()) / p.camera_calib.fy(); out[2] = 1;
	}
	template <class PRM> static void fill_params(srba_hip_params &hp, const PRM &p) { hp.cam_left[0] = p.camera_calib.fx(); hp.cam_left[1] = p.camera_calib.fy(); hp.cam_left[2] = p.camera_calib.cx();
		hp.cam_left[3] = p.camera_calib.cy(); }
};
template <> struct sensor_model<landmarks::Euclidean3D, observations::StereoCamera> {
	static const int family = SRBA_SE3_STEREO;
	template <class LM, class OBSD, class PRM> static void inverse_sensor_model(LM &out, const OBSD &obs, const PRM &p) { // sensors.h:291-315
		const double fxl = p.camera_calib.leftCamera.fx(), fyl = p.camera_calib.leftCamera.fy(), cxl = p.camera_calib.leftCamera.cx(), cyl = p.camera_calib.leftCamera.cy();
		const double disparity = std::max(0.001f, obs.left_px.x - obs.right_px.x);
		const double baseline = p.camera_calib.rightCameraPose.x(); ASSERT_(baseline != 0);
		const double Z = fxl * baseline / disparity;
		out[0] = (obs.left_px.x - cxl) * Z / fxl; out[1] = (obs.left_px.y - cyl) * Z / fyl; out[2] = Z;
	}
	template <class PRM> static void fill_params(srba_hip_params &hp, const PRM &p) {
		const mrpt::utils::TCamera &l = p.camera_calib.leftCamera, &r = p.camera_calib.rightCamera;
		hp.cam_left[0] = l.fx(); hp.cam_left[1] = l.fy(); hp.cam_left[2] = l.cx(); hp.cam_left[3] = l.cy();
		hp.cam_right[0] = r.fx(); hp.cam_right[1] = r.fy(); hp.cam_right[2] = r.cx(); hp.cam_right[3] = r.cy();
		const mrpt::poses::CPose3DQuat &q = p.camera_calib.rightCameraPose;
		hp.right_cam_pose[0] = q.x(); hp.right_cam_pose[1] = q.y(); hp.right_cam_pose[2] = q.z();
		hp.right_cam_pose[3] = q.quat().r(); hp.right_cam_pose[4] = q.quat().x(); hp.right_cam_pose[5] = q.quat().y(); hp.right_cam_pose[6] = q.quat().z();
	}
};
template <> struct sensor_model<landmarks::Euclidean3D, observations::Cartesian_3D> {
	static const int family = SRBA_SE3_CART3D;
	template <class LM, class OBSD, class PRM> static void inverse_sensor_model(LM &out, const OBSD &obs, const PRM &) { out[0] = obs.pt.x; out[1] = obs.pt.y; out[2] = obs.pt.z; } // sensors.h:396-407
	template <class PRM> static void fill_params(srba_hip_params &, const PRM &) {}
};
template <> struct sensor_model<landmarks::Euclidean3D, observations::RangeBearing_3D> {
	static const int family = SRBA_SE3_RB3D;
	template <class LM, class OBSD, class PRM> static void inverse_sensor_model(LM &out, const OBSD &obs, const PRM &) { // sensors.h:620-634
		const double cy = std::cos(obs.yaw), sy = std::sin(obs.yaw), cp = std::cos(obs.pitch), sp = std::sin(obs.pitch);
		out[0] = obs.range * cy * cp; out[1] = obs.range * sy * cp; out[2] = -obs.range * sp;
	}
	template <class PRM> static void fill_params(srba_hip_params &, const PRM &) {}
};
template <> struct sensor_model<landmarks::Euclidean2D, observations::Cartesian_2D> {
	static const int family = SRBA_SE2_CART2D;
	template <class LM, class OBSD, class PRM> static void inverse_sensor_model(LM &out, const OBSD &obs, const PRM &) { out[0] = obs.pt.x; out[1] = obs.pt.y; }
	template <class PRM> static void fill_params(srba_hip_params &, const PRM &) {}
};
template <> struct sensor_model<landmarks::Euclidean2D, observations::RangeBearing_2D> {
	static const int family = SRBA_SE2_RB2D;
	template <class LM, class OBSD, class PRM> static void inverse_sensor_model(LM &out, const OBSD &obs, const PRM &) { out[0] = obs.range * std::cos(obs.yaw);
		out[1] = obs.range * std::sin(obs.yaw); } // sensors.h:726-736
	template <class PRM> static void fill_params(srba_hip_params &, const PRM &) {}
};
template <> struct sensor_model<landmarks::RelativePoses3D, observations::RelativePoses_3D> {
	static const int family = SRBA_SE3_RELPOSE3D;
	template <class LM, class OBSD, class PRM> static void inverse_sensor_model(LM &out, const OBSD &obs, const PRM &) { out[0] = obs.x; out[1] = obs.y; out[2] = obs.z; out[3] = obs.yaw;
		out[4] = obs.pitch; out[5] = obs.roll; } // sensors.h:915-927
	template <class PRM> static void fill_params(srba_hip_params &, const PRM &) {}
};
template <> struct sensor_model<landmarks::RelativePoses2D, observations::RelativePoses_2D> {
	static const int family = SRBA_SE2_RELPOSE2D;
	template <class LM, class OBSD, class PRM> static void inverse_sensor_model(LM &out, const OBSD &obs, const PRM &) { out[0] = obs.x; out[1] = obs.y; out[2] = obs.yaw; } // sensors.h:825-834
	template <class PRM> static void fill_params(srba_hip_params &, const PRM &) {}
};

/** Device family of a <key-frame pose, landmark, observation> triple: the sensor model's family, except where the pose parameterisation changes the kernels */
template <class KF, class LM, class OBS> struct device_family { static const int value = sensor_model<LM, OBS>::family; };
template <> struct device_family<kf2kf_poses::SE2, landmarks::Euclidean3D, observations::StereoCamera> { static const int value = SRBA_SE2_STEREO; }; // SE(2) key-frames,
	// 3D points (tutorial-srba-stereo-se2.cpp)
/** REL_POSE_DIMS the kernels of a family are written for (checked against KF::REL_POSE_DIMS at compile time in RbaEngine) */
constexpr int family_pose_dims(int family) { return (family == SRBA_SE2_RELPOSE2D || family == SRBA_SE2_RB2D || family == SRBA_SE2_CART2D || family == SRBA_SE2_STEREO) ? 3 : 6; }

} // namespace srba
