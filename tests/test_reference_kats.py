"""The reference's two full-solve mini-problems restated on its own literal input tables (tests/golden/reference_test_tables.json, extracted by
tests/golden/make_reference_test_tables.py): MiniProblems.FixedTransformations (tests/fixed-transformations_unittest.cpp:45-171, tolerance 1e-3) and
MiniProblems.SensorAtRobot_vs_SensorDisplaced (tests/sensor-pose_unittest.cpp:216-319, tolerance 1e-2). They pin the yaw/pitch/roll and quaternion
conventions, the SE(3) exponential update, the sensor-pose algebra and the inverse sensor model through complete define_new_keyframe() runs --
on CPU with the oracle as numeric back-end, on the GPU with the HIP back-end."""
import json
import os

import numpy as np
import pytest

from srba_amd import capi, runner
import _oracle  # tests/_oracle.py: the CPU checker


def _be(backend):
    return _oracle.BACKEND if backend == "oracle" else backend

T = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_test_tables.json")))


def rot_ypr(yaw, pitch, roll):  # CPose3D(x,y,z,yaw,pitch,roll): R = Rz(yaw) Ry(pitch) Rx(roll)
    cy, sy, cp, sp, cr, sr = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch), np.cos(roll), np.sin(roll)
    return np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr], [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr], [-sp, cp * sr, cp * cr]])


def rot_quat(qr, qx, qy, qz):
    return np.array([[qr * qr + qx * qx - qy * qy - qz * qz, 2 * (qx * qy - qr * qz), 2 * (qz * qx + qr * qy)],
                     [2 * (qx * qy + qr * qz), qr * qr - qx * qx + qy * qy - qz * qz, 2 * (qy * qz - qr * qx)],
                     [2 * (qz * qx - qr * qy), 2 * (qy * qz + qr * qx), qr * qr - qx * qx - qy * qy + qz * qz]])


def ypr_of(R):
    return np.array([np.arctan2(R[1, 0], R[0, 0]), np.arcsin(-R[2, 0]), np.arctan2(R[2, 1], R[2, 2])])


def hom(t, R):
    H = np.eye(4); H[:3, :3] = R; H[:3, 3] = t; return H


def edge0_inv_pose(eng):
    fr, to, pose = eng.edges()
    assert len(fr) == 1 and (fr[0], to[0]) == (0, 1)
    return hom(pose[0, :3], pose[0, 3:].reshape(3, 3))


def run_fixed(backend, incr, inverse):
    ft = T["fixed_transformations"]
    eng = runner.landmark_engine("cart3d", backend=_be(backend), depth=3, sigma=1.0, robust=0, harvest=0, with_sensor_pose=False, max_error_per_obs_to_stop=1e-9)
    ids = [r[0] for r in ft["landmarks"]]; pts = np.array([r[1:] for r in ft["landmarks"]])
    eng.add_keyframe(ids, pts, flags=np.zeros(len(ids)))
    R = rot_ypr(*incr[3:]); t = np.array(incr[:3])
    z = (pts - t) @ R if inverse else pts @ R.T + t   # inverseComposePoint / composePoint
    info = eng.add_keyframe(ids, z, flags=np.zeros(len(ids)))
    est = edge0_inv_pose(eng)
    if inverse:
        est = np.linalg.inv(est)
    return np.abs(hom(t, R) - est).sum(), info


def run_sensor(backend, displaced):
    sp = T["sensor_pose"]
    kw = dict(depth=3, sigma=sp["std_noise_observations"], robust=1 if sp["use_robust_kernel"] else 0, harvest=0, max_error_per_obs_to_stop=1e-9)
    if displaced:
        eng = runner.landmark_engine("cart3d", backend=_be(backend), with_sensor_pose=True, sensor_pose_xyzypr=sp["sensor_pose_on_robot_xyz_ypr"], **kw)
    else:
        eng = runner.landmark_engine("cart3d", backend=_be(backend), with_sensor_pose=False, **kw)
    for key in (("obs_kf0_displaced", "obs_kf1_displaced") if displaced else ("obs_kf0", "obs_kf1")):
        tab = sp[key]
        eng.add_keyframe([r[0] for r in tab], np.array([r[1:] for r in tab]), flags=np.zeros(len(tab)))
    P = np.linalg.inv(edge0_inv_pose(eng))   # -inv_pose: pose of KF#1 wrt KF#0
    g0, g1 = sp["gt_kf0_xyz_qrxyz"], sp["gt_kf1_xyz_qrxyz"]
    GT = np.linalg.inv(hom(g0[:3], rot_quat(*g0[3:]))) @ hom(g1[:3], rot_quat(*g1[3:]))   # GT1 - GT0
    v = np.concatenate([P[:3, 3], ypr_of(P[:3, :3])]); vg = np.concatenate([GT[:3, 3], ypr_of(GT[:3, :3])])
    return np.abs(v - vg).sum()


@pytest.mark.parametrize("inverse", [False, True])
def test_fixed_transformations_oracle_backend(inverse):
    for incr in T["fixed_transformations"]["increments_xyz_ypr"]:
        err, info = run_fixed("oracle", incr, inverse)
        assert err < T["fixed_transformations"]["tolerance_sum_abs_homogeneous"], (incr, inverse, err)


@pytest.mark.parametrize("displaced", [False, True])
def test_sensor_at_robot_vs_displaced_oracle_backend(displaced):
    assert run_sensor("oracle", displaced) < T["sensor_pose"]["tolerance_sum_abs_xyzypr"]


@pytest.mark.gpu
@pytest.mark.parametrize("inverse", [False, True])
def test_fixed_transformations_hip_backend(inverse):
    for incr in T["fixed_transformations"]["increments_xyz_ypr"]:
        err, info = run_fixed("hip", incr, inverse)
        assert err < T["fixed_transformations"]["tolerance_sum_abs_homogeneous"], (incr, inverse, err)


@pytest.mark.gpu
@pytest.mark.parametrize("displaced", [False, True])
def test_sensor_at_robot_vs_displaced_hip_backend(displaced):
    assert run_sensor("hip", displaced) < T["sensor_pose"]["tolerance_sum_abs_xyzypr"]
