"""Pins the oracle (CPU) and the HIP kernels (GPU) on the BASELINE model families with the REFERENCE'S OWN literal data: the two-key-frame datasets of
examples/cpp/tutorial-srba-range-bearing-se2.cpp:50-78 (config 1 family), tutorial-srba-stereo-se3.cpp:41-147 (config 3 family) and
tutorial-srba-monocular-se3.cpp:46-131 (config 4 family), each printed next to the simulator's ground-truth poses (SURVEY App. C-5/C-6/C-7;
numbers extracted by tests/golden/make_reference_test_tables.py into tests/golden/reference_tutorial_tables.json).

Two kinds of check, both independent of this repo's own data generators:
  * residual at ground truth -- with the kf2kf edge set to the ground-truth relative pose and the landmarks placed by plain numpy geometry, the
    observation model h() of the oracle / of the device must reproduce the tutorial's SECOND key-frame observations (<= 2e-3 px, 1e-5 m / rad):
    this pins sign and frame conventions of sensor_model<>::observe_error, the sensor-pose-on-robot algebra, the stereo right-camera pose and the
    direction of k2k_edge_t::inv_pose on data that no code of this repo produced;
  * full solve -- define_new_keyframe() x 2 on the tutorial's observations recovers the ground-truth pose of KF#1 wrt KF#0 (what the tutorials print,
    tutorial-srba-range-bearing-se2.cpp:179-183).
The simulator's ground truth is the SENSOR pose (for the cameras it already contains the (-90,0,-90) deg mounting rotation); the monocular dataset was
generated with cy = 300 (datasets/tutorials_dataset-monocular.cfg:66-69) although the tutorial's main() sets cy = 320 -- the generator's value is used here."""
import json
import os

import numpy as np
import pytest

from srba_amd import capi, runner
import _oracle  # tests/_oracle.py: the CPU checker

T = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_tutorial_tables.json")))
MONO_GENERATOR_CY = 300.0


def rot_quat(qr, qx, qy, qz):
    return np.array([[qr * qr + qx * qx - qy * qy - qz * qz, 2 * (qx * qy - qr * qz), 2 * (qz * qx + qr * qy)],
                     [2 * (qx * qy + qr * qz), qr * qr - qx * qx + qy * qy - qz * qz, 2 * (qy * qz - qr * qx)],
                     [2 * (qz * qx - qr * qy), 2 * (qy * qz + qr * qx), qr * qr - qx * qx - qy * qy + qz * qz]])


def rot_ypr(yaw, pitch, roll):
    cy, sy, cp, sp, cr, sr = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch), np.cos(roll), np.sin(roll)
    return np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr], [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr], [-sp, cp * sr, cp * cr]])


def hom(g):
    H = np.eye(4); H[:3, :3] = rot_quat(*g[3:]); H[:3, 3] = g[:3]; return H


def case(kind):
    """ids/z per key-frame, ground-truth pose of robot KF#0 seen from robot KF#1 (= k2k_edge_t::inv_pose of edge 0 -> 1), landmark positions in robot KF#0."""
    if kind == "rb2d":
        d = T["range_bearing_se2"]; k0 = np.array(d["obs_kf0_id_range_yaw_pitch"]); k1 = np.array(d["obs_kf1_id_range_yaw_pitch"])
        inv_pose = np.linalg.inv(hom(d["gt_xyz_qrxyz"][1])) @ hom(d["gt_xyz_qrxyz"][0])      # the 2D simulator's ground truth is the robot pose itself
        lm = {int(r[0]): [r[1] * np.cos(r[2]), r[1] * np.sin(r[2])] for r in k0}
        return dict(kind=kind, ids=[k0[:, 0].astype(int), k1[:, 0].astype(int)], z=[k0[:, 1:3], k1[:, 1:3]], inv_pose=inv_pose, lm=lm, eng=dict(sigma=d["std_noise_observations"]), tol=1e-5)
    d = T["stereo_se3"] if kind == "stereo" else T["monocular_se3"]
    S = np.eye(4); S[:3, :3] = rot_ypr(*np.radians(d["sensor_pose_on_robot_xyz_ypr_deg"][3:]))
    C0, C1 = hom(d["gt_xyz_qrxyz"][0]), hom(d["gt_xyz_qrxyz"][1])                                 # camera poses; robot = camera (+) (-)S
    rel_cam = np.linalg.inv(C1) @ C0
    inv_pose = S @ rel_cam @ np.linalg.inv(S)
    fx, fy, cx, cy = d["camera_fx_fy_cx_cy"]
    if kind == "stereo":
        k0 = np.array(d["obs_kf0_id_lx_ly_rx_ry"]); k1 = np.array(d["obs_kf1_id_lx_ly_rx_ry"]); b = d["right_camera_pose_xyz_qrxyz"][0]
        lm = {}
        for r in k0:
            Z = fx * b / (r[1] - r[3]); lm[int(r[0])] = (S @ np.array([(r[1] - cx) * Z / fx, (r[2] - cy) * Z / fy, Z, 1.0]))[:3]
        return dict(kind=kind, ids=[k0[:, 0].astype(int), k1[:, 0].astype(int)], z=[k0[:, 1:5], k1[:, 1:5]], inv_pose=inv_pose, lm=lm, tol=2e-3,
                    eng=dict(sigma=d["std_noise_observations"], cam=(fx, fy, cx, cy), baseline=b, sensor_pose_xyzypr=list(np.radians(d["sensor_pose_on_robot_xyz_ypr_deg"]))))
    cy = MONO_GENERATOR_CY
    k0 = np.array(d["obs_kf0_id_px_py"]); k1 = np.array(d["obs_kf1_id_px_py"]); o1 = {int(r[0]): r[1:] for r in k1}
    lm = {}
    for r in k0:   # depth along the KF#0 ray: least squares on the KF#1 reprojection (Gauss-Newton on one scalar), landmarks seen once keep depth 1
        ray = np.array([(r[1] - cx) / fx, (r[2] - cy) / fy, 1.0]); dep = 1.0
        if int(r[0]) in o1:
            f = lambda t: (lambda p: np.array([cx + fx * p[0] / p[2], cy + fy * p[1] / p[2]]) - o1[int(r[0])])(rel_cam[:3, :3] @ (ray * t) + rel_cam[:3, 3])
            dep = min(np.linspace(0.5, 40, 400), key=lambda t: np.linalg.norm(f(t)))
            for _ in range(20):
                J = (f(dep + 1e-6) - f(dep - 1e-6)) / 2e-6; dep -= float(J @ f(dep)) / float(J @ J)
        lm[int(r[0])] = (S @ np.append(ray * dep, 1.0))[:3]
    return dict(kind=kind, ids=[k0[:, 0].astype(int), k1[:, 0].astype(int)], z=[k0[:, 1:3], k1[:, 1:3]], inv_pose=inv_pose, lm=lm, tol=2e-3,
                eng=dict(sigma=d["std_noise_observations"], cam=(fx, fy, cx, cy), sensor_pose_xyzypr=list(np.radians(d["sensor_pose_on_robot_xyz_ypr_deg"]))))


def engine(c, backend, **kw):
    name = {"rb2d": "rb2d", "stereo": "stereo", "mono": "mono"}[c["kind"]]
    return runner.landmark_engine(name, backend=backend, depth=3, max_error_per_obs_to_stop=1e-9, **dict(c["eng"], **kw))


def edge_storage(c):
    H = c["inv_pose"]
    if c["kind"] == "rb2d":
        return np.array([H[0, 3], H[1, 3], np.arctan2(H[1, 0], H[0, 0])])
    return np.concatenate([H[:3, 3], H[:3, :3].reshape(-1)])


def capsule_at_ground_truth(c, backend):
    """The optimize_local_area capsule of KF#1 (harvested before it is optimised), with the edge moved to the ground truth; landmarks entered at their true positions."""
    eng = engine(c, backend, robust=0, harvest=1)
    relpos = np.array([c["lm"][int(i)] for i in c["ids"][0]])
    eng.add_keyframe(c["ids"][0], c["z"][0], flags=np.full(len(c["ids"][0]), 2, np.uint8), relpos=relpos)   # first sighting with caller-supplied initial value
    eng.add_keyframe(c["ids"][1], c["z"][1], flags=np.zeros(len(c["ids"][1]), np.uint8))
    b = eng.harvest(); work = b.clone(b.n - 1, 1); work.engine = eng
    cap = work[0]
    assert cap.n_unk_edges == 1 and cap.n_unk_lms == len(set(c["ids"][0]) & set(c["ids"][1]))
    PD = capi.DIMS[work.family][3]
    np.ctypeslib.as_array(cap.edge_pose, shape=(cap.n_edges * PD,))[:PD] = edge_storage(c)
    return work


@pytest.mark.parametrize("kind", ["rb2d", "stereo", "mono"])
def test_oracle_residuals_vanish_at_tutorial_ground_truth(kind):
    c = case(kind); work = capsule_at_ground_truth(c, _oracle.BACKEND)
    a = _oracle.stage(work, 0)
    assert work[0].n_obs >= 2 * work[0].n_unk_lms and np.abs(a["resid"]).max() < c["tol"], np.abs(a["resid"]).max()


def recovered_error(c, backend):
    eng = engine(c, backend, robust=(1 if c["kind"] == "stereo" else 0), harvest=0)
    for k in range(2):
        eng.add_keyframe(c["ids"][k], c["z"][k], flags=np.zeros(len(c["ids"][k]), np.uint8))
    fr, to, pose = eng.edges()
    assert len(fr) == 1 and (fr[0], to[0]) == (0, 1)
    return np.abs(pose[0] - edge_storage(c)).max()


@pytest.mark.parametrize("kind", ["rb2d", "stereo"])
def test_oracle_backend_recovers_tutorial_ground_truth(kind):
    assert recovered_error(case(kind), _oracle.BACKEND) < (1e-5 if kind == "rb2d" else 1e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["rb2d", "stereo", "mono"])
def test_hip_residuals_vanish_at_tutorial_ground_truth(kind):
    c = case(kind); work = capsule_at_ground_truth(c, "hip")
    ctx = runner.HipContext(work.params); ctx.upload(work)
    assert ctx.lib.srba_hip_update_spantree(ctx.ctx, 0) == 0
    chi2 = np.zeros(1); assert ctx.lib.srba_hip_eval_residuals(ctx.ctx, chi2.ctypes.data_as(capi.PF64)) == 0
    res = ctx.debug(0); ctx.close()
    assert np.abs(res).max() < c["tol"] and chi2[0] < work[0].n_obs * c["tol"] ** 2 * 4


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["rb2d", "stereo"])
def test_hip_backend_recovers_tutorial_ground_truth(kind):
    assert recovered_error(case(kind), "hip") < (1e-5 if kind == "rb2d" else 1e-3)
