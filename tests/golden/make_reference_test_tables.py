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
