"""Input data for tests and benchmarks.

* Literal datasets quoted from the reference's own tests/tutorials (DATA, not code; origin cited per item).
* Deterministic synthetic generators for the BASELINE.json configurations (the reference ships only generator
  configs for the external RWT tool, datasets/README.txt:1-5, so the 30k-keyframe graph-SLAM data is produced here).

A dataset is a list with one entry per keyframe: dict(feat_ids uint64[n], z float64[n,O], flags uint8[n], relpos float64[n,L] or None)
in the order the reference apps feed define_new_keyframe (apps/srba-slam/srba-run-generic-impl.h:365-461).
"""
import math
import numpy as np

FLAG_FIXED, FLAG_INIT = 1, 2

# tests/submaps_edge_init_values.cpp:61-79 -- (current_kf, observed_kf, x, y, yaw): pose of observed_kf seen from current_kf
C1_SUBMAPS = [(1, 0, -1.0, 0.0, 0.0), (2, 1, -1.0, 0.0, 0.0), (3, 2, -1.0, 0.0, 0.0), (4, 3, -1.0, 0.0, 0.0), (5, 4, -1.0, 0.0, 0.0), (6, 5, -1.0, 0.0, 0.0),
              (7, 6, -1.0, 0.0, 0.0), (8, 7, -1.0, 0.0, 0.0), (9, 8, -1.0, 0.0, 0.0), (10, 9, -1.0, 0.0, 0.0), (11, 10, -1.0, 0.0, 0.0), (11, 1, -10.05, 0.0, 0.0),
              (12, 11, -1.0, 0.0, 0.0), (13, 12, -1.0, 0.0, 0.0), (14, 13, -1.0, 0.0, 0.0), (15, 14, -1.0, 0.0, 0.0), (16, 15, -1.0, 0.0, 0.0)]

# examples/cpp/tutorial-srba-relative-graph-slam-se2.cpp:48-76 -- (current_kf, observed_kf, x, y, yaw)
C2_TUTORIAL_SE2 = [
    (1, 0, -1.78055512, 1.11331694, -0.37399920), (2, 1, -1.71545942, 2.05914961, -0.37399882), (2, 0, -2.96619160, 3.74601658, -0.74799802),
    (3, 2, -1.15014065, 2.45631509, -0.37399920), (4, 3, -0.71839088, 2.17858845, -0.37399920), (4, 2, -0.89163376, 4.88530127, -0.74799839),
    (5, 4, -1.33870852, 1.73631597, -0.37399882), (5, 3, -1.21151269, 4.02676447, -0.74799802), (6, 5, -1.67977719, 2.03565806, -0.37399920),
    (6, 4, -2.29159821, 4.14103420, -0.74799802), (7, 6, -1.49006905, 2.30876608, -0.37399920), (8, 7, -1.15992524, 2.21845386, -0.37399882),
    (9, 8, -1.28889269, 1.78614744, -0.37399920), (9, 7, -1.55814427, 4.27501619, -0.74799802), (10, 9, -1.67026750, 1.96210498, -0.37399920),
    (10, 8, -2.21751078, 4.09566815, -0.74799839), (11, 10, -1.55210516, 2.27651848, -0.37399882), (12, 11, -1.21625554, 2.27164636, -0.37399920),
    (13, 12, -1.45455725, 1.51179033, -0.22440012), (13, 11, -2.13482825, 3.99712454, -0.59839931), (14, 13, -2.36655195, 0.41536284, 0.00000000),
    (14, 12, -3.82110920, 1.92715317, -0.22440012), (15, 14, -2.74448431, -0.11373775, 0.00000000), (15, 0, 4.16910212, 0.67638546, 1.57079633),
    (16, 0, 1.58658626, 0.30349575, 1.57079633), (16, 15, -2.58251586, -0.37288971, 0.00000000), (16, 1, 1.97243380, 2.36770815, 1.94479552)]


def graph_slam_from_entries(entries, sigma_xy=0.0, sigma_yaw=0.0, seed=0):
    """(current, observed, x, y, yaw) rows -> per-keyframe observation lists, the way the reference test/tutorial builds them
    (tests/submaps_edge_init_values.cpp:115-146): first the keyframe's own fixed "fake landmark", then its relative-pose observations."""
    rng = np.random.RandomState(seed)
    n_kf = max(e[0] for e in entries) + 1
    out = []
    for kf in range(n_kf):
        rows = [e for e in entries if e[0] == kf]
        ids = [kf] + [e[1] for e in rows]
        z = [[0.0, 0.0, 0.0]] + [[e[2] + sigma_xy * rng.randn(), e[3] + sigma_xy * rng.randn(), e[4] + sigma_yaw * rng.randn()] for e in rows]
        flags = [FLAG_FIXED] + [0] * len(rows)
        out.append(dict(feat_ids=np.array(ids, np.uint64), z=np.array(z, np.float64).reshape(-1, 3), flags=np.array(flags, np.uint8), relpos=None))
    return out


def _compose2(a, b):
    c, s = math.cos(a[2]), math.sin(a[2])
    return (a[0] + b[0] * c - b[1] * s, a[1] + b[0] * s + b[1] * c, a[2] + b[2])


def _inv_compose2(a, b):
    """a (-) b : pose a as seen from b"""
    c, s = math.cos(b[2]), math.sin(b[2])
    dx, dy = a[0] - b[0], a[1] - b[1]
    ang = (a[2] - b[2] + math.pi) % (2 * math.pi) - math.pi
    return (dx * c + dy * s, -dx * s + dy * c, ang)


def manhattan_path(n_kf, seed=1, block=100.0, grid=16, step=2.0):
    """Ground-truth SE2 path: random walk on a `grid` x `grid` Manhattan street map with `block`-metre blocks, <= `step` m and <= 30 deg
    per keyframe (datasets/world-2d-30k-rel-graph-slam.cfg:26-27 max_step_lin=2.0, max_step_ang=30)."""
    rng = np.random.RandomState(seed)
    ix, iy = grid // 2, grid // 2  # current intersection
    heading = 0  # 0:+x 1:+y 2:-x 3:-y
    poses = []
    x, y, th = ix * block, iy * block, 0.0
    dirs = [(1, 0), (0, 1), (-1, 0), (0, -1)]
    while len(poses) < n_kf:
        # choose the next street: no U-turn, stay inside the map
        cand = [h for h in ((heading + k) % 4 for k in (0, 1, 3)) if 0 <= ix + dirs[h][0] <= grid and 0 <= iy + dirs[h][1] <= grid]
        if not cand:
            cand = [(heading + 2) % 4]
        new_h = cand[rng.randint(len(cand))]
        # turn in place in <= 30 deg increments, creeping 0.5 m forward per keyframe
        dth = ((new_h - heading + 1) % 4 - 1) * (math.pi / 2) if new_h != (heading + 2) % 4 else math.pi
        nturn = int(round(abs(dth) / (math.pi / 6)))
        for _ in range(nturn):
            th += dth / nturn
            x += 0.5 * math.cos(th); y += 0.5 * math.sin(th)
            poses.append((x, y, th))
        heading = new_h
        tx, ty = (ix + dirs[heading][0]) * block, (iy + dirs[heading][1]) * block
        th = heading * math.pi / 2
        dist = math.hypot(tx - x, ty - y)
        nstep = max(1, int(math.ceil(dist / step)))
        for k in range(1, nstep + 1):
            poses.append((x + (tx - x) * k / nstep, y + (ty - y) * k / nstep, th))
        x, y = tx, ty
        ix += dirs[heading][0]; iy += dirs[heading][1]
    return poses[:n_kf]


def manhattan_tour(n_kf, block=100.0, step=2.0):
    """Ground-truth SE2 path of the benchmark map: a serpentine tour over all east-west streets of a square Manhattan grid followed by a
    serpentine over all north-south streets.  The first half is pure exploration, the second half crosses an already mapped street at every
    intersection, i.e. one loop-closure event per `block` metres -- the "corridor loop walk" of SURVEY 8d for which the reference's RWT path
    (datasets/world-2d-30k-rel-graph-slam.cfg:22) is the model.  The grid size is chosen so that the tour is just long enough for n_kf keyframes."""
    G = 1
    while ((G + 1) * (G + 1) + G * (G + 2)) * block / step < n_kf:
        G += 1
    wp = []
    for r in range(G + 1):  # phase 1: east-west streets y = r*block, alternating direction, joined along x = 0 / x = G*block
        xs = (0.0, G * block) if r % 2 == 0 else (G * block, 0.0)
        wp += [(xs[0], r * block), (xs[1], r * block)]
    # phase 2: north-south streets half a block off the phase-1 connectors (x = (c+0.5)*block), running from half a block above the
    # top row to half a block below the bottom row: every crossing of a phase-1 street is perpendicular, the joins are new ground.
    end_x = wp[-1][0]
    cols = list(range(G)) if end_x == 0.0 else list(range(G - 1, -1, -1))
    top, bot = (G + 0.5) * block, -0.5 * block
    wp.append((end_x, top))
    down = True
    for cidx in cols:
        xc = (cidx + 0.5) * block
        wp += [(xc, top if down else bot), (xc, bot if down else top)]
        down = not down
    poses = []; x, y = wp[0]; th = 0.0
    for (tx, ty) in wp[1:]:
        d = math.hypot(tx - x, ty - y)
        if d < 1e-9:
            continue
        new_th = math.atan2(ty - y, tx - x)
        dth = (new_th - th + math.pi) % (2 * math.pi) - math.pi
        nturn = int(math.ceil(abs(dth) / (math.pi / 6) - 1e-9))
        for _ in range(nturn):  # turn in <= 30 degree increments while creeping forward
            th += dth / nturn
            x += 0.25 * math.cos(th); y += 0.25 * math.sin(th)
            poses.append((x, y, th))
        th = new_th
        d = math.hypot(tx - x, ty - y); nstep = max(1, int(math.ceil(d / step)))
        sx, sy = x, y
        for k in range(1, nstep + 1):
            poses.append((sx + (tx - sx) * k / nstep, sy + (ty - sy) * k / nstep, th))
        x, y = tx, ty
        if len(poses) >= n_kf:
            break
    assert len(poses) >= n_kf
    return poses[:n_kf]


def graph_slam_se2(n_kf=30000, seed=1, sigma_xy=1e-3, sigma_yaw_deg=0.2, max_range=9.0, grid=16, block=100.0, path="random"):
    """cfg2 of BASELINE.md: SE2 relative graph-SLAM, every keyframe observes all EARLIER keyframes within `max_range` metres
    ("relative_poses" sensor, datasets/world-2d-30k-rel-graph-slam.cfg:42-45, maxRange 9 m), noise 0.001 m / 0.2 deg (README.md:68)."""
    rng = np.random.RandomState(seed + 7919)
    gt = manhattan_tour(n_kf, block=block) if path == "tour" else manhattan_path(n_kf, seed=seed, block=block, grid=grid)
    sig_yaw = math.radians(sigma_yaw_deg)
    cell = max_range
    buckets = {}
    out = []
    for kf, p in enumerate(gt):
        cx, cy = int(math.floor(p[0] / cell)), int(math.floor(p[1] / cell))
        near = []
        for dx in (-1, 0, 1):
            for dy in (-1, 0, 1):
                for j in buckets.get((cx + dx, cy + dy), ()):
                    q = gt[j]
                    if math.hypot(q[0] - p[0], q[1] - p[1]) <= max_range:
                        near.append(j)
        near.sort()
        ids = [kf] + near
        z = [[0.0, 0.0, 0.0]]
        for j in near:
            r = _inv_compose2(gt[j], p)
            z.append([r[0] + sigma_xy * rng.randn(), r[1] + sigma_xy * rng.randn(), r[2] + sig_yaw * rng.randn()])
        flags = [FLAG_FIXED] + [0] * len(near)
        out.append(dict(feat_ids=np.array(ids, np.uint64), z=np.array(z, np.float64).reshape(-1, 3), flags=np.array(flags, np.uint8), relpos=None))
        buckets.setdefault((cx, cy), []).append(kf)
    return out


def graph_slam_lambda(sigma_xy=1e-3, sigma_yaw_deg=0.2):
    """Information matrix of apps/srba-slam/CDatasetParser_RelGraphSLAM2D.h:45-52"""
    s = math.radians(sigma_yaw_deg)
    return np.diag([1.0 / sigma_xy ** 2, 1.0 / sigma_xy ** 2, 1.0 / s ** 2])


# ------------------------------------------------------------------------------------------------- SE3 / point-landmark generators
def rot_ypr(yaw, pitch, roll):
    """R = Rz(yaw) Ry(pitch) Rx(roll) (MRPT CPose3D convention)"""
    cy, sy, cp, sp, cr, sr = math.cos(yaw), math.sin(yaw), math.cos(pitch), math.sin(pitch), math.cos(roll), math.sin(roll)
    return np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr], [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr], [-sp, cp * sr, cp * cr]])


def pose3(x, y, z, yaw=0.0, pitch=0.0, roll=0.0):
    T = np.eye(4); T[:3, :3] = rot_ypr(yaw, pitch, roll); T[:3, 3] = (x, y, z); return T


def pose3_to_pd(T):
    """4x4 -> the PD=12 boundary layout [t, R row-major]"""
    return np.concatenate([T[:3, 3], T[:3, :3].reshape(-1)])


CAMERA_ON_ROBOT = (0.0, 0.0, 0.0, math.radians(-90), 0.0, math.radians(-90))  # apps/srba-slam/srba-run-generic-impl.h:79 ; tutorial-srba-stereo-se3.cpp:147


def spiral_path_se3(n_kf, seed=1, radius=6.0, step=0.35):
    """Forward-looking spiral inside a room: smooth 3D trajectory with gentle pitch/roll."""
    rng = np.random.RandomState(seed)
    out = []
    for k in range(n_kf):
        a = step * k / radius
        r = radius * (0.6 + 0.4 * math.cos(0.13 * a))
        x, y, z = r * math.cos(a), r * math.sin(a), 0.6 * math.sin(0.37 * a)
        yaw = a + math.pi / 2 + 0.05 * rng.randn()
        out.append(pose3(x, y, z, yaw, 0.05 * math.sin(0.5 * a), 0.04 * math.cos(0.3 * a)))
    return out


def landmarks_dataset_se3(kind="cart3d", n_kf=40, n_lm=400, seed=1, max_range=5.0, noise=0.0, cam=(200.0, 150.0, 512.0, 384.0), baseline=0.2, with_sensor_pose=None,
                          known_first=0, room=9.0, init_from_gt_noise=None):
    """SE3 keyframes observing 3D point landmarks.
    kind: 'cart3d' (observations = landmark in the robot frame), 'rb3d' (range / yaw / pitch of the same point), 'stereo' or 'mono' (pinhole, camera looking along robot +X through CAMERA_ON_ROBOT).
    known_first: the first `known_first` landmarks seen from keyframe 0 are given as FIXED (known relative position), like the reference tutorials do to fix the gauge.
    Returns (dataset, gt_poses)."""
    rng = np.random.RandomState(seed)
    gt = spiral_path_se3(n_kf, seed)
    lms = np.column_stack([rng.uniform(-room, room, n_lm), rng.uniform(-room, room, n_lm), rng.uniform(-2.0, 2.0, n_lm)])
    if with_sensor_pose is None:
        with_sensor_pose = kind in ("stereo", "mono")
    S = pose3(*CAMERA_ON_ROBOT) if with_sensor_pose else np.eye(4)
    fx, fy, cx, cy = cam
    O = {"cart3d": 3, "rb3d": 3, "stereo": 4, "mono": 2}[kind]
    seen = set(); out = []
    for kf, T in enumerate(gt):
        Tinv = np.linalg.inv(T @ S)
        ids, zs, flags, rel = [], [], [], []
        p_s = (Tinv[:3, :3] @ lms.T).T + Tinv[:3, 3]  # landmarks in the sensor frame
        p_r = (np.linalg.inv(T)[:3, :3] @ lms.T).T + np.linalg.inv(T)[:3, 3]  # in the robot frame
        for j in range(n_lm):
            q = p_s[j]
            d = np.linalg.norm(q)
            if d > max_range or d < 0.4:
                continue
            if kind == "cart3d":
                if with_sensor_pose is False and q[0] < 0.2:  # looks along +X of the robot
                    continue
                z = q + noise * rng.randn(3)
            elif kind == "rb3d":   # range, yaw, pitch = -asin(z/range) (models/sensors.h:545-566); keep away from the poles of the parameterisation
                if q[0] < 0.2 or abs(q[2]) > 0.8 * d:
                    continue
                z = np.array([d, math.atan2(q[1], q[0]), -math.asin(q[2] / d)]) + noise * rng.randn(3) * np.array([1.0, 0.2, 0.2])
            else:
                if q[2] < 0.5:
                    continue
                u, v = cx + fx * q[0] / q[2], cy + fy * q[1] / q[2]
                if not (0 <= u < 2 * cx and 0 <= v < 2 * cy):
                    continue
                if kind == "stereo":
                    ur = cx + fx * (q[0] - baseline) / q[2]
                    z = np.array([u, v, ur, v]) + noise * rng.randn(4)
                else:
                    z = np.array([u, v]) + noise * rng.randn(2)
            first = j not in seen
            fl = 0; rp = np.zeros(3)
            if first and len(seen) < known_first and kf == 0:
                fl = FLAG_FIXED; rp = p_r[j]
            elif first and init_from_gt_noise is not None:
                fl = FLAG_INIT; rp = p_r[j] + init_from_gt_noise * rng.randn(3)
            seen.add(j)
            ids.append(j); zs.append(z); flags.append(fl); rel.append(rp)
        out.append(dict(feat_ids=np.array(ids, np.uint64), z=np.array(zs, np.float64).reshape(-1, O), flags=np.array(flags, np.uint8), relpos=np.array(rel, np.float64).reshape(-1, 3)))
    return out, gt


def mono_deep_window(n_kf=500, n_lm=20000, seed=1, step=0.35, max_range=6.0, noise=0.5, cam=(200.0, 200.0, 400.0, 320.0), init_noise=0.05, fix_first_kf=True):
    """BASELINE config 4 at any scale ("synthetic monocular SE3, 5k KFs x 200k landmarks, deep local window"): one lap of a wide circle with gentle 3D motion, point landmarks in a
    band around the path, forward-looking monocular camera (fx fy cx cy of tutorial-srba-monocular-se3.cpp:125-131, mounted through CAMERA_ON_ROBOT), pixel noise, every landmark
    entered with an initial value near the truth (is_unknown_with_init_val, as SURVEY 8d prescribes). The landmarks first seen from key-frame 0 are given as FIXED (known
    relative position) when fix_first_kf is set: a monocular map has a free global scale, and along that flat direction the reference's LM takes wild rejected steps whose
    spanning-tree twins are never restored and then seed the next edge -- pinning the scale keeps the synthetic run meaningful.
    Meant for max_tree_depth = max_optimize_depth = 8, submap 20.
    Returns (dataset, gt_poses)."""
    rng = np.random.RandomState(seed)
    radius = max(20.0, step * n_kf / (2 * math.pi) * 1.15)   # short maps walk an arc of a 20 m circle, long ones a full lap
    gt = []
    for k in range(n_kf):
        a = step * k / radius
        gt.append(pose3(radius * math.cos(a), radius * math.sin(a), 0.4 * math.sin(3 * a), a + math.pi / 2 + 0.03 * rng.randn(), 0.04 * math.sin(5 * a), 0.03 * math.cos(4 * a)))
    ang = rng.uniform(-0.1, step * n_kf / radius + 0.35, n_lm); rad = radius + rng.uniform(-5.0, 5.0, n_lm)
    lms = np.column_stack([rad * np.cos(ang), rad * np.sin(ang), rng.uniform(-2.0, 2.0, n_lm)])
    S = pose3(*CAMERA_ON_ROBOT); fx, fy, cx, cy = cam
    seen = np.zeros(n_lm, bool); out = []
    for kf, T in enumerate(gt):
        Ts = np.linalg.inv(T @ S); Tr = np.linalg.inv(T)
        q = (Ts[:3, :3] @ lms.T).T + Ts[:3, 3]
        ok = (q[:, 2] > 0.5) & (np.linalg.norm(q, axis=1) < max_range)
        u = cx + fx * q[:, 0] / np.where(ok, q[:, 2], 1.0); v = cy + fy * q[:, 1] / np.where(ok, q[:, 2], 1.0)
        ok &= (u >= 0) & (u < 2 * cx) & (v >= 0) & (v < 2 * cy)
        idx = np.flatnonzero(ok)
        z = np.column_stack([u[idx], v[idx]]) + noise * rng.randn(len(idx), 2)
        first = ~seen[idx]; seen[idx] = True
        rel = (Tr[:3, :3] @ lms[idx].T).T + Tr[:3, 3] + init_noise * rng.randn(len(idx), 3)
        if kf == 0 and fix_first_kf:
            rel = (Tr[:3, :3] @ lms[idx].T).T + Tr[:3, 3]; flags = np.full(len(idx), FLAG_FIXED, np.uint8)
        else:
            flags = np.where(first, FLAG_INIT, 0).astype(np.uint8)
        out.append(dict(feat_ids=idx.astype(np.uint64), z=z, flags=flags, relpos=np.where(first[:, None], rel, 0.0)))
    return out, gt


def ypr_of(R):
    return np.array([math.atan2(R[1, 0], R[0, 0]), math.asin(-R[2, 0]), math.atan2(R[2, 1], R[2, 2])])


def graph_slam_se3(n_kf=60, seed=1, sigma_xyz=1e-3, sigma_ang_deg=0.1, max_range=2.5, max_back=40):
    """SE(3) relative graph-SLAM (the 3D counterpart of graph_slam_se2; examples/cpp/tutorial-srba-relative-graph-slam-se3.cpp feeds the engine the same way):
    every key-frame first lists itself as a fixed "fake landmark" at the null pose, then the relative pose (x y z yaw pitch roll) of every earlier key-frame
    closer than max_range, with Gaussian noise. Returns (dataset, gt_poses)."""
    rng = np.random.RandomState(seed)
    gt = spiral_path_se3(n_kf, seed)
    out = []
    for kf, T in enumerate(gt):
        ids, zs, flags = [kf], [np.zeros(6)], [FLAG_FIXED]
        Tinv = np.linalg.inv(T)
        for j in range(max(0, kf - max_back), kf):
            rel = Tinv @ gt[j]
            if np.linalg.norm(rel[:3, 3]) > max_range:
                continue
            z = np.concatenate([rel[:3, 3], ypr_of(rel[:3, :3])]) + np.concatenate([sigma_xyz * rng.randn(3), math.radians(sigma_ang_deg) * rng.randn(3)])
            ids.append(j); zs.append(z); flags.append(0)
        out.append(dict(feat_ids=np.array(ids, np.uint64), z=np.array(zs, np.float64).reshape(-1, 6), flags=np.array(flags, np.uint8), relpos=np.zeros((len(ids), 6))))
    return out, gt


def graph_slam_lambda_se3(sigma_xyz=1e-3, sigma_ang_deg=0.1):
    a = 1.0 / sigma_xyz ** 2; b = 1.0 / math.radians(sigma_ang_deg) ** 2
    return np.diag([a, a, a, b, b, b])


def landmarks_dataset_se2_stereo(n_kf=30, n_lm=500, seed=1, max_range=6.0, noise=0.0, cam=(200.0, 150.0, 512.0, 384.0), baseline=0.2, init_from_gt_noise=None):
    """Planar robot (SE(2) key-frames) with a forward-looking stereo camera mounted through CAMERA_ON_ROBOT observing 3D point landmarks
    (the problem type of examples/cpp/tutorial-srba-stereo-se2.cpp). Returns (dataset, gt as (x, y, phi))."""
    rng = np.random.RandomState(seed)
    gt = [(0.0, 0.0, 0.0)]
    for _ in range(n_kf - 1):
        gt.append(_compose2(gt[-1], (rng.uniform(0.3, 0.6), 0.0, math.radians(rng.uniform(-20, 20)))))
    xs = np.array([p[0] for p in gt]); ys = np.array([p[1] for p in gt])
    lms = np.column_stack([rng.uniform(xs.min() - 5, xs.max() + 5, n_lm), rng.uniform(ys.min() - 5, ys.max() + 5, n_lm), rng.uniform(-1.5, 1.5, n_lm)])
    S = pose3(*CAMERA_ON_ROBOT); fx, fy, cx, cy = cam
    seen = set(); out = []
    for kf, p in enumerate(gt):
        T = pose3(p[0], p[1], 0.0, p[2], 0.0, 0.0)
        Ts = np.linalg.inv(T @ S); Tr = np.linalg.inv(T)
        ids, zs, flags, rel = [], [], [], []
        for j in range(n_lm):
            q = Ts[:3, :3] @ lms[j] + Ts[:3, 3]
            if q[2] < 0.6 or np.linalg.norm(q) > max_range:
                continue
            u, v, ur = cx + fx * q[0] / q[2], cy + fy * q[1] / q[2], cx + fx * (q[0] - baseline) / q[2]
            if not (0 <= u < 2 * cx and 0 <= v < 2 * cy and 0 <= ur < 2 * cx):
                continue
            first = j not in seen; fl = 0; rp = np.zeros(3)
            if first and init_from_gt_noise is not None:
                fl = FLAG_INIT; rp = Tr[:3, :3] @ lms[j] + Tr[:3, 3] + init_from_gt_noise * rng.randn(3)
            seen.add(j)
            ids.append(j); zs.append(np.array([u, v, ur, v]) + noise * rng.randn(4)); flags.append(fl); rel.append(rp)
        out.append(dict(feat_ids=np.array(ids, np.uint64), z=np.array(zs, np.float64).reshape(-1, 4), flags=np.array(flags, np.uint8), relpos=np.array(rel, np.float64).reshape(-1, 3)))
    return out, gt


def landmarks_dataset_se2(kind="rb2d", n_kf=50, n_lm=300, seed=1, max_range=4.0, fov_deg=100.0, noise=1e-3, known_first=0):
    """cfg1 of BASELINE.md: planar random walk, step U(0.4,0.8) m, turn U(-30,30) deg; point landmarks; range-bearing (or cartesian) sensor
    (datasets/tutorials_dataset-range-bearing-2d.cfg:27-28,44-46)."""
    rng = np.random.RandomState(seed)
    gt = [(0.0, 0.0, 0.0)]
    for _ in range(n_kf - 1):
        gt.append(_compose2(gt[-1], (rng.uniform(0.4, 0.8), 0.0, math.radians(rng.uniform(-30, 30)))))
    xs = np.array([p[0] for p in gt]); ys = np.array([p[1] for p in gt])
    lms = np.column_stack([rng.uniform(xs.min() - 4, xs.max() + 4, n_lm), rng.uniform(ys.min() - 4, ys.max() + 4, n_lm)])
    seen = set(); out = []
    for kf, p in enumerate(gt):
        ids, zs, flags, rel = [], [], [], []
        c, s = math.cos(p[2]), math.sin(p[2])
        for j in range(n_lm):
            dx, dy = lms[j, 0] - p[0], lms[j, 1] - p[1]
            lx, ly = dx * c + dy * s, -dx * s + dy * c
            r, b = math.hypot(lx, ly), math.atan2(ly, lx)
            if r > max_range or r < 0.2 or abs(b) > math.radians(fov_deg) / 2:
                continue
            z = np.array([r + noise * rng.randn(), b + noise * rng.randn()]) if kind == "rb2d" else np.array([lx + noise * rng.randn(), ly + noise * rng.randn()])
            first = j not in seen
            fl = 0; rp = np.zeros(2)
            if first and len(seen) < known_first and kf == 0:
                fl = FLAG_FIXED; rp = np.array([lx, ly])
            seen.add(j)
            ids.append(j); zs.append(z); flags.append(fl); rel.append(rp)
        out.append(dict(feat_ids=np.array(ids, np.uint64), z=np.array(zs, np.float64).reshape(-1, 2), flags=np.array(flags, np.uint8), relpos=np.array(rel, np.float64).reshape(-1, 2)))
    return out, gt


# ------------------------------------------------------------------------------------------------ text formats of apps/srba-slam
def write_text_dataset(frames, path, kind):
    """Writes key-frame dictionaries in the text format read by srba-slam (apps/srba-slam/CDatasetParserBase.h:82: FRAME_ID FEAT_ID fields...).
    kind: 'graph-slam' (12 columns: X Y Z YAW PITCH ROLL QR QX QY QZ, CDatasetParser_RelGraphSLAM2D.h:28-32; the fixed self-landmark rows are
    not part of the file), or a landmark sensor: 'rb2d' / 'cart2d' (4 columns), 'cart3d' (5), 'mono' (4), 'stereo' (6)."""
    with open(path, "w") as f:
        f.write("% FRAME_ID FEAT_ID sensor-specific fields\n")
        for kf, fr in enumerate(frames):
            z = np.asarray(fr["z"], np.float64).reshape(len(fr["feat_ids"]), -1)
            for fid, zi in zip(fr["feat_ids"], z):
                if kind == "graph-slam":
                    if int(fid) == kf:
                        continue
                    h = 0.5 * zi[2]
                    f.write("%d %d %.17g %.17g 0 %.17g 0 0 %.17g 0 0 %.17g\n" % (kf, int(fid), zi[0], zi[1], zi[2], math.cos(h), math.sin(h)))
                else:
                    f.write("%d %d %s\n" % (kf, int(fid), " ".join("%.17g" % v for v in zi)))


def write_gt_path(poses_xyz_quat, path):
    """ground-truth path file: idx x y z qr qx qy qz (CDatasetParserBase.h:213-226)"""
    with open(path, "w") as f:
        for i, p in enumerate(poses_xyz_quat):
            f.write("%d %s\n" % (i, " ".join("%.17g" % v for v in p)))


def write_stereo_cfg(path, cam=(200.0, 150.0, 512.0, 384.0), baseline=0.2):
    """sensor parameter file of --sensor-params-cfg-file ([EXT] TStereoCamera::loadFromConfigFile("CAMERA"): fx fy cx cy per camera, left-to-right pose)"""
    with open(path, "w") as f:
        for sec in ("CAMERA", "CAMERA_LEFT", "CAMERA_RIGHT"):
            f.write("[%s]\nfx = %.17g\nfy = %.17g\ncx = %.17g\ncy = %.17g\n\n" % ((sec,) + tuple(cam)))
        f.write("[CAMERA_LEFT2RIGHT_POSE]\npose_quaternion = [%.17g 0 0 1 0 0 0]\n" % baseline)
