"""Script-level drivers over the C ABIs (tests / bench plumbing).

Engine        : srba::RbaEngine<> through libsrba_engine.so (the source-compatible C++ front-end; GPU back-end by default)
CapsuleBatch  : an array of srba_problem_capsule harvested from an Engine (pre-optimisation state of every optimize_local_area call)
run_batch_hip : upload + srba_hip_lm_run on the GPU                       (product path)
The CPU oracle is NOT reachable from this package: tests plug it in from tests/_oracle.py (Engine(backend=<plug object>)).
"""
import ctypes as C
import numpy as np
from . import capi

_RES_FIELDS = ("status", "num_iters", "num_trials", "num_not_pd", "num_accepted", "num_relinearized", "num_invalid_jacobs", "stop_reason",
               "num_observations", "num_jacobians", "num_span_tree_numeric_updates", "total_sqr_error_init", "total_sqr_error_final", "obs_rmse", "lambda_init", "lambda_final")


def results_to_dict(res, n):
    out = {}
    for f in _RES_FIELDS:
        out[f] = np.array([getattr(res[i], f) for i in range(n)])
    out["chi2_init"] = out["total_sqr_error_init"]; out["chi2_final"] = out["total_sqr_error_final"]
    out["trace_chi2"] = np.array([list(res[i].trace_chi2) for i in range(n)]).reshape(n, capi.TRACE_LEN)
    out["trace_lambda"] = np.array([list(res[i].trace_lambda) for i in range(n)]).reshape(n, capi.TRACE_LEN)
    out["trace_rho"] = np.array([list(res[i].trace_rho) for i in range(n)]).reshape(n, capi.TRACE_LEN)
    return out


class Engine(object):
    def __init__(self, family, backend="hip", **kw):
        self.lib = capi.engine_lib()
        self.family = family
        self.P, self.L, self.O, self.PD = capi.DIMS[family]
        cfg = capi.EngineConfig()
        self.lib.srba_engine_config_default(C.byref(cfg), family)
        lam = kw.pop("lambda_", None)
        if lam is not None:
            lam = np.asarray(lam, np.float64).reshape(-1)
            for i, v in enumerate(lam):
                cfg.lambda_[i] = v
        for k in ("sensor_pose_xyzypr", "cam_left", "cam_right", "right_cam_pose"):
            if k in kw:
                for i, v in enumerate(kw.pop(k)):
                    getattr(cfg, k)[i] = v
        for k, v in kw.items():
            if not hasattr(cfg, k):
                raise KeyError(k)
            setattr(cfg, k, v)
        self.cfg = cfg
        self.h = self.lib.srba_engine_create(C.byref(cfg))
        if not self.h:
            raise RuntimeError(self.lib.srba_engine_last_error(None).decode())
        self._cb = None
        # backend: "hip" (the GPU, default and only built-in back-end) or an object with .plug(engine) that installs an external numeric
        # back-end through srba_engine_set_backend_fn (how tests/ run the same front-end against the CPU oracle)
        if hasattr(backend, "plug"):
            backend.plug(self)
        elif backend != "hip":
            raise ValueError("unknown back-end %r: the product has no CPU path" % (backend,))

    def close(self):
        if self.h:
            self.lib.srba_engine_destroy(self.h); self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add_keyframe(self, feat_ids, z, flags=None, relpos=None):
        n = len(feat_ids)
        ids = np.ascontiguousarray(feat_ids, np.uint64); z = np.ascontiguousarray(z, np.float64).reshape(n, self.O)
        fl = np.ascontiguousarray(flags if flags is not None else np.zeros(n), np.uint8)
        rp = None if relpos is None else np.ascontiguousarray(relpos, np.float64).reshape(n, self.L)
        info = capi.KfInfo()
        rc = self.lib.srba_engine_add_keyframe(self.h, n, ids.ctypes.data_as(C.POINTER(C.c_uint64)), z.ctypes.data_as(capi.PF64), fl.ctypes.data_as(capi.PU8),
                                                rp.ctypes.data_as(capi.PF64) if rp is not None else None, C.byref(info))
        if rc != 0:
            raise RuntimeError(self.lib.srba_engine_last_error(self.h).decode())
        return info

    def run(self, dataset):
        return [self.add_keyframe(k["feat_ids"], k["z"], k["flags"], k.get("relpos")) for k in dataset]

    def edges(self):
        n = self.lib.srba_engine_num_edges(self.h)
        fr = np.zeros(n, np.uint64); to = np.zeros(n, np.uint64); pose = np.zeros((n, self.PD))
        f, t = C.c_uint64(), C.c_uint64(); buf = (C.c_double * self.PD)()
        for i in range(n):
            self.lib.srba_engine_get_edge(self.h, i, C.byref(f), C.byref(t), buf)
            fr[i], to[i], pose[i] = f.value, t.value, list(buf)
        return fr, to, pose

    def unknown_lms(self):
        n = self.lib.srba_engine_num_unknown_lms(self.h)
        ids = np.zeros(n, np.uint64); base = np.zeros(n, np.uint64); pos = np.zeros((n, self.L))
        if n:
            self.lib.srba_engine_get_unknown_lms(self.h, ids.ctypes.data_as(C.POINTER(C.c_uint64)), base.ctypes.data_as(C.POINTER(C.c_uint64)), pos.ctypes.data_as(capi.PF64))
        return ids, base, pos

    def st_dump(self, what):
        n = self.lib.srba_engine_st_dump(self.h, what, None, 0)
        out = np.zeros(n, np.int64)
        self.lib.srba_engine_st_dump(self.h, what, out.ctypes.data_as(C.POINTER(C.c_int64)), n)
        return out

    def rel_pose(self, query, reference):
        buf = (C.c_double * self.PD)()
        if self.lib.srba_engine_get_rel_pose(self.h, query, reference, buf) != 0:
            return None
        return np.array(list(buf))

    def global_graphslam_problem(self, root=0):
        """RbaEngine<>::get_global_graphslam_problem(): (node ids, node poses [n, PD], edge (from, to) [m, 2], edge poses [m, PD])"""
        n_kf = int(max(self.edges()[0].max(), self.edges()[1].max())) + 1 if self.lib.srba_engine_num_edges(self.h) else 1; m = self.lib.srba_engine_num_edges(self.h)
        ids = np.zeros(n_kf, np.uint64); poses = np.zeros((n_kf, self.PD)); ft = np.zeros((max(m, 1), 2), np.uint64); ep = np.zeros((max(m, 1), self.PD))
        n = self.lib.srba_engine_export_graphslam(self.h, root, ids.ctypes.data_as(C.POINTER(C.c_uint64)), poses.ctypes.data_as(capi.PF64), n_kf, ft.ctypes.data_as(C.POINTER(C.c_uint64)),
                ep.ctypes.data_as(capi.PF64), m)
        return ids[:n], poses[:n], ft[:m], ep[:m]

    def eval_overall_squared_error(self):
        """RbaEngine<>::eval_overall_squared_error(): squared error of every observation of the map with the current estimate"""
        v = C.c_double()
        if self.lib.srba_engine_eval_overall_sqr_error(self.h, C.byref(v)) != 0:
            raise RuntimeError(self.lib.srba_engine_last_error(self.h).decode())
        return v.value

    # ---- map sweeps (RbaEngine<>::plan_local_area_sweep / optimize_local_areas_batch; srba_amd/multi.py shards them over GPUs)
    def _chk(self, rc):
        if rc < 0:
            raise RuntimeError(self.lib.srba_engine_last_error(self.h).decode())
        return rc

    def plan_sweep(self, roots, win):
        """rounds of mutually independent local areas: (round_of [n], touch_off [n + 1], touch [edge id | 0x80000000 if written], n_rounds)"""
        roots = np.ascontiguousarray(roots, np.uint64); n = len(roots); PU64 = C.POINTER(C.c_uint64)
        round_of = np.zeros(n, np.int32); off = np.zeros(n + 1, np.int64); cap = max(1024, 64 * n)
        while True:
            touch = np.zeros(cap, np.uint32)
            rc = self.lib.srba_engine_plan_sweep(self.h, roots.ctypes.data_as(PU64), n, win, round_of.ctypes.data_as(capi.PI32), off.ctypes.data_as(C.POINTER(C.c_int64)),
                                                  touch.ctypes.data_as(C.POINTER(C.c_uint32)), cap)
            if rc <= -2:
                cap = -2 - rc; continue
            self._chk(rc)
            return round_of, off, touch[:off[n]], int(rc)

    def plan_sweep_lms(self, n):
        """landmarks the windows of the LAST plan_sweep (n roots) touch: (off [n + 1], touch [landmark id | 0x80000000 if the window optimises it])"""
        off = np.zeros(n + 1, np.int64); cap = 1024
        while True:
            touch = np.zeros(cap, np.uint32)
            rc = self.lib.srba_engine_plan_sweep_lms(self.h, off.ctypes.data_as(C.POINTER(C.c_int64)), touch.ctypes.data_as(C.POINTER(C.c_uint32)), cap)
            if rc <= -2:
                cap = -2 - rc; continue
            self._chk(rc)
            return off, touch[:rc]

    def get_lm_positions(self, ids):
        ids = np.ascontiguousarray(ids, np.uint64); out = np.zeros((len(ids), self.L))
        if len(ids):
            self._chk(self.lib.srba_engine_get_lm_positions(self.h, ids.ctypes.data_as(C.POINTER(C.c_uint64)), len(ids), out.ctypes.data_as(capi.PF64)))
        return out

    def set_lm_positions(self, ids, pos):
        ids = np.ascontiguousarray(ids, np.uint64); pos = np.ascontiguousarray(pos, np.float64).reshape(len(ids), self.L)
        if len(ids):
            self._chk(self.lib.srba_engine_set_lm_positions(self.h, ids.ctypes.data_as(C.POINTER(C.c_uint64)), len(ids), pos.ctypes.data_as(capi.PF64)))

    def optimize_batch(self, roots, win):
        """optimize_local_area() of mutually independent roots as ONE batch of the numeric back-end; returns the KfInfo records"""
        roots = np.ascontiguousarray(roots, np.uint64); n = len(roots); out = (capi.KfInfo * max(n, 1))()
        if n:
            self._chk(self.lib.srba_engine_optimize_batch(self.h, roots.ctypes.data_as(C.POINTER(C.c_uint64)), n, win, out))
        return out

    def optimize_local_area(self, root, win):
        info = capi.KfInfo(); self._chk(self.lib.srba_engine_optimize_local_area(self.h, int(root), win, C.byref(info))); return info

    def get_edge_poses(self, ids):
        ids = np.ascontiguousarray(ids, np.uint64); out = np.zeros((len(ids), self.PD))
        if len(ids):
            self._chk(self.lib.srba_engine_get_edge_poses(self.h, ids.ctypes.data_as(C.POINTER(C.c_uint64)), len(ids), out.ctypes.data_as(capi.PF64)))
        return out

    def set_edge_poses(self, ids, poses):
        ids = np.ascontiguousarray(ids, np.uint64); poses = np.ascontiguousarray(poses, np.float64).reshape(len(ids), self.PD)
        if len(ids):
            self._chk(self.lib.srba_engine_set_edge_poses(self.h, ids.ctypes.data_as(C.POINTER(C.c_uint64)), len(ids), poses.ctypes.data_as(capi.PF64)))

    def harvest(self):
        return CapsuleBatch(self)


class CapsuleBatch(object):
    """View over capsules owned by an Engine (harvest) or by a capsule file / clone."""
    def __init__(self, owner=None, handle=None, params=None, family=None):
        self.lib = capi.engine_lib()
        self.owner = owner; self.handle = handle
        if owner is not None:
            self.n = int(self.lib.srba_engine_harvest_count(owner.h))
            self.ptr = self.lib.srba_engine_harvest_capsules(owner.h)
            self.params = capi.HipParams(); self.lib.srba_engine_get_hip_params(owner.h, C.byref(self.params))
            self.family = owner.family
        else:
            self.n = int(self.lib.srba_capsule_file_count(handle))
            self.ptr = self.lib.srba_capsule_file_capsules(handle)
            self.params = params; self.family = family if family is not None else params.family

    @staticmethod
    def load(path):
        lib = capi.engine_lib()
        h = lib.srba_capsule_file_load(path.encode())
        if not h:
            raise RuntimeError(lib.srba_engine_last_error(None).decode())
        p = capi.HipParams(); lib.srba_capsule_file_params(h, C.byref(p))
        return CapsuleBatch(handle=h, params=p)

    def clone(self, first=0, count=None):
        count = self.n - first if count is None else count
        sub = C.cast(C.addressof(self.ptr.contents) + first * C.sizeof(capi.Capsule), capi.PCAP)
        h = self.lib.srba_capsule_clone(sub, count, self.family)
        return CapsuleBatch(handle=h, params=self.params, family=self.family)

    def sub(self, first, count):
        b = CapsuleBatch.__new__(CapsuleBatch)
        b.lib = self.lib; b.owner = self.owner; b.handle = None; b._parent = self
        b.n = count; b.ptr = C.cast(C.addressof(self.ptr.contents) + first * C.sizeof(capi.Capsule), capi.PCAP); b.params = self.params; b.family = self.family
        return b

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        return self.ptr[i]

    def array(self, i, field, dtype, count):
        p = getattr(self.ptr[i], field)
        if not p or count == 0:
            return np.zeros(0, dtype)
        return np.ctypeslib.as_array(p, shape=(count,)).copy()

    def __del__(self):
        try:
            if self.handle:
                self.lib.srba_capsule_file_free(self.handle); self.handle = None
        except Exception:
            pass


class HipContext(object):
    def __init__(self, params, device=-1):
        self.lib = capi.hip_lib()
        self.params = params
        self.ctx = self.lib.srba_hip_create(device, C.byref(params))
        if not self.ctx:
            raise RuntimeError("srba_hip_create failed: " + self.lib.srba_hip_last_error(None).decode())
        self.n = 0

    def _chk(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s failed: %s" % (what, self.lib.srba_hip_last_error(self.ctx).decode()))

    def upload(self, batch):
        self._chk(self.lib.srba_hip_upload_problems(self.ctx, batch.ptr, batch.n), "upload"); self.n = batch.n; self.batch = batch

    def lm_run(self):
        res = (capi.LmResult * self.n)()
        self._chk(self.lib.srba_hip_lm_run(self.ctx, res), "lm_run")
        return results_to_dict(res, self.n)

    def optimize_capsule(self, batch):
        """srba_hip_optimize_capsule: upload + LM loop + write-back of ONE capsule (written into the capsule's own arrays), one wait for the device"""
        res = (capi.LmResult * 1)()
        self._chk(self.lib.srba_hip_optimize_capsule(self.ctx, batch.ptr, res), "optimize_capsule"); self.n = 1; self.batch = batch
        return results_to_dict(res, 1)

    def stats(self):
        s = capi.BatchStats(); self.lib.srba_hip_batch_stats(self.ctx, C.byref(s))
        return {f[0]: getattr(s, f[0]) for f in capi.BatchStats._fields_}

    def debug(self, what):
        n = self.lib.srba_hip_debug_size(self.ctx, what)
        out = np.zeros(max(n, 1))
        self._chk(self.lib.srba_hip_debug_read(self.ctx, what, out.ctypes.data_as(capi.PF64), out.size), "debug_read")
        return out[:n]

    def close(self):
        if self.ctx:
            self.lib.srba_hip_destroy(self.ctx); self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def run_batch_hip(batch, download=False):
    ctx = HipContext(batch.params)
    ctx.upload(batch)
    out = ctx.lm_run()
    out["kernel_ms"] = ctx.lib.srba_hip_last_kernel_ms(ctx.ctx)
    if download:
        work = batch.clone()
        ctx._chk(ctx.lib.srba_hip_download_state(ctx.ctx, work.ptr, work.n), "download_state")
        out["state"] = work
    ctx.close()
    return out


def graph_slam_engine(backend="hip", submap=10, depth=3, sigma_xy=1e-3, sigma_yaw_deg=0.2, solver=capi.SOLVER_NO_SCHUR_SPARSE, harvest=1, **kw):
    """srba-slam --se2 --graph-slam configuration (apps/srba-slam/instance_relative_graph_slam_se2.cpp:17-24, srba-run-generic-impl.h:127-182)."""
    from . import datasets
    args = dict(solver=solver, noise=capi.NOISE_MATRIX, lambda_=_lam36(datasets.graph_slam_lambda(sigma_xy, sigma_yaw_deg), 3),
                max_tree_depth=depth, max_optimize_depth=depth, submap_size=submap, min_obs_to_loop_closure=1, optimize_new_edges_alone=1,
                use_robust_kernel=0, max_error_per_obs_to_stop=1e-8, harvest=harvest)
    args.update(kw)
    return Engine(capi.SE2_RELPOSE2D, backend=backend, **args)


def graph_slam_engine_se3(backend="hip", submap=5, depth=3, sigma_xyz=1e-3, sigma_ang_deg=0.1, solver=capi.SOLVER_NO_SCHUR_SPARSE, harvest=1, **kw):
    """SE(3) relative graph-SLAM: <SE3, RelativePoses3D, RelativePoses_3D> with a constant 6x6 information matrix (examples/cpp/tutorial-srba-relative-graph-slam-se3.cpp:20-33)."""
    from . import datasets
    args = dict(solver=solver, noise=capi.NOISE_MATRIX, lambda_=_lam36(datasets.graph_slam_lambda_se3(sigma_xyz, sigma_ang_deg), 6), max_tree_depth=depth, max_optimize_depth=depth,
                submap_size=submap, min_obs_to_loop_closure=1, optimize_new_edges_alone=1, use_robust_kernel=0, max_error_per_obs_to_stop=1e-8, harvest=harvest)
    args.update(kw)
    return Engine(capi.SE3_RELPOSE3D, backend=backend, **args)


def _lam36(m, O):
    out = np.zeros(36); out[:O * O] = np.asarray(m, np.float64).reshape(-1); return out


def harvest_graph_slam(dataset, backend="hip", **kw):
    eng = graph_slam_engine(backend=backend, **kw)
    eng.run(dataset)
    b = eng.harvest(); b.engine = eng
    return b


def landmark_engine(kind, backend="hip", depth=3, submap=15, sigma=None, robust=0, harvest=1, with_sensor_pose=None, cam=(200.0, 150.0, 512.0, 384.0), baseline=0.2, **kw):
    """Engines for the point-landmark families with the reference apps' policies: identity noise, Schur + dense Cholesky
    (apps/srba-slam/instance_se3_lm3d_stereo.cpp:36-45, instance_se3_lm3d_monocular.cpp:33-42, instance_se2_lm2d_rangebearing2d.cpp:17)."""
    from . import datasets
    fam = {"cart3d": capi.SE3_CART3D, "rb3d": capi.SE3_RB3D, "stereo": capi.SE3_STEREO, "mono": capi.SE3_MONO, "rb2d": capi.SE2_RB2D, "cart2d": capi.SE2_CART2D, "stereo_se2": capi.SE2_STEREO}[kind]
    if with_sensor_pose is None:
        with_sensor_pose = kind in ("stereo", "mono", "stereo_se2")
    if sigma is None:
        sigma = {"cart3d": 0.01, "rb3d": 0.01, "stereo": 0.5, "mono": 0.5, "rb2d": 0.05, "cart2d": 0.05, "stereo_se2": 0.5}[kind]
    args = dict(solver=capi.SOLVER_SCHUR_DENSE, noise=capi.NOISE_IDENTITY, std_noise_observations=sigma, max_tree_depth=depth, max_optimize_depth=depth, submap_size=submap,
                use_robust_kernel=robust, harvest=harvest, cam_left=cam, cam_right=cam, right_cam_pose=(baseline, 0, 0, 1, 0, 0, 0))
    if with_sensor_pose:
        args.update(sensor_pose=capi.SENSOR_POSE_SE3, sensor_pose_xyzypr=datasets.CAMERA_ON_ROBOT)
    args.update(kw)
    return Engine(fam, backend=backend, **args)
