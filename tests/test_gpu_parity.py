"""GPU parity tests (-m gpu): the HIP path (through the C ABI) against the CPU oracle on the same capsules.

Tolerances: chi2 / residual-derived scalars 1e-6 relative (BASELINE.json north_star); integer outputs (trial counts, accept /
relinearise decisions, stop reasons) exact; unknowns 1e-6 relative to their scale.
"""
import numpy as np
import pytest

from srba_amd import capi, datasets, runner
import _oracle  # tests/_oracle.py: the CPU checker

pytestmark = pytest.mark.gpu

REL = 1e-6


def _close(a, b, rel=REL, abs_=1e-15):
    a = np.asarray(a, float); b = np.asarray(b, float)
    return np.all(np.abs(a - b) <= rel * np.maximum(np.abs(a), np.abs(b)) + abs_)


def _replay_exact(b, gpu, tol_trace=1e-9, tol_floor=1e-9, tol_final=1e-6, tol=None, tol_trial=None, threads=8):
    """Exact form of "the two runs may part only at a rounding-floor decision" (round 4; oracle/srba_oracle.cpp Problem::run, Replay): the oracle re-runs every window taking the GPU's
    own not-PD / reject / accept sequence and reports its own rho and chi2 per trial. For EVERY window: (i) the chi2 of every accepted trial of the whole run agrees to tol_trace,
    (ii) every decision the oracle would have taken the other way is a step that moves chi2 by less than tol_floor of its value in BOTH runs, (iii) where the whole sequence fits
    the trace, the final chi2 agrees to tol_final; the oracle can follow every sequence (it never calls a system the GPU solved not positive definite, nor the other way round).
    tol / tol_trial: optional per-window / per-trial arrays that replace the scalars (windows whose rounding sensitivity is measured, see the lost monocular map)."""
    rep = _oracle.run_batch_replay(b, gpu, threads=threads); R = _oracle.replay_report(gpu, rep)
    tt = np.full(b.n, tol_trace) if tol is None else tol; te = np.full(b.n, tol_final) if tol is None else np.maximum(tol, tol_final)
    tf = np.full(R["floor_move"].shape, tol_floor) if tol_trial is None else tol_trial
    assert (R["diverged_at"] < 0).all() and R["forced_notpd"].sum() == 0, (np.flatnonzero(R["diverged_at"] >= 0), R["forced_notpd"].sum())
    bad = np.flatnonzero(R["worst_trace"] > tt); assert len(bad) == 0, ("accepted-trial chi2", bad[:8], R["worst_trace"][bad[:8]])
    bad = np.argwhere(R["floor_move"] > tf); assert len(bad) == 0, ("decision not at the rounding floor", bad[:8], R["floor_move"][R["floor_move"] > tf][:8])
    bad = np.flatnonzero(R["complete"] & (R["final_rel"] > te)); assert len(bad) == 0, ("final chi2", bad[:8], R["final_rel"][bad[:8]])
    return rep, R


@pytest.fixture(scope="module")
def se2_batch():
    # capsules harvested with the oracle as numeric back-end (tests may use the oracle); 240 keyframes with loop closures
    ds = datasets.graph_slam_se2(n_kf=240, seed=5, grid=2, block=30.0)
    return runner.harvest_graph_slam(ds, backend=_oracle.BACKEND, submap=10, depth=3)


def _compare_lm(b, gpu, cpu):
    """chi2 parity (1e-6 relative, BASELINE.json) + exact agreement of everything that is well conditioned.
    Trial COUNTS are not compared: with max_error_per_obs_to_stop=1e-8 (srba-slam default) the reference keeps iterating at the noise floor,
    where accept/reject (sign of rho = (E-E')/...) is decided by the last bits of E-E' and legitimately differs between any two
    floating-point evaluation orders.  The per-trial chi2 traces must agree on the whole prefix where the two runs take the same decisions."""
    assert np.all(gpu["status"] == cpu["status"])
    for k in ("num_observations", "num_jacobians", "num_span_tree_numeric_updates", "num_invalid_jacobs"):
        assert np.array_equal(gpu[k], cpu[k]), k
    for k in ("chi2_init", "lambda_init"):
        assert _close(gpu[k], cpu[k], rel=1e-9), k
    assert _close(gpu["chi2_final"], cpu["chi2_final"], rel=1e-6, abs_=1e-20)
    assert _close(gpu["obs_rmse"], cpu["obs_rmse"], rel=1e-6, abs_=1e-12)
    n_full = 0
    for i in range(b.n):
        m = int(min(gpu["num_trials"][i], cpu["num_trials"][i], capi.TRACE_LEN))
        g, c = gpu["trace_chi2"][i][:m], cpu["trace_chi2"][i][:m]
        same_dec = (np.sign(gpu["trace_rho"][i][:m]) == np.sign(cpu["trace_rho"][i][:m])) & (np.isnan(g) == np.isnan(c))
        k = m if same_dec.all() else int(np.argmin(same_dec))  # first trial where the decisions differ
        assert k >= min(m, 2), (i, k, m)                      # the descent phase is always identical
        if k < m:   # the two runs may part ways only at the rounding floor: the step they disagree on changes chi2 by less than 1e-6 of its value in BOTH runs
            acc = np.flatnonzero(cpu["trace_rho"][i][:k] > 0); e_prev = c[acc[-1]] if len(acc) else cpu["chi2_init"][i]
            for e_k in (g[k], c[k]):
                assert np.isnan(e_k) or abs(e_k - e_prev) <= 1e-6 * max(e_prev, 1e-300) + 1e-20, (i, k, e_prev, g[k], c[k])
        ok = cpu["trace_rho"][i][:k] > 0  # accepted trials (the chi2 of a rejected overshoot is chaotic; only its decision is compared)
        assert _close(g[:k][ok], c[:k][ok], rel=1e-6, abs_=1e-20), i
        assert _close(gpu["trace_lambda"][i][:k], cpu["trace_lambda"][i][:k], rel=1e-9), i
        n_full += int(k == m and gpu["num_trials"][i] == cpu["num_trials"][i])
    _replay_exact(b, gpu)   # ... and over the WHOLE run of every window, the oracle following the GPU's decisions: 1e-9 on accepted trials, 1e-9 at every disputed decision, 1e-6 at the end
    return n_full


@pytest.mark.gpu
def test_lean_instantiation_of_the_small_classes_matches_oracle(se2_batch, monkeypatch):
    """k_lm_run_lean (round 4: three wavefronts per SIMD, fewer loads in flight per lane; what a big batch runs its small size classes on) forced onto every class it can take of this batch:
    the same parity statements (decision replay against the oracle) for it and for k_lm_run on the same batch, and the two kernels within rounding of each other."""
    b = se2_batch
    monkeypatch.setenv("SRBA_HIP_LEAN_MIN_COUNT", "1"); lean = runner.run_batch_hip(b)
    monkeypatch.setenv("SRBA_HIP_LEAN", "0"); plain = runner.run_batch_hip(b)
    cpu = _oracle.run_batch(b)
    _compare_lm(b, lean, cpu)
    assert _close(lean["chi2_final"], plain["chi2_final"], rel=1e-9, abs_=1e-20) and _close(lean["chi2_init"], plain["chi2_init"], rel=1e-12)
    _compare_lm(b, plain, cpu)   # (both kernels carry the exact statement -- decision replay -- on their own; how often their trial counts coincide is not asserted)


@pytest.mark.gpu
def test_speculative_single_capsule_run_is_the_sequential_loop_bit_for_bit(se2_batch, monkeypatch):
    """A batch of ONE capsule (what define_new_keyframe -> optimize_edges hands over) runs as spec_w replicas that evaluate the next steps of the lambda ladder at once (k_lm_spec,
    optimize_edges.h:685-687: a rejected trial only moves lambda). Every result field, trace entry, unknown edge and spanning-tree pose must be bit-identical to the plain
    two-wavefront run of the same capsule (SRBA_HIP_SPEC=0), for 4, 8 and 16 replicas; and that run agrees with the oracle (decision replay)."""
    b = se2_batch; P, L, O, PD = capi.DIMS[b.family]
    idx = list(range(0, b.n, 3))
    def run(spec):
        monkeypatch.setenv("SRBA_HIP_SPEC", str(spec)); out = []
        ctx = runner.HipContext(b.params)
        for i in idx:
            s = b.sub(i, 1); ctx.upload(s); r = ctx.lm_run(); w = s.clone(); ctx._chk(ctx.lib.srba_hip_download_state(ctx.ctx, w.ptr, 1), "download_state")
            r["edge"] = w.array(0, "edge_pose", np.float64, s[0].n_unk_edges * PD); r["pose"] = w.array(0, "pose", np.float64, 2 * s[0].n_pairs * PD); out.append(r)
        ctx.close(); return out
    plain = run(0)
    assert np.mean([r["num_trials"][0] for r in plain]) > 10
    for W in (8, 4, 16):
        got = run(W)
        for i, (p, g) in enumerate(zip(plain, got)):
            for k in p:
                assert np.array_equal(np.asarray(p[k]), np.asarray(g[k]), equal_nan=True), (W, idx[i], k)
    monkeypatch.setenv("SRBA_HIP_SPEC", "8")
    for i in idx[:40]:
        s = b.sub(i, 1); _replay_exact(s, runner.run_batch_hip(s))


@pytest.mark.gpu
def test_speculative_run_whose_replicas_lose_step_falls_back_to_the_sequential_loop(se2_batch, monkeypatch):
    """k_lm_spec needs all its replicas resident; when one does not come (a shared GPU, another context on the CUs) the waiters give up after the spin bound
    and the capsule used to FAIL (status 2, RbaEngine throws). Now the library restores the unknown edges the launch started from and runs the capsule once on the sequential
    path: same record, same written-back state as a plain run, bit for bit. The test knob SRBA_HIP_SPEC_TEST_DROP never launches the last replica."""
    import ctypes as C
    b = se2_batch; P, L, O, PD = capi.DIMS[b.family]
    idx = [i for i in range(b.n) if b[i].n_unk_edges >= 3][:3]
    def run(drop, one_call):
        monkeypatch.setenv("SRBA_HIP_SPEC", "0" if drop is None else "8")
        if drop: monkeypatch.setenv("SRBA_HIP_SPEC_TEST_DROP", "1")
        else: monkeypatch.delenv("SRBA_HIP_SPEC_TEST_DROP", raising=False)
        out = []; ctx = runner.HipContext(b.params)
        for i in idx:
            w = b.clone(i, 1)
            if one_call: r = ctx.optimize_capsule(w)
            else: ctx.upload(w); r = ctx.lm_run(); ctx._chk(ctx.lib.srba_hip_download_state(ctx.ctx, w.ptr, 1), "download_state")
            r["edge"] = w.array(0, "edge_pose", np.float64, w[0].n_unk_edges * PD).copy(); r["pose"] = w.array(0, "pose", np.float64, 2 * w[0].n_pairs * PD).copy(); out.append(r)
        st = (C.c_int64 * 2)(); ctx.lib.srba_hip_spec_stats(ctx.ctx, st); ctx.close(); return out, (int(st[0]), int(st[1]))
    plain, st0 = run(None, False); assert st0 == (0, 0)
    for one_call in (False, True):
        got, st = run(True, one_call)
        assert st[0] == len(idx) and st[1] >= 1, st          # every capsule was tried speculatively; at least one run consulted the missing replica and fell back
        for i, (p_, g) in enumerate(zip(plain, got)):
            assert g["status"][0] == 0
            for k in p_:
                assert np.array_equal(np.asarray(p_[k]), np.asarray(g[k]), equal_nan=True), (one_call, idx[i], k)


@pytest.mark.gpu
def test_class_launches_start_largest_footprint_first(se2_batch, monkeypatch):
    """the staggered start of the class launches (plan_launches: every class stream held back by a delay kernel so that the big, LDS-bound classes get
    their workgroups placed before the wave-slot-bound small ones fill the CUs) was a timing heuristic nothing checked. Every persistent launch now stamps the device time at
    which its first capsule was taken (srba_hip_launch_order). What the stamps show (tools/diag_launch_stamps.py, profiles/r05_launch_order.txt) and what is asserted here, on 24
    consecutive launches of a batch with many size classes: the LARGEST-footprint launch takes its first capsule before every other launch does, every time. On the benchmark batch
    the three largest start in plan order at their programmed delays; whatever comes after the chip is full does not start at its delay but when the dispatcher finds room, in no
    particular order (wave-slot-bound fillers: their order is immaterial) -- reported by bench.py (`launch_order`), not asserted."""
    import ctypes as C
    fb = _replicate(se2_batch, 40)
    monkeypatch.setenv("SRBA_HIP_LEAN_MIN_COUNT", "64"); monkeypatch.setenv("SRBA_HIP_TWO_MIN_COUNT", "64")
    ctx = runner.HipContext(se2_batch.params); ctx.upload(fb); lib = ctx.lib
    stamp = (C.c_int64 * 64)(); wgs = (C.c_int32 * 64)(); dly = (C.c_int32 * 64)(); held = 0; njobs = 0; log = []
    for it in range(25):
        lib.srba_hip_reset_state(ctx.ctx); lib.srba_hip_lm_run_async(ctx.ctx); lib.srba_hip_sync(ctx.ctx)
        njobs = lib.srba_hip_launch_order(ctx.ctx, stamp, wgs, dly, 64)
        assert njobs >= 5, njobs                                     # several size classes, or the test says nothing
        t = np.array([stamp[j] for j in range(njobs)]); assert (t > 0).all(), t   # every launch took a capsule
        ok = bool(t[0] == t.min())
        if it > 0: held += int(ok); log.append(((t - t.min()) / 100.0).round().astype(int).tolist())   # (the first launch pays for module loading)
    ctx.close()
    assert held == 24, (held, njobs, [dly[j] for j in range(njobs)], log[:3])


@pytest.mark.gpu
def test_optimize_capsule_is_upload_run_download_in_one_call(se2_batch):
    """srba_hip_optimize_capsule (what RbaEngine<>::optimize_edges binds per key-frame: one wait for the device) against the three calls it stands for, on the same context, capsule
    after capsule (the staging buffers are reused from one call to the next): result records and written-back unknowns / spanning-tree poses bit-identical; and a landmark family,
    which takes the three calls inside."""
    b = se2_batch; P, L, O, PD = capi.DIMS[b.family]
    ctx = runner.HipContext(b.params)
    for i in range(0, b.n, 5):
        one = b.clone(i, 1); three = b.clone(i, 1)
        ctx.upload(three); r3 = ctx.lm_run(); ctx._chk(ctx.lib.srba_hip_download_state(ctx.ctx, three.ptr, 1), "download_state")
        r1 = ctx.optimize_capsule(one)
        for k in r3:
            assert np.array_equal(np.asarray(r3[k]), np.asarray(r1[k]), equal_nan=True), (i, k)
        for field, cnt in (("edge_pose", one[0].n_unk_edges * PD), ("pose", 2 * one[0].n_pairs * PD)):
            assert np.array_equal(one.array(0, field, np.float64, cnt), three.array(0, field, np.float64, cnt)), (i, field)
        assert not np.array_equal(one.array(0, "edge_pose", np.float64, one[0].n_unk_edges * PD), b.array(i, "edge_pose", np.float64, one[0].n_unk_edges * PD)) or r1["num_accepted"][0] == 0
    ctx.close()
    ds, _ = datasets.landmarks_dataset_se2("rb2d", n_kf=12, n_lm=200, seed=3, noise=1e-3)
    eng = runner.landmark_engine("rb2d", backend=_oracle.BACKEND, solver=capi.SOLVER_SCHUR_DENSE, depth=2); eng.run(ds); lb = eng.harvest(); lb.engine = eng
    P, L, O, PD = capi.DIMS[lb.family]; ctx = runner.HipContext(lb.params)
    for i in range(max(0, lb.n - 4), lb.n):
        one = lb.clone(i, 1); three = lb.clone(i, 1)
        ctx.upload(three); r3 = ctx.lm_run(); ctx._chk(ctx.lib.srba_hip_download_state(ctx.ctx, three.ptr, 1), "download_state")
        r1 = ctx.optimize_capsule(one)
        for k in ("num_trials", "chi2_final", "trace_chi2"):
            assert np.array_equal(np.asarray(r3[k]), np.asarray(r1[k]), equal_nan=True), (i, k)
        for field, cnt in (("edge_pose", one[0].n_unk_edges * PD), ("ulm_pos", one[0].n_unk_lms * L), ("pose", 2 * one[0].n_pairs * PD)):
            assert np.array_equal(one.array(0, field, np.float64, cnt), three.array(0, field, np.float64, cnt)), (i, field)
    ctx.close()


@pytest.mark.gpu
def test_two_wavefronts_per_capsule_match_oracle(se2_batch, monkeypatch):
    """k_lm_run2 (round 4: two wavefronts on one capsule -- what a big relative-pose batch runs its big, LDS-bound windows on) forced onto every class of this batch, loop-closure
    windows included: the same parity statements as k_lm_run (decision replay over the whole run of every window), reproducible run to run (its group reductions and the
    split of the Hessian terms between the wavefronts have a fixed order), and within rounding of the one-wavefront kernel."""
    b = se2_batch
    monkeypatch.setenv("SRBA_HIP_TWO_FROM_KB", "1"); monkeypatch.setenv("SRBA_HIP_TWO_MIN_COUNT", "1"); monkeypatch.setenv("SRBA_HIP_LEAN", "0")
    two = runner.run_batch_hip(b, download=True); again = runner.run_batch_hip(b)
    monkeypatch.setenv("SRBA_HIP_TWO", "0"); plain = runner.run_batch_hip(b, download=True)
    cpu = _oracle.run_batch(b)
    _compare_lm(b, two, cpu)
    for k in ("num_trials", "chi2_final", "trace_chi2", "trace_rho"):
        assert np.array_equal(two[k], again[k], equal_nan=True), k
    assert _close(two["chi2_final"], plain["chi2_final"], rel=1e-9, abs_=1e-20) and _close(two["chi2_init"], plain["chi2_init"], rel=1e-12)
    P, L, O, PD = capi.DIMS[b.family]
    for i in range(b.n):
        g = two["state"].array(i, "edge_pose", np.float64, b[i].n_unk_edges * PD); c = plain["state"].array(i, "edge_pose", np.float64, b[i].n_unk_edges * PD)
        assert np.allclose(g, c, rtol=1e-7, atol=1e-8), i
        g = two["state"].array(i, "pose", np.float64, 2 * b[i].n_pairs * PD); c = plain["state"].array(i, "pose", np.float64, 2 * b[i].n_pairs * PD)
        assert np.allclose(g, c, rtol=1e-7, atol=1e-7), i


def test_lm_run_matches_oracle_se2(se2_batch):
    b = se2_batch
    assert b.n > 200
    gpu = runner.run_batch_hip(b, download=True)
    cpu = _oracle.run_batch(b, keep_state=True)
    n_full = _compare_lm(b, gpu, cpu)
    assert n_full > b.n // 5  # a good share of the capsules follow the identical trial sequence to the end
    # final unknowns and spanning-tree poses (metres / radians; both runs stop at the same minimum)
    P, L, O, PD = capi.DIMS[b.family]
    for i in range(b.n):
        nk = b[i].n_unk_edges
        g = gpu["state"].array(i, "edge_pose", np.float64, nk * PD); c = cpu["state"].array(i, "edge_pose", np.float64, nk * PD)
        assert np.allclose(g, c, rtol=1e-6, atol=1e-7), i
        g = gpu["state"].array(i, "pose", np.float64, 2 * b[i].n_pairs * PD); c = cpu["state"].array(i, "pose", np.float64, 2 * b[i].n_pairs * PD)
        assert np.allclose(g, c, rtol=1e-6, atol=1e-6), i


@pytest.mark.parametrize("fused", ["1", "0", "tiny"])
def test_stepwise_kernels_match_oracle_se2(se2_batch, fused, monkeypatch):
    """K1, K4, then srba_hip_linearize: for this family the fused normal-equations kernel (srba_assemble.hpp: Jacobian blocks stay in LDS, read back on demand), with
    SRBA_HIP_ASSEMBLE=0 the unfused one (blocks through HBM), "tiny": a 10 KB image limit, so that the larger capsules overflow the classes and the two kernels share the batch."""
    b = se2_batch
    if fused == "tiny": monkeypatch.setenv("SRBA_HIP_ASSEMBLE_MAX_KB", "10")
    else: monkeypatch.setenv("SRBA_HIP_ASSEMBLE", fused)
    ctx = runner.HipContext(b.params); ctx.upload(b)
    lib = ctx.lib
    assert lib.srba_hip_update_spantree(ctx.ctx, 0) == 0
    chi2 = np.zeros(b.n); assert lib.srba_hip_eval_residuals(ctx.ctx, chi2.ctypes.data_as(capi.PF64)) == 0
    assert lib.srba_hip_linearize(ctx.ctx) == 0
    res, Jp, HAp, grad, poses = ctx.debug(0), ctx.debug(1), ctx.debug(3), ctx.debug(6), ctx.debug(9)
    assert lib.srba_hip_solve(ctx.ctx, None, None) == 0  # lambda = the guess computed by linearize
    delta = ctx.debug(7)
    P, L, O, PD = capi.DIMS[b.family]
    o = dict(res=0, Jp=0, HAp=0, grad=0, poses=0)
    for i in range(b.n):
        c = b[i]
        ref = _oracle.stage(b, i, do_solve=True, lam=0.0)  # lam filled below
        n = P * c.n_unk_edges + L * c.n_unk_lms
        sl = lambda key, cnt: slice(o[key], o[key] + cnt)
        assert _close(chi2[i], ref["scalars"][0]), i
        assert np.allclose(res[sl("res", c.n_obs * O)], ref["resid"], rtol=1e-9, atol=1e-12), i
        assert np.allclose(Jp[sl("Jp", c.n_bp * O * P)], ref["Jp"], rtol=1e-9, atol=1e-12), i
        assert np.allclose(poses[sl("poses", 2 * c.n_pairs * PD)], ref["poses"], rtol=1e-9, atol=1e-12), i
        noise_only = ref["scalars"][0] < 1e-24   # e.g. the 2-key-frame window of KF#1: the residual is rounding noise, so are g and delta; nothing to compare relatively
        assert noise_only or np.allclose(grad[sl("grad", n)], ref["grad"], rtol=1e-7, atol=1e-9 * np.abs(ref["grad"]).max()), i
        # HAp (no Schur here: unchanged by the solve) and the LM step for lambda0
        ref2 = _oracle.stage(b, i, do_solve=True, lam=ref["scalars"][1])
        assert np.allclose(HAp[sl("HAp", c.n_hap * P * P)], ref2["HAp"], rtol=1e-9, atol=1e-9 * np.abs(ref2["HAp"]).max()), i
        assert noise_only or np.allclose(delta[sl("grad", n)], ref2["delta"], rtol=1e-6, atol=1e-9 * max(1e-30, np.abs(ref2["delta"]).max())), i
        o["res"] += c.n_obs * O; o["Jp"] += c.n_bp * O * P; o["HAp"] += c.n_hap * P * P; o["grad"] += n; o["poses"] += 2 * c.n_pairs * PD
    ctx.close()


def test_engine_with_hip_backend_submaps_loop_closure():
    """tests/submaps_edge_init_values.cpp:81-173 (MiniProblems.SubmapsEdgesInitValues) through the product front-end + GPU."""
    ds = datasets.graph_slam_from_entries(datasets.C1_SUBMAPS, 1e-3, np.radians(0.05), seed=1)
    eng = runner.graph_slam_engine(backend="hip", submap=5, depth=3, sigma_xy=1e-3, sigma_yaw_deg=0.05, solver=capi.SOLVER_SCHUR_DENSE, max_error_per_obs_to_stop=1e-6)
    seen_lc = False
    for k in ds:
        info = eng.add_keyframe(k["feat_ids"], k["z"], k["flags"])
        if info.n_new_edges == 2:
            seen_lc = True
            inv = 2 ** 64 - 1
            assert info.lc_base[0] != inv or info.lc_base[1] != inv
            assert info.lc_observer[0] != inv or info.lc_observer[1] != inv
            assert info.num_observations > 1 and info.obs_rmse < 1e-6
    assert seen_lc


@pytest.mark.gpu
def test_eval_overall_squared_error_hip_vs_oracle():
    """srba_hip_eval_overall_sqr_error (K1 + K4 over the whole map) against the oracle's restatement, through the same front-end."""
    from srba_amd import datasets
    ds = datasets.graph_slam_se2(n_kf=400, seed=5, path="tour", sigma_xy=0.02, sigma_yaw_deg=0.5)
    vals = {}
    for backend in ("oracle", "hip"):
        eng = runner.graph_slam_engine(backend=_oracle.BACKEND if backend == "oracle" else "hip", submap=10, depth=3, sigma_xy=0.02, sigma_yaw_deg=0.5, harvest=0)
        eng.run(ds)
        vals[backend] = eng.eval_overall_squared_error()
    assert vals["oracle"] > 0 and abs(vals["hip"] - vals["oracle"]) <= 1e-6 * vals["oracle"], vals
    # landmark family (SE3 + stereo, sensor pose on the robot)
    ds3, _ = datasets.landmarks_dataset_se3("stereo", n_kf=12, n_lm=300, seed=5, noise=0.3)
    vals = {}
    for backend in ("oracle", "hip"):
        eng = runner.landmark_engine("stereo", backend=_oracle.BACKEND if backend == "oracle" else "hip", harvest=0)
        eng.run(ds3)
        vals[backend] = eng.eval_overall_squared_error()
    assert vals["oracle"] > 0 and abs(vals["hip"] - vals["oracle"]) <= 1e-6 * vals["oracle"] + 1e-9, vals


@pytest.mark.gpu
def test_big_capsule_path_and_mixed_classes(se2_batch, monkeypatch):
    """Capsules whose system does not fit one wavefront's LDS run on the multi-workgroup path (srba_big.hpp: grid-wide phase kernels, dense blocked Cholesky with FP64
    MFMA updates, LM control on the host). The knob SRBA_HIP_MAX_LDS_KB forces ordinary capsules onto it: all of a batch (0), or only its larger ones (10 KB -> a mix of
    persistent one-wavefront launches and big-path capsules in the same call)."""
    sub = se2_batch.sub(se2_batch.n - 60, 60)
    ref = _oracle.run_batch(sub)
    for kb in ("0", "10"):
        monkeypatch.setenv("SRBA_HIP_MAX_LDS_KB", kb)
        gpu = runner.run_batch_hip(sub)
        _compare_lm(sub, gpu, ref)
    monkeypatch.delenv("SRBA_HIP_MAX_LDS_KB")


@pytest.mark.gpu
@pytest.mark.parametrize("kind,solver", [("stereo", capi.SOLVER_SCHUR_DENSE), ("mono", capi.SOLVER_SCHUR_DENSE), ("cart3d", capi.SOLVER_NO_SCHUR_SPARSE), ("rb2d", capi.SOLVER_SCHUR_SPARSE)])
def test_big_capsule_path_landmark_families(kind, solver, monkeypatch):
    """The same path with landmarks: Schur reduction, dense reduced system, landmark back-substitution (and the full system for the no-Schur solver), forced on small capsules."""
    from test_oracle_numeric import _harvest
    b = _harvest(kind, solver=solver, n_kf=12)
    sub = b.sub(max(0, b.n - 5), min(5, b.n)); ref = _oracle.run_batch(sub)
    monkeypatch.setenv("SRBA_HIP_MAX_LDS_KB", "0")
    gpu = runner.run_batch_hip(sub)
    monkeypatch.delenv("SRBA_HIP_MAX_LDS_KB")
    assert np.all(gpu["status"] == ref["status"]) and _close(gpu["chi2_init"], ref["chi2_init"], rel=1e-9)
    assert _close(gpu["chi2_final"], ref["chi2_final"], rel=1e-6, abs_=1e-18)
    assert np.array_equal(gpu["num_observations"], ref["num_observations"]) and np.array_equal(gpu["num_jacobians"], ref["num_jacobians"])


@pytest.mark.gpu
def test_large_windows_depth5():
    """Maximum-size case for the LDS path: depth-5 windows of 20-key-frame sub-maps (systems of 100+ block rows), SE2 graph-SLAM."""
    from srba_amd import datasets
    ds = datasets.graph_slam_se2(n_kf=260, seed=9, path="tour")
    eng = runner.graph_slam_engine(backend=_oracle.BACKEND, submap=20, depth=5)
    eng.run(ds)
    b = eng.harvest(); b.engine = eng
    sub = b.sub(b.n - 40, 40)
    ref = _oracle.run_batch(sub); gpu = runner.run_batch_hip(sub)
    assert max(sub.ptr[i].n_unk_edges for i in range(sub.n)) >= 60
    _compare_lm(sub, gpu, ref)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,solver", [(k, s) for k in ("rb2d", "cart3d", "stereo", "rb3d", "stereo_se2") for s in (capi.SOLVER_SCHUR_DENSE, capi.SOLVER_SCHUR_SPARSE, capi.SOLVER_NO_SCHUR_SPARSE)])
def test_landmark_families_all_solvers(kind, solver):
    """Every reference solver (lev-marq_solvers.h: Schur+dense LLT, Schur+sparse, full sparse) on landmark problems: the device factors the
    same SPD system in one way, so all three must reproduce the oracle's chi2 (which runs the reference's three code paths)."""
    if kind in ("rb2d",):
        ds, _ = datasets.landmarks_dataset_se2(kind, n_kf=30, n_lm=900, seed=7, noise=1e-3)
    elif kind == "stereo_se2":
        ds, _ = datasets.landmarks_dataset_se2_stereo(n_kf=16, n_lm=90, seed=7, noise=0.1)
    else:
        ds, _ = datasets.landmarks_dataset_se3(kind, n_kf=16, n_lm=320, seed=7, noise=(0.1 if kind == "stereo" else 1e-3))
    eng = runner.landmark_engine(kind, backend=_oracle.BACKEND, solver=solver)
    eng.run(ds)
    b = eng.harvest(); b.engine = eng
    sub = b.sub(max(0, b.n - 8), min(8, b.n))
    ref = _oracle.run_batch(sub); gpu = runner.run_batch_hip(sub)
    assert np.all(gpu["status"] == ref["status"])
    assert _close(gpu["chi2_init"], ref["chi2_init"], rel=1e-9)
    assert _close(gpu["chi2_final"], ref["chi2_final"], rel=1e-6, abs_=1e-18)
    assert np.array_equal(gpu["num_observations"], ref["num_observations"]) and np.array_equal(gpu["num_jacobians"], ref["num_jacobians"])


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["stereo", "cart3d"])
def test_workgroup_landmark_kernels_repeat_to_rounding(kind):
    """k_lm_wg sums the U_Ap blocks and the Schur gradient correction with LDS atomics from several wavefronts: the order of the additions is not fixed, so two runs of one
    batch agree to rounding, not bit for bit -- unlike the one-wavefront kernels and the deep-window gang, which are reproducible. What is asserted: same status, same initial chi2 bit
    for bit (no atomics before the first Hessian), chi2 of the first accepted trial to 1e-12, the final chi2 of windows that converged in both runs to 1e-6 (the parity bound itself)."""
    ds, _ = datasets.landmarks_dataset_se3(kind, n_kf=16, n_lm=320, seed=7, noise=(0.1 if kind == "stereo" else 1e-3))
    eng = runner.landmark_engine(kind, backend=_oracle.BACKEND, solver=capi.SOLVER_SCHUR_DENSE); eng.run(ds)
    b = eng.harvest(); b.engine = eng; sub = b.sub(max(0, b.n - 8), min(8, b.n))
    r1 = runner.run_batch_hip(sub); r2 = runner.run_batch_hip(sub)
    assert np.array_equal(r1["status"], r2["status"]) and np.array_equal(r1["chi2_init"], r2["chi2_init"])
    t1, t2 = np.asarray(r1["trace_chi2"]), np.asarray(r2["trace_chi2"])
    first = np.isfinite(t1[:, 0]) & np.isfinite(t2[:, 0])
    assert np.allclose(t1[first, 0], t2[first, 0], rtol=1e-12, atol=0)
    conv = (r1["chi2_final"] < 0.5 * r1["chi2_init"]) & (r2["chi2_final"] < 0.5 * r2["chi2_init"])
    assert conv.any() and _close(r1["chi2_final"][conv], r2["chi2_final"][conv], rel=1e-6, abs_=1e-18)


@pytest.mark.gpu
def test_abi_misuse_is_reported_not_fatal(se2_batch):
    """error behaviour of the C ABI: run before upload, empty upload, malformed capsule, under-determined problem (optimize_edges.h:355)"""
    import ctypes as C
    ctx = runner.HipContext(se2_batch.params)
    lib = ctx.lib
    assert lib.srba_hip_lm_run(ctx.ctx, None) != 0 and b"no batch" in lib.srba_hip_last_error(ctx.ctx)
    assert lib.srba_hip_upload_problems(ctx.ctx, se2_batch.ptr, 0) != 0
    bad = se2_batch.clone(0, 1); bad.ptr[0].n_unk_edges = bad.ptr[0].n_edges + 1
    assert lib.srba_hip_upload_problems(ctx.ctx, bad.ptr, 1) != 0 and b"malformed" in lib.srba_hip_last_error(ctx.ctx)
    bad = se2_batch.clone(40, 1); assert bad.ptr[0].n_obs > 1; bad.ptr[0].n_obs = 1      # block tables now point past the observation table
    assert lib.srba_hip_upload_problems(ctx.ctx, bad.ptr, 1) != 0 and b"malformed" in lib.srba_hip_last_error(ctx.ctx)
    ctx.upload(se2_batch.sub(0, 4)); r = ctx.lm_run()   # the context stays usable
    assert np.all(r["status"] == 0)


@pytest.mark.gpu
def test_underdetermined_problem_raises_like_the_reference():
    """optimize_edges.h:355 ASSERT_ABOVEEQ_(OBS_DIMS*nObs, nUnknowns): one range-bearing landmark shared by two key-frames gives 4 observation
    scalars for 5 unknowns; the kernel reports status 1 without iterating and the front-end throws."""
    eng = runner.landmark_engine("rb2d", backend="hip", harvest=0, min_obs_to_loop_closure=1)
    eng.add_keyframe([7], np.array([[2.0, 0.1]]), flags=np.zeros(1))
    with pytest.raises(RuntimeError, match="OBS_DIMS"):
        eng.add_keyframe([7], np.array([[1.5, 0.2]]), flags=np.zeros(1))


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["rb2d", "cart3d", "stereo"])
def test_schur_complement_on_device_equals_dense(kind):
    """The reference's SchurTests (tests/schur_unittest.cpp:147-267: block-sparse Schur complement == dense formula, 1e-10 relative, lambda = 1e3) on the
    device: K7 reduces HAp in place, so after srba_hip_solve the HAp read-back must equal Hpp - Hpf (Hff + lambda I)^-1 Hpf^t built in numpy from the
    un-reduced device blocks, and the step must solve the full damped system."""
    from test_oracle_numeric import _dense_system, _harvest
    b = _harvest(kind)
    P, L, O, PD = capi.DIMS[b.family]
    checked = 0
    for i in range(max(0, b.n - 4), b.n):
        c = b[i]
        if c.n_unk_lms == 0 or c.n_unk_edges < 2:
            continue
        sub = b.sub(i, 1)
        ctx = runner.HipContext(b.params); ctx.upload(sub); lib = ctx.lib
        assert lib.srba_hip_update_spantree(ctx.ctx, 0) == 0 and lib.srba_hip_eval_residuals(ctx.ctx, None) == 0 and lib.srba_hip_linearize(ctx.ctx) == 0
        a0 = dict(HAp=ctx.debug(3).copy(), Hf=ctx.debug(4).copy(), HApf=ctx.debug(5).copy(), grad=ctx.debug(6).copy())
        H, n, nK, nF = _dense_system(sub, 0, a0)
        lam = np.array([1e3]); notpd = np.zeros(1, np.int32)
        assert lib.srba_hip_solve(ctx.ctx, lam.ctypes.data_as(capi.PF64), notpd.ctypes.data_as(capi.PI32)) == 0 and notpd[0] == 0
        HAp1, delta = ctx.debug(3), ctx.debug(7)
        Hpp, Hpf, Hff = H[:P * nK, :P * nK], H[:P * nK, P * nK:], H[P * nK:, P * nK:]
        S = Hpp - Hpf @ np.linalg.inv(Hff + lam[0] * np.eye(L * nF)) @ Hpf.T
        hi, hj = sub.array(0, "hap_i", np.int32, c.n_hap), sub.array(0, "hap_j", np.int32, c.n_hap)
        for k in range(c.n_hap):
            blk = HAp1[k * P * P:(k + 1) * P * P].reshape(P, P); ref = S[P * hi[k]:P * hi[k] + P, P * hj[k]:P * hj[k] + P]
            if hi[k] == hj[k]:
                blk = np.triu(blk); ref = np.triu(ref)
            assert np.abs(blk - ref).max() <= 1e-10 * np.abs(S).max(), (kind, i, k)
        full = np.linalg.solve(H + lam[0] * np.eye(n), a0["grad"])
        assert np.allclose(delta, full, rtol=1e-7, atol=1e-9 * np.abs(full).max()), (kind, i)
        ctx.close(); checked += 1
    assert checked >= 2


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["rb2d", "cart2d", "cart3d", "rb3d", "stereo", "mono", "stereo_se2"])
def test_stepwise_kernels_match_oracle_landmark_families(kind):
    """K1, K4, K2, K3, K6, K5 one launch each (srba_hip_update_spantree / eval_residuals / linearize) against the oracle's arrays of the same capsule:
    residuals, dh_dAp and dh_df blocks, the three Hessian block sets and the gradient, for every landmark family."""
    from test_oracle_numeric import _harvest
    b = _harvest(kind)
    P, L, O, PD = capi.DIMS[b.family]
    for i in range(max(0, b.n - 3), b.n):
        c = b[i]; sub = b.sub(i, 1)
        ctx = runner.HipContext(b.params); ctx.upload(sub); lib = ctx.lib
        chi = np.zeros(1)
        assert lib.srba_hip_update_spantree(ctx.ctx, 0) == 0 and lib.srba_hip_eval_residuals(ctx.ctx, chi.ctypes.data_as(capi.PF64)) == 0 and lib.srba_hip_linearize(ctx.ctx) == 0
        ref = _oracle.stage(b, i)
        assert _close(chi[0], ref["scalars"][0], rel=1e-9)
        for what, key in ((0, "resid"), (1, "Jp"), (2, "Jf"), (3, "HAp"), (4, "Hf"), (5, "HApf"), (6, "grad"), (9, "poses")):
            g = ctx.debug(what); r = ref[key]
            assert g.shape == r.shape, (kind, key)
            if r.size:
                assert np.allclose(g, r, rtol=1e-8, atol=1e-9 * max(1e-300, np.abs(r).max())), (kind, i, key, float(np.abs(g - r).max()), float(np.abs(r).max()))
        ctx.close()


@pytest.mark.gpu
def test_relpose_se3_family_matches_oracle():
    """<SE3, RelativePoses3D, RelativePoses_3D> (SE(3) relative graph-SLAM, 6x6 blocks, constant 6x6 information matrix): per-kernel arrays and the whole LM loop against the oracle."""
    ds, _ = datasets.graph_slam_se3(n_kf=40, seed=6)
    eng = runner.graph_slam_engine_se3(backend=_oracle.BACKEND); eng.run(ds)
    b = eng.harvest(); b.engine = eng
    for i in (b.n - 1, b.n - 7):
        sub = b.sub(i, 1); ctx = runner.HipContext(b.params); ctx.upload(sub); lib = ctx.lib; chi = np.zeros(1)
        assert lib.srba_hip_update_spantree(ctx.ctx, 0) == 0 and lib.srba_hip_eval_residuals(ctx.ctx, chi.ctypes.data_as(capi.PF64)) == 0 and lib.srba_hip_linearize(ctx.ctx) == 0
        ref = _oracle.stage(b, i)
        assert _close(chi[0], ref["scalars"][0], rel=1e-9)
        for what, key in ((0, "resid"), (1, "Jp"), (3, "HAp"), (6, "grad"), (9, "poses")):
            g = ctx.debug(what); r = ref[key]
            assert g.shape == r.shape and np.allclose(g, r, rtol=1e-8, atol=1e-9 * max(1e-300, np.abs(r).max())), (i, key, float(np.abs(g - r).max()))
        ctx.close()
    sub = b.sub(b.n - 24, 24)
    ref = _oracle.run_batch(sub); gpu = runner.run_batch_hip(sub)
    assert np.all(gpu["status"] == ref["status"]) and _close(gpu["chi2_init"], ref["chi2_init"], rel=1e-9)
    assert _close(gpu["chi2_final"], ref["chi2_final"], rel=1e-6, abs_=1e-18)


@pytest.mark.gpu
def test_new_families_through_the_engine_with_hip_backend():
    """define_new_keyframe() with the GPU back-end for the two families added in round 2: SE(3) relative graph-SLAM recovers a noise-free map; the planar robot with
    a stereo camera (SE(2) key-frames, 3D landmarks) ends at the noise floor and at the ground-truth edges."""
    ds, gt = datasets.graph_slam_se3(n_kf=8, seed=1, sigma_xyz=0, sigma_ang_deg=0)
    eng = runner.graph_slam_engine_se3(backend="hip", harvest=0); infos = eng.run(ds)
    assert all(i.chi2_final < 1e-12 for i in infos)
    fr, to, pose = eng.edges()
    for k in range(len(fr)):
        T = np.linalg.inv(gt[int(to[k])]) @ gt[int(fr[k])]
        assert np.abs(pose[k][:3] - T[:3, 3]).max() < 1e-7 and np.abs(pose[k][3:].reshape(3, 3) - T[:3, :3]).max() < 1e-7
    ds2, gt2 = datasets.landmarks_dataset_se2_stereo(n_kf=14, n_lm=120, seed=2, noise=0.05)
    eng2 = runner.landmark_engine("stereo_se2", backend="hip", robust=0, harvest=0); infos = eng2.run(ds2)
    assert infos[-1].obs_rmse < 0.15
    fr, to, pose = eng2.edges()
    for k in range(len(fr)):
        assert np.abs(pose[k] - np.array(datasets._inv_compose2(gt2[int(fr[k])], gt2[int(to[k])]))).max() < 5e-3


@pytest.mark.gpu
def test_deep_monocular_window_on_the_big_path_matches_oracle():
    """BASELINE config 4 at reduced scale (100 key-frames, 2 000 landmarks, max_tree_depth = max_optimize_depth = 8, sub-maps of 20, Schur + dense Cholesky): the local area
    covers the whole map -- ~100 unknown edges (a dense reduced system of ~600), ~1 600 unknown landmarks, ~35 000 observations per capsule -- and runs on the multi-workgroup
    path (grid-wide phases, blocked Cholesky with FP64 MFMA updates). chi2 against the oracle at 1e-6, identical counters, same accept / reject prefix."""
    ds, _ = datasets.mono_deep_window(n_kf=100, n_lm=2000, seed=1)
    eng = runner.landmark_engine("mono", backend=_oracle.BACKEND, depth=8, submap=20, sigma=0.5, robust=0, harvest=1, cam=(200., 200., 400., 320.), refresh_all_read_poses=2)
    eng.run(ds); b = eng.harvest(); b.engine = eng
    sub = b.sub(b.n - 2, 2)
    assert sub[1].n_unk_edges == 99 and sub[1].n_unk_lms > 1000 and sub[1].n_obs > 20000
    ref = _oracle.run_batch(sub, keep_state=True)
    ctx = runner.HipContext(sub.params); ctx.upload(sub); gpu = ctx.lm_run()
    import ctypes as C
    st = (C.c_double * 4)(); ctx.lib.srba_hip_big_path_stats(ctx.ctx, st)
    assert st[2] >= gpu["num_trials"].sum() - gpu["num_not_pd"].sum() and st[3] == 6 * 99   # every solve went through the dense blocked factorisation of the 594-unknown reduced system
    assert np.all(gpu["status"] == 0) and np.array_equal(gpu["num_observations"], ref["num_observations"]) and np.array_equal(gpu["num_jacobians"], ref["num_jacobians"])
    assert _close(gpu["chi2_init"], ref["chi2_init"], rel=1e-9) and _close(gpu["lambda_init"], ref["lambda_init"], rel=1e-9)
    assert _close(gpu["chi2_final"], ref["chi2_final"], rel=1e-6)
    for i in range(sub.n):   # accepted / rejected decisions agree while both are away from the rounding floor
        m = min(gpu["num_trials"][i], ref["num_trials"][i], 4)
        assert np.array_equal(np.sign(gpu["trace_rho"][i][:m]), np.sign(ref["trace_rho"][i][:m]))
        assert _close(gpu["trace_chi2"][i][:m], ref["trace_chi2"][i][:m], rel=1e-6)
    work = sub.clone(); ctx._chk(ctx.lib.srba_hip_download_state(ctx.ctx, work.ptr, work.n), "download"); ctx.close()
    for i in range(sub.n):
        g = work.array(i, "edge_pose", np.float64, sub[i].n_unk_edges * 12); c = ref["state"].array(i, "edge_pose", np.float64, sub[i].n_unk_edges * 12)
        assert np.allclose(g, c, rtol=1e-5, atol=1e-6), i


@pytest.mark.gpu
def test_deep_windows_schur_reduction_forms_and_gang_count_agree(monkeypatch):
    """Round 5, multi-workgroup path: the Schur reduction with a wavefront per U_Ap block (kb_schur_reduce_wave, default) against the workgroup-per-block kernel of rounds 3-4
    (SRBA_HIP_SCHUR_WAVE=0) -- different summation trees: chi2 to 1e-9 of each other, both within 1e-6 of the oracle -- and one gang against two and four gangs side by side
    (SRBA_HIP_BIG_GANGS): a window's numbers do not depend on which gang it ran in, bit for bit. Four deep monocular windows of different sizes."""
    ds, _ = datasets.mono_deep_window(n_kf=100, n_lm=2000, seed=1)
    eng = runner.landmark_engine("mono", backend=_oracle.BACKEND, depth=8, submap=20, sigma=0.5, robust=0, harvest=1, cam=(200., 200., 400., 320.), refresh_all_read_poses=2)
    eng.run(ds); b = eng.harvest(); b.engine = eng
    sub = b.sub(b.n - 4, 4); ref = _oracle.run_batch(sub)
    def run(wave, gangs):
        monkeypatch.setenv("SRBA_HIP_SCHUR_WAVE", wave); monkeypatch.setenv("SRBA_HIP_BIG_GANGS", gangs)
        ctx = runner.HipContext(sub.params); ctx.upload(sub); out = ctx.lm_run(); ctx.close(); return out
    w1g1, w1g2, w1g4, w0g1 = run("1", "1"), run("1", "2"), run("1", "4"), run("0", "1")
    monkeypatch.delenv("SRBA_HIP_SCHUR_WAVE"); monkeypatch.delenv("SRBA_HIP_BIG_GANGS")
    for other in (w1g2, w1g4):
        for k in w1g1:
            assert np.array_equal(np.nan_to_num(np.asarray(w1g1[k], float), nan=-1.0), np.nan_to_num(np.asarray(other[k], float), nan=-1.0)), k
    for out in (w1g1, w0g1):
        assert np.all(out["status"] == 0) and _close(out["chi2_init"], ref["chi2_init"], rel=1e-9) and _close(out["chi2_final"], ref["chi2_final"], rel=1e-6)
    assert _close(w1g1["chi2_final"], w0g1["chi2_final"], rel=1e-9)
    for i in range(sub.n):
        m = min(w1g1["num_trials"][i], w0g1["num_trials"][i], 4); assert _close(w1g1["trace_chi2"][i][:m], w0g1["trace_chi2"][i][:m], rel=1e-9)


@pytest.mark.gpu
def test_many_mid_size_windows_keep_one_wavefront_each():
    """A batch with MANY windows whose Schur-reduced system is too large for one wavefront's LDS (more than 63 block rows: > 31 SE3 edges) but far from a deep window:
    they must not be serialised through the host-driven multi-workgroup path. Such capsules stay on the fused kernel with a dense block system in an HBM workspace;
    smaller ones use the dense block layout in LDS. Every replica must reproduce the oracle's chi2 of its original."""
    import ctypes as C
    ds, _ = datasets.landmarks_dataset_se3("stereo", n_kf=60, n_lm=600, seed=5, noise=0.1)
    eng = runner.landmark_engine("stereo", backend=_oracle.BACKEND); eng.run(ds); b = eng.harvest(); b.engine = eng
    nk = np.array([b[i].n_unk_edges for i in range(b.n)])
    big = np.flatnonzero(2 * nk > 63); small = np.flatnonzero((2 * nk <= 63) & (nk >= 20))[:6]
    assert len(big) >= 5 and len(small) >= 3
    pick = list(big[:5]) + list(small); copies = 8   # 40 replicas of the mid-size windows (> 4 x the lanes of the multi-workgroup path)
    ref = _oracle.run_batch(b)
    arr = (capi.Capsule * (len(pick) * copies))()
    for r in range(copies):
        for j, i in enumerate(pick): arr[r * len(pick) + j] = b.ptr[i]
    class Rep: pass
    fb = Rep(); fb.ptr = C.cast(arr, capi.PCAP); fb.n = len(pick) * copies; fb.params = b.params; fb.family = b.family
    ctx = runner.HipContext(b.params); ctx.upload(fb); gpu = ctx.lm_run()
    st = (C.c_double * 4)(); ctx.lib.srba_hip_big_path_stats(ctx.ctx, st); ctx.close()
    assert st[2] == 0                                                        # nothing went through the dense Cholesky of the multi-workgroup path
    for r in range(copies):
        for j, i in enumerate(pick):
            q = r * len(pick) + j
            assert gpu["status"][q] == 0 and gpu["num_observations"][q] == ref["num_observations"][i]
            assert abs(gpu["chi2_init"][q] - ref["chi2_init"][i]) <= 1e-9 * ref["chi2_init"][i]
            assert abs(gpu["chi2_final"][q] - ref["chi2_final"][i]) <= 1e-6 * ref["chi2_final"][i] + 1e-20, (q, i)


def _mono_dataset(gauge_fixed):
    """60-key-frame monocular map. gauge_fixed: the landmarks first seen from key-frame 0 are given with their known relative position (what the reference's monocular
    tutorial does, tutorial-srba-monocular-se3.cpp) and new landmarks enter 5 cm from the truth -> every window converges to the pixel noise and the reference's two Schur
    solvers agree to 1e-13. Without it (free global scale, 20 cm depth noise) the map is lost (RMSE ~ 100 px) and the windows are ill-conditioned: the oracle's Schur+dense
    and Schur+sparse code paths -- the same algebra in two elimination orders -- end 1e-5 apart (tests/test_conditioning.py)."""
    if gauge_fixed:
        return datasets.landmarks_dataset_se3("mono", n_kf=60, n_lm=600, seed=5, noise=0.1, init_from_gt_noise=0.05, known_first=1000)[0]
    return datasets.landmarks_dataset_se3("mono", n_kf=60, n_lm=600, seed=5, noise=0.1, init_from_gt_noise=0.2)[0]


def _replicate(b, copies):
    import ctypes as C
    n0 = b.n; arr = (capi.Capsule * (n0 * copies))()
    for r in range(copies):
        for i in range(n0): arr[r * n0 + i] = b.ptr[i]
    class Rep: pass
    fb = Rep(); fb.ptr = C.cast(arr, capi.PCAP); fb.n = n0 * copies; fb.params = b.params; fb.family = b.family; fb._keep = (arr, b)
    return fb


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["stereo", "mono"])
def test_large_batch_moves_wide_lds_images_to_hbm(kind):
    """In batches of 1 024 capsules or more, landmark windows whose LDS image would need 48 KB or more keep one wavefront but hold their dense block system in HBM
    (left-looking sweeps), so that LDS does not cap the wavefronts per CU. EVERY window of a 60-key-frame map x 18 replicas must reproduce the oracle's chi2 of its
    original at 1e-6, whichever layout it got (sparse in LDS, dense in LDS, dense in HBM)."""
    import ctypes as C
    ds = _mono_dataset(True) if kind == "mono" else datasets.landmarks_dataset_se3(kind, n_kf=60, n_lm=600, seed=5, noise=0.1)[0]
    eng = runner.landmark_engine(kind, backend=_oracle.BACKEND); eng.run(ds); b = eng.harvest(); b.engine = eng
    ref = _oracle.run_batch(b); n0 = b.n; copies = 18
    assert n0 * copies >= 1024
    fb = _replicate(b, copies)
    ctx = runner.HipContext(b.params); ctx.upload(fb); gpu = ctx.lm_run()
    st = (C.c_double * 4)(); ctx.lib.srba_hip_big_path_stats(ctx.ctx, st); ctx.close()
    assert st[2] == 0                                                        # nothing went through the multi-workgroup path
    for q in range(fb.n):
        i = q % n0
        assert gpu["status"][q] == 0 and gpu["num_observations"][q] == ref["num_observations"][i]
        assert abs(gpu["chi2_init"][q] - ref["chi2_init"][i]) <= 1e-9 * ref["chi2_init"][i], (q, i)
        assert abs(gpu["chi2_final"][q] - ref["chi2_final"][i]) <= 1e-6 * ref["chi2_final"][i] + 1e-20, (q, i, gpu["chi2_final"][q], ref["chi2_final"][i])


@pytest.mark.gpu
def test_lost_monocular_map_agrees_up_to_a_rounding_floor_decision():
    """The gauge-free monocular map (round 2's dataset of the test above) is LOST (RMSE ~ 100 px, chi2 ~ 1e7): its windows are not problems whose minimum the reference's arithmetic
    defines to 1e-6 -- on some hosts the oracle's own Schur+dense and Schur+sparse code paths end 1e-5 apart on them. What two correct implementations must still share:
    every trial up to the first accept / reject decision they take differently, that decision being a coin toss at the rounding floor (the trial moves chi2 by less than 1e-6 of its
    value in BOTH runs), and on that common prefix every accepted chi2 at 1e-6 and the same lambda schedule. Where the runs never part (most windows) the final chi2 agrees at 1e-6;
    where they do, the continuations are different but both legitimate (one run may stop on rho > max_rho or lambda > max_lambda a dozen trials before the other: round 2's
    "4e-4 window")."""
    ds = _mono_dataset(False)
    eng = runner.landmark_engine("mono", backend=_oracle.BACKEND); eng.run(ds); b = eng.harvest(); b.engine = eng
    cpu = _oracle.run_batch(b); gpu = runner.run_batch_hip(b)
    assert np.all(gpu["status"] == cpu["status"]) and np.array_equal(gpu["num_observations"], cpu["num_observations"]) and np.array_equal(gpu["num_jacobians"], cpu["num_jacobians"])
    # (chi2_init ~ 1e7 px^2 comes from points that start almost in the camera plane: their pixel coordinates, and the 1/z^2 of their Jacobians that sets
    #  lambda_0 = 1e-3 max diag H, amplify the last bits of the composed pose)
    assert _close(gpu["chi2_init"], cpu["chi2_init"], rel=1e-6) and _close(gpu["lambda_init"], cpu["lambda_init"], rel=1e-4)
    # Exact, window by window (round 4): the oracle follows the GPU's own decision sequence (tests/_oracle.py run_batch_replay), so the two runs are compared over their WHOLE length
    # instead of up to their first different decision. EVERY window of this lost map (RMSE 60 .. 140 px, chi2 ~ 1e7, points next to the camera plane), nothing counted:
    # accepted trials and final chi2 within 1e-5, every disputed decision a step that moves chi2 by less than 1e-6 of its value in both runs (measured on the MI355X box: 2.4e-6 /
    # 2.4e-6 / 2.5e-8; 23 of the 59 windows sit below 1e-9). The 1e-9 / 1e-9 / 1e-6 of the other tests is not attainable here by ANY two evaluations: the oracle following the same
    # decisions on inputs (observations, initial unknowns) moved by one unit in the last place ends 1e-9 .. 1e-6 away from itself on the accepted trials (checked below; tools/diag_replay.py prints the
    # GPU-vs-oracle distance and that sensitivity side by side, window by window: they go together).
    rep, R = _replay_exact(b, gpu, tol_trace=1e-5, tol_floor=1e-6, tol_final=1e-5)
    sens, _ = _oracle.rounding_sensitivity(b, gpu, seeds=(0, 1))
    assert sens.max() > 1e-8, sens.max()   # (the oracle against itself, same decisions, inputs moved by one ulp: this map amplifies rounding by seven orders of magnitude or more)
    assert R["complete"].sum() >= b.n - 2   # nearly every window's whole run fits the trial trace and is compared to its end


@pytest.mark.gpu
def test_schur_gradient_defect_is_reproduced_and_its_repair_matches_the_oracle(monkeypatch):
    """BASELINE cfg3 stereo room up to the key-frame where the reference's algorithm loses it (tests/test_reference_defects.py): the Schur solvers
    reduce minus_grad in place (schur.h:248-265, :294) and a rejected trial retries on that gradient (optimize_edges.h:658-690). Faithful default: the device retries on the
    reduced gradient too -- window 76 rejects every retry after its 13th trial exactly like the oracle. Extension bit SRBA_EXT_SCHUR_KEEPS_GRADIENT: the
    device restores the gradient before every solve, on the one-wavefront path and on the multi-workgroup path, and matches the oracle running the same repair on all windows."""
    from test_reference_defects import _room, _first_lost
    b0, ref0 = _room(82, 0); lost = _first_lost(ref0)
    assert lost == 76
    sub = b0.sub(0, lost); ref = _oracle.run_batch(sub); gpu = runner.run_batch_hip(sub)
    _compare_lm(sub, gpu, ref)
    # the window that breaks: its first trial is rejected, so every later solve works on a gradient that was reduced twice or more -- differences of nearly equal numbers, which
    # amplify the rounding differences between two implementations (1e-7 at trial 4, 2e-5 at trial 12 on the MI355X). What is compared is the behaviour: the same
    # accept / reject decision at every one of the 24 trials, the same lambda schedule, the same stop, accepted chi2 within 1e-4, the same lost window at the end.
    one = b0.sub(lost, 1); r = _oracle.run_batch(one); g = runner.run_batch_hip(one)
    m = int(r["num_trials"][0]); assert g["num_trials"][0] == m and m <= capi.TRACE_LEN and g["status"][0] == r["status"][0]
    assert np.array_equal(np.sign(g["trace_rho"][0][:m]), np.sign(r["trace_rho"][0][:m])) and _close(g["trace_lambda"][0][:m], r["trace_lambda"][0][:m], rel=1e-9)
    acc = r["trace_rho"][0][:m] > 0
    assert _close(g["trace_chi2"][0][:m][acc], r["trace_chi2"][0][:m][acc], rel=1e-4) and _close(g["chi2_final"], r["chi2_final"], rel=1e-4)
    assert g["obs_rmse"][0] > 5.0 and (g["trace_chi2"][0][:m][~acc][2:] > 50.0 * g["chi2_final"][0]).all()
    b1, ref1 = _room(82, 4)
    assert b1.params.extensions & capi.EXT_SCHUR_KEEPS_GRADIENT
    gpu1 = runner.run_batch_hip(b1)
    _compare_lm(b1, gpu1, ref1)
    assert gpu1["obs_rmse"].max() < 1.5
    tail = b1.sub(72, 8); reft = _oracle.run_batch(tail)
    monkeypatch.setenv("SRBA_HIP_MAX_LDS_KB", "0")
    gput = runner.run_batch_hip(tail)
    monkeypatch.delenv("SRBA_HIP_MAX_LDS_KB")
    assert _close(gput["chi2_final"], reft["chi2_final"], rel=1e-6) and gput["obs_rmse"].max() < 1.5


@pytest.mark.gpu
def test_deep_monocular_window_with_the_reference_defaults(monkeypatch):
    """The deep window (max_tree_depth = max_optimize_depth = 8) with the reference's spanning-tree refresh to the letter (no extension bit): the map holds for 25 key-frames
    (the stale twin of DESIGN section 8, item 2, ends it at the 26th); the last windows before that -- every key-frame of the map is in the window -- on the one-wavefront
    path and on the multi-workgroup path against the oracle at 1e-6."""
    ds, _ = datasets.mono_deep_window(n_kf=29, n_lm=600, seed=1)
    eng = runner.landmark_engine("mono", backend=_oracle.BACKEND, depth=8, submap=20, sigma=0.5, robust=0, harvest=1, cam=(200., 200., 400., 320.), refresh_all_read_poses=0)
    eng.run(ds); b = eng.harvest(); b.engine = eng
    assert b.params.extensions == 0
    sub = b.sub(20, 5); ref = _oracle.run_batch(sub)
    assert ref["obs_rmse"].max() < 1.0 and sub[4].n_unk_edges == 25
    gpu = runner.run_batch_hip(sub)
    _compare_lm(sub, gpu, ref)
    monkeypatch.setenv("SRBA_HIP_MAX_LDS_KB", "0")
    big = runner.run_batch_hip(sub)
    monkeypatch.delenv("SRBA_HIP_MAX_LDS_KB")
    assert np.all(big["status"] == ref["status"]) and _close(big["chi2_init"], ref["chi2_init"], rel=1e-9) and _close(big["chi2_final"], ref["chi2_final"], rel=1e-6)
    assert np.array_equal(big["num_observations"], ref["num_observations"]) and np.array_equal(big["num_jacobians"], ref["num_jacobians"])


@pytest.mark.gpu
def test_big_path_with_the_one_launch_factorisation(monkeypatch):
    """SRBA_HIP_BIG_PERSISTENT=1: the blocked Cholesky of the multi-workgroup path as one persistent launch with grid barriers (srba_big.hpp, k_chol_persistent) instead of one
    launch per panel step and trailing update. Same arithmetic in the same order: the results equal those of the default launches bit for bit, and the oracle's at 1e-6."""
    from test_oracle_numeric import _harvest
    b = _harvest("stereo", solver=capi.SOLVER_SCHUR_DENSE, n_kf=14)
    sub = b.sub(max(0, b.n - 4), min(4, b.n)); ref = _oracle.run_batch(sub)
    monkeypatch.setenv("SRBA_HIP_MAX_LDS_KB", "0")
    base = runner.run_batch_hip(sub)
    monkeypatch.setenv("SRBA_HIP_BIG_PERSISTENT", "1")
    one = runner.run_batch_hip(sub)
    monkeypatch.delenv("SRBA_HIP_BIG_PERSISTENT"); monkeypatch.delenv("SRBA_HIP_MAX_LDS_KB")
    assert np.all(one["status"] == ref["status"]) and _close(one["chi2_final"], ref["chi2_final"], rel=1e-6, abs_=1e-18)
    for k in ("chi2_final", "chi2_init", "num_trials", "trace_chi2", "trace_lambda"):
        assert np.array_equal(np.nan_to_num(np.asarray(one[k], float), nan=-1.0), np.nan_to_num(np.asarray(base[k], float), nan=-1.0)), k


@pytest.mark.gpu
@pytest.mark.parametrize("kind,solver", [("stereo", "dense"), ("mono", "dense"), ("stereo", "sparse")])
def test_big_path_gang_equals_one_window_per_stream(monkeypatch, kind, solver):
    """The large windows of a batch run as a GANG in lock-step (srba_big.hpp `Gang`, srba_hip.hip `big_gang_run`): one launch per phase for every window that is at that
    point of its LM loop, slots refilled as windows end. A window's numbers must not depend on its gang: every result field, every trace entry and the written-back state equal,
    bit for bit, those of the earlier scheme (SRBA_HIP_BIG_GANG=0: one host thread + stream per window), with more windows than slots (refills) and with a single slot;
    and the oracle's at 1e-6. Windows of different sizes (the map grows), two Schur solvers."""
    from test_oracle_numeric import _harvest
    b = _harvest(kind, solver=capi.SOLVER_SCHUR_DENSE if solver == "dense" else capi.SOLVER_SCHUR_SPARSE, n_kf=14)
    sub = b.sub(max(0, b.n - 9), min(9, b.n)); ref = _oracle.run_batch(sub)
    monkeypatch.setenv("SRBA_HIP_MAX_LDS_KB", "0")   # every window on the multi-workgroup path
    import ctypes as C
    def run(gang, lanes):
        monkeypatch.setenv("SRBA_HIP_BIG_GANG", gang); monkeypatch.setenv("SRBA_HIP_BIG_LANES", lanes)
        ctx = runner.HipContext(sub.params); ctx.upload(sub); out = ctx.lm_run()
        st = (C.c_double * 8)(); ctx.lib.srba_hip_big_path_stats2(ctx.ctx, st)
        work = sub.clone(); ctx._chk(ctx.lib.srba_hip_download_state(ctx.ctx, work.ptr, work.n), "download"); ctx.close()
        state = [work.array(i, "edge_pose", np.float64, sub[i].n_unk_edges * capi.DIMS[sub.family][3]).copy() for i in range(sub.n)]
        return out, state, list(st)
    base, base_state, st0 = run("0", "4")
    assert st0[5] == 0 and st0[4] == st0[2] and st0[2] > 0        # one launch sequence per factorisation
    for lanes in ("16", "4", "1"):
        g, g_state, st = run("1", lanes)
        assert st[5] == 1 and st[2] == st0[2]                      # the same factorisations ...
        if lanes != "1": assert st[4] < 0.5 * st[2]                # ... in far fewer launch sequences
        for k in base:
            assert np.array_equal(np.nan_to_num(np.asarray(g[k], float), nan=-1.0), np.nan_to_num(np.asarray(base[k], float), nan=-1.0)), (lanes, k)
        for i in range(sub.n): assert np.array_equal(g_state[i], base_state[i]), (lanes, i)
    monkeypatch.delenv("SRBA_HIP_BIG_GANG"); monkeypatch.delenv("SRBA_HIP_BIG_LANES"); monkeypatch.delenv("SRBA_HIP_MAX_LDS_KB")
    assert np.all(base["status"] == ref["status"]) and _close(base["chi2_final"], ref["chi2_final"], rel=1e-6, abs_=1e-18)


@pytest.mark.gpu
def test_fused_linearisation_and_k6_on_materialised_blocks_agree(se2_batch):
    """srba_hip_linearize for <SE2, RelativePoses2D> keeps the Jacobian blocks on the chip; srba_hip_hessian_from_jacobians (K6 alone) afterwards first has them written
    by the unfused kernel and must reproduce the Hessian blocks of the fused launch (different summation trees: equal to rounding)."""
    b = se2_batch
    ctx = runner.HipContext(b.params); ctx.upload(b); lib = ctx.lib
    assert lib.srba_hip_update_spantree(ctx.ctx, 0) == 0 and lib.srba_hip_eval_residuals(ctx.ctx, None) == 0 and lib.srba_hip_linearize(ctx.ctx) == 0
    fused = ctx.debug(3).copy()
    assert lib.srba_hip_hessian_from_jacobians(ctx.ctx) == 0
    k6 = ctx.debug(3)
    assert np.abs(fused).max() > 0 and np.allclose(k6, fused, rtol=1e-12, atol=1e-12 * np.abs(fused).max())
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["identity", "full"])
def test_fused_linearisation_noise_instantiations(se2_batch, mode):
    """k_assemble_se2rel has three instantiations: Lambda = identity (scaled by 1 / sigma afterwards), diagonal (the benchmark's, covered above) and full symmetric matrix.
    The other two against the oracle: Hessian blocks and gradient of every capsule."""
    b = se2_batch; P, L, O, PD = capi.DIMS[b.family]
    saved = (b.params.noise, b.params.std_noise_observations, [b.params.lambda_[k] for k in range(36)])
    try:
        if mode == "identity":
            b.params.noise = capi.NOISE_IDENTITY; b.params.std_noise_observations = 0.05
        else:   # a symmetric positive definite information matrix with off-diagonal entries
            A = np.array([[2.0e5, 3.0e4, -1.0e4], [3.0e4, 1.5e5, 2.0e4], [-1.0e4, 2.0e4, 8.0e4]])
            for i in range(3):
                for j in range(3): b.params.lambda_[3 * i + j] = A[i, j]
        ctx = runner.HipContext(b.params); ctx.upload(b); lib = ctx.lib
        assert lib.srba_hip_update_spantree(ctx.ctx, 0) == 0 and lib.srba_hip_eval_residuals(ctx.ctx, None) == 0 and lib.srba_hip_linearize(ctx.ctx) == 0
        HAp, grad = ctx.debug(3), ctx.debug(6)
        oh = og = 0
        for i in range(b.n):
            c = b[i]; ref = _oracle.stage(b, i, do_solve=False, lam=0.0); n = P * c.n_unk_edges
            assert np.allclose(HAp[oh:oh + c.n_hap * P * P], ref["HAp"], rtol=1e-9, atol=1e-9 * np.abs(ref["HAp"]).max()), (mode, i)
            assert ref["scalars"][0] < 1e-24 or np.allclose(grad[og:og + n], ref["grad"], rtol=1e-7, atol=1e-9 * np.abs(ref["grad"]).max()), (mode, i)
            oh += c.n_hap * P * P; og += n
        ctx.close()
    finally:
        b.params.noise, b.params.std_noise_observations = saved[0], saved[1]
        for k in range(36): b.params.lambda_[k] = saved[2][k]


@pytest.mark.gpu
def test_flat_valley_windows_part_at_a_rounding_floor_decision():
    """Found by the 40-seed soak (profiles/r03_soak_parity_40seeds.log): in 3 of 10 000 well-conditioned windows chi2_final differs from the oracle's by 2e-6 .. 9e-6.
    Range-bearing 2D, seed 32, window 7: the two runs take the same decisions for seven trials with accepted chi2 equal to 1e-13; at the eighth the step changes chi2 by
    3e-15 (oracle) / 3e-14 (device) of its value -- the sign of rho is rounding -- and the run that rejects it stops while the other one goes on creeping down a flat valley
    for twenty more trials (9e-6 in total). What 1e-6 parity means here is the prefix, and that the split is a rounding-floor decision in BOTH runs."""
    seed = 32
    ds, _ = datasets.landmarks_dataset_se2("rb2d", n_kf=30, n_lm=800, seed=seed, noise=1e-3)
    eng = runner.landmark_engine("rb2d", backend=_oracle.BACKEND, solver=capi.SOLVER_SCHUR_DENSE, depth=2 + seed % 3); eng.run(ds); b = eng.harvest(); b.engine = eng
    sub = b.sub(max(0, b.n - 40), min(40, b.n)); ref = _oracle.run_batch(sub); gpu = runner.run_batch_hip(sub)
    rel = np.abs(gpu["chi2_final"] - ref["chi2_final"]) / ref["chi2_final"]
    assert rel.max() > 1e-6   # (the dataset still shows what it was chosen for: the two runs' OWN final chi2 differ by more than the parity tolerance in at least one window ...)
    # ... and the oracle following the GPU's decisions lands on the GPU's chi2: every accepted trial of every window, every disputed decision and the end at 1e-9 (measured: 3e-13, 1e-13, 1e-13)
    _replay_exact(sub, gpu, tol_trace=1e-9, tol_floor=1e-9, tol_final=1e-9)
    for i in range(sub.n):
        m = int(min(gpu["num_trials"][i], ref["num_trials"][i], capi.TRACE_LEN)); g, c = gpu["trace_chi2"][i][:m], ref["trace_chi2"][i][:m]
        same = (np.sign(gpu["trace_rho"][i][:m]) == np.sign(ref["trace_rho"][i][:m])) & (np.isnan(g) == np.isnan(c)); k = m if same.all() else int(np.argmin(same))
        ok = ref["trace_rho"][i][:k] > 0
        assert _close(g[:k][ok], c[:k][ok], rel=1e-9), i                         # the common prefix is the same descent
        if k < m:                                                                  # ... and it ends where a step no longer changes chi2 beyond rounding, in both runs
            e_prev = c[np.flatnonzero(ok)[-1]] if ok.any() else ref["chi2_init"][i]
            assert all(np.isnan(e) or abs(e - e_prev) <= 1e-9 * e_prev for e in (g[k], c[k])), (i, k)
