"""CPU oracle bindings -- TEST INFRASTRUCTURE (oracle/srba_oracle.cpp -> oracle/_build/libsrba_oracle.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; nothing under srba_amd/ does.
The oracle is the checker, never the thing measured or shipped: the product path (libsrba_hip.so) has no CPU fallback.

BACKEND          : plug object for runner.Engine(backend=...) -- runs the C++ front-end with the oracle as numeric back-end
run_batch        : the oracle's LM loop on a deep copy of a capsule batch (optionally multi-threaded: bench cpu_baseline)
stage            : initial linearisation (S5..S14) of one capsule (+ optionally one solve): arrays for per-kernel parity tests
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from srba_amd import capi, runner  # noqa: E402

_lib = None


def lib(opt="O2"):
    """libsrba_oracle.so (g++ -O2, the reference's default flags CMakeLists.txt:48-50); opt="O3" loads the -O3 build (bench only)."""
    global _lib
    if opt != "O2":
        path = os.path.join(ROOT, "oracle", "_build", "libsrba_oracle_%s.so" % opt)
        if not os.path.exists(path):
            raise RuntimeError("oracle (%s) not built: run `python __graft_entry__.py`" % opt)
        l = C.CDLL(path)
        _proto(l)
        return l
    if _lib is None:
        path = os.path.join(ROOT, "oracle", "_build", "libsrba_oracle.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle not built (%s): run `python __graft_entry__.py`" % path)
        _lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
        _proto(_lib)
    return _lib


def _proto(l):
    l.srba_oracle_lm_run.argtypes = [C.POINTER(capi.HipParams), capi.PCAP, capi.c_i32, C.POINTER(capi.LmResult), capi.c_i32]
    l.srba_oracle_run_one.argtypes = [C.POINTER(capi.HipParams), capi.PCAP, C.POINTER(capi.LmResult)]
    l.srba_oracle_take_symbolic_seconds.restype = capi.c_f64
    l.srba_oracle_lm_run_replay.argtypes = [C.POINTER(capi.HipParams), capi.PCAP, capi.c_i32, capi.PI32, capi.PI32, capi.c_i32, capi.PF64, capi.PF64, capi.PF64, capi.PI32, capi.PI32,
            C.POINTER(capi.LmResult), capi.c_i32]
    l.srba_oracle_schur_from_jacobians.argtypes = [C.POINTER(capi.HipParams), capi.PCAP, capi.PF64, capi.PF64, capi.PF64, capi.c_f64] + [capi.PF64] * 4
    l.srba_oracle_stage.argtypes = [C.POINTER(capi.HipParams), capi.PCAP, capi.c_i32, capi.c_f64] + [capi.PF64] * 10


class _Backend(object):
    """runner.Engine(backend=BACKEND): plugs srba_oracle_run_one / srba_oracle_eval_overall into the engine through srba_engine_set_backend_fn."""
    name = "oracle"

    def plug(self, eng):
        ora = lib()
        eng.lib.srba_engine_set_backend_fn(eng.h, C.cast(ora.srba_oracle_run_one, C.c_void_p), b"cpu-oracle")
        eng.lib.srba_engine_set_overall_fn(eng.h, C.cast(ora.srba_oracle_eval_overall, C.c_void_p))


BACKEND = _Backend()


def run_batch(batch, threads=1, keep_state=False, opt="O2"):
    ora = lib(opt)
    work = batch.clone()
    res = (capi.LmResult * work.n)()
    rc = ora.srba_oracle_lm_run(C.byref(batch.params), work.ptr, work.n, res, threads)
    if rc != 0:
        raise RuntimeError("oracle failed")
    out = runner.results_to_dict(res, work.n)
    if keep_state:
        out["state"] = work
    return out


def decisions_of(res):
    """The trial sequence of a run, from its trace (first TRACE_LEN trials): 0 = solve failed (not positive definite), 1 = rejected, 2 = accepted; and how many of them there are."""
    n = len(res["num_trials"]); k = np.minimum(res["num_trials"], capi.TRACE_LEN).astype(np.int32)
    rho = res["trace_rho"]; dec = np.where(np.isnan(rho), 0, np.where(rho > 0, 2, 1)).astype(np.int32)
    dec[np.arange(capi.TRACE_LEN)[None, :] >= k[:, None]] = -1
    return np.ascontiguousarray(dec), k


def run_batch_replay(batch, other, threads=1):
    """The oracle's LM loop on a deep copy of the batch, taking the accept / reject / not-PD sequence of ANOTHER run (`other`: a results dict, normally the GPU's) instead of its own
    decisions (oracle/srba_oracle.cpp, Problem::run). Returns the usual results dict of the replayed run plus, per trial: own_rho, own_chi2 (chi2 of the trial point), own_E (chi2 the
    trial started from), flags (1 own factorisation PD, 2 own rho sign differs from the replayed decision, 4 not-PD forced on a PD system), and per capsule diverged_at, replayed (trials
    replayed), complete (the other run's whole sequence fitted the trace and was followed to its end)."""
    ora = lib(); work = batch.clone(); n = work.n; T = capi.TRACE_LEN
    dec, k = decisions_of(other)
    own_rho = np.full((n, T), np.nan); own_chi2 = np.full((n, T), np.nan); own_E = np.full((n, T), np.nan); flags = np.zeros((n, T), np.int32); div = np.full(n, -1, np.int32)
    res = (capi.LmResult * n)()
    rc = ora.srba_oracle_lm_run_replay(C.byref(batch.params), work.ptr, n, dec.ctypes.data_as(capi.PI32), k.ctypes.data_as(capi.PI32), T, own_rho.ctypes.data_as(capi.PF64),
            own_chi2.ctypes.data_as(capi.PF64),
                                       own_E.ctypes.data_as(capi.PF64), flags.ctypes.data_as(capi.PI32), div.ctypes.data_as(capi.PI32), res, threads)
    if rc != 0:
        raise RuntimeError("oracle replay failed")
    out = runner.results_to_dict(res, n)
    out.update(own_rho=own_rho, own_chi2=own_chi2, own_E=own_E, flags=flags, diverged_at=div, decisions=dec, replayed=k, complete=(div < 0) & (other["num_trials"] <= T), state=work)
    return out


def replay_report(gpu, rep, tol_trace=1e-9, tol_floor=1e-9):
    """What the replay of a GPU run by the oracle proves, window by window:
      trace_ok  -- on every ACCEPTED trial the chi2 of the two runs agree to tol_trace (relative)
      floor_ok  -- wherever the oracle's own rho sign disagrees with the GPU's decision, the step moves chi2 by less than tol_floor of its value in BOTH runs (a rounding-floor decision)
      final_rel -- relative distance of the final chi2 (replayed oracle vs GPU); only meaningful where `complete`
    Returns a dict of per-capsule arrays."""
    T = capi.TRACE_LEN; n = len(gpu["num_trials"])
    dec = rep["decisions"]; acc = dec == 2; ev = dec >= 1
    g_chi2 = gpu["trace_chi2"]; o_chi2 = rep["own_chi2"]
    with np.errstate(invalid="ignore", divide="ignore"):
        rel = np.abs(g_chi2 - o_chi2) / np.maximum(np.abs(o_chi2), 1e-300)
        rel = np.where(np.abs(g_chi2 - o_chi2) < 1e-20, 0.0, rel)   # (absolute floor: noise-free windows end at chi2 ~ 1e-25)
    trace_err = np.where(acc, rel, 0.0); trace_err[np.isnan(trace_err)] = np.inf
    worst_trace = trace_err.max(axis=1)
    # the chi2 the GPU trial started from = chi2 of its last accepted trial before (or the initial one)
    g_E = np.empty((n, T)); cur = gpu["chi2_init"].astype(float).copy()
    for t in range(T):
        g_E[:, t] = cur; a = acc[:, t]; cur = np.where(a, g_chi2[:, t], cur)
    dis = (rep["flags"] & 2) != 0
    with np.errstate(invalid="ignore", divide="ignore"):
        move_o = np.abs(rep["own_E"] - o_chi2) / np.maximum(np.abs(rep["own_E"]), 1e-300); move_g = np.abs(g_E - g_chi2) / np.maximum(np.abs(g_E), 1e-300)
    floor_move = np.where(dis, np.maximum(move_o, move_g), 0.0); floor_move[np.isnan(floor_move)] = np.inf
    worst_floor = floor_move.max(axis=1)
    fin = np.abs(rep["chi2_final"] - gpu["chi2_final"]); final_rel = np.where(fin < 1e-20, 0.0, fin / np.maximum(np.abs(rep["chi2_final"]), 1e-300))
    return dict(worst_trace=worst_trace, worst_floor=worst_floor, floor_move=floor_move, final_rel=final_rel, complete=rep["complete"], diverged_at=rep["diverged_at"], n_disagree=dis.sum(axis=1),
            forced_notpd=((rep["flags"] & 4) != 0).sum(axis=1),
                trace_ok=worst_trace <= tol_trace, floor_ok=worst_floor <= tol_floor)


def perturbed(batch, eps=2.220446049250313e-16, seed=0):
    """A deep copy of the batch whose observations, initial unknowns and known landmarks are moved by +-eps relative (one unit in the last place by default): the input of a
    rounding-sensitivity measurement."""
    P, L, O, PD = capi.DIMS[batch.family]; w = batch.clone(); rng = np.random.RandomState(seed)
    for i in range(w.n):
        c = w.ptr[i]
        for ptr, cnt in ((c.obs_z, c.n_obs * O), (c.edge_pose, c.n_edges * PD), (c.ulm_pos, c.n_unk_lms * L), (c.klm_pos, c.n_known_lms * L)):   # observations, initial unknowns, known landmarks
            if cnt:
                z = np.ctypeslib.as_array(ptr, shape=(cnt,)); z *= 1.0 + eps * rng.choice([-1.0, 1.0], size=z.shape)
    return w


def rounding_sensitivity(batch, other, seeds=(0, 1, 2, 3, 4), threads=8):
    """How far the chi2 of every evaluated trial moves (relative) when the observations are perturbed by one unit in the last place and the oracle follows the SAME decision sequence
    (`other`) -- the distance at which two correct evaluations of that trial can be expected to sit. Returns (per window: worst ACCEPTED trial, per trial [n, TRACE_LEN]).
    Well-conditioned windows: 1e-13; accepted trials of a lost map: 1e-9 .. 1e-7 (not proportional to the perturbation: these windows hold points next to the camera plane);
    rejected overshoots of such a map can be chaotic (the trial point itself moves)."""
    base = run_batch_replay(batch, other, threads=threads); ev = base["decisions"] >= 1; acc = base["decisions"] == 2; per_trial = np.zeros(base["own_chi2"].shape)
    for sd in seeds:
        rp = run_batch_replay(perturbed(batch, seed=sd), other, threads=threads)
        with np.errstate(invalid="ignore", divide="ignore"):
            d = np.where(ev, np.abs(base["own_chi2"] - rp["own_chi2"]) / np.abs(base["own_chi2"]), 0.0)
        d[np.isnan(d)] = 0.0; per_trial = np.maximum(per_trial, d)
    return np.where(acc, per_trial, 0.0).max(axis=1), per_trial


def stage(batch, i, do_solve=False, lam=0.0):
    ora = lib()
    work = batch.clone(i, 1)
    c = work.ptr[0]; P, L, O, PD = capi.DIMS[batch.family]
    n = P * c.n_unk_edges + L * c.n_unk_lms
    arr = dict(resid=np.zeros(c.n_obs * O), Jp=np.zeros(c.n_bp * O * P), Jf=np.zeros(c.n_bf * O * L), HAp=np.zeros(c.n_hap * P * P), Hf=np.zeros(c.n_hf * L * L),
               HApf=np.zeros(c.n_hapf * P * L), grad=np.zeros(n), delta=np.zeros(n), poses=np.zeros(2 * c.n_pairs * PD), scalars=np.zeros(4))
    p = lambda a: a.ctypes.data_as(capi.PF64)
    rc = ora.srba_oracle_stage(C.byref(batch.params), work.ptr, 1 if do_solve else 0, lam, p(arr["resid"]), p(arr["Jp"]), p(arr["Jf"]), p(arr["HAp"]), p(arr["Hf"]), p(arr["HApf"]),
                               p(arr["grad"]), p(arr["delta"]), p(arr["poses"]), p(arr["scalars"]))
    if rc != 0:
        raise RuntimeError("oracle stage failed")
    return arr


def schur_from_jacobians(batch, i, Jp, Jf, grad, lam):
    """Hessians over the capsule's plan from GIVEN Jacobian blocks + SchurComplement::numeric_build_reduced_system(lam): reduced HAp blocks, Hf, HApf, reduced gradient."""
    ora = lib(); c = batch[i]; P, L, O, PD = capi.DIMS[batch.family]
    out = dict(HAp=np.zeros(c.n_hap * P * P), Hf=np.zeros(c.n_hf * L * L), HApf=np.zeros(c.n_hapf * P * L), grad=np.zeros(P * c.n_unk_edges + L * c.n_unk_lms))
    p = lambda a: np.ascontiguousarray(a, np.float64).ctypes.data_as(capi.PF64)
    Jp = np.ascontiguousarray(Jp, np.float64); Jf = np.ascontiguousarray(Jf, np.float64); grad = np.ascontiguousarray(grad, np.float64)
    sub = C.cast(C.addressof(batch.ptr.contents) + i * C.sizeof(capi.Capsule), capi.PCAP)
    rc = ora.srba_oracle_schur_from_jacobians(C.byref(batch.params), sub, Jp.ctypes.data_as(capi.PF64), Jf.ctypes.data_as(capi.PF64), grad.ctypes.data_as(capi.PF64), lam,
                                              out["HAp"].ctypes.data_as(capi.PF64), out["Hf"].ctypes.data_as(capi.PF64), out["HApf"].ctypes.data_as(capi.PF64), out["grad"].ctypes.data_as(capi.PF64))
    if rc != 0:
        raise RuntimeError("oracle schur_from_jacobians failed")
    return out
