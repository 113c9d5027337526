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
