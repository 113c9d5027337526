"""Window 76 of the cfg3 room with the reference's defaults (the Schur-gradient defect bites here): GPU and oracle traces side by side."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
from srba_amd import capi, runner
import _oracle
from test_reference_defects import _room
b0, ref0 = _room(82, 0)
sub = b0.sub(76, 1); ref = _oracle.run_batch(sub); gpu = runner.run_batch_hip(sub)
np.set_printoptions(linewidth=250)
for name, r in (("cpu", ref), ("gpu", gpu)):
    m = int(min(r["num_trials"][0], capi.TRACE_LEN))
    print(name, "trials", r["num_trials"][0], "chi2_final %.9e" % r["chi2_final"][0], "rmse", r["obs_rmse"][0], "stop", r.get("stop_reason", [None])[0])
    print(" chi2  ", np.array2string(r["trace_chi2"][0][:m], precision=6))
    print(" rho   ", np.array2string(r["trace_rho"][0][:m], precision=3))
    print(" lambda", np.array2string(r["trace_lambda"][0][:m], precision=3))
