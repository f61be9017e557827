#!/usr/bin/env python3
"""Where bench.py's CALIBRATION constants come from: the oracle (oracle/mgoracle.c, the CPU port) timed in the development container on
the workload the survey timed the real reference on (BASELINE.md sections 2 and 4: seamount 512x512x64, 4x2 ranks on 8 cores,
flang -O2 + MPICH), next to the reference's recorded times.  Prints reference time / port time per quantity.
  python scripts/cpu_calibration.py [FC|RB]          (8 cores, about a minute)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["OMP_NUM_THREADS"] = "8"
from oracle.mgoracle import make_seamount  # noqa: E402

REFERENCE_S = {"FC": {"solve_iteration": 0.66, "level1_sweep": 0.075, "level1_residual": 0.036},
               "RB": {"solve_iteration": 0.68, "level1_sweep": 0.067, "level1_residual": 0.044}}
method = sys.argv[1] if len(sys.argv) > 1 else "FC"
o = make_seamount(128, 256, 64, 4, 2, relax_method=method)
o.compute_rhs()
o.solve_p(1e-30, 1)  # first touch


def timed(f, reps):
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    return (time.perf_counter() - t0) / reps


port = {"solve_iteration": timed(lambda: o.solve_p(1e-30, 1), 2), "level1_sweep": timed(lambda: o.relax(1, 1), 5), "level1_residual": timed(lambda: o.residual(1), 5)}
for k, v in port.items():
    print(f"{method} {k}: reference {REFERENCE_S[method][k]:.3f} s, port {v:.3f} s, reference/port = {REFERENCE_S[method][k] / v:.3f}")
