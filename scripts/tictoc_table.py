#!/usr/bin/env python3
"""The reference's timer table (mg_tictoc.f90 format) of a 5-iteration solve_p at 512x512x64, after an UNTIMED warm-up iteration (the
first launch of every kernel loads its code object: tens of milliseconds that a timer around it would book on that level -- what
profiles/r03_tictoc_512x512x64_FC_5it.txt showed on level 2), next to the same five iterations without timers.
python3 scripts/tictoc_table.py <out.txt> [method]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import mgroms_amd as mg  # noqa: E402
from mgroms_amd import nhydro  # noqa: E402
from mgroms_amd.testcases import seamount_geometry, resting_column_state  # noqa: E402

out = sys.argv[1]
method = sys.argv[2] if len(sys.argv) > 2 else "FC"
torch.cuda.set_device(0)
nhydro.set_verbose(0)
nhydro.set_option("tictoc", 0)
mg.nhydro_init(512, 512, 64, 1, 1, 0, nhydro.default_params(relax_method=method))
mg.nhydro_matrices(*seamount_geometry(512, 512), None, 4e3, 0.0, 0.0)
nhydro.compute_rhs(*resting_column_state(512, 512, 64))
mg.solve_p(1e-12, 2)                       # warm-up, no timers
torch.cuda.synchronize(); t0 = time.perf_counter()
mg.solve_p(1e-12, 5)
torch.cuda.synchronize(); plain = time.perf_counter() - t0
nhydro.set_option("tictoc", 1)
torch.cuda.synchronize(); t0 = time.perf_counter()
mg.solve_p(1e-12, 5)
torch.cuda.synchronize(); timed = time.perf_counter() - t0
nhydro.print_tictoc(out)
nhydro.set_option("tictoc", 0)
mg.nhydro_clean()
with open(out, "a") as f:
    f.write(f"# five solve_p iterations ({method}): {plain * 1e3:.2f} ms without timers, {timed * 1e3:.2f} ms with them (host wall clock)\n")
print(open(out).read())
