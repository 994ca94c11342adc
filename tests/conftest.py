import os
import sys

import pytest

# torch bundles its own HIP runtime: it has to be loaded BEFORE libkjarni_ffi.so (which then binds to the same
# runtime, as in bench.py); the other way round torch finds no GPU in a process that already initialised HIP.
try:
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# ---- observed parity errors: tests call tests.parity_report.report(case, err); the terminal summary prints the largest
# ---- error per case (also with -q) and gpurun_out/parity_report.jsonl keeps them, so a green run says HOW green.
def pytest_terminal_summary(terminalreporter):
    from tests.parity_report import PARITY
    if not PARITY:
        return
    import json
    tr = terminalreporter
    tr.section("observed max |GPU - oracle / fixture| per case")
    for case in sorted(PARITY):
        err, tol = PARITY[case]
        tr.write_line(f"{case}: {err:.3e}" + (f"  (bar {tol:g})" if tol is not None else ""))
    try:
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_report.jsonl"), "w") as f:
            for case in sorted(PARITY):
                f.write(json.dumps({"case": case, "max_abs_err": PARITY[case][0], "bar": PARITY[case][1]}) + "\n")
    except OSError:
        pass
