"""Observed parity errors of a test run (see tests/conftest.py: printed in the terminal summary, kept in
gpurun_out/parity_report.jsonl)."""
PARITY = {}


def report(case: str, err, tol=None) -> float:
    """Records the largest error seen for `case` and returns err, so a test writes `assert report(case, err, tol) < tol`."""
    err = float(err)
    cur = PARITY.get(case)
    if cur is None or err > cur[0] or err != err:
        PARITY[case] = (err, None if tol is None else float(tol))
    return err
