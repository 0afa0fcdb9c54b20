"""Per-sample parity bookkeeping shared by the GPU tests: how many samples are bit-identical to the oracle's, and how the rest differ.
`record` appends one line per call to gpurun_out/parity_report.txt (scripts and DESIGN.md quote it); `check` is the assertion the tests
use: at least `min_exact` of the samples bit-identical and every other sample within the given tolerance (a last-place difference in an
elementary function that did not flip any decision), except for at most `max_path` samples whose paths took different branches."""
import inspect
import os

import numpy as np

_REPORT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_report.txt")


def classify(got, want):
    got, want = np.asarray(got), np.asarray(want)
    if got.ndim == 1:
        got, want = got[:, None], want[:, None]
    exact = (got == want).all(1) | (np.isnan(got) & np.isnan(want)).all(1)
    rel = np.abs(got - want).max(1) / np.maximum(np.abs(want).max(1), 1e-6)
    return exact, rel


def record(label, got, want):
    exact, rel = classify(got, want)
    bad = ~exact
    caller = inspect.stack()[1]
    line = "%-100s exact %.6f  n %7d  differ %6d  ulp-level(<1e-5) %6d  path-level %5d  worst rel %.3g" % (
        (os.environ.get("PYTEST_CURRENT_TEST", "%s:%s" % (os.path.basename(caller.filename), caller.function)).split(" (")[0].replace("tests/", "") + " " + label).strip(), exact.mean(), exact.size, bad.sum(),
        (bad & (rel < 1e-5)).sum(), (bad & (rel >= 1e-5)).sum(), rel.max() if rel.size else 0.0)
    try:
        os.makedirs(os.path.dirname(_REPORT), exist_ok=True)
        with open(_REPORT, "a") as f:
            f.write(line + "\n")
    except OSError:
        pass
    return exact, rel


def check(label, got, want, min_exact=1.0, max_path=0, ulp_rel=1e-5):
    exact, rel = record(label, got, want)
    bad = ~exact
    path_level = int((bad & (rel >= ulp_rel)).sum())
    assert exact.mean() >= min_exact, (label, "exact fraction", float(exact.mean()), "required", min_exact, "first differing", np.nonzero(bad)[0][:8].tolist())
    assert path_level <= max_path, (label, "samples whose paths diverged", path_level, "allowed", max_path, np.nonzero(bad & (rel >= ulp_rel))[0][:8].tolist())
