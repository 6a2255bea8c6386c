import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
        return cache[name]

    return load


# Parity report: tests append (case, quantity, error, bound) rows; the rows are printed in pytest's terminal summary
# (also under -q), so the log of a run shows what bound every case passed, not only that it passed.
_PARITY_ROWS = []


@pytest.fixture(scope="session")
def parity_report():
    def add(case, quantity, err, bound, note=""):
        _PARITY_ROWS.append((str(case), str(quantity), float(err), float(bound), str(note)))
    return add


def pytest_terminal_summary(terminalreporter):
    if not _PARITY_ROWS:
        return
    tr = terminalreporter
    tr.section("parity report: error vs bound (ratio = error / bound, must be <= 1)")
    for case, q, err, bound, note in _PARITY_ROWS:
        ratio = err / bound if bound > 0 else float("inf")
        tr.write_line(f"{case:44s} {q:22s} err {err:9.3e}  bound {bound:9.3e}  ratio {ratio:5.2f}  {note}")
    out = os.environ.get("NF_PARITY_REPORT")
    if out:
        with open(out, "w") as f:
            for row in _PARITY_ROWS:
                f.write("\t".join(str(v) for v in row) + "\n")
