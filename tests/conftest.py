import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by `pytest -m gpu` on the GPU box)")
    config.addinivalue_line("markers", "slow: takes more than ~20 s on CPU")


import pytest  # noqa: E402


@pytest.fixture(autouse=True)
def _pin_candidates_per_proposal(monkeypatch):
    """The parity tests compare with oracle calls that say how many candidates a match proposal carries (one, unless the
    call says otherwise); the library's own choice (`alternatives = 0`: two on deep-coverage pools) would make the
    expected streams depend on the pool.  Here an opts object that leaves the field at 0 runs with ONE candidate;
    tests/test_gpu_alternatives.py covers two, and the automatic choice through `alternatives = -1` (negative = the
    library's choice, like 0).  The same for `phases` (chain groups): 0 runs ONE group here."""
    try:
        import spring_amd.reorder as R
    except Exception:  # (CPU-only collections that never touch the package)
        yield
        return
    orig = R.ReorderOpts.to_c

    def to_c(self):
        o = orig(self)
        if o.alternatives == 0:
            o.alternatives = 1
        if o.phases == 0:  # likewise the chain schedule: one group unless the test says otherwise (tests/test_gpu_phases.py;
            o.phases = 1   # negative = the library's choice: two groups from 16 384 chains on)
        return o
    monkeypatch.setattr(R.ReorderOpts, "to_c", to_c)
    yield
