import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes more than ~20 s on CPU")


def pytest_sessionstart(session):
    """A fresh checkout has no built artefacts (they are git-ignored): build the HIP library (hipcc cross-compiles gfx950
    without a GPU) and the CPU oracle once, exactly as __graft_entry__.build() does."""
    lib = os.path.join(ROOT, "go-snark-study_amd", "libgosnark_hip.so")
    ora = os.path.join(ROOT, "oracle", "libgs_oracle.so")
    if not (os.path.exists(lib) and os.path.exists(ora)):
        import __graft_entry__
        __graft_entry__.build()
