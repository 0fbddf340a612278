"""pytest configuration: the `gpu` marker and shared golden-vector fixtures.

CPU tier  (`-m "not gpu"`): oracle vs golden vectors, host logic, C-ABI symbol checks,
                            world_size-2 gloo tests.  No HIP compute is called.
GPU tier  (`-m gpu`):       parity tests proper -- HIP kernels, called through the C ABI,
                            compared with the oracle and the committed golden vectors.
"""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


class Golden:
    """Vectors captured from the real reference by tests/golden/generate_golden.py."""

    def __init__(self):
        self.arrays = np.load(os.path.join(GOLDEN_DIR, "reference_vectors.npz"))
        with open(os.path.join(GOLDEN_DIR, "reference_vectors.json")) as fh:
            self.manifest = json.load(fh)
        self.cases = {c["name"]: c for c in self.manifest["cases"]}

    def get(self, name, field):
        return self.arrays["%s/%s" % (name, field)]

    def inputs(self, name):
        return self.get(name, "scores"), self.get(name, "relevance"), self.get(name, "n")

    def by_op(self, op):
        return [c for c in self.manifest["cases"] if c["op"] == op]


_GOLDEN = None


def load_golden():
    global _GOLDEN
    if _GOLDEN is None:
        _GOLDEN = Golden()
    return _GOLDEN


@pytest.fixture(scope="session")
def golden():
    return load_golden()


def synth(B, L, seed, F=None):
    """Deterministic synthetic batch, SURVEY.md section 8(d) recipe (CPU generator)."""
    import torch
    g = torch.Generator().manual_seed(seed)
    scores = torch.randn(B, L, generator=g)
    relevance = torch.randint(0, 5, (B, L), generator=g)
    n = torch.randint(1, L + 1, (B,), generator=g)
    out = [scores, relevance, n]
    if F is not None:
        X = torch.randn(B, L, F, generator=g)
        bound = 1.0 / (F ** 0.5)
        W = (torch.rand(F, generator=g) * 2 - 1) * bound
        b = (torch.rand(1, generator=g) * 2 - 1) * bound
        out += [X, W, b]
    return out


@pytest.fixture(scope="session", autouse=True)
def _native_libraries_present():
    """A fresh checkout has no .so files (they are git-ignored): build what is missing once per
    session (hipcc cross-compiles gfx950 without a GPU; the parser needs only g++).  Existing
    libraries are left alone -- on the GPU box the prebuilt ones travel with the tree."""
    from pytorchltr_amd import build
    if not os.path.exists(build.LIB_PATH):
        build.build_extension()
    if not os.path.exists(build.IO_LIB_PATH):
        build.build_io()


@pytest.fixture(autouse=True)
def _index_tie_breaking():
    """The parity tests compare with the oracle's deterministic index tie rule; the reference-like
    random tie-break (the package default) is exercised explicitly by tests/test_ties.py."""
    from pytorchltr_amd import _ties
    prev = _ties.set_tie_breaking("index")
    yield
    _ties.set_tie_breaking(prev)
