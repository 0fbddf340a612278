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


def assert_rank_dependent_losses(kind, got, X, W, bias, y, n, want_l, want_s, rtol, atol=1e-5, sigma=1.0):
    """Every row of a fused LambdaNDCG step at the STATED tolerance (1e-5 up to 128 documents, 5e-4 on
    long lists), in stages (VERDICT r4 item 4 -- rounds 3-4 accepted 3 % of the rows at 5e-3 on the ASSERTION that
    those were rank flips of nearly tied fp32 scores; this tests it):
      0. against the fp64 oracle (scores and loss in fp64) -- all rows but a few pass here;
      i. the library's own fp32 scores of the batch (ltr_linear_scores_f32) agree with the oracle's to 1e-5, and
     ii. the oracle's loss ON THOSE fp32 scores explains rows that stage 0 does not: the ranks inside the loss are
         ranks of fp32 scores (loss/pairwise_lambda.py:68-70 sorts what nn.Linear returned, also fp32);
    iii. a row still off must contain documents whose oracle scores are closer than the 1e-5 of stage i, and
         exchanging the scores of one or two such adjacent pairs (a perturbation below that tolerance which flips
         their ranks -- the fused kernel adds the dot product in another order than the scorer kernel) must
         reproduce the kernel's loss at the same rtol.  Anything else fails."""
    import numpy as np
    import torch
    from oracle import ltr_oracle as O
    got = np.asarray(got, dtype=np.float64)
    ok = np.isclose(got, want_l, rtol=rtol, atol=atol)
    if ok.all():
        return 0
    from pytorchltr_amd.fused import LinearScorer
    dev = torch.device("cuda:0")
    sc = LinearScorer(X.shape[2]).to(dev)
    with torch.no_grad():
        sc.weight.copy_(W.reshape(1, -1).to(dev))
        sc.bias.copy_(bias.reshape(1).to(dev))
        s32 = sc(X.to(dev), n.to(dev)).squeeze(-1).cpu().numpy()
    nn_ = n.numpy()
    real = np.arange(X.shape[1])[None, :] < nn_[:, None]
    assert np.allclose(np.where(real, s32, 0.0), np.where(real, want_s, 0.0), rtol=1e-5, atol=1e-5), "stage i: fp32 scores"
    l32, _ = O.pairwise_loss(kind, s32.astype(np.float64), y.numpy(), nn_, sigma=sigma, need_grad=False)
    ok |= np.isclose(got, l32, rtol=rtol, atol=atol)
    left = np.nonzero(~ok)[0]
    for r in left:
        nr = int(nn_[r])
        s = np.array(want_s[r, :nr], dtype=np.float64)
        order = np.argsort(-s, kind="stable")
        gaps = s[order[:-1]] - s[order[1:]]
        delta = 2e-5 * max(1.0, float(np.max(np.abs(s))))
        cand = [int(k) for k in np.argsort(gaps)[:8] if gaps[k] <= delta]
        assert cand, "row %d: loss %r vs %r / %r and no nearly tied scores to explain it" % (r, got[r], want_l[r], l32[r])
        subsets = [(a,) for a in cand] + [(a, c) for i, a in enumerate(cand) for c in cand[i + 1:] if abs(a - c) > 1]
        explained = False
        for sub in subsets:
            sp = np.zeros((1, want_s.shape[1]))
            sp[0, :nr] = s
            for k in sub:
                i, j = order[k], order[k + 1]
                sp[0, i], sp[0, j] = s[j], s[i]
            lp, _ = O.pairwise_loss(kind, sp, y.numpy()[r:r + 1], nn_[r:r + 1], sigma=sigma, need_grad=False)
            if np.isclose(got[r], lp[0], rtol=rtol, atol=atol):
                explained = True
                break
        assert explained, "row %d: loss %r vs oracle %r (fp32 scores: %r): not a rank flip of nearly tied scores" % (
            r, got[r], want_l[r], l32[r])
    # (and they stay rare: a kernel that mis-ranks wholesale must not pass as "ties")
    assert len(left) <= max(2, got.size // 50), "%d of %d rows needed a tie explanation" % (len(left), got.size)
    return int((~np.isclose(got, want_l, rtol=rtol, atol=atol)).sum())
