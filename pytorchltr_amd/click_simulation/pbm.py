"""Position-based-model click simulators on MI355X (reference: click_simulation/pbm.py).

One HIP launch (``ltr_pbm_clicks``) replaces the reference's chain of
repeat_interleave / gather / bernoulli / argsort / gather ops (:39-63): for every rank position
it looks up the document and its label, forms the observation propensity ``1/(1+rank)^eta``
(zero beyond ``min(n, cutoff)``), draws the click against a uniform sample and writes both
results back at the document's slot.  The uniform samples come from torch's device generator.
"""
from typing import Optional
from typing import Tuple

import torch as _torch

from pytorchltr_amd import _C
from pytorchltr_amd._prepare import prepare_n as _prepare_n

_SIM_RETURN_TYPE = Tuple[_torch.LongTensor, _torch.FloatTensor]
_validate = True


def set_index_validation(on):
    """Switches the per-call index check of the click simulators (labels index `relevance_probs`, rankings
    index the list) on or off; returns the previous setting.  Off: no host-device synchronisation in
    simulate_pbm and friends (out-of-range indices then read clamped entries instead of raising)."""
    global _validate
    prev, _validate = _validate, bool(on)
    return prev



def _simulate_pbm_from_uniform(rankings, ys, n, relevance_probs, uniform, cutoff, eta):
    """Deterministic core: clicks = uniform[b, rank] < P(click | rank, label)."""
    _C.require_device(rankings, "rankings")
    B, L = rankings.shape
    rk = rankings.to(_torch.int64).contiguous()
    yy = ys.reshape(B, L).to(_torch.int64).contiguous()
    nn = _prepare_n(n, B)
    probs = relevance_probs.to(device=rk.device, dtype=_torch.float32).contiguous()
    if B > 0 and L > 0 and _validate and not _torch.cuda.is_current_stream_capturing():
        # the reference's gathers raise on bad indices; one combined device-side check here.  It costs a
        # host-device synchronisation per call: `set_index_validation(False)` drops it, and it is
        # skipped under stream capture, where a synchronisation is not allowed
        bad = ((yy < 0) | (yy >= probs.numel())).any() | ((rk < 0) | (rk >= L)).any()
        if bool(bad):
            raise IndexError("simulate_pbm: labels must index relevance_probs (%d entries) and rankings "
                             "the %d list positions" % (probs.numel(), L))
    uu = uniform.to(device=rk.device, dtype=_torch.float32).contiguous()
    clicks = _torch.empty(B, L, dtype=_torch.int64, device=rk.device)
    props = _torch.empty(B, L, dtype=_torch.float32, device=rk.device)
    if B > 0 and L > 0:
        with _C.device_ctx(rk):
            _C.check(_C.lib().ltr_pbm_clicks(
                _C.ptr(rk), _C.ptr(yy), _C.ptr(nn), _C.ptr(probs), int(probs.numel()), _C.ptr(uu),
                B, L, -1 if cutoff is None else int(cutoff), float(eta), _C.ptr(clicks),
                _C.ptr(props), _C.stream_of(rk)))
    return clicks, props


def simulate_pbm(rankings: _torch.LongTensor, ys: _torch.LongTensor,
                 n: _torch.LongTensor, relevance_probs: _torch.FloatTensor,
                 cutoff: Optional[int] = None,
                 eta: float = 1.0) -> _SIM_RETURN_TYPE:
    """Simulates clicks according to a position-biased user model (reference :12-63).

    Args:
        rankings: (batch_size, list_size) rankings (document index per rank).
        ys: (batch_size, list_size) relevance labels.
        n: (batch_size) number of documents per query.
        relevance_probs: (max_relevance) click probability per label, given observation.
        cutoff: maximum list size to simulate.
        eta: severity of the position bias (0.0 = none).

    Returns:
        (clicks, propensities), both (batch_size, list_size) in document order: clicks int64 in
        {0, 1}; propensities float32.
    """
    uniform = _torch.rand(rankings.shape, device=rankings.device, dtype=_torch.float32)
    return _simulate_pbm_from_uniform(rankings, ys, n, relevance_probs, uniform, cutoff, eta)


def simulate_perfect(rankings, ys, n, cutoff: Optional[int] = None):
    """Perfect user: clicks follow relevance only, no position bias (reference :66-84)."""
    probs = _torch.tensor([0.0, 0.2, 0.4, 0.8, 1.0], device=rankings.device)
    return simulate_pbm(rankings, ys, n, probs, cutoff, 0.0)


def simulate_position(rankings, ys, n, cutoff: Optional[int] = None,
                      eta: float = 1.0) -> _SIM_RETURN_TYPE:
    """Binarised relevance with position bias (reference :87-105)."""
    probs = _torch.tensor([0.1, 0.1, 0.1, 1.0, 1.0], device=rankings.device)
    return simulate_pbm(rankings, ys, n, probs, cutoff, eta)


def simulate_nearrandom(rankings, ys, n, cutoff: Optional[int] = None,
                        eta: float = 1.0) -> _SIM_RETURN_TYPE:
    """Near-random user with position bias (reference :108-127)."""
    probs = _torch.tensor([0.4, 0.45, 0.5, 0.55, 0.6], device=rankings.device)
    return simulate_pbm(rankings, ys, n, probs, cutoff, eta)
