"""How equal scores are ordered by the ranking kernels (rank_by_score, dcg/ndcg, arp).

The reference breaks ties with ONE random permutation per call, shared by all rows
(tiebreak_argsort, utils/tensor_operations.py:29-45; rank_by_score and every metric draw it from
torch's global RNG, :48-64, evaluation/dcg.py:85, evaluation/arp.py:32), so that a scorer whose
outputs tie -- constant, zero-initialised, ReLU-dead -- gets an unbiased expected metric instead of
whatever the storage order of the documents happens to reward.  The kernels take that permutation
as an int32 priority per list position (include/ltr_hip.h: ltr_*_tie_f32): of two documents with
equal score the one with the smaller priority ranks first.

mode "random" (default, the reference's behaviour): a fresh pseudo-random order of the tied positions
    per call.  Round 3: no permutation is drawn on the device any more (torch.randperm cost five kernels
    per metric call, 35 us against a 6 us kernel) -- ONE 62-bit seed is drawn on the host from `generator`
    or torch's default CPU generator (the reference draws its randperm from the same global CPU RNG;
    `torch.manual_seed` makes it reproducible) and the kernels hash (seed, position) into the tie word
    themselves (include/ltr_hip.h: ltr_*_seed_f32).  A generator living on a device is honoured too:
    its draw stays on the device and the kernel reads it there;
mode "index": deterministic document-index order, no draw (the round-1 behaviour; bit-reproducible
    metrics, biased on tied scores).
Rows without ties give the same result in both modes.
"""
import torch

_MODES = ("random", "index")
_mode = "random"


def get_tie_breaking():
    return _mode


def set_tie_breaking(mode):
    """Sets the process-wide tie-break mode ("random" or "index"); returns the previous one."""
    global _mode
    if mode not in _MODES:
        raise ValueError("tie-breaking mode must be one of %s" % (_MODES,))
    prev, _mode = _mode, mode
    return prev


class tie_breaking:
    """Context manager: `with tie_breaking("index"): ...`."""

    def __init__(self, mode):
        if mode not in _MODES:
            raise ValueError("tie-breaking mode must be one of %s" % (_MODES,))
        self.mode = mode
        self.prev = None

    def __enter__(self):
        self.prev = set_tie_breaking(self.mode)
        return self

    def __exit__(self, *exc):
        set_tie_breaking(self.prev)
        return False


def draw_seed(L, device, generator=None):
    """(seed, seed_tensor) for one call, or None for index order.  `seed_tensor` is a device int64[1] when
    the generator lives on a device (the kernel reads it there, no host round trip), else None and `seed`
    is a python int drawn on the host.  An explicit `generator` is always honoured."""
    if generator is None:
        if _mode == "index" or L <= 1:
            return None
        return int(torch.randint(0, 1 << 62, (1,), dtype=torch.int64).item()), None
    gdev = getattr(generator, "device", torch.device("cpu"))
    if gdev.type == "cpu":
        return int(torch.randint(0, 1 << 62, (1,), dtype=torch.int64, generator=generator).item()), None
    t = torch.empty(1, dtype=torch.int64, device=gdev).random_(0, 1 << 62, generator=generator)
    return 0, t.to(device)


def hash_words(seed, L):
    """The tie words the kernels make from `seed` for positions 0..L-1 (numpy int32; for tests / oracles)."""
    import numpy as np
    from . import _C
    lib = _C.lib()
    return np.array([lib.ltr_tie_hash_word(seed, j) for j in range(L)], dtype=np.int32)


def draw_priorities(L, device, generator=None):
    """int32 (L) tie priorities on `device` for one call, or None for index order.  An explicit
    `generator` is always honoured (on whichever device it lives)."""
    if generator is None:
        if _mode == "index" or L <= 1:
            return None
        return torch.randperm(L, dtype=torch.int32, device=device)
    gdev = getattr(generator, "device", torch.device("cpu"))
    perm = torch.randperm(L, generator=generator, device=gdev)
    return perm.to(device=device, dtype=torch.int32)
