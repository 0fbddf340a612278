"""pytorchltr_amd: MI355X-native ranking losses and ranking metrics.

Drop-in for the per-batch hot path of rjagerman/pytorchltr: ``pytorchltr_amd.loss``,
``pytorchltr_amd.evaluation`` and ``pytorchltr_amd.utils`` keep the reference's class and
function signatures (``pytorchltr/loss/__init__.py:1-7``, ``pytorchltr/evaluation/__init__.py:1-4``,
``pytorchltr/utils/__init__.py:1-5``) and run hand-written HIP kernels for gfx950 through the
C ABI declared in ``include/ltr_hip.h``.  There is no CPU fallback: tensors must live on a
ROCm device and the in-tree extension (``python -m pytorchltr_amd.build``) must be present.
"""
__version__ = "0.1.0"
