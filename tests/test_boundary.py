"""CPU tier: the drop-in boundary.  The C-ABI library loads and exports every symbol that
include/ltr_hip.h declares; the Python surface mirrors the reference's names and signatures;
the product never touches the oracle; CPU tensors are refused (no fallback).  No HIP compute
is launched here."""
import ast
import ctypes
import inspect
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "ltr_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ltr_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_all_exported_and_bound():
    from pytorchltr_amd import _C
    from pytorchltr_amd.build import build_extension
    build_extension()
    declared = _declared_symbols()
    assert len(declared) >= 15
    handle = ctypes.CDLL(_C.LIB_PATH)
    # (the ltr_debug_* hooks are exported by the default build -- this tier and bench.py use them -- and left out by a
    # production build, LTR_NO_DEBUG_HOOKS=1: all of them or none)
    hooks = [name for name in declared if name.startswith("ltr_debug_")]
    assert len(hooks) >= 8 and len({hasattr(handle, name) for name in hooks}) == 1
    for name in declared:
        if name.startswith("ltr_debug_") and os.environ.get("LTR_NO_DEBUG_HOOKS") == "1":
            assert not hasattr(handle, name), "a production build exports %s" % name
            continue
        assert hasattr(handle, name), "libltr_hip.so does not export %s" % name
    # the ctypes table covers exactly the header
    assert sorted(_C.SIGNATURES) == declared
    lib = _C.lib()
    assert lib.ltr_version() == 114
    assert lib.ltr_max_list_len() >= 1024
    assert b"NULL" in lib.ltr_error_string(-1)
    assert lib.ltr_linear_workspace_bytes(1024, 128, 136) >= 1024 * 137 * 4


def test_argument_validation_without_a_gpu():
    """Argument errors are decided on the host before any launch."""
    from pytorchltr_amd import _C
    lib = _C.lib()
    assert lib.ltr_pairwise_loss_f32(99, 1.0, None, None, 0, None, 1, 8, None, None, None) == -3
    assert lib.ltr_pairwise_loss_f32(0, 1.0, None, None, 7, None, 1, 8, None, None, None) == -3
    assert lib.ltr_pairwise_loss_f32(0, 1.0, None, None, 0, None, -1, 8, None, None, None) == -2
    assert lib.ltr_pairwise_loss_f32(0, 1.0, None, None, 0, None, 1, 0, None, None, None) == -2
    assert lib.ltr_pairwise_loss_f32(0, 1.0, None, None, 0, None, 1, 100000, None, None, None) == -4
    assert lib.ltr_pairwise_loss_f32(0, 1.0, None, None, 0, None, 1, 8, None, None, None) == -1
    assert lib.ltr_pairwise_loss_f32(0, 1.0, None, None, 0, None, 0, 8, None, None, None) == 0
    assert lib.ltr_pairwise_loss_f32_cfg(0, 1.0, 1, 1, 0, 1, 1, 8, 1, None, 96, 1, 1, None) == -6
    assert lib.ltr_pairwise_loss_f32_cfg(0, 1.0, 1, 1, 0, 1, 1, 8, 1, None, 64, 3, 1, None) == -6
    assert lib.ltr_dcg_f32(None, None, 0, None, 1, 8, -1, 1, 0, None, None) == -2
    assert lib.ltr_batch_pairs(None, 2, 1, 8, None, None) == -3
    assert lib.ltr_linear_pairwise_f32(0, 1.0, 1, 1, 1, 1, 0, 1, None, 4, 8, 4, 1, None, 1, 1,
                                       None, 0, None) == -5
    with pytest.raises(RuntimeError, match="list_len"):
        _C.check(-4)


def test_python_surface_mirrors_reference_signatures():
    import pytorchltr_amd.evaluation as ev
    import pytorchltr_amd.loss as losses
    import pytorchltr_amd.utils as utils
    # pytorchltr/loss/__init__.py:1-7
    for name in ("PairwiseHingeLoss", "PairwiseDCGHingeLoss", "PairwiseLogisticLoss",
                 "LambdaARPLoss1", "LambdaARPLoss2", "LambdaNDCGLoss1", "LambdaNDCGLoss2"):
        cls = getattr(losses, name)
        assert issubclass(cls, torch.nn.Module)
        assert list(inspect.signature(cls.forward).parameters) == ["self", "scores", "relevance", "n"]
        assert len(cls().state_dict()) == 0            # parameter-free, checkpoints interchangeable
    assert issubclass(losses.PairwiseDCGHingeLoss, losses.PairwiseHingeLoss)
    assert losses.PairwiseLogisticLoss(sigma=2.0).sigma == 2.0
    assert losses.LambdaNDCGLoss2().sigma == 1.0
    assert [p.default for p in inspect.signature(losses.PairwiseLogisticLoss.__init__).parameters.values()][1] == 1.0
    # pytorchltr/evaluation/__init__.py:1-3, evaluation/dcg.py:8-10,41-43, arp.py:7-8
    for fn in (ev.dcg, ev.ndcg):
        sig = inspect.signature(fn)
        assert list(sig.parameters) == ["scores", "relevance", "n", "k", "exp"]
        assert sig.parameters["k"].default is None and sig.parameters["exp"].default is True
    assert list(inspect.signature(ev.arp).parameters) == ["scores", "relevance", "n"]
    # pytorchltr/utils/tensor_operations.py:6-8,29-32,48-51,94
    sig = inspect.signature(utils.mask_padded_values)
    assert list(sig.parameters) == ["xs", "n", "mask_value", "mutate"]
    assert sig.parameters["mask_value"].default == -float("inf") and sig.parameters["mutate"].default is False
    assert list(inspect.signature(utils.tiebreak_argsort).parameters) == ["x", "descending", "generator"]
    assert list(inspect.signature(utils.rank_by_score).parameters) == ["scores", "n", "generator"]
    assert list(inspect.signature(utils.batch_pairs).parameters) == ["x"]


def test_cpu_tensors_are_refused_not_silently_computed():
    import pytorchltr_amd.evaluation as ev
    import pytorchltr_amd.loss as losses
    from pytorchltr_amd.utils import batch_pairs, rank_by_score
    s = torch.zeros(2, 4)
    y = torch.zeros(2, 4, dtype=torch.int64)
    n = torch.tensor([4, 2])
    for call in (lambda: losses.PairwiseHingeLoss()(s, y, n), lambda: losses.LambdaNDCGLoss2()(s, y, n),
                 lambda: ev.ndcg(s, y, n, k=10), lambda: ev.arp(s, y, n),
                 lambda: rank_by_score(s, n), lambda: batch_pairs(s)):
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            call()


def test_missing_extension_fails_loudly(monkeypatch):
    from pytorchltr_amd import _C
    monkeypatch.setattr(_C, "_lib", None)
    monkeypatch.setattr(_C, "LIB_PATH", os.path.join(ROOT, "pytorchltr_amd", "csrc", "nope.so"))
    with pytest.raises(ImportError, match="no CPU fallback"):
        _C.lib()


def test_product_never_imports_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use oracle/."""
    pkg = os.path.join(ROOT, "pytorchltr_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if not f.endswith(".py"):
                continue
            tree = ast.parse(open(os.path.join(dirpath, f)).read())
            for node in ast.walk(tree):
                mods = []
                if isinstance(node, ast.Import):
                    mods = [a.name for a in node.names]
                elif isinstance(node, ast.ImportFrom):
                    mods = [node.module or ""]
                for m in mods:
                    assert not m.split(".")[0] == "oracle", "%s imports %s" % (f, m)
            assert "/root/reference" not in open(os.path.join(dirpath, f)).read()
    for f in ("bench.py", "__graft_entry__.py"):
        assert "/root/reference" not in open(os.path.join(ROOT, f)).read()


def test_cutoff_semantics_follow_python_slicing():
    from pytorchltr_amd.evaluation.dcg import _cutoff
    assert _cutoff(None, 10) == 0            # full curve
    assert _cutoff(3, 10) == 3
    assert _cutoff(50, 10) == 10             # k > L silently means DCG@L (reference dcg.py:97-98)
    assert _cutoff(-2, 10) == 8              # dcg[:, :-2][:, -1]
    with pytest.raises(IndexError):
        _cutoff(0, 10)
    with pytest.raises(IndexError):
        _cutoff(-10, 10)


def test_prepare_shapes_and_dtypes_host_logic(monkeypatch):
    """Shape/dtype normalisation, exercised with the device check stubbed out."""
    from pytorchltr_amd import _C, _prepare
    monkeypatch.setattr(_C, "require_device", lambda t, what: None)
    s = torch.zeros(3, 5, 1, dtype=torch.float64)
    y = torch.zeros(3, 5, 1, dtype=torch.int16)
    n = torch.tensor([5, 2, 0], dtype=torch.int32)
    s2, y2, n2 = _prepare.prepare(s, y, n)
    assert s2.shape == (3, 5) and s2.dtype == torch.float32 and s2.is_contiguous()
    assert y2.shape == (3, 5) and y2.dtype == torch.float32
    assert n2.dtype == torch.int64
    with pytest.raises(ValueError):
        _prepare.prepare(torch.zeros(3, 5, 2), y, n)
    with pytest.raises(ValueError):
        _prepare.prepare(torch.zeros(3, 5), torch.zeros(3, 4), n)
    with pytest.raises(ValueError):
        _prepare.prepare(torch.zeros(3, 5), torch.zeros(3, 5), torch.tensor([1, 2]))
    with pytest.raises(TypeError):
        _prepare.prepare(torch.zeros(3, 5, dtype=torch.int64), torch.zeros(3, 5), n)
    with pytest.raises(TypeError):
        _prepare.prepare(torch.zeros(3, 5), torch.zeros(3, 5), torch.tensor([1.0, 2.0, 3.0]))
    with pytest.raises(ValueError):
        _prepare.prepare(torch.zeros(1, 5000), torch.zeros(1, 5000), torch.tensor([1]))
    assert _C.label_dtype(torch.zeros(1, dtype=torch.int64)) == _C.LABEL_I64
    assert _C.label_dtype(torch.zeros(1, dtype=torch.int32)) == _C.LABEL_I32
    assert _C.label_dtype(torch.zeros(1)) == _C.LABEL_F32


def test_fused_plan_for_the_baseline_shapes():
    """ltr_linear_fused_plan is host logic (no launch): which kernel the fused Linear scorer + loss
    step takes for the BASELINE.json shapes (256 CUs assumed when no device is present)."""
    from pytorchltr_amd import _C
    lib = _C.lib()
    assert lib.ltr_linear_fused_plan(_C.HINGE, 1024, 128, 136) == _C.PLAN_REGISTER_TILE      # C2
    assert lib.ltr_linear_fused_plan(_C.NDCG2, 1024, 128, 136) == _C.PLAN_REGISTER_TILE      # C3
    assert lib.ltr_linear_fused_plan(_C.DCG_HINGE, 256, 1000, 220) == _C.PLAN_CLUSTER        # C4
    assert lib.ltr_linear_fused_plan(_C.HINGE, 512, 512, 700) == _C.PLAN_PARTS               # C5
    assert lib.ltr_linear_fused_plan(_C.HINGE, 512, 512, 220) == _C.PLAN_GENERAL             # narrow rows
    assert lib.ltr_linear_fused_plan(_C.HINGE, 64, 512, 700) == _C.PLAN_CLUSTER              # C5 shard of 8 GPUs
    assert lib.ltr_linear_fused_plan(_C.DCG_HINGE, 32, 1000, 220) == _C.PLAN_CLUSTER         # C4 shard of 8 GPUs
    assert lib.ltr_linear_fused_plan(_C.NDCG2, 512, 512, 700) == _C.PLAN_PARTS               # round 4: every part ranks the query itself
    assert lib.ltr_linear_fused_plan(_C.NDCG1, 256, 1000, 700) == _C.PLAN_PARTS
    assert lib.ltr_linear_fused_plan(_C.NDCG2, 512, 512, 220) == _C.PLAN_GENERAL             # narrow rows, as for the other kinds
    assert lib.ltr_linear_fused_plan(_C.HINGE, 48, 2000, 64) == _C.PLAN_PARTS                # beyond the symmetric pass
    # round 4: the rank exchange lets the cluster kernel take the NDCG kinds like the other heavy kinds
    assert lib.ltr_linear_fused_plan(_C.NDCG2, 256, 1000, 220) == _C.PLAN_CLUSTER            # C4's shape with LambdaNDCG2
    assert lib.ltr_linear_fused_plan(_C.NDCG2, 384, 1000, 220) == _C.PLAN_GENERAL            # (measured: loses there)
    assert lib.ltr_linear_fused_plan(_C.NDCG1, 384, 1000, 220) == _C.PLAN_CLUSTER
    assert lib.ltr_linear_fused_plan(_C.NDCG1, 128, 1000, 136) == _C.PLAN_CLUSTER
    assert lib.ltr_linear_fused_plan(_C.NDCG2, 128, 512, 700) == _C.PLAN_PARTS                # wide rows: the parts kernel
    assert lib.ltr_linear_fused_plan(_C.NDCG2, 32, 1000, 220) == _C.PLAN_CLUSTER
    # where the parts kernel is measured to lose it is not picked (round 4: cold parts / general sweeps)
    assert lib.ltr_linear_fused_plan(_C.HINGE, 256, 300, 700) == _C.PLAN_GENERAL              # short lists, one workgroup per CU
    assert lib.ltr_linear_fused_plan(_C.LOGISTIC, 512, 512, 448) == _C.PLAN_GENERAL           # 112 float4 per row: the heavy kinds lose
    assert lib.ltr_linear_fused_plan(_C.ARP1, 768, 1000, 448) == _C.PLAN_GENERAL
    assert lib.ltr_linear_fused_plan(_C.LOGISTIC, 256, 1000, 700) == _C.PLAN_PARTS
    assert lib.ltr_linear_fused_plan(_C.HINGE, 160, 1000, 512) == _C.PLAN_PARTS               # the general kernel would leave CUs empty
    assert lib.ltr_linear_fused_plan(_C.HINGE, 1024, 256, 700) == _C.PLAN_PARTS               # short lists, wide rows, large batch
    assert lib.ltr_linear_fused_plan(_C.HINGE, 512, 128, 700) == _C.PLAN_GENERAL
    old = lib.ltr_debug_parts_all(1)                                                          # tests: every shape it can take
    assert lib.ltr_linear_fused_plan(_C.HINGE, 256, 300, 700) == _C.PLAN_PARTS
    assert lib.ltr_debug_parts_all(old) == 1
    assert lib.ltr_linear_fused_plan(_C.HINGE, 8, 5000, 136) == _C.PLAN_NONE                 # list_len > 4096
    assert lib.ltr_linear_fused_plan(99, 8, 100, 136) == _C.PLAN_NONE
    # the cluster kernel's scratch rides behind the (F+1, B) partials
    assert lib.ltr_linear_workspace_bytes(256, 1000, 220) > lib.ltr_linear_workspace_bytes(256, 128, 220)


def test_mlp_shape_limits_are_host_logic():
    """Which shapes the fused MLP step takes is decided on the host (no launch): lists up to 256
    documents while the feature rows fit the tile kernel (F <= 144), 128 for wider rows; the Python
    mirror (fused.mlp_supported) agrees with the library (ltr_mlp_max_list_len)."""
    from pytorchltr_amd import _C, fused
    lib = _C.lib()
    for F in (4, 48, 136, 144, 148, 200, 224):
        assert lib.ltr_mlp_max_list_len(F) == fused.mlp_max_list_len(F) == (256 if F <= 144 else 128)
    assert lib.ltr_mlp_max_list_len(6) == 0 and lib.ltr_mlp_max_list_len(228) == 0
    assert fused.mlp_supported(256, 136, 50, 10) and not fused.mlp_supported(257, 136, 50, 10)
    assert fused.mlp_supported(128, 220, 64, 16) and not fused.mlp_supported(129, 220, 64, 16)
    assert not fused.mlp_supported(64, 136, 65, 10) and not fused.mlp_supported(64, 138, 50, 10)
    # the workspace covers the larger of the two kernel grids, rows padded to 16 bytes
    P = lib.ltr_mlp_param_count(136, 50, 10)
    assert P == 50 * 136 + 50 + 10 * 50 + 10 + 10 + 1
    assert lib.ltr_mlp_workspace_bytes(3, 136, 50, 10) == 3 * ((P + 3) // 4 * 4) * 4
    z = 8
    assert lib.ltr_mlp_pairwise_f32(0, 1.0, z, z, z, z, z, z, z, z, 0, z, None, 1, 16, 8, 4, 4, z, None,
                                    z, None, z, 4, None) == -5              # LTR_ERR_WORKSPACE
    assert lib.ltr_mlp_scores_f32(z, z, z, z, z, z, z, z, 1, 257, 8, 4, 4, z, None) == -4
    assert lib.ltr_mlp_scores_f32(z, z, z, z, z, z, z, z, 1, 129, 200, 4, 4, z, None) == -4
