"""CPU tier: register spills of the built library, read from the code-object notes.

`pytorchltr_amd._codeobj.kernel_records` takes the gfx950 ELF images out of libltr_hip.so and returns the
assembler's per-kernel metadata.  A spilled VGPR in one of these kernels is scratch traffic through the very
memory pipe the kernel is bound by -- and, in the register-tile kernels, an `s_waitcnt vmcnt(0)` inside the load
burst -- so the kernels on the plans of the BASELINE configs must not spill at all, and the count of spilling
kernels in the library may only go down (a ratchet: round 4 started at 99 of 430, the parts kernel's tile
registers being live around its whole loop body on one path).
No GPU needed: the library is cross-compiled by __graft_entry__.build()."""
import os

import pytest


def _records():
    from pytorchltr_amd import _codeobj
    from pytorchltr_amd.build import LIB_PATH
    if not os.path.exists(LIB_PATH):
        pytest.skip("libltr_hip.so not built")
    try:
        recs = _codeobj.kernel_records(LIB_PATH)
    except FileNotFoundError as exc:          # no llvm tools on this machine
        pytest.skip(str(exc))
    assert len(recs) > 100
    return recs


# kernels the BASELINE configs run on (plan table: tests/test_boundary.py; instantiations: csrc/*.inc launchers)
BASELINE_KERNELS = [
    "linear_regtile2_kernel<0, 9, 34, 512>",        # C2  1024 x 128 x 136 hinge (the headline)
    "linear_regtile2w_kernel<6, 19, 34, 256>",      # C3  LambdaNDCG2 (the uncapped entry point of the same body)
    "linear_cluster_kernel<1, 512, 8>",             # C4  256 x 1000 x 220 DCG-hinge (integer labels: sorted runs) and its shard of 8 GPUs (32 queries)
    "linear_cluster_kernel<1, 512, 12>",            # C4 on float labels (the pair pass)
    "linear_parts_kernel<0, 16, 3, 2, 0, false>",   # C5  512 x 512 x 700 hinge
    "linear_cluster_kernel<0, 1024, 19>",           # C5 shard of 8 GPUs (64 queries)
    "linear_reduce_kernel(",                        # the cross-query reduction (+ SGD update) of every fused step
    "sgd_update_kernel(",
    "pairwise_loss_kernel<0, 0, 8>",                # loss only, C2 shape (symmetric pass, 8 waves)
    "pairwise_loss_kernel80<6, 4>",                 # loss only, C3 shape (LambdaNDCG2: the entry point capped at 80 SGPRs)
    "pairwise_loss_kernel80<6, 8>",
    "metric_kernel<1, 0>",                          # ndcg@10 (C3)
    "mlp_tile_kernel<0, 9, 34, 128, false, false>",  # f-2: the guide's MLP + hinge at the C2 shape
    "mlp_reduce4_kernel",
]


def test_baseline_plan_kernels_do_not_spill():
    recs = _records()
    by_name = {}
    for r in recs:
        by_name.setdefault(r.get("demangled", r["name"]).replace("(anonymous namespace)::", ""), r)
    for pat in BASELINE_KERNELS:
        hits = [(n, r) for n, r in by_name.items() if pat in n]
        assert hits, "no kernel matching %r in the library (renamed? update BASELINE_KERNELS)" % pat
        for n, r in hits:
            assert r.get("vgpr_spill_count", 0) == 0 and r.get("private_segment_fixed_size", 0) == 0, \
                "%s spills %d VGPRs (%d B scratch)" % (n, r.get("vgpr_spill_count", -1), r.get("private_segment_fixed_size", -1))


def test_no_fused_scorer_kernel_outside_the_known_exceptions_spills():
    """Every instantiation of the fused Linear scorer + loss kernels that a plan can pick: zero spills, except the
    listed ones -- the counting formulation of the hinge kinds on lists beyond 1024 documents is a real call under
    the register-resident tile (its callee-saved registers are stored around it by the ABI), and the LambdaNDCG
    instantiations of the older register-tile / cluster layouts at their largest tile."""
    recs = _records()
    known = ("linear_parts_kernel<0, 40, 1, 2, 1, false>", "linear_parts_kernel<0, 20, 2, 2, 1, false>", "linear_parts_kernel<0, 14, 3, 2, 1, false>",
             "linear_parts_kernel<1, 40, 1, 2, 1, false>", "linear_parts_kernel<1, 20, 2, 2, 1, false>", "linear_parts_kernel<1, 14, 3, 2, 1, false>",
             "linear_regtile_kernel<", "linear_regtile2w_kernel<5, 24,", "linear_regtile2w_kernel<6, 24,",
             "linear_regtile2_kernel<1, 3, 0, 512>", "linear_regtile2_kernel<1, 5, 0, 512>", "linear_regtile2_kernel<1, 9, 0, 512>",     # (five VGPRs under the eight-wave cap)
             "linear_regtile2w_kernel<2, 12, 0, 512>", "linear_regtile2w_kernel<4, 12, 0, 512>",       # (one float4 each: see the ratchet below)       # (the NDCG kinds on 24 sweeps: 4-12 VGPRs, and faster than the two-pass kernel)
            
             # (the general kernel's 16-byte-row passes, held at 64 VGPRs for eight waves per SIMD: 1-3 VGPRs each and 10-12 % faster)
             "linear_pairwise_kernel<0, 0, 4>", "linear_pairwise_kernel<0, 1, 4>", "linear_pairwise_kernel<1, 0, 4>", "linear_pairwise_kernel<1, 1, 4>",
             "linear_pairwise_kernel<2, 0, 4>", "linear_pairwise_kernel<2, 1, 4>", "linear_pairwise_kernel<3, 0, 4>", "linear_pairwise_kernel<3, 1, 4>",
             "linear_pairwise_kernel<4, 0, 4>", "linear_pairwise_kernel<4, 1, 4>",
             "linear_cluster_kernel<5,", "linear_cluster_kernel<6,", "linear_cluster_kernel<2, 512, 12>",
             "linear_cluster_kernel<4, 512, 12>")
    bad = []
    for r in recs:
        n = r.get("demangled", r["name"]).replace("(anonymous namespace)::", "")
        if not any(k in n for k in ("linear_parts_kernel", "linear_cluster_kernel", "linear_regtile", "linear_pairwise_kernel")):
            continue
        if r.get("vgpr_spill_count", 0) and not any(k in n for k in known):
            bad.append((n[:80], r["vgpr_spill_count"]))
    assert not bad, bad


def test_spilling_kernel_count_only_goes_down():
    recs = _records()
    spilling = [r for r in recs if r.get("vgpr_spill_count", 0) > 0]
    # (round 6: 29 -> 31 -- the data-parallel lazy step's in-launch all-reduce, code only the reducer workgroups run, costs the
    # 12-sweep generic-width tiles of the logistic / LambdaARP2 kinds, which sit at their 80-VGPR cap, one spilled float4 each)
    # (round 6, last session: + 10 -- the general fused kernel's 16-byte-row passes are held at 64 VGPRs, eight waves per SIMD; one to
    # three spilled VGPRs each buy a fourth resident workgroup per CU: 10-12 % faster on every shape measured)
    # (+ 3: the DCG-hinge generic-width register tiles under the eight-wave cap)
    assert len(spilling) <= 44, sorted((r.get("demangled", r["name"])[:70], r["vgpr_spill_count"]) for r in spilling)


def test_eight_wave_entry_points_stay_under_the_sgpr_cliff():
    """a wave's SGPRs are allocated in sixteens and the trap handler takes sixteen more out of 800 per SIMD -- more
    than 80 SGPRs run SEVEN waves per SIMD, not eight (round 6: the headline tile at 82 lost a quarter of its resident
    workgroups, 11.7 -> 13.3 us with an identical hot path).  The entry points compiled for eight waves must stay at
    <= 80 SGPRs and <= 64 VGPRs, unspilled."""
    recs = _records()
    seen = 0
    for r in recs:
        n = r.get("demangled", r["name"]).replace("(anonymous namespace)::", "")
        if "linear_regtile2_kernel<" in n or "pairwise_loss_kernel80<" in n or "linear_pairwise_kernel80<" in n:
            seen += 1
            assert r["sgpr_count"] <= 80, (n, r["sgpr_count"])
            assert r["vgpr_count"] + r.get("agpr_count", 0) <= 64, (n, r["vgpr_count"])
            # (the DCG-hinge generic-width tiles keep five VGPRs in scratch under the cap -- and are 16 % faster for it)
            assert r.get("vgpr_spill_count", 0) <= (5 if "linear_regtile2_kernel<1, " in n and ", 0, 512>" in n else 0), n
    assert seen >= 10, seen
