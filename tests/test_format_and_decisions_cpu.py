"""CPU: (1) the operand-format switch (footprints_amd/_format.py): the default is the exact bf16x3 split, fp16 pairs have to be asked for, the legacy
spelling still works; (2) the decision-forced oracle (oracle/restatement.py ReluDecisions, tests/parity.py decision_forced_report): with its OWN ReLU /
max-pool decisions imposed the float64 oracle reproduces itself, with one decision flipped the gradients behind it move -- the tool the GPU parity
cases use to separate an implementation's decisions from its arithmetic (network.py:48-59: ResNet-34 encoder behind train-mode BatchNorm)."""
import pytest
import torch
import torch.nn.functional as F

from footprints_amd._format import DTYPE_LABEL, FORMATS, format_env, operand_format


def test_operand_format_defaults_to_exact_and_reads_both_spellings():
    assert FORMATS == ("exact", "fp16_pair") and set(DTYPE_LABEL) == set(FORMATS)
    assert operand_format({}) == "exact"
    assert operand_format({"FP_OPERANDS": "fp16_pair"}) == "fp16_pair" and operand_format({"FP_OPERANDS": " Exact "}) == "exact"
    assert operand_format({"FP_HP": "1"}) == "fp16_pair" and operand_format({"FP_HP": "0"}) == "exact"
    assert operand_format({"FP_OPERANDS": "exact", "FP_HP": "1"}) == "exact"          # the new spelling wins over a stale legacy variable
    with pytest.raises(ValueError):
        operand_format({"FP_OPERANDS": "bf16"})
    for f in FORMATS:
        assert operand_format(format_env(f)) == f
    assert "opt-in" in DTYPE_LABEL["fp16_pair"] and "exact" in DTYPE_LABEL["exact"]


def _pool_winners(x):
    """window position ky * 3 + kx of max_pool2d(3, 2, 1)'s winner per output element (the engine's encoding, csrc/bn_pool.hip)"""
    N, C, H, W = x.shape
    y, idx = F.max_pool2d(x, 3, 2, 1, return_indices=True)
    OH, OW = y.shape[2:]
    iy, ix = idx // W, idx % W
    oy, ox = torch.arange(OH).view(1, 1, OH, 1), torch.arange(OW).view(1, 1, 1, OW)
    return (iy - (oy * 2 - 1)) * 3 + (ix - (ox * 2 - 1))


def test_the_oracle_under_its_own_decisions_is_itself_and_a_flipped_decision_moves_the_gradients():
    from oracle import restatement as R
    from tests.parity import count_decision_flips, oracle_grads, rel_l2
    P, B = R.make_state(tag="dec")
    batch = R.make_batch(1, 64, 96, tag="dec")
    rec = R.ReluDecisions()
    out, _, g, _, _ = oracle_grads(P, B, batch, torch.float64, relu_decisions=rec)
    assert len(rec.taken) == 33                                      # stem + 2 per BasicBlock of ResNet-34 (3 + 4 + 6 + 3 blocks)
    # the stem's output feeds the max-pool: its winners from the recorded stem activation
    stem = R.resnet_encoder(batch["image"].double(), {k: v.double() for k, v in P.items()},
                            {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in B.items()}, True)[0]
    pool = _pool_winners(stem.detach())
    out2, _, g2, _, _ = oracle_grads(P, B, batch, torch.float64, relu_decisions=R.ReluDecisions(impose=rec.taken, pool_impose=pool))
    assert all(torch.equal(out[k], out2[k]) for k in out)
    assert max(rel_l2(g2[k], g[k]) for k in g if g[k] is not None) < 1e-12
    assert count_decision_flips({"relu": rec.taken, "pool": pool}, rec.taken) == (0, sum(m.numel() for m in rec.taken))
    # flip ONE decision of the last block's output ReLU (an active element switched off): the forward hardly moves, gradients in front of it do
    flipped = [m.clone() for m in rec.taken]
    pos = flipped[-1].nonzero()[0]
    flipped[-1][tuple(pos)] = False
    assert count_decision_flips(flipped, rec.taken)[0] == 1
    _, _, g3, _, _ = oracle_grads(P, B, batch, torch.float64, relu_decisions=R.ReluDecisions(impose=flipped))
    moved = [k for k in g if g[k] is not None and k.startswith("encoder.layer4.2") and rel_l2(g3[k], g[k]) > 1e-9]
    assert moved, "switching off an active element of the last encoder block must move that block's gradients"
    # decoder parameters behind the flipped activation see a different feature, the stem in front of everything still gets a gradient
    assert g3["encoder.layer0.0.weight"] is not None


def test_imposed_decisions_are_bounded_to_roundoff_ties():
    """Round 6 (VERDICT r5 "Next" 2): the forced oracle must not be able to absorb a wrong mask.  decision_forced_report returns how many
    imposed decisions differ from the float64 run's own and how far from the boundary the float64 value of the worst one sits;
    assert_decisions_at_roundoff accepts the run's own decisions (0 flips), rejects an ordinary active element switched off (|z| ~ RMS, not a
    tie), rejects a thin slab of wrong masks by COUNT, and rejects a max-pool winner that is not a tie."""
    import pytest
    from oracle import restatement as R
    from tests.parity import FLIP_MAX_DISTANCE, assert_decisions_at_roundoff, decision_forced_report, oracle_grads
    P, B = R.make_state(tag="dec")
    batch = R.make_batch(1, 64, 96, tag="dec")
    rec = R.ReluDecisions()
    _, _, g, _, _ = oracle_grads(P, B, batch, torch.float64, relu_decisions=rec)
    stem = R.resnet_encoder(batch["image"].double(), {k: v.double() for k, v in P.items()},
                            {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in B.items()}, True)[0]
    pool = _pool_winners(stem.detach())
    own = {"relu": rec.taken, "pool": pool}
    _, _, _, st = decision_forced_report(P, B, batch, own, g, g, g)
    assert st["relu_flips"] == 0 and st["pool_flips"] == 0 and st["relu_flip_worst_distance"] == 0.0 and st["pool_flip_worst_distance"] == 0.0
    assert st["relu_decisions"] == sum(m.numel() for m in rec.taken) and st["pool_decisions"] == pool.numel()
    assert_decisions_at_roundoff(st, "own decisions")
    # one ordinary active element switched off: the forward moves, so masks downstream of it stop matching the forced run's own judgement as
    # well -- a wrong decision of ordinary magnitude announces itself in the count AND in the distance
    flipped = [m.clone() for m in rec.taken]
    pos = flipped[5].nonzero()[0]
    flipped[5][tuple(pos)] = False
    _, _, _, st = decision_forced_report(P, B, batch, {"relu": flipped, "pool": pool}, g, g, g)
    assert st["relu_flips"] >= 1 and st["relu_flip_worst_distance"] > 100 * FLIP_MAX_DISTANCE and st["relu_flip_worst_where"][0] >= 5
    with pytest.raises(AssertionError, match="not a round-off tie"):
        assert_decisions_at_roundoff(st, "one wrong mask", max_fraction=1.0)     # count check disabled: the DISTANCE must catch it
    with pytest.raises(AssertionError, match="ReLU decisions differ"):
        assert_decisions_at_roundoff(st, "one wrong mask")
    # a thin slab (one row of one channel) mis-masked: fails on the count as well
    slab = [m.clone() for m in rec.taken]
    slab[2][0, 3, 1, :] = ~slab[2][0, 3, 1, :]
    _, _, _, st = decision_forced_report(P, B, batch, {"relu": slab, "pool": pool}, g, g, g)
    assert st["relu_flips"] >= slab[2].shape[-1]
    with pytest.raises(AssertionError):
        assert_decisions_at_roundoff(st, "slab", max_distance=1e9)       # distance check disabled: the COUNT must catch it
    # a pool window told to pick a non-maximal element
    wrong = pool.clone()
    wrong[0, 0, 4, 4] = (wrong[0, 0, 4, 4] + 1) % 9 if (wrong[0, 0, 4, 4] + 1) % 9 != pool[0, 0, 4, 4] else (wrong[0, 0, 4, 4] + 2) % 9
    _, _, _, st = decision_forced_report(P, B, batch, {"relu": rec.taken, "pool": wrong}, g, g, g)
    if st["pool_flips"]:                                                  # (the neighbour may hold exactly the same value: then it IS a tie)
        assert st["pool_flip_worst_distance"] > FLIP_MAX_DISTANCE
        with pytest.raises(AssertionError, match="max-pool winner"):
            assert_decisions_at_roundoff(dict(st, relu_flips=0, relu_flip_worst_distance=0.0), "wrong pool winner")   # the pool's own check


def test_flip_distance_is_anchored_on_the_reference_arithmetic():
    """the relative half of the bound: the engine's worst flipped decision may sit no further from zero (in float64) than FACTOR x the CPU fp32
    run's own worst flip (floor 1e-5), and it may not flip many more decisions than the CPU fp32 run does"""
    import pytest
    from oracle import restatement as R
    from tests.parity import assert_decisions_at_roundoff, oracle_grads, reference_flip_stats
    base = {"relu_flips": 20, "relu_decisions": 10 ** 8, "relu_flip_worst_distance": 2e-5, "relu_flip_worst_where": (28, 83),
            "pool_flips": 0, "pool_decisions": 10 ** 6, "pool_flip_worst_distance": 0.0}
    assert_decisions_at_roundoff(base, "t", reference={"relu_flips": 15, "relu_flip_worst_distance": 1.2e-5})
    assert_decisions_at_roundoff(dict(base, relu_flip_worst_distance=9e-6), "t", reference={"relu_flips": 15, "relu_flip_worst_distance": 1e-7})   # floor
    with pytest.raises(AssertionError, match="CPU fp32"):
        assert_decisions_at_roundoff(dict(base, relu_flip_worst_distance=6e-5), "t", reference={"relu_flips": 15, "relu_flip_worst_distance": 1.2e-5})
    with pytest.raises(AssertionError, match="differ from float64"):
        assert_decisions_at_roundoff(dict(base, relu_flips=100), "t", reference={"relu_flips": 15, "relu_flip_worst_distance": 1.2e-5})
    with pytest.raises(AssertionError, match="not a round-off tie"):
        assert_decisions_at_roundoff(dict(base, relu_flip_worst_distance=2e-4), "t")                                                               # absolute
    # the measurement itself on the CPU: the fp32 oracle's decisions imposed on the float64 oracle
    P, B = R.make_state(tag="dec")
    batch = R.make_batch(1, 64, 96, tag="dec")
    rec = R.ReluDecisions()
    oracle_grads(P, B, batch, torch.float32, relu_decisions=rec)
    st = reference_flip_stats(P, B, batch, rec.taken)
    assert st["relu_decisions"] == sum(m.numel() for m in rec.taken) and st["relu_flips"] <= 8 and st["relu_flip_worst_distance"] <= 1e-4


def test_probe_measurement_equals_the_forced_run():
    """ReluDecisions(probe=...): the float64 recording run measures somebody else's masks without imposing them -- same flips and (to ~1e-7)
    the same worst distance as a float64 run that has them imposed (reference_flip_stats)"""
    from oracle import restatement as R
    from tests.parity import fp32_forward_decisions, oracle_grads, probe_flip_stats, reference_flip_stats
    P, B = R.make_state(tag="dec")
    batch = R.make_batch(2, 64, 96, tag="probe")
    dec32 = fp32_forward_decisions(P, B, batch)
    rec32 = R.ReluDecisions()
    oracle_grads(P, B, batch, torch.float32, relu_decisions=rec32)
    assert all(torch.equal(a, b) for a, b in zip(dec32, rec32.taken))              # the forward-only pass takes the training run's decisions
    dec64 = R.ReluDecisions(probe=dec32)
    oracle_grads(P, B, batch, torch.float64, relu_decisions=dec64)
    a, b = probe_flip_stats(dec64), reference_flip_stats(P, B, batch, dec32)
    assert a["relu_decisions"] == b["relu_decisions"] and abs(a["relu_flips"] - b["relu_flips"]) <= 2
    assert abs(a["relu_flip_worst_distance"] - b["relu_flip_worst_distance"]) <= 1e-6 + 0.05 * b["relu_flip_worst_distance"]
