"""Shared parity helpers for the -m gpu tests (test infrastructure).

The fp64-anchored rule: the CPU oracle run in float64 is the truth; the CPU oracle run in float32 (the arithmetic the
reference itself uses, training/train.py:150-156 on torch CPU ops) shows how far a correct fp32 implementation sits from it.
The HIP engine -- another fp32 implementation with a different summation order -- has to be as close to the truth as the
reference's own arithmetic is:   err(GPU vs fp64) <= FACTOR * err(CPU-fp32 vs fp64)   per parameter tensor, with a floor far
below every other tolerance of the suite for tensors where both errors are at round-off level.  No hand-picked constant
per layer: ill-conditioned tensors (train-mode BatchNorm over few samples, ReLU masks next to 0) get exactly the slack the
reference's own fp32 arithmetic needs on the same inputs."""
from collections import OrderedDict

import torch

FACTOR = 2.0
FLOOR = 2e-5          # relative L2; both implementations at fp32 round-off


def rel_l2(t, ref64):
    ref64 = ref64.double()
    return ((t.detach().double().cpu() - ref64).norm() / ref64.norm().clamp_min(1e-300)).item()


def chan_relerr(got, ref):
    """per-channel max|got - ref| / max|ref| of [B,C,H,W] tensors -> list of C floats (a channel whose magnitude is far below the
    tensor's max cannot hide behind it)"""
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    d = (got - ref).abs().amax(dim=(0, 2, 3))
    s = ref.abs().amax(dim=(0, 2, 3)).clamp_min(1e-30)
    return (d / s).tolist()


def oracle_grads(P, B, cpu_batch, dtype):
    """one train-mode fwd + loss + bwd of the CPU oracle in `dtype` -> (outputs, losses, {name: grad})"""
    from oracle import restatement as R
    Pd = OrderedDict((k, v.to(dtype)) for k, v in P.items())
    Bd = OrderedDict((k, v.to(dtype) if v.is_floating_point() else v.clone()) for k, v in B.items())
    batch = OrderedDict((k, v.to(dtype)) for k, v in cpu_batch.items())
    tr = R.OracleTrainer(Pd, Bd)
    out, losses = tr.forward_backward(batch)
    grads = OrderedDict((k, p.grad) for k, p in tr.P.items())
    return {k: v.detach() for k, v in out.items()}, losses, grads, tr


def anchored_report(gpu, cpu32, ref64, factor=FACTOR, floor=FLOOR):
    """gpu / cpu32 / ref64: {name: tensor or None}.  Returns (failures, rows) with rows = (ratio, name, err_gpu, err_cpu)."""
    rows, bad = [], []
    for n, r in ref64.items():
        if r is None:
            assert gpu.get(n) is None, "%s: the oracle has no gradient here, the engine produced one" % n
            continue
        assert gpu.get(n) is not None, "%s: missing gradient" % n
        eg, ec = rel_l2(gpu[n], r), rel_l2(cpu32[n], r)
        bound = max(factor * ec, floor)
        rows.append((eg / max(ec, 1e-30), n, eg, ec))
        if not eg <= bound:
            bad.append("%s gpu %.2e cpu32 %.2e bound %.2e" % (n, eg, ec, bound))
    rows.sort(reverse=True)
    return bad, rows
