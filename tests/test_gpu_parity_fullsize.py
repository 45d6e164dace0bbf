"""GPU: train-mode forward + loss + backward of the HIP engine at the BASELINE.json workloads against the CPU oracle, with
float64-anchored gradient tolerances (tests/parity.py) instead of hand-picked constants.

Why these sizes: tile plans, split-K factors, weight-gradient split counts and the phase-kernel fallbacks all depend on
N, H, W (conv3x3_tile_bf3.hip plan3, wgrad3x3_bf3.hip, Engine._phase_ok), so parity at 2x96x128 says nothing about the
launches the benchmark actually runs:
  * 12x192x640  -- BASELINE configs[2] (KITTI train step): outputs, 21 losses, every parameter gradient, BN running statistics;
  *  1x512x640  -- the Matterport resolution of configs[4] (16x20 .. 512x640 pyramid: other plans / splits than KITTI);
  *  1x256x448  -- predict_simple's `handheld` model size (8x14 pyramid top: the phase kernels' padding fallbacks).
Bars: outputs per CHANNEL within 1e-4 of the fp32 CPU path (north_star) and of the float64 truth; losses 1e-4 relative; masks
bit-exact outside the |logit - thr| < 1e-4 max tie band; every parameter gradient fp64-anchored:
err(GPU vs fp64) <= 4 x max(err(CPU fp32 vs fp64), its stage median), relative L2 per tensor, floor 2e-5 (tests/parity.py says why).
"""
from collections import OrderedDict

import numpy as np
import pytest
import torch

from tests.parity import anchored_report, chan_relerr, oracle_grads, rel_l2, tie_free_batch

pytestmark = pytest.mark.gpu


def _gpu_step(P, B, cpu_batch):
    from footprints_amd import FootprintNetwork
    from footprints_amd.training.losses import LossManager
    model = FootprintNetwork(pretrained=False)
    model.load_state_dict({**P, **B})
    model.cuda().train()
    batch = {k: v.cuda() for k, v in cpu_batch.items()}
    out = model(batch["image"])
    losses = LossManager((0.1, 100), 0.25, compute_viz=False)(out, batch)
    losses["loss"].backward()
    torch.cuda.synchronize()
    grads = OrderedDict((n, p.grad) for n, p in model.named_parameters())
    return model, out, losses, grads


@pytest.mark.parametrize("Bn,Hn,Wn", [(1, 256, 448), (1, 512, 640), (12, 192, 640)])
def test_train_step_fp64_anchored(Bn, Hn, Wn):
    from oracle import restatement as R
    P, B = R.make_state(tag="anch")
    cpu_batch = R.make_batch(Bn, Hn, Wn, tag="anch%d" % Hn)
    removed = []

    def fix(batch, out64):
        b, n = tie_free_batch(batch, out64)
        removed.append(n)
        return b
    out64, l64, g64, _, cpu_batch = oracle_grads(P, B, cpu_batch, torch.float64, fix_batch=fix)      # float64 first: it defines the tie pixels
    out32, l32, g32, tr32, _ = oracle_grads(P, B, cpu_batch, torch.float32)
    model, out, losses, g_gpu = _gpu_step(P, B, cpu_batch)
    print("\n[%dx%dx%d] |.|-kink pixels removed from the depth masks: %d of %d" % (Bn, Hn, Wn, removed[0], 2 * Bn * Hn * Wn))
    # ---- outputs: per channel, against the reference's fp32 CPU arithmetic and against the float64 truth ---------------
    for k in R.SCALES:
        e32, e64 = chan_relerr(out[k], out32[k]), chan_relerr(out[k], out64[k])
        assert max(e32) <= 1e-4 and max(e64) <= 1e-4, "output %s per-channel rel err vs fp32 %s vs fp64 %s" % (k, e32, e64)
        ref = out32[k]
        for thr in (0.0, 0.5):          # sigmoid(logit) > 0.5 (losses.py:78) and predict_simple's logit > 0.5 (predict_simple.py:77)
            band = (ref[:, :2] - thr).abs() < 1e-4 * ref[:, :2].abs().max()
            assert torch.equal((out[k][:, :2].cpu() > thr) | band, (ref[:, :2] > thr) | band), "mask bits (%s, thr %.1f)" % (k, thr)
    # ---- 21 losses -----------------------------------------------------------------------------------------------------
    for key in R.LOSS_KEYS:
        ref = float(l32[key])
        assert abs(float(losses[key]) - ref) <= 1e-4 * max(abs(ref), 1e-3), (key, float(losses[key]), ref)
        assert abs(float(losses[key]) - float(l64[key])) <= 1e-4 * max(abs(float(l64[key])), 1e-3), key
    # ---- every parameter gradient, fp64-anchored -------------------------------------------------------------------------
    bad, rows = anchored_report(g_gpu, g32, g64)
    print("\n[%dx%dx%d] worst GPU/CPU32 error ratios (vs fp64): %s" % (Bn, Hn, Wn, ["%s %.2f (gpu %.1e cpu %.1e)" % (n, r, eg, ec)
                                                                             for r, n, eg, ec in rows[:6]]))
    print("median ratio %.2f, tensors %d" % (float(np.median([r for r, *_ in rows])), len(rows)))
    assert not bad, "gradients farther from the float64 truth than the reference's own fp32 arithmetic allows: %s" % bad[:10]
    # ---- BatchNorm running statistics after the step (train-mode side effect, network.py:40-44 / nn.BatchNorm2d) ----------
    sd = model.state_dict()
    for k, v in tr32.B.items():
        if k.endswith("num_batches_tracked"):
            assert int(sd[k]) == int(v), k
        elif "encoder" in k:
            assert rel_l2(sd[k], v) <= 1e-5, k


def test_g5_gradients_and_adam_state_fp64_anchored():
    """The G5 fixture's inputs (2x64x96: 12 BatchNorm samples per channel at layer4 -- ill-conditioned in fp32) through two
    drop-in steps; instead of loose constants the per-tensor gradients AND Adam's exp_avg / exp_avg_sq after two steps are held
    to the fp64-anchored rule against the oracle run in float64."""
    from footprints_amd.model_manager import ModelManager
    from footprints_amd.training.losses import LossManager
    from oracle import restatement as R
    P, B = R.make_state()
    batches = [R.make_batch(2, 64, 96, tag="g5.step%d" % s) for s in range(2)]
    mm = ModelManager(use_cuda=True, learning_rate=1e-4)
    mm.model.load_state_dict({**P, **B})
    mm.model.train()
    lm = LossManager((0.1, 100), 0.25, compute_viz=False)
    trs = {}
    for dt in (torch.float32, torch.float64):
        trs[dt] = R.OracleTrainer(OrderedDict((k, v.to(dt)) for k, v in P.items()),
                                  OrderedDict((k, v.to(dt) if v.is_floating_point() else v.clone()) for k, v in B.items()))
    names = [k for k, _ in mm.model.named_parameters()]
    for s, cb in enumerate(batches):
        batch = {k: v.cuda() for k, v in cb.items()}
        out = mm.model(batch["image"])
        losses = lm(out, batch)
        mm.model.zero_grad()
        losses["loss"].backward()
        g_gpu = OrderedDict((n, None if p.grad is None else p.grad.clone()) for n, p in mm.model.named_parameters())
        mm.optimiser.step()
        gs = {}
        for dt, tr in trs.items():
            tr.step(OrderedDict((k, v.to(dt)) for k, v in cb.items()))
            gs[dt] = OrderedDict((k, p.grad) for k, p in tr.P.items())
        if s == 0:          # the second step's gradients depend on sign-like first Adam updates of round-off-level gradients
            bad, rows = anchored_report(g_gpu, gs[torch.float32], gs[torch.float64])
            print("\nG5 step 0 worst ratios:", ["%s %.2f" % (n, r) for r, n, *_ in rows[:5]])
            assert not bad, bad[:10]
    st = mm.optimiser.state_dict()["state"]
    for key in ("exp_avg", "exp_avg_sq"):
        gpu, c32, c64 = {}, {}, {}
        for i, n in enumerate(names):
            p32, p64 = trs[torch.float32].P[n], trs[torch.float64].P[n]
            s32, s64 = trs[torch.float32].opt.state.get(p32), trs[torch.float64].opt.state.get(p64)
            if not s64:
                assert i not in st, n
                c64[n] = None
                continue
            gpu[n], c32[n], c64[n] = st[i][key], s32[key], s64[key]
        bad, rows = anchored_report(gpu, c32, c64, floor=1e-4)      # two steps deep: first-step sign flips of ~0 gradients
        print("G5 Adam %s worst ratios:" % key, ["%s %.2f (gpu %.1e cpu %.1e)" % (n, r, eg, ec) for r, n, eg, ec in rows[:4]])
        assert not bad, (key, bad[:10])
