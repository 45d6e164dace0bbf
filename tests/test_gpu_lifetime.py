"""GPU: buffer lifetimes under launches still in flight (round 6, VERDICT r5 "Next" 9).

Round 5's whole-suite failure (profiles/round5_notes.md section 7) was a host-side free racing kernels: a split-K workspace was dropped while
the previous convolution was still reducing its partial sums in it, the caching allocator handed the block out again, and the late writes landed
on somebody else's data.  It was only seen through the garbage collector's timing under one exact test selection.  This test makes the class
deterministic:

  * every place that lets go of a device buffer goes through `ops.release` (workspace growth, arena re-allocation); a hook keeps each released
    block alive, fills it with a sentinel on an otherwise idle stream THE MOMENT the host lets go of it -- exactly what a new owner would do --
    and, after the run, checks that the sentinel is intact: a kernel that was still using the block would have written into it;
  * a spin kernel (`torch.cuda._sleep`) in front of every step keeps the launch queue deep, so "still in flight when the host gets there" does
    not depend on timing;
  * the steps run at four pyramid sizes (the tiny ones are the r5 repro: every convolution a first touch and a split-K grid) in three orders,
    each on a fresh engine, and their outputs and gradients must be bit-identical to the same steps run one at a time with a device
    synchronize in between (same kernels, same order => same bits, whatever the allocator did);
  * the detector has teeth: with the wait in `ops.release` switched off the same run damages the sentinel (asserted).
"""
import random
from collections import OrderedDict

import pytest
import torch

pytestmark = pytest.mark.gpu

SIZES = [(1, 64, 64), (2, 64, 96), (1, 96, 128), (2, 128, 160)]          # (batch, H, W): growing and shrinking pyramids (the tiny ones: every conv a split-K grid)
SENTINEL = 0xA5
SPIN = 4e8               # cycles of torch.cuda._sleep in front of every step (~0.2 s): the host finishes issuing a step long before the device starts it, even on a loaded host
_BATCHES = {}


def _state():
    """parameters and buffers resident on the device: re-loading them before every step is then a set of asynchronous device copies (a
    pageable host copy would drain the launch queue the test wants deep)"""
    from oracle import restatement as R
    P, B = R.make_state(tag="life")
    return OrderedDict((k, v.cuda()) for k, v in P.items()), OrderedDict((k, v.cuda()) for k, v in B.items())


def _batch(size):
    from oracle import restatement as R
    b, h, w = size
    if size not in _BATCHES:
        _BATCHES[size] = {k: v.cuda() for k, v in R.make_batch(b, h, w, tag="life%dx%dx%d" % size).items()}
    return _BATCHES[size]


def _step(model, lm, P, B, batch, spin, reload=True):
    """(reload: the state again -- weights dirty -> repack on the side stream, lazily packed layouts -> first touches;) a spin kernel; one
    train-mode forward + loss + backward.  No synchronize anywhere: the caller decides when the device drains."""
    if reload:
        model.load_state_dict({**P, **B})
    model.train()
    if spin:
        torch.cuda._sleep(int(spin))                     # the launches below queue up behind this: the host runs ahead of the device
    out = model(batch["image"])
    losses = lm(out, batch)
    model.zero_grad()
    losses["loss"].backward()
    res = OrderedDict(("out." + k, v.detach().clone()) for k, v in out.items())
    res.update(("grad." + n, p.grad.detach().clone()) for n, p in model.named_parameters() if p.grad is not None)
    return res


class _Sentinels:
    """ops.release hook: own every released block, fill it at once from an idle stream, verify later"""

    def __init__(self):
        self.stream = torch.cuda.Stream()
        self.held = []

    def __call__(self, old, what):
        flat = old.view(-1).view(torch.uint8)
        with torch.cuda.stream(self.stream):
            flat.fill_(SENTINEL)
        self.held.append((what, old, flat))

    def damaged(self):
        torch.cuda.synchronize()
        bad = []
        for what, _, flat in self.held:
            n = int((flat != SENTINEL).sum())
            if n:
                bad.append((what, n, flat.numel()))
        return bad


def _run(orders, spin, reload_every=3):
    """per order: a fresh engine; the state loaded behind a spin kernel (work pending on the stream while the engine is built: the start-up
    race of Engine._flatten, see its comment); two settled steps at the smallest size; then the order's sizes back to back -- a spin kernel in
    front of each, no synchronize in between, the state re-loaded every `reload_every`-th step -- so that every growth of an arena buffer or
    a workspace happens while the previous size's launches are still queued."""
    from footprints_amd import FootprintNetwork, ops
    from footprints_amd.training.losses import LossManager
    P, B = _state()
    lm = LossManager((0.1, 100), 0.25, compute_viz=False)
    hook = _Sentinels()
    ops._release_hooks.append(hook)
    results = []
    small = min(SIZES, key=lambda s: s[0] * s[1] * s[2])
    try:
        for order in orders:
            ops._workspaces.clear()                      # a fresh engine AND fresh scratch: every repetition grows its buffers again
            model = FootprintNetwork(pretrained=False).cuda()
            results.append((small, _step(model, lm, P, B, _batch(small), spin)))           # engine built with the spin kernel pending
            results.append((small, _step(model, lm, P, B, _batch(small), 0, reload=False)))
            torch.cuda.synchronize()                     # settled: no first touch, no table rebuild left at this size
            for i, size in enumerate(order):
                results.append((size, _step(model, lm, P, B, _batch(size), spin, reload=(i % reload_every == reload_every - 1))))
            torch.cuda.synchronize()
            del model
    finally:
        ops._release_hooks.remove(hook)
    return results, hook


def _orders():
    rng = random.Random(6)
    asc = sorted(SIZES, key=lambda s: s[0] * s[1] * s[2])
    o2, o3 = SIZES[:], SIZES[:]
    rng.shuffle(o2)
    rng.shuffle(o3)
    return [asc + asc[::-1], o2 + o3, o3[::-1] + asc]


def test_no_buffer_is_released_under_launches_still_in_flight():
    from footprints_amd import FootprintNetwork
    from footprints_amd.training.losses import LossManager
    P, B = _state()
    lm = LossManager((0.1, 100), 0.25, compute_viz=False)
    # calm reference: one size at a time on its own engine, device drained around every step (nothing updates the weights: a size's outputs
    # and gradients are the same at every repetition)
    ref = {}
    for size in SIZES:
        model = FootprintNetwork(pretrained=False).cuda()
        torch.cuda.synchronize()
        ref[size] = _step(model, lm, P, B, _batch(size), spin=0)
        torch.cuda.synchronize()
        del model
    results, hook = _run(_orders(), spin=SPIN)
    print("\n[lifetime] released: %s" % sorted({(w, f.numel()) for w, _, f in hook.held})[:60])
    assert len(hook.held) >= 8, "the stress run must actually release buffers (got %d)" % len(hook.held)
    bad = hook.damaged()
    assert not bad, "blocks written to AFTER the host released them (what, bytes damaged, bytes): %s" % bad[:6]
    wrong = []
    for i, (size, res) in enumerate(results):
        for k, v in res.items():
            if not torch.equal(v, ref[size][k]):
                wrong.append((i, size, k, float("%.3e" % float((v - ref[size][k]).abs().max())), float("%.3e" % float(ref[size][k].abs().max()))))
    assert not wrong, "%d tensors differ from the calm run (step, size, tensor, max |d|, max |ref|): %s" % (len(wrong), wrong[:12])
    print("[lifetime] %d steps, %d released blocks (%s), all sentinels intact, all results bit-identical to the calm run" % (
        len(results), len(hook.held), sorted({w.split(":")[0] for w, _, _ in hook.held})))


def test_the_detector_fires_without_the_wait():
    """negative control: the same run with ops.release's device wait switched off must leave damaged sentinels -- launches queued behind the spin
    kernels write into blocks the host has already given away (the ascending order: every step grows what the queued one is still using)"""
    from footprints_amd import ops
    was = ops._RELEASE_SYNC
    ops._RELEASE_SYNC = False
    try:
        _, hook = _run(_orders()[:1], spin=SPIN, reload_every=10 ** 6)
    finally:
        ops._RELEASE_SYNC = was
        torch.cuda.synchronize()
        ops._workspaces.clear()                          # nothing of this run is reused
    bad = hook.damaged()
    print("\n[lifetime, wait off] %d released blocks, %d damaged: %s" % (len(hook.held), len(bad), bad[:4]))
    assert bad, "with the wait off, in-flight launches must have written into released blocks -- the detector saw nothing"


def test_train_step_plan_record_and_replay_behind_a_deep_queue():
    """the bench / trainer path: TrainStep issues two eager steps, records a launch plan on the third and replays it from C afterwards
    (footprints_amd/training/train.py, csrc/plan.cpp) -- with a spin kernel in front of EVERY step, i.e. the engine is built, the plan is
    recorded and the first replays run while earlier work is still queued.  Seven steps with Adam must leave the same 21 losses per step and
    the same parameters, bit for bit, as the same seven steps with the device drained around each."""
    from footprints_amd.model_manager import ModelManager
    from footprints_amd.training.train import TrainStep
    P, B = _state()
    size = (2, 96, 128)
    batch = _batch(size)

    def run(spin, drain):
        mm = ModelManager(use_cuda=True, learning_rate=1e-4)
        mm.model.load_state_dict({**P, **B})
        if spin:
            torch.cuda._sleep(int(spin))
        ts = TrainStep(mm.model, mm.optimiser)
        losses = []
        for _ in range(7):
            if spin:
                torch.cuda._sleep(int(spin))
            losses.append(ts(batch).clone())
            if drain:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        sd = OrderedDict((k, v.detach().clone()) for k, v in mm.model.state_dict().items())
        used_plan = bool(getattr(ts, "use_plan", False))
        del ts, mm
        return losses, sd, used_plan

    calm_l, calm_sd, _ = run(0, True)
    deep_l, deep_sd, used_plan = run(SPIN / 4, False)
    assert used_plan, "TrainStep is expected to record and replay a launch plan by default"
    for i, (a, b) in enumerate(zip(calm_l, deep_l)):
        assert torch.equal(a, b), "losses of step %d differ behind a deep queue: max |d| %.3e" % (i, float((a - b).abs().max()))
    bad = [k for k in calm_sd if not torch.equal(calm_sd[k], deep_sd[k])]
    assert not bad, "%d state tensors differ after seven steps behind a deep queue: %s" % (len(bad), bad[:6])
