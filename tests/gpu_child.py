"""One train-mode forward + loss + backward of the HIP engine in a CHOSEN operand format (test infrastructure).

The operand format (footprints_amd/_format.py: exact bf16x3 split = default, scaled fp16 pairs = opt-in) is fixed when footprints_amd.engine is
imported, so the format the test session itself runs in executes in-process and the other one in a child process:
`gpu_step(P, B, cpu_batch, fmt)` hides the difference and returns plain CPU tensors either way --

    out        {scale: [B,4,H,W]}            the network outputs (network.py:26-30)
    losses     {key: float}                  the 21 scalars of LossManager (training/losses.py:31-92)
    grads      {name: tensor | None}         d loss / d parameter
    decisions  {"relu": [33 bool NCHW masks], the engine's discrete decisions: ReLU masks (stem, then (bn1's ReLU, block output) per BasicBlock) and the
                "pool": int64 [N,64,OH,OW]}  max-pool's winning window positions -- what tests/parity.py imposes on the float64 oracle to separate
                                             decisions from arithmetic
    state      {key: tensor}                 state_dict after the step (BatchNorm running statistics)
    taps       {name: tensor}                (tap_block = i) g / d out / z2 / out of encoder block i, see test_gpu_parity_fullsize.py
    format     str                           what the engine that ran really used
"""
import os
import subprocess
import sys
import tempfile
from collections import OrderedDict

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _nchw_mask(t):
    return (t > 0).permute(0, 3, 1, 2).contiguous().cpu()


def engine_decisions(eng):
    """the discrete decisions of the engine's last train-mode forward (call before backward): {"relu": 33 bool NCHW masks, "pool": winners}"""
    S = eng.saved
    relu = [_nchw_mask(S["feats"][0])]
    for Bk in S["blocks"]:
        relu += [_nchw_mask(Bk["a1"]), _nchw_mask(Bk["out"])]
    f0 = S["feats"][0]
    hp, wp = (f0.shape[1] + 1) // 2, (f0.shape[2] + 1) // 2
    am = eng._bufs["pool.argmax"][:f0.shape[0] * hp * wp * 64].view(f0.shape[0], hp, wp, 64)          # uint8 ky * 3 + kx (csrc/bn_pool.hip maxpool_fwd_kernel)
    return {"relu": relu, "pool": am.permute(0, 3, 1, 2).contiguous().cpu().to(torch.int64)}


def _step_here(P, B, cpu_batch, tap_block=None):
    from footprints_amd import FootprintNetwork
    from footprints_amd._format import operand_format
    from footprints_amd.training.losses import LossManager
    model = FootprintNetwork(pretrained=False)
    model.load_state_dict({**P, **B})
    model.cuda().train()
    eng = model.engine()
    taps = {}
    if tap_block is not None:
        def hook(i, d):
            if i == tap_block:
                taps.update(dout=d["dout"].detach().cpu(), g=d["g"].detach().cpu(), z2=d["B"]["z2"].detach().cpu(), out=d["B"]["out"].detach().cpu())
        eng.debug_hook = hook
    batch = {k: v.cuda() for k, v in cpu_batch.items()}
    out = model(batch["image"])
    torch.cuda.synchronize()
    decisions = engine_decisions(eng)
    losses = LossManager((0.1, 100), 0.25, compute_viz=False)(out, batch)
    losses["loss"].backward()
    torch.cuda.synchronize()
    res = {"out": OrderedDict((k, v.detach().cpu()) for k, v in out.items()),
           "losses": OrderedDict((k, float(v)) for k, v in losses.items()),
           "grads": OrderedDict((n, None if p.grad is None else p.grad.detach().cpu()) for n, p in model.named_parameters()),
           "decisions": decisions,
           "state": OrderedDict((k, v.detach().cpu()) for k, v in model.state_dict().items()),
           "taps": taps, "format": operand_format()}
    eng.debug_hook = None
    del model, eng, out, losses, batch
    torch.cuda.empty_cache()
    return res


def gpu_step(P, B, cpu_batch, fmt, tap_block=None):
    from footprints_amd._format import format_env, operand_format
    if fmt == operand_format():
        return _step_here(P, B, cpu_batch, tap_block)
    with tempfile.TemporaryDirectory(prefix="fp_gpu_child_") as tmp:
        src, dst = os.path.join(tmp, "in.pt"), os.path.join(tmp, "out.pt")
        torch.save({"P": P, "B": B, "batch": cpu_batch, "tap_block": tap_block}, src)
        env = dict(os.environ)
        env.update(format_env(fmt))
        r = subprocess.run([sys.executable, "-m", "tests.gpu_child", src, dst], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
        assert r.returncode == 0, "child engine (%s) failed:\n%s" % (fmt, (r.stderr or r.stdout)[-3000:])
        res = torch.load(dst, weights_only=False)
    assert res["format"] == fmt, (res["format"], fmt)
    return res


if __name__ == "__main__":
    job = torch.load(sys.argv[1], weights_only=False)
    torch.save(_step_here(job["P"], job["B"], job["batch"], job["tap_block"]), sys.argv[2])
