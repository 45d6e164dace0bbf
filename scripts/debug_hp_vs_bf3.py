"""GPU box: where does the fp16-pair path leave the exact-split path in the backward pass of the natural-statistics case?
    python scripts/debug_hp_vs_bf3.py           runs both operand formats in child processes (same box), then compares every tensor the
                                                engine's debug hook sees per encoder block (dout, g, dz2, da1, dz1, dnext) + dF + parameter grads
"""
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def dump(path):
    from oracle.cpu_threads import effective_cores
    from tests.parity import oracle_grads, tie_free_batch
    from tests.test_gpu_parity_fullsize import _natural_batch, _wide_range_state
    torch.set_num_threads(min(effective_cores(), 32))
    P, B = _wide_range_state()
    batch = _natural_batch(4, 192, 640)
    cache = "/tmp/natural_tiefree.pt"
    if os.path.exists(cache):
        batch = torch.load(cache)
    else:
        _, _, _, _, batch = oracle_grads(P, B, batch, torch.float64, fix_batch=lambda b, o: tie_free_batch(b, o)[0])
        torch.save(batch, cache)
    from footprints_amd import FootprintNetwork
    from footprints_amd.training.losses import LossManager
    model = FootprintNetwork(pretrained=False)
    model.load_state_dict({**P, **B})
    model.cuda().train()
    gb = {k: v.cuda() for k, v in batch.items()}
    eng = model.engine()
    keep = {}

    def hook(i, d):
        torch.cuda.synchronize()
        for k in ("dout", "g", "dz2", "da1", "dz1", "dnext"):
            if d.get(k) is not None:
                keep["blk%02d.%s" % (i, k)] = d[k].detach().float().cpu().clone()
    eng.debug_hook = hook
    out = model(gb["image"])
    for v in out.values():
        v.retain_grad()
    losses = LossManager((0.1, 100), 0.25, compute_viz=False)(out, gb)
    losses["loss"].backward()
    torch.cuda.synchronize()
    for k, v in out.items():
        keep["out." + k] = v.detach().cpu()
        keep["dout." + k] = v.grad.detach().cpu()
    for i, f in enumerate(eng.saved["feats"]):
        keep["dF%d" % i] = eng._bufs["dF%d" % i][:f.numel()].view(f.shape).detach().cpu().clone()
    for n, p in model.named_parameters():
        if p.grad is not None:
            keep["grad." + n] = p.grad.detach().cpu().clone()
    torch.save(keep, path)


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "dump":
        return dump(sys.argv[2])
    for mode, env in (("hp", {}), ("bf3", {"FP_HP": "0"})):
        subprocess.run([sys.executable, os.path.abspath(__file__), "dump", "/tmp/dbg_%s.pt" % mode], env=dict(os.environ, **env), check=True)
    a, b = torch.load("/tmp/dbg_hp.pt"), torch.load("/tmp/dbg_bf3.pt")
    rows = []
    for k in a:
        if k not in b:
            continue
        x, y = a[k].double(), b[k].double()
        rel = ((x - y).norm() / y.norm().clamp_min(1e-300)).item()
        mx = ((x - y).abs().max() / y.abs().max().clamp_min(1e-300)).item()
        rows.append((k, rel, mx, y.abs().max().item(), y.abs().median().item()))
    print("%-60s %10s %10s %10s %10s" % ("tensor", "relL2", "max/max", "max|ref|", "med|ref|"))
    for k, rel, mx, m, md in rows:
        if k.startswith("grad.") and rel < 2e-5:
            continue
        print("%-60s %10.2e %10.2e %10.2e %10.2e" % (k, rel, mx, m, md))


if __name__ == "__main__":
    main()
