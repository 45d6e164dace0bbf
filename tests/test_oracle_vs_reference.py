"""CPU, build container only: restatement vs the imported reference (skips when /root/reference is absent)."""
from collections import OrderedDict

import pytest
import torch

from oracle import ref_import, restatement as R

pytestmark = pytest.mark.skipif(not ref_import.available(), reason="/root/reference not present (GPU box)")


def test_state_dict_layout_matches_reference():
    net, _ = ref_import.load_reference()
    sd = net.FootprintNetwork(pretrained=False).state_dict()
    spec = R.state_spec()
    assert [k for k in sd] == [s[0] for s in spec]
    assert [tuple(v.shape) for v in sd.values()] == [tuple(s[1]) for s in spec]


def test_forward_backward_and_loss_match_reference():
    net, loss_mod = ref_import.load_reference()
    P, B = R.make_state(tag="vsref")
    batch = R.make_batch(2, 64, 64, tag="vsref")
    m = net.FootprintNetwork(pretrained=False)
    m.load_state_dict({**P, **B})
    m.train()
    out_ref = m(batch["image"])
    l_ref = loss_mod.LossManager((0.1, 100), 0.25)(dict(out_ref), batch)
    l_ref["loss"].backward()
    tr = R.OracleTrainer(P, B)
    out, l = tr.forward_backward(batch)
    for k in out:
        assert torch.equal(out[k], out_ref[k])
    for k in l_ref:
        assert abs(float(l[k]) - float(l_ref[k])) <= 2e-6 * max(1.0, abs(float(l_ref[k])))
    g_ref = dict(m.named_parameters())
    for k, p in tr.P.items():
        if g_ref[k].grad is None:
            assert p.grad is None
        else:
            scale = g_ref[k].grad.abs().max().item() + 1e-30
            assert (p.grad - g_ref[k].grad).abs().max().item() / scale < 1e-4, k
