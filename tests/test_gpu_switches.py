"""GPU: the A/B switches documented in DESIGN.md select working code paths.  Each case runs four training steps of a small batch in a
fresh process (the switches are read at import / first use) and must reproduce its reference run's losses -- the default build (exact bf16x3
split operands) or the opt-in fp16-pair format: bit-identically where only the schedule changes, within 1e-4 where the arithmetic of the
convolutions changes (fp16 pairs vs the exact bf16 split vs fp32 MFMA)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROG = """
import json, torch
from footprints_amd.model_manager import ModelManager
from footprints_amd.training.train import SEED, TrainStep, synthetic_batch
torch.manual_seed(SEED)
mm = ModelManager(use_cuda=True)
ts = TrainStep(mm.model, mm.optimiser)
batch = synthetic_batch(2, 128, 192, "cuda")
out = []
for _ in range(4):
    ts(batch)
    out.append([float(v) for v in ts.losses.cpu()])
print("LOSSES " + json.dumps(out))
"""


def run(env_extra):
    env = dict(os.environ)
    for k in ("FP_OPERANDS", "FP_HP", "FP_BN_BWD_EPI", "FP_BF3_IGEMM", "FP_NO_BF3", "FP_SERIAL", "FP_NO_PHASE", "FP_DS_AUX", "FP_PLAN", "FP_NO_WBF3", "FP_WGRAD_PF", "FP_BN_EPI", "FP_ADAM_STAGED"):
        env.pop(k, None)
    env.update(env_extra)
    r = subprocess.run([sys.executable, "-c", PROG], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("LOSSES ")][-1]
    return json.loads(line[len("LOSSES "):])


PAIR = {"FP_OPERANDS": "fp16_pair"}            # the opt-in operand format (footprints_amd/_format.py); the default is the exact bf16x3 split
_BASELINES = {}


def baseline(env):
    key = tuple(sorted(env.items()))
    if key not in _BASELINES:
        _BASELINES[key] = run(env)
    return _BASELINES[key]


# (switch, the run it is compared with, bit-identical?)
@pytest.mark.parametrize("env,ref,exact", [
    ({"FP_SERIAL": "1"}, {}, False),           # one stream instead of five (the downsample branch then runs in line: another accumulation order)
    ({"FP_SERIAL": "1", "FP_DS_AUX": "0"}, {}, False),
    ({"FP_PLAN": "0"}, {}, True),              # launches issued from Python instead of the recorded plan (step 4 is a replay by default)
    ({"FP_DS_AUX": "0"}, {}, False),           # downsample branch in line: its data gradient accumulates after conv1's instead of before
    ({"FP_NO_PHASE": "1"}, {}, False),         # fused nearest-x2 gather instead of the phase decomposition
    ({"FP_NO_BF3": "1"}, {}, False),           # fp32-MFMA kernels everywhere
    ({"FP_BN_EPI": "0"}, {}, False),           # BatchNorm statistics by a pass over the activation instead of the conv epilogue's partials
    ({"FP_ADAM_STAGED": "1"}, {}, True),       # a piece of the Adam update per stage under the backward pass instead of one launch after it (element-wise)
    ({"FP_WGRAD_PF": "0"}, {}, True),          # exact split: third-generation weight-gradient kernel instead of the ring (round 5): same sums, bit for bit
    ({"FP_BF3_IGEMM": "0"}, {}, False),        # stride-2 3x3 / 1x1 convolutions on fp32 MFMA instead of exactly split bf16x3 operands (round 5)
    ({"FP_BN_BWD_EPI": "0"}, {}, False),       # BatchNorm backward sums by their own reduction pass instead of the data gradient's epilogue
    ({"FP_HP": "0"}, {}, True),                # the legacy spelling of the default format
    (PAIR, {}, False),                         # scaled fp16 pairs (opt-in) against the exact split (default): the 1e-4 contract
    ({"FP_HP": "1"}, PAIR, True),              # the legacy spelling of the opt-in format
    ({**PAIR, "FP_NO_PHASE": "1"}, PAIR, False),
    ({**PAIR, "FP_PLAN": "0"}, PAIR, True),
    ({**PAIR, "FP_SERIAL": "1"}, PAIR, False),
    ({**PAIR, "FP_BN_EPI": "0"}, PAIR, False),
    ({**PAIR, "FP_WGRAD_PF": "0"}, PAIR, True),   # third-generation weight-gradient kernel: same products and summation order, other load schedule
    ({**PAIR, "FP_WGRAD_PF": "3"}, PAIR, True),   # prefetch ring of depth three
    ({**PAIR, "FP_ADAM_STAGED": "1"}, PAIR, True),
])
def test_switch_reproduces_the_reference_run(env, ref, exact):
    got, want = run(env), baseline(ref)
    assert len(got) == len(want) == 4
    for step, (a, b) in enumerate(zip(got, want)):
        for x, y in zip(a, b):
            if exact:
                assert x == y, (env, step, x, y)
            else:
                assert abs(x - y) <= 1e-4 * max(abs(y), 1e-3), (env, step, x, y)


def test_network_and_trainer_suites_in_the_optin_format():
    """the session itself runs in the default (exact) operand format; the network- and trainer-level tests once more in a child pytest with
    FP_OPERANDS=fp16_pair, so that the driver's `pytest -m gpu` exercises the opt-in format end to end as well (golden vectors G2 / G3 / G5, drop-in
    surface, TrainStep = drop-in path, launch plans, eval fast path; the full-size parity cases cover both formats themselves)"""
    env = dict(os.environ)
    for k in ("FP_OPERANDS", "FP_HP"):
        env.pop(k, None)
    env.update(PAIR)
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_network.py", "tests/test_gpu_trainer.py", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, (r.stdout or "")[-3000:] + (r.stderr or "")[-1000:]
    assert " passed" in r.stdout, r.stdout[-500:]
