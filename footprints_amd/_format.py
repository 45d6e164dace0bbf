"""Operand format of the split-operand convolution kernels -- one place that reads the environment (no torch, no HIP: bench.py and the tests
import it before anything else).

  exact      (default)  every fp32 operand of a 3x3 stride-1 convolution (forward, data- and weight-gradient) is split EXACTLY into three
             bf16 terms (x = h + m + l, 8 + 8 + 8 significant bits) and six bf16 MFMA products are accumulated in fp32; every other
             convolution runs on native fp32 MFMA.  No operand bit of the reference's fp32 arithmetic (training/train.py:150-156) is dropped.
  fp16_pair  opt-in fast mode: operands scaled by a per-tensor power of two and split into two fp16 terms (22 significant bits, three
             products); also used for the stem, the stride-2 3x3 and the 1x1 convolutions.  Meets the 1e-4 output contract, is NOT
             fp32-faithful -- like TF32, it has to be asked for.

FP_OPERANDS=exact|fp16_pair selects; the legacy switch FP_HP=1 / FP_HP=0 (rounds 2-4, where fp16 pairs were the default) is still read
when FP_OPERANDS is absent."""
import os

FORMATS = ("exact", "fp16_pair")
DTYPE_LABEL = {
    "exact": "f32 (tensors, accumulation; 3x3 stride-1 conv operands as exact bf16x3 splits: 24 significant bits kept, 6 MFMA products)",
    "fp16_pair": "f32 tensors and accumulation; conv operands as scaled fp16 pairs (22 significant bits, 3 MFMA products) -- opt-in, below fp32",
}


def operand_format(env=None):
    env = os.environ if env is None else env
    v = env.get("FP_OPERANDS")
    if v is not None:
        v = v.strip().lower()
        if v not in FORMATS:
            raise ValueError("FP_OPERANDS=%r: expected one of %s" % (v, FORMATS))
        return v
    hp = env.get("FP_HP")
    if hp is not None:
        return "fp16_pair" if bool(int(hp)) else "exact"
    return "exact"


def format_env(fmt):
    """environment entries that select `fmt` in a child process (both spellings, so that a stale FP_HP of the parent cannot win)"""
    if fmt not in FORMATS:
        raise ValueError(fmt)
    return {"FP_OPERANDS": fmt, "FP_HP": "1" if fmt == "fp16_pair" else "0"}
