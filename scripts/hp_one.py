"""GPU box: fp_conv3x3_hp against fp_conv3x3_bf3 and a float64 reference on one shape: accuracy + time.
   python scripts/hp_one.py Cin Cout H W [N] [reps] [mode: fwd|dgrad] [scale_x] [scale_w]"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from footprints_amd import ops, _lib as L      # noqa: E402

C, Co, H, W = (int(v) for v in sys.argv[1:5])
N = int(sys.argv[5]) if len(sys.argv) > 5 else 12
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 20
mode = sys.argv[7] if len(sys.argv) > 7 else "fwd"
sx = float(sys.argv[8]) if len(sys.argv) > 8 else 1.0
sw = float(sys.argv[9]) if len(sys.argv) > 9 else 0.1
torch.manual_seed(1)
x = (torch.randn(N, H, W, C, device="cuda") * sx)
x = x * (torch.rand_like(x) > 0.3)            # post-ReLU-like sparsity
if os.environ.get("HP_HEAVY"):                # heavy-tailed magnitudes: |x| spans 2^-p .. 1 (p uniform in [0, HP_HEAVY])
    x = x * torch.exp2(-torch.rand_like(x) * float(os.environ["HP_HEAVY"]))
w = torch.randn(Co, C, 3, 3, device="cuda") * sw
b = torch.randn(Co, device="cuda") * 0.1
dg = mode == "dgrad"
wt = w if not dg else w.permute(1, 0, 2, 3).contiguous()
wp3 = ops.pack_conv_weight_bf3(wt, torch.empty(ops.packed_weight_elems_bf3(Co, C, 3, False), device="cuda"), False)
slot_w = torch.zeros(ops.amax_elems(), dtype=torch.int32, device="cuda")
wph = ops.pack_conv_weight_hp(wt, torch.empty(ops.packed_weight_elems_hp(Co, C, 3, False), device="cuda"), slot_w, False)
slot_x = ops.amax_f32(x, torch.zeros(ops.amax_elems(), dtype=torch.int32, device="cuda"))
slot_y = torch.zeros(ops.amax_elems(), dtype=torch.int32, device="cuda")
d = ops.make_desc(N, H, W, H, W, C, 0, Co, 3, 1, 1, L.GATHER_DGRAD_REFLECT if dg else L.GATHER_FWD_REFLECT, act=0)
y3 = torch.empty(N, H, W, Co, device="cuda")
yh = torch.empty(N, H, W, Co, device="cuda")
kw = dict(bias=None if dg else b)
ops.conv3x3_bf3(d, x, wp3, y3, **kw)
ops.conv3x3_hp(d, x, wph, yh, slot_x, slot_w, amax_out=slot_y, **kw)
torch.cuda.synchronize()
print("amax: x %.6g (true %.6g)  w %.6g (true %.6g)  y %.6g (true %.6g)" % (ops.amax_value(slot_x), x.abs().max().item(), ops.amax_value(slot_w),
                                                                          w.abs().max().item(), ops.amax_value(slot_y), yh.abs().max().item()))
# float64 reference (reflection padding; the data gradient of a reflect-padded conv = conv_transpose of the padded gradient folded back)
x64, w64 = x.double().permute(0, 3, 1, 2), w.double()
if not dg:
    ref = F.conv2d(F.pad(x64, (1, 1, 1, 1), mode="reflect"), w64, b.double()).permute(0, 2, 3, 1)
else:
    xin = torch.zeros(N, Co, H, W, dtype=torch.float64, device="cuda", requires_grad=True)
    out = F.conv2d(F.pad(xin, (1, 1, 1, 1), mode="reflect"), w64.permute(1, 0, 2, 3).contiguous())
    out.backward(x64)
    ref = xin.grad.permute(0, 2, 3, 1)
den = ref.norm()
print("rel L2 error vs float64: bf16x3 %.3e   fp16-pair %.3e   | max abs / max|ref|: bf16x3 %.3e  fp16-pair %.3e" % (
    ((y3.double() - ref).norm() / den).item(), ((yh.double() - ref).norm() / den).item(),
    ((y3.double() - ref).abs().max() / ref.abs().max()).item(), ((yh.double() - ref).abs().max() / ref.abs().max()).item()))
# fp32 torch conv on the GPU for scale
y32 = F.conv2d(F.pad(x.permute(0, 3, 1, 2), (1, 1, 1, 1), mode="reflect"), w, b).permute(0, 2, 3, 1) if not dg else None
if y32 is not None:
    print("rel L2 error of torch fp32 conv2d (MIOpen): %.3e" % ((y32.double() - ref).norm() / den).item())


def timeit(f):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        f()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


t3 = timeit(lambda: ops.conv3x3_bf3(d, x, wp3, y3, **kw))
th = timeit(lambda: ops.conv3x3_hp(d, x, wph, yh, slot_x, slot_w, amax_out=slot_y, **kw))
th0 = timeit(lambda: ops.conv3x3_hp(d, x, wph, yh, slot_x, slot_w, **kw))
ta = timeit(lambda: ops.amax_f32(x, slot_x))


def cold():
    ops.zero_u32(slot_y)
    ops.conv3x3_hp(d, x, wph, yh, slot_x, slot_w, amax_out=slot_y, **kw)


tz = timeit(lambda: ops.zero_u32(slot_y))
tc = timeit(cold) - tz
fl = 2.0 * N * H * W * C * Co * 9
print("%s %d->%d @%dx%dx%d: bf16x3 %.1f us (%.1f TF/s)  fp16-pair %.1f us (%.1f TF/s; without amax_out %.1f us; slot zeroed before every launch %.1f us)  standalone amax(x) %.1f us" % (
    mode, C, Co, H, W, N, t3, fl / t3 / 1e6, th, fl / th / 1e6, th0, tc, ta))
