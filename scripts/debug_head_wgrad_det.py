"""Reproducer of the packed-fp32 / bf16-MFMA hazard (see profiles/round1_notes.md): run head_wgrad (or any kernel that uses
v_pk_fma_f32) on one stream while conv3x3_bf3 (v_mfma_f32_32x32x16_bf16) runs on another and compare results bitwise.
With the library built without -fno-slp-vectorize the head_wgrad lines report "5 of 5 differ"; FP_LIB=<other .so> selects a build."""
import sys, torch
sys.path.insert(0, "/root/repo")
from footprints_amd import ops, _lib as L
import os
L.LIB_PATH = os.environ.get("FP_LIB", L.LIB_PATH)
torch.manual_seed(0)
N, H, W, C = 12, 192, 640, 32
x = torch.rand(N, H, W, C, device="cuda") - 0.5
dz = (torch.rand(N, H, W, 2, device="cuda") - 0.5) * 1e-3
dw = torch.empty(2, C, 3, 3, device="cuda"); db = torch.empty(2, device="cuda")
# a heavy conv to run concurrently
xin = torch.rand(N, 96, 320, 64, device="cuda"); w = torch.rand(64, 64, 3, 3, device="cuda") * 0.05
wp3 = ops.pack_conv_weight_bf3(w, torch.empty(ops.packed_weight_elems_bf3(64, 64, 3), device="cuda"))
wp = ops.pack_conv_weight(w, torch.empty(ops.packed_weight_elems(64, 64, 3), device="cuda"))
y = torch.empty(N, 96, 320, 64, device="cuda")
d = ops.make_desc(N, 96, 320, 96, 320, 64, 0, 64, 3, 1, 1, L.GATHER_FWD_REFLECT, act=L.ACT_ELU)
side = torch.cuda.Stream()
def run(conc, bf3):
    outs = []
    for it in range(6):
        if conc:
            with torch.cuda.stream(side):
                for _ in range(3):
                    (ops.conv3x3_bf3(d, xin, wp3, y) if bf3 else ops.conv_igemm(d, xin, None, wp, y))
        ops.head_wgrad(x, dz, dw, db)
        torch.cuda.synchronize()
        outs.append(dw.clone())
    return sum(int(not torch.equal(outs[0], o)) for o in outs[1:])
print("alone: %d of 5 differ" % run(False, False))
print("with fp32 tile conv concurrently: %d of 5 differ" % run(True, False))
print("with bf3 tile conv concurrently: %d of 5 differ" % run(True, True))

def victim_test(name, fn, agg):
    outs = []
    for it in range(6):
        with torch.cuda.stream(side):
            for _ in range(3):
                agg()
        r = fn()
        torch.cuda.synchronize()
        outs.append(r.clone())
    print("%-50s %d of 5 differ" % (name, sum(int(not torch.equal(outs[0], o)) for o in outs[1:])))

bf3 = lambda: ops.conv3x3_bf3(d, xin, wp3, y)
big = torch.rand(64 << 20, device="cuda")
victim_test("torch sum (bf3 concurrent)", lambda: big.sum(), bf3)
M = N * H * W
cs = torch.empty(32, device="cuda")
victim_test("colsum (bf3 concurrent)", lambda: ops.colsum(x.view(M, C), cs), bf3)
def hw():
    ops.head_wgrad(x, dz, dw, db)
    return dw
victim_test("head_wgrad (bf3 concurrent)", hw, bf3)
# is it the partial kernel or the reduce?  partials live in the stream workspace
def hw_part():
    ops.head_wgrad(x, dz, dw, db)
    return ops.workspace(1, x.device)[:1 << 20]
victim_test("head_wgrad partials (bf3 concurrent)", hw_part, bf3)
d32 = ops.make_desc(N, 192, 640, 192, 640, 32, 0, 32, 3, 1, 1, L.GATHER_FWD_REFLECT, act=L.ACT_ELU)
x32 = torch.rand(N, 192, 640, 32, device="cuda"); w32 = torch.rand(32, 32, 3, 3, device="cuda") * 0.05
wp32 = ops.pack_conv_weight_bf3(w32, torch.empty(ops.packed_weight_elems_bf3(32, 32, 3), device="cuda"))
y32 = torch.empty(N, 192, 640, 32, device="cuda")
victim_test("head_wgrad (bf3 Nout=32 concurrent)", hw, lambda: ops.conv3x3_bf3(d32, x32, wp32, y32))
# does bf3 itself stay deterministic next to head_wgrad?
def bf3_out():
    ops.conv3x3_bf3(d, xin, wp3, y)
    return y
victim_test("bf3 conv output (head_wgrad concurrent)", bf3_out, lambda: ops.head_wgrad(x, dz, dw, db))

# where do the partials differ?
per = 9 * C * 2 + 2
nblk = 1024
def parts():
    ops.head_wgrad(x, dz, dw, db)
    torch.cuda.synchronize()
    return ops.workspace(1, x.device)[:nblk * per * 4].view(torch.float32).view(nblk, per).clone()
p0 = parts()
with torch.cuda.stream(side):
    for _ in range(3):
        bf3()
p1 = parts()
torch.cuda.synchronize()
diff = (p0 != p1)
print("blocks with differences:", int(diff.any(1).sum()), "of", nblk, " elements:", int(diff.sum()), "of", diff.numel())
bad = diff.any(1).nonzero().flatten()[:8].tolist()
for b in bad:
    idx = diff[b].nonzero().flatten()
    print(" block", b, "n diff", len(idx), "first idx", idx[:6].tolist(), "vals", p0[b][idx[:3]].tolist(), p1[b][idx[:3]].tolist())
