"""GPU box: per-tap cycle stamps (s_memtime) of wave 0 of three workgroups of the bf16x3 tile kernel (instrumented build
scripts/ubench/bin/lib_ts.so, see profiles/round1_notes.md)."""
import sys, os, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from footprints_amd import ops, _lib as L
L.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", "lib_ts.so")
N, H, W, C = 12, 96, 320, 64
x = torch.rand(N, H, W, C, device="cuda") - 0.5
w = (torch.rand(C, C, 3, 3, device="cuda") - 0.5) * 0.1
b = torch.zeros(C, device="cuda"); y = torch.empty(N, H, W, C, device="cuda")
d = ops.make_desc(N, H, W, H, W, C, 0, C, 3, 1, 1, L.GATHER_FWD_REFLECT, act=L.ACT_ELU)
wp3 = ops.pack_conv_weight_bf3(w, torch.empty(ops.packed_weight_elems_bf3(C, C, 3), device="cuda"))
for _ in range(3):
    ops.conv3x3_bf3(d, x, wp3, y, bias=b)
torch.cuda.synchronize()
lib = L.load()
buf = (ctypes.c_longlong * 8192)()
lib.fp_dbg_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
print("rc", lib.fp_dbg_read(buf, 8192))
for base, name in ((0, "wg0"), (1024, "wg1500"), (2048, "wg2800")):
    e = [buf[base + 900 + i] for i in range(6)]
    print("%s: entry->setup done %d | prologue loads issued+halo stored %d | barrier %d | main loop %d | epilogue %d | total %d cycles" % (
        name, e[1] - e[0], e[2] - e[1], e[3] - e[2], e[4] - e[3], e[5] - e[4], e[5] - e[0]))
