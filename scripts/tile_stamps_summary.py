"""summarise a conv3x3_tile_bf3 stamp dump (diagnostics build: scripts/build_variant.sh stamps conv3x3_tile_bf3.hip -DFP_TILE_STAMPS;
FP_LIB=scripts/ubench/bin/lib_stamps.so FP_TILE_STAMPS_FILE=out.txt python scripts/tile_one.py ...).  One line per wave:
id, st[0..15] = start, loop start, taps-done of chunk 0..5, barrier-done of chunk 0..5, end, HW id."""
import sys
import numpy as np
d = np.loadtxt(sys.argv[1], dtype=np.float64)
st = d[:, 1:]
t0 = st[:, 0].min()
q = lambda v: "p10 %.0f p50 %.0f p90 %.0f max %.0f" % (np.percentile(v, 10), np.median(v), np.percentile(v, 90), v.max())
print("waves %d  kernel span %.0f cycles (100 MHz-independent shader clock)" % (len(d), st[:, 14].max() - t0))
print("start    :", q(st[:, 0] - t0))
print("prologue :", q(st[:, 1] - st[:, 0]))
nch = min(4, int((st[:, 2:8] > 0).sum(axis=1).max()))
prev = st[:, 1]
for k in range(nch):
    ok = st[:, 2 + k] > 0
    print("chunk %d taps   :" % k, q((st[:, 2 + k] - prev)[ok]))
    print("chunk %d restage:" % k, q((st[:, 8 + k] - st[:, 2 + k])[ok]))
    prev = st[:, 8 + k]
last = np.max(st[:, 8:12], axis=1)
print("epilogue :", q(st[:, 14] - last))
print("lifetime :", q(st[:, 14] - st[:, 0]))
hw = st[:, 15].astype(np.int64)
cu = (hw >> 8) & 0xf; se = (hw >> 13) & 0x7; simd = (hw >> 4) & 0x3
print("waves per simd id:", dict(zip(*np.unique(simd, return_counts=True))))
rt0, rt1 = st[:, 12], st[:, 13]
span_us = (rt1.max() - rt0.min()) / 100.0
f = (st[:, 14] - st[:, 0]) / np.maximum(rt1 - rt0, 1) * 100.0      # MHz
print("kernel span (100 MHz clock): %.1f us; shader clock while resident: p10 %.0f p50 %.0f p90 %.0f MHz" % (span_us, np.percentile(f, 10), np.median(f), np.percentile(f, 90)))
life_us = (rt1 - rt0) / 100.0
print("wave lifetime: p50 %.1f us; mean resident workgroups per CU = %.2f" % (np.median(life_us), life_us.sum() / 4 / 256 / span_us))
