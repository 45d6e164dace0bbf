"""summarise FP_W3_STAMPS dumps: start skew, phase durations, placement"""
import sys
import numpy as np
d = np.loadtxt(sys.argv[1], dtype=np.float64)
t0 = d[:, 1].min()
start, lstart, lend, end = (d[:, i] - t0 for i in (1, 2, 3, 4))
print("workgroups %d  kernel span %.0f cycles" % (len(d), end.max()))
print("start   : min %.0f  p50 %.0f  p90 %.0f  max %.0f" % (start.min(), np.median(start), np.percentile(start, 90), start.max()))
print("prologue: p50 %.0f  max %.0f" % (np.median(lstart - start), (lstart - start).max()))
print("loop    : p10 %.0f  p50 %.0f  p90 %.0f  max %.0f" % (np.percentile(lend - lstart, 10), np.median(lend - lstart), np.percentile(lend - lstart, 90), (lend - lstart).max()))
print("epilogue: p50 %.0f  max %.0f" % (np.median(end - lend), (end - lend).max()))
print("end     : min %.0f  p50 %.0f  max %.0f" % (end.min(), np.median(end), end.max()))
hw = d[:, 5].astype(np.int64)
xcc = d[:, 6].astype(np.int64) & 0xf
cu = (hw >> 8) & 0xf
se = (hw >> 13) & 0x7
key = xcc * 10000 + se * 100 + cu
u, cnt = np.unique(key, return_counts=True)
print("distinct (xcc, se, cu): %d; workgroups per CU: min %d max %d; histogram %s" % (len(u), cnt.min(), cnt.max(), dict(zip(*np.unique(cnt, return_counts=True)))))
late = start > 0.2 * end.max()
print("workgroups starting after 20 %% of the span: %d" % late.sum())
