#!/usr/bin/env python
"""Per-workgroup phase stamps of the fused weight-gradient + Adam kernel (tests/native/test_gemm dwx ... with DWX_DUMP=prefix):
   [virtual workgroup][8] u64 = 100-MHz times of: start, loads issued, slice 0 landed, K walk done, parked, stores issued, stores
   acknowledged; HW_ID | XCC_ID << 32.  Groups the tiles by the hardware slot they ran in (a persistent workgroup keeps its slot)
   and prints, per slot, tiles, busy time, and the gaps between consecutive tiles.    usage: tools/dw_stamps.py FILE.bin"""
import sys
import numpy as np

a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8)
a = a[a[:, 0] != 0]
t0 = a[:, 0].min()
st = (a[:, 0] - t0) * 0.01
en = (a[:, 6] - t0) * 0.01
hw = a[:, 7]
print("%d tiles, span %.1f us, mean life %.2f us" % (len(a), en.max(), (en - st).mean()))
# hardware slot of thread 0's wave: XCC | SE | CU | SIMD | wave slot  (HW_ID bits: wave 3:0, simd 5:4, cu 11:8, sh 12, se 15:13)
key = (hw >> 32) << 32 | (hw & 0xffff)
slots = {}
for k, s, e, i in zip(key, st, en, range(len(a))):
    slots.setdefault(int(k), []).append((s, e, i))
ntile = np.array([len(v) for v in slots.values()])
print("%d distinct slots; tiles per slot: min %d max %d; histogram %s" % (len(slots), ntile.min(), ntile.max(), np.bincount(ntile).tolist()))
gaps, busy, last_end = [], [], []
for v in slots.values():
    v.sort()
    busy.append(sum(e - s for s, e, _ in v))
    last_end.append(v[-1][1])
    for (s0, e0, _), (s1, e1, _) in zip(v, v[1:]):
        gaps.append(s1 - e0)
gaps = np.array(gaps)
print("gap between consecutive tiles of a slot: n %d mean %.2f median %.2f p90 %.2f max %.2f us" % (len(gaps), gaps.mean(), np.median(gaps), np.percentile(gaps, 90), gaps.max()))
print("slot busy time: mean %.1f us; last tile ends: mean %.1f, p10 %.1f, p50 %.1f, p90 %.1f, max %.1f us" % (np.mean(busy), np.mean(last_end), *np.percentile(last_end, [10, 50, 90]), np.max(last_end)))
ph = (a[:, 1:7] - a[:, 0:6]).astype(np.float64) * 0.01
names = ["issue", "first slice", "K walk", "park", "adam+stores", "store ack"]
for lo, hi in ((0, 25), (25, 50), (50, 75), (75, 1e9)):
    m = (st >= lo) & (st < hi)
    if m.any():
        print("tiles started in [%g, %g) us: %4d | " % (lo, hi, m.sum()) + " | ".join("%s %.2f" % (n, x) for n, x in zip(names, ph[m].mean(0))) + " | life %.2f" % (en[m] - st[m]).mean())
