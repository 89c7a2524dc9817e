#!/usr/bin/env python3
"""Developer tool: per-wave timeline of one k_dyn_stream launch from a -DQS_STAMPS build (QD_STAMPS_FILE).  python scripts/stamps_report.py file [waves_per_wg]"""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8)
wpw = int(sys.argv[2]) if len(sys.argv) > 2 else 5
a = a[a[:, 0] > 0]
t0, t1, t2, hw, xcc = (a[:, k].astype(np.int64) for k in (0, 1, 2, 4, 5))
base = t0.min()
us = lambda t: (t - base) / 100.0          # s_memrealtime: 100 MHz
print(f"waves {len(a)}  launch span {us(t2.max()):.2f} us")
print(f"start   : min {us(t0).min():.2f} p50 {np.median(us(t0)):.2f} p90 {np.percentile(us(t0), 90):.2f} max {us(t0).max():.2f}")
if a.shape[1] > 3 and (a[:, 3] > 0).any():
    t3 = a[:, 3].astype(np.int64)
    print(f"stamp1 - stamp0 p50 {np.median((t1 - t0) / 100.0):.2f}  stamp3 - stamp1 p50 {np.median((t3 - t1) / 100.0):.2f}  stamp2 - stamp3 p50 {np.median((t2 - t3) / 100.0):.2f} us")
print(f"prologue: p10 {np.percentile((t1 - t0) / 100.0, 10):.2f} p50 {np.median((t1 - t0) / 100.0):.2f} p90 {np.percentile((t1 - t0) / 100.0, 90):.2f} us")
print(f"lifetime: p10 {np.percentile((t2 - t0) / 100.0, 10):.2f} p50 {np.median((t2 - t0) / 100.0):.2f} p90 {np.percentile((t2 - t0) / 100.0, 90):.2f} max {((t2 - t0) / 100.0).max():.2f} us")
print(f"end     : p10 {np.percentile(us(t2), 10):.2f} p50 {np.median(us(t2)):.2f} p90 {np.percentile(us(t2), 90):.2f} max {us(t2).max():.2f}")
role = np.arange(len(a)) % wpw
for r in range(wpw):
    m = role == r
    print(f"  wave role {r}: lifetime p50 {np.median((t2[m] - t0[m]) / 100.0):.2f}  end p50 {np.median(us(t2[m])):.2f} max {us(t2[m]).max():.2f}")
# placement: HW_ID bits (gfx9): wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh_id[12] se_id[15:13]
cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 7; simd = (hw >> 4) & 3
key = xcc * 10000 + se * 1000 + sh * 100 + cu
u, cnt = np.unique(key, return_counts=True)
print(f"distinct CUs seen {len(u)}; waves per CU: min {cnt.min()} p50 {int(np.median(cnt))} max {cnt.max()}  histogram {dict(zip(*np.unique(cnt, return_counts=True)))}")
ks = key * 10 + simd
u2, c2 = np.unique(ks, return_counts=True)
print(f"waves per SIMD: histogram {dict(zip(*np.unique(c2, return_counts=True)))}")
# does a CU's load decide when its waves end?
for n in np.unique(cnt):
    cus = u[cnt == n]
    m = np.isin(key, cus)
    print(f"  CUs with {n} waves: end p50 {np.median(us(t2[m])):.2f} max {us(t2[m]).max():.2f}  lifetime p50 {np.median((t2[m] - t0[m]) / 100.0):.2f}")
# which strips are the stragglers?  (blockIdx -> strip through qd_xcd_chunk, as qs_strip does)
ntc = int(sys.argv[3]) if len(sys.argv) > 3 else 25
wg = np.arange(len(a)) // wpw
nb = wg.max() + 1
per, rem = nb >> 3, nb & 7
x = wg & 7
w = x * per + np.minimum(x, rem) + (wg >> 3)
rs, cs = w // ntc, w % ntc
end = us(t2); life = (t2 - t0) / 100.0
print("end time by row strip (p50 / max):")
print("  " + "  ".join(f"{r}:{np.median(end[rs == r]):.1f}/{end[rs == r].max():.1f}" for r in np.unique(rs)))
print("end time by XCC (p50 / max): " + "  ".join(f"{k}:{np.median(end[xcc == k]):.1f}/{end[xcc == k].max():.1f}" for k in np.unique(xcc)))
print("end time by column strip (p50): " + " ".join(f"{np.median(end[cs == k]):.1f}" for k in np.unique(cs)))
late = np.argsort(end)[-20:]
print("latest 20 waves: (rs, cs, role, xcc, start, prologue, life, end)")
for i in late:
    print(f"  rs {rs[i]:2d} cs {cs[i]:2d} role {role[i]} xcc {xcc[i]} start {us(t0)[i]:.2f} pro {(t1[i] - t0[i]) / 100.0:.2f} life {life[i]:.2f} end {end[i]:.2f}")
