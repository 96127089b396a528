"""Diagnostic (TRPX_DEC_STAMPS build): per-frame walker / extraction cycles by XCD, CU and SIMD placement.  usage: dec_stamps2.py [frames]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trpx_amd import codec
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
n = 512 * 512
px = codec.synth(np.uint16, 0, frames, n)
enc = codec.encode(px); torch.cuda.synchronize()
ws = codec.Workspace("cuda")
for _ in range(3):
    back, st = codec.decode(enc.data, enc.frame_offsets, n, frames, np.uint16, workspace=ws); torch.cuda.synchronize()
d = back.cpu().numpy().reshape(frames, n).view(np.uint32)[:, :32].reshape(frames, 4, 8).astype(np.int64)
work, wait, hwid, total, start, xcc = d[..., 0], d[..., 1], d[..., 2], d[..., 3], d[..., 4], d[..., 5] & 0xF
for r in range(4):
    print(f"role {r}: work {work[:, r].mean():9.0f} (p95 {np.percentile(work[:, r], 95):9.0f})  wait {wait[:, r].mean():9.0f}  total {total[:, r].mean():9.0f} max {total[:, r].max():9.0f} cycles")
h = hwid[:, 0]
simd = (h >> 4) & 3; cu = (h >> 8) & 0xF; sh = (h >> 12) & 1; se = (h >> 13) & 7; slot = h & 0xF
x = xcc[:, 0]
cuid = ((x * 8 + se) * 2 + sh) * 16 + cu
t = total[:, 0]
s0 = (start[:, 0] - start[:, 0].min()) & 0xFFFFFFFF
end = s0 + t
print("frames", frames, " kernel span (cycles)", int(end.max()), " start spread", int(s0.max()))
print("frame total percentiles 0/10/50/90/99/100:", [int(np.percentile(t, q)) for q in (0, 10, 50, 90, 99, 100)])
print("walker work percentiles:", [int(np.percentile(work[:, 0], q)) for q in (0, 10, 50, 90, 99, 100)])
print("per XCD: frames, mean total, max end:", [(int((x == k).sum()), int(t[x == k].mean()) if (x == k).any() else 0, int(end[x == k].max()) if (x == k).any() else 0) for k in range(8)])
ucu, inv, cnt = np.unique(cuid, return_inverse=True, return_counts=True)
print("CUs used", len(ucu), " WGs per CU histogram", np.bincount(cnt))
# walkers per SIMD on the frame's CU (all WGs that ran there, whole launch) vs frame time
wps = np.zeros((len(ucu), 4), dtype=int)
np.add.at(wps, (inv, simd), 1)
mx = wps.max(axis=1)[inv]
print("max walkers on one SIMD of the CU -> frames, mean total:", [(int(k), int((mx == k).sum()), int(t[mx == k].mean())) for k in np.unique(mx)])
print("by WGs on the CU -> mean total:", [(int(k), int(t[cnt[inv] == k].mean())) for k in np.unique(cnt)])
late = s0 > 1000000
print("late starters (second round):", int(late.sum()))
