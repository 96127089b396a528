"""Diagnostic (TRPX_DEC_STAMPS build of decode_frame.hip): per wave role, cycles working / waiting at the super-step barrier."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trpx_amd import codec, _lib
frames, n = 2000, 512 * 512
px = codec.synth(np.uint16, 0, frames, n)
enc = codec.encode(px); torch.cuda.synchronize()
ws = codec.Workspace("cuda")
for _ in range(3):
    back, st = codec.decode(enc.data, enc.frame_offsets, n, frames, np.uint16, workspace=ws); torch.cuda.synchronize()
d = back.cpu().numpy().reshape(frames, n).view(np.uint32)[:, :32].reshape(frames, 4, 8).astype(np.int64)
work, wait, hwid, total, start = d[..., 0], d[..., 1], d[..., 2], d[..., 3], d[..., 4]
for r in range(4):
    print(f"role {r}: work {work[:, r].mean():9.0f} (p95 {np.percentile(work[:, r], 95):9.0f})  wait {wait[:, r].mean():9.0f}  total {total[:, r].mean():9.0f} max {total[:, r].max():9.0f} cycles")
for r in range(1, 4):
    print(f"role {r}: fetch wait + raw read {d[:, r, 5].mean():9.0f}  extraction {d[:, r, 6].mean():9.0f}  staged stores {d[:, r, 7].mean():9.0f}  (cycles per frame and wave)")
simd = (hwid[:, 0] >> 4) & 3; cu = (hwid[:, 0] >> 8) & 0xF; se = (hwid[:, 0] >> 13) & 7; slot = hwid[:, 0] & 0xF
print("walker slots", np.bincount(slot, minlength=8), "walker simds", np.bincount(simd, minlength=4))
t = total[:, 0]
print("walker total by slot:", [int(t[slot == k].mean()) for k in range(8) if (slot == k).any()])
print("start spread (cycles):", int((start[:, 0] - start[:, 0].min()).max() & 0xFFFFFFFF), " frame total cycles min/mean/max", t.min(), int(t.mean()), t.max())
