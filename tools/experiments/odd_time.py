"""Frames whose pixel count is not a multiple of 4 (most detectors: 1030 x 1065, 2463 x 2527 ...): which kernels run, how fast?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trpx_amd import codec
frames, h, w = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
n = h * w
px = codec.synth(np.uint16, 0, frames, n)
ws = codec.Workspace("cuda")
enc = codec.encode(px, workspace=ws); torch.cuda.synchronize(); enc.check()
back = torch.empty_like(px); st = torch.empty(8, dtype=torch.int32, device="cuda")
def timed(fn, reps=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps
te = timed(lambda: codec.encode(px, out=enc.data, workspace=ws, frame_offsets=enc.frame_offsets, status=enc.status))
td = timed(lambda: codec.decode(enc.data, enc.frame_offsets, n, frames, np.uint16, out=back, workspace=ws, status=st))
gb = frames * n * 2 / 1e6
print(f"{frames} frames {h}x{w} (n % 4 = {n % 4}): encode {te:.3f} ms ({gb / te:.0f} GB/s pixels) decode {td:.3f} ms ({gb / td:.0f} GB/s) exact {bool(torch.equal(back.view(torch.int16), px.view(torch.int16)))}")
