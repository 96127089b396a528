"""Encode-only timing, launches queued back to back (no host sync in between): separates per-launch fixed cost from idle-gap effects."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trpx_amd import codec
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
px = codec.synth(np.uint16, 0, frames, 512 * 512)
ws = codec.Workspace("cuda"); enc = codec.encode(px, workspace=ws); torch.cuda.synchronize()
for _ in range(3):
    codec.encode(px, out=enc.data, workspace=ws, frame_offsets=enc.frame_offsets, status=enc.status)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    codec.encode(px, out=enc.data, workspace=ws, frame_offsets=enc.frame_offsets, status=enc.status)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print(f"frames {frames} back-to-back {reps} launches: {ms:.4f} ms per encode call ({frames / ms / 1e3:.3f} Mfps), status {enc.status[:2].tolist()}")
