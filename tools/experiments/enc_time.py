"""Quick encode-only timing (bench helper for kernel experiments)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trpx_amd import codec, _lib
L = _lib.lib()
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
px = codec.synth(np.uint16, 0, frames, 512 * 512)
ws = codec.Workspace("cuda"); enc = codec.encode(px, workspace=ws); torch.cuda.synchronize()
L.trpx_profile_enable(1)
buf = (C.c_float * 8)(); acc = []
for _ in range(10):
    codec.encode(px, out=enc.data, workspace=ws, frame_offsets=enc.frame_offsets, status=enc.status)
    n = L.trpx_profile_read(buf, 8); acc.append([buf[k] for k in range(n)])
m = np.median(np.array(acc), axis=0)
print("stages ms", m, "status", enc.status[:2].tolist(), "total", int(enc.frame_offsets[-1]),
      "Mfps", frames / m.sum() / 1e3, "pixel GB/s", frames * 524288 / m.sum() / 1e6)
