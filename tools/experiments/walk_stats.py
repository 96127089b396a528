import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trpx_amd import codec
frames, n = 2000, 512 * 512
px = codec.synth(np.uint16, 0, frames, n)
enc = codec.encode(px); torch.cuda.synchronize()
back, st = codec.decode(enc.data, enc.frame_offsets, n, frames, np.uint16); torch.cuda.synchronize()
s = st.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
print("steps/frame", s[2] / frames, "refills/frame", s[3] / frames, "refill cycles/frame", s[4] * 16 / frames, "total cycles/frame", s[5] * 16 / frames)
