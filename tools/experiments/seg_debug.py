"""Debug helper: tiled decode (position-parallel walk) of a few large frames; prints status words and the first mismatch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trpx_amd import codec
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 2
side = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
n = side * side
px = codec.synth(np.int32, 0, frames, n)
enc = codec.encode(px, index=True); torch.cuda.synchronize(); enc.check()
for rep in range(3):
    back, st = codec.decode(enc.data, enc.frame_offsets, n, frames, np.int32); torch.cuda.synchronize()
    ok = torch.equal(back, px)
    print("rep", rep, "status", st.cpu().numpy().tolist(), "exact", ok)
    if not ok:
        bad = (back != px).nonzero()
        print("  mismatches", bad.shape[0], "first", bad[0].tolist(), "block", int(bad[0][1]) // 12, "last", bad[-1].tolist())
walked = codec.build_index(enc.stack(), enc.frame_offsets, n, frames, np.int32); torch.cuda.synchronize()
nb = (n + 11) // 12; ng = (nb + 255) // 256
w_off = (8 * frames * ng + 15) // 16 * 16
a = enc.index[w_off: w_off + frames * nb]; b = walked[w_off: w_off + frames * nb]
d = (a != b).nonzero()
print("width mismatches", d.shape[0], d[:5].flatten().tolist())
ga = enc.index[: 8 * frames * ng].view(torch.int64); gb = walked[: 8 * frames * ng].view(torch.int64)
dg = (ga != gb).nonzero()
print("group offset mismatches", dg.shape[0], dg[:5].flatten().tolist())
