"""Round trip with and without the decode index, decode right behind the encode (the bench loop's cache state): event times of the decode alone."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trpx_amd import codec
frames, n = (int(sys.argv[1]) if len(sys.argv) > 1 else 2000), 512 * 512
px = codec.synth(np.uint16, 0, frames, n)
ws = codec.Workspace("cuda")
enc = codec.encode(px, workspace=ws, index=True); torch.cuda.synchronize()
back = torch.empty((frames, n), dtype=torch.uint16, device="cuda"); st = torch.empty(8, dtype=torch.int32, device="cuda")
def run(with_index, reps=20):
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(reps)]
    for r in range(reps + 2):
        e = ev[max(r - 2, 0)]
        e[0].record()
        codec.encode(px, out=enc.data, workspace=ws, frame_offsets=enc.frame_offsets, status=enc.status, index=enc.index if with_index else None)
        e[1].record()
        codec.decode(enc.data, enc.frame_offsets, n, frames, np.uint16, out=back, status=st, workspace=ws, index=enc.index if with_index else None)
        e[2].record()
    torch.cuda.synchronize()
    return np.median([e[0].elapsed_time(e[1]) for e in ev]), np.median([e[1].elapsed_time(e[2]) for e in ev])
for wi in (False, True, False, True):
    e, d = run(wi)
    print("index" if wi else "plain", "encode ms %.4f decode ms %.4f round trip %.4f -> %.3f M frames/s" % (e, d, e + d, frames / (e + d) / 1e3),
          "exact", bool(torch.equal(back.view(torch.int16), px.view(torch.int16))))
