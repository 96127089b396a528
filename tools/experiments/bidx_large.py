"""trpx_build_index on stacks of large frames: `python3 tools/bidx_large.py` (ms per call; the index route since round 5)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trpx_amd import codec
from leg_prof import make
dev = torch.device("cuda:0")
for leg in sys.argv[1:] or ["c4", "midsize", "midsize_p3"]:
    px, dt = make(leg, dev)
    nf, nv = px.shape[0], px[0].numel()
    enc = codec.encode(px, index=True); torch.cuda.synchronize(); enc.check()
    idx = codec.build_index(enc.data, enc.frame_offsets, nv, nf, dt); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): idx = codec.build_index(enc.data, enc.frame_offsets, nv, nf, dt)
    e1.record(); torch.cuda.synchronize()
    nb = (nv + 11) // 12; ng = (nb + 255) // 256; w_off = (8 * nf * ng + 15) // 16 * 16
    same = torch.equal(idx[: 8 * nf * ng], enc.index[: 8 * nf * ng]) and torch.equal(idx[w_off: w_off + nf * nb], enc.index[w_off: w_off + nf * nb])
    print(f"{leg}: build_index {e0.elapsed_time(e1) / 5:.4f} ms per call (incl. the index allocation), == the encoder's index: {same}")
