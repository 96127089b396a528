"""Verdicts of k_part_resolve (decode_part.hip) for a leg: `TRPX_LIB=tools/variants/libtrpx_partstats.so python3 tools/part_stats.py <leg>`."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trpx_amd import codec
from leg_prof import make
leg = sys.argv[1]
dev = torch.device("cuda:0")
px, dt = make(leg, dev)
nf, nv = px.shape[0], px[0].numel()
enc = codec.encode(px); torch.cuda.synchronize(); enc.check()
back, st = codec.decode(enc.data, enc.frame_offsets, nv, nf, dt)
torch.cuda.synchronize()
s = st.cpu().numpy()
print(f"{leg}: frames {nf}, status {s[0]}, fallback frames {s[2]}, plain guesses {s[3]}, bad/dense walks {s[4]}, repaired links {s[5]}, failed repairs {s[6]}, oversized {s[7]}, exact {torch.equal(back.view(torch.uint8), px.view(torch.uint8))}")
