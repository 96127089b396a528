import os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
import numpy as np, torch
from trpx_amd import codec
from leg_prof import make
dev = torch.device("cuda:0")
for leg in sys.argv[1:]:
    px, dt = make(leg, dev)
    nf, nv = px.shape[0], px[0].numel()
    enc = codec.encode(px); torch.cuda.synchronize(); enc.check()
    back, st = codec.decode(enc.data, enc.frame_offsets, nv, nf, dt)
    torch.cuda.synchronize()
    s = st.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    nb = (nv + 11) // 12
    print(f"{leg}: frames {nf} blocks/frame {nb}: rounds(sum over waves) {s[2]}, wave-steps {s[3]}, lane-walks {s[4]}; per frame: rounds {s[2]/nf:.1f} wave-steps {s[3]/nf:.0f}")
