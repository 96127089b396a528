import sys
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
    s = st.cpu().numpy()
    print(f"{leg}: handed over after step 0 / 3 / 11: {s[2]} / {s[3]} / {s[4]}; decisions by the stack's count {s[5]}; mean own density at step 0: {s[6] / nf / 10:.1f} %")
