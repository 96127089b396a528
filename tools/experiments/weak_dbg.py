import sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from trpx_amd import codec, _lib
rng = np.random.RandomState(5)
for dtype, n, frames in ((np.uint16, 1030 * 1065, 12),):
    a = rng.randint(0, 8, (frames, n)).astype(dtype)
    a[:, 0::192] = 9
    px = torch.from_numpy(a).cuda()
    enc = codec.encode(px); torch.cuda.synchronize(); enc.check()
    back, st = codec.decode(enc.stack(), enc.frame_offsets, n, frames, dtype)
    torch.cuda.synchronize()
    print(st.cpu().numpy(), torch.equal(back, px))
