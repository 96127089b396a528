import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from trpx_amd import codec
frames, n = 200, 512 * 512
g = torch.Generator(device="cuda"); g.manual_seed(1)
bg = torch.poisson(torch.full((frames, n), 1.5, device="cuda"), generator=g).clamp_(0, 6).to(torch.int32)
hot = torch.rand((frames, n), device="cuda", generator=g) < (1.0 / 4096)
pxn = torch.where(hot, torch.randint(0, 4000, (frames, n), device="cuda", generator=g, dtype=torch.int32), bg).to(torch.int16).view(torch.uint16)
ws = codec.Workspace("cuda")
en = codec.encode(pxn, workspace=ws); torch.cuda.synchronize()
back, st = codec.decode(en.data, en.frame_offsets, n, frames, np.uint16, workspace=ws); torch.cuda.synchronize()
s = st.cpu().numpy().view(np.uint32)
print("status", s[0], "per frame: rounds %.2f wave-steps %.0f lane-walks %.1f merged %.1f with-B %.1f merge-iters %.0f" % tuple(x / frames for x in s[2:8]))
