"""Decode time against the density of width changes: 2000 x 512^2 u16 frames whose blocks change their width with probability d
(independently: runs are geometric, like detector noise, not regular).  `python3 tools/dens_time.py [d ...]`"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trpx_amd import codec
dev = torch.device("cuda:0")
F, N = 2000, 512 * 512
nb = (N + 11) // 12
for d in [float(x) for x in sys.argv[1:]] or [0.05, 0.1, 0.15, 0.2, 0.25, 0.3]:
    g = torch.Generator(device=dev); g.manual_seed(7)
    flip = torch.rand((F, nb), device=dev, generator=g) < d
    wide = (torch.cumsum(flip.to(torch.int32), 1) & 1).to(torch.bool)          # the block's width: 3 or 4 bits
    px = torch.randint(0, 8, (F, nb, 12), device=dev, generator=g, dtype=torch.int32)
    px[:, :, 0] = torch.where(wide, torch.full_like(px[:, :, 0], 9), torch.full_like(px[:, :, 0], 5))
    px = px.reshape(F, nb * 12)[:, :N].to(torch.int16).view(torch.uint16).contiguous()
    enc = codec.encode(px); torch.cuda.synchronize(); enc.check()
    back = torch.empty_like(px); st = torch.empty(8, dtype=torch.int32, device=dev); ws = codec.Workspace(dev)
    fn = lambda: codec.decode(enc.data, enc.frame_offsets, N, F, np.uint16, out=back, status=st, workspace=ws)
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"d={d:.2f}: {e0.elapsed_time(e1) / 10:.4f} ms, exact={bool(int(st[0].item()) == 0 and torch.equal(back.view(torch.int16), px.view(torch.int16)))}")
