import os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
import numpy as np, torch
from trpx_amd import codec
dev = torch.device("cuda:0")
for name, F, N, lo, hi in (("ped midsize", 200, 1030*1065, 100, 108), ("ped 2048", 64, 2048*2048, 100, 108), ("ped wide", 200, 1030*1065, 1000, 1064)):
    g = torch.Generator(device=dev); g.manual_seed(3)
    px = torch.randint(lo, hi, (F, N), device=dev, generator=g, dtype=torch.int32).to(torch.int16).view(torch.uint16)
    enc = codec.encode(px); torch.cuda.synchronize(); enc.check()
    back = torch.empty_like(px); st = torch.empty(8, dtype=torch.int32, device=dev); ws = codec.Workspace(dev)
    fn = lambda: codec.decode(enc.data, enc.frame_offsets, N, F, np.uint16, out=back, status=st, workspace=ws)
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    s = st.cpu().numpy()
    alg = F * N * 2 + enc.total_bytes()
    print(f"{name}: {e0.elapsed_time(e1)/5:.4f} ms ({alg/(e0.elapsed_time(e1)/5)/1e6/8000:.3f} of peak), status {s[:8]}, exact={bool(torch.equal(back.view(torch.int16), px.view(torch.int16)))}")
