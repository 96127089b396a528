"""Header-dense frames larger than one wavefront's worth of the position-parallel walk (> 32 K blocks): decode time by route."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trpx_amd import codec
frames, n = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda")
g = torch.Generator(device=dev); g.manual_seed(1)
bg = torch.poisson(torch.full((frames, n), 1.5, device=dev), generator=g).clamp_(0, 6).to(torch.int32)
hot = torch.rand((frames, n), device=dev, generator=g) < (1.0 / 4096)
px = torch.where(hot, torch.randint(0, 4000, (frames, n), device=dev, generator=g, dtype=torch.int32), bg).to(torch.int16).view(torch.uint16)
del bg, hot
ws = codec.Workspace(dev)
enc = codec.encode(px, workspace=ws); torch.cuda.synchronize(); enc.check()
back = torch.empty_like(px); st = torch.empty(8, dtype=torch.int32, device=dev)
def timed(fn, reps=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps
t = timed(lambda: codec.decode(enc.data, enc.frame_offsets, n, frames, np.uint16, out=back, workspace=ws, status=st))
print(f"noisy u16 frames {frames} n {n} route {os.environ.get('TRPX_DECODE_PATH', 'default')}: decode {t:.3f} ms exact {bool(torch.equal(back.view(torch.int16), px.view(torch.int16)))} status {int(st[0])}")
