"""Where should the per-frame decoder hand a frame to the position-parallel walk?  Stacks whose block width changes every k-th block
(k = 2 .. 12), decoded with the library in $TRPX_LIB (a -DTRPX_DIAGNOSTICS build of api.hip honours TRPX_NO_DEFER=1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trpx_amd import codec
frames, n = 2000, 512 * 512
dev = torch.device("cuda")
g = torch.Generator(device=dev); g.manual_seed(3)
nblk = (n + 11) // 12
def timed(fn, reps=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps
for k in (2, 3, 4, 5, 6, 8, 12):
    # runs of k blocks, the width of a run drawn from {2, 3} so that neighbouring runs differ half of the time -> a change every ~2k blocks;
    # alternate strictly instead: change every k blocks
    run = (torch.arange(nblk, device=dev) // k) % 2
    hi = torch.where(run == 0, 3, 7).repeat_interleave(12)[:n]                    # widths 2 / 3
    px = (torch.randint(0, 1 << 20, (frames, n), device=dev, generator=g, dtype=torch.int32) % (hi + 1)).to(torch.int32)
    top = torch.zeros(n, dtype=torch.bool, device=dev); top[::12] = True
    px = torch.where(top, hi, px).to(torch.int16).view(torch.uint16)               # first value of a block pins the width
    ws = codec.Workspace(dev)
    enc = codec.encode(px, workspace=ws); torch.cuda.synchronize(); enc.check()
    back = torch.empty_like(px); st = torch.empty(8, dtype=torch.int32, device=dev)
    t = timed(lambda: codec.decode(enc.data, enc.frame_offsets, n, frames, np.uint16, out=back, workspace=ws, status=st))
    ok = bool(torch.equal(back.view(torch.int16), px.view(torch.int16)))
    print(f"width change every {k:2d} blocks: decode {t:.3f} ms exact {ok} ratio {enc.total_bytes() / (frames * n * 2):.3f}", flush=True)
    del px, enc, back
