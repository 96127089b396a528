"""Why does bench.py report the indexed decode of the Poisson(3) stack at 0.30-0.32 ms when tools/leg_prof.py (always run under
rocprofv3 by tools/profile_legs.sh) reports 0.24-0.25?  Same call, same data.  Plain run: `python3 tools/experiments/idx_gap.py`."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from trpx_amd import codec, workloads
dev = torch.device("cuda:0")
F, N = 2000, 512 * 512
px = workloads.poisson_u16(3.0, 0, F, N, device=dev)
ws_e, ws_d = codec.Workspace(dev), codec.Workspace(dev)
enc = codec.encode(px, workspace=ws_e, index=True); torch.cuda.synchronize(); enc.check()
back = torch.empty_like(px); st = torch.empty(8, dtype=torch.int32, device=dev)
idx = lambda: codec.decode(enc.data, enc.frame_offsets, N, F, np.uint16, out=back, status=st, index=enc.index)
free = lambda: codec.decode(enc.data, enc.frame_offsets, N, F, np.uint16, out=back, status=st, workspace=ws_d)
def timed(fn, n):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
def each(fn, n):                      # every call on its own pair of events: is it the loop or every call?
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return ts
print(f"idx x10 (leg_prof's loop)        {timed(idx, 10):.4f} ms")
print(f"idx x20 (bench's loop length)    {timed(idx, 20):.4f} ms")
print(f"idx x100                         {timed(idx, 100):.4f} ms")
print(f"free x20                         {timed(free, 20):.4f} ms")
print(f"idx x20 right behind free        {timed(idx, 20):.4f} ms")
print("idx, one call per event pair:   ", " ".join(f"{t:.3f}" for t in each(idx, 12)))
ok = int(st[0].item()) == 0 and torch.equal(back.view(torch.int16), px.view(torch.int16))
print("exact:", ok)
