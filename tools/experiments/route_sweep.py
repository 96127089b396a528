"""Index-free decode of one stack: kind (p3 | synth | noisy) h w frames.  Run once per value of $TRPX_SINGLE_PART (read at library load)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from trpx_amd import codec, workloads, _lib
kind, h, w, F = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
dev = torch.device("cuda:0"); N = h * w
if kind == "p3": px = workloads.poisson_u16(3.0, 0, F, N, device=dev, chunk_frames=max(1, 125 * 262144 // N))
elif kind == "synth": px = codec.synth(np.uint16, 0, F, N, device=dev)
else:
    g = torch.Generator(device=dev); g.manual_seed(1)
    px = torch.poisson(torch.full((F, N), 1.5, device=dev), generator=g).clamp_(0, 6).to(torch.int16).view(torch.uint16)
ws_e, ws_d = codec.Workspace(dev), codec.Workspace(dev)
enc = codec.encode(px, workspace=ws_e); torch.cuda.synchronize(); enc.check()
back = torch.empty_like(px); st = torch.empty(8, dtype=torch.int32, device=dev)
fn = lambda: codec.decode(enc.data, enc.frame_offsets, N, F, np.uint16, out=back, status=st, workspace=ws_d)
fn(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(8): fn()
e1.record(); torch.cuda.synchronize()
ok = int(st[0]) == 0 and torch.equal(back, px)
P = _lib.lib().trpx_decode_parts_per_frame(codec.dtype_code(np.uint16), N, F, 12)
print(f"{kind} {F} x ({h} x {w}) [{os.environ.get('TRPX_SINGLE_PART', 'default')}]: parts/frame {P}, {e0.elapsed_time(e1) / 8:.4f} ms for {F * N * 2 / 1e9:.3f} GB of pixels, exact={ok}, listed {int(st[2])}")
