"""Does the large-frame extraction (k_decode_units_indexed, 128 x 2048^2 Poisson(3)) depend on where its buffers sit?  The same decode
into output tensors at different offsets inside one allocation, and from stream copies at different offsets."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from trpx_amd import codec, workloads
dev = torch.device("cuda:0")
F, N = 128, 2048 * 2048
px = workloads.poisson_u16(3.0, 0, F, N, device=dev, chunk_frames=8)
ws_e, ws_d = codec.Workspace(dev), codec.Workspace(dev)
enc = codec.encode(px, workspace=ws_e); torch.cuda.synchronize(); enc.check()
total = enc.total_bytes()
st = torch.empty(8, dtype=torch.int32, device=dev)
def timed(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
big = torch.empty(F * N * 2 + (64 << 20), dtype=torch.uint8, device=dev)
for off in (0, 4096, 65536, 1 << 20, (3 << 20) + 128, (17 << 20) + 8192):
    back = big[off: off + F * N * 2].view(torch.uint16).view(F, N)
    t = timed(lambda: codec.decode(enc.data, enc.frame_offsets, N, F, np.uint16, out=back, status=st, workspace=ws_d))
    ok = int(st[0]) == 0 and torch.equal(back, px)
    print(f"output at +{off:>9d}: {t:.4f} ms exact={ok}", flush=True)
sbig = torch.empty(total + (64 << 20) + 64, dtype=torch.uint8, device=dev)
back = torch.empty_like(px)
for off in (0, 4096, 65536, (5 << 20) + 16):
    s = sbig[off: off + total + 16]; s[:total].copy_(enc.data[:total])
    t = timed(lambda: codec.decode(s, enc.frame_offsets, N, F, np.uint16, out=back, status=st, workspace=ws_d))
    print(f"stream at +{off:>9d}: {t:.4f} ms exact={int(st[0]) == 0 and torch.equal(back, px)}", flush=True)
