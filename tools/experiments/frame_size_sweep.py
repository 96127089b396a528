"""Decode throughput against the frame size at (nearly) equal stack bytes: is the 512 x 512 stack's 2^19-byte frame special?
Poisson(3) counts, decode with the index (k_decode_frames_indexed, one workgroup per frame) and without."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from trpx_amd import codec, workloads
dev = torch.device("cuda:0")
def timed(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
ws_e, ws_d = codec.Workspace(dev), codec.Workspace(dev)
st = torch.empty(8, dtype=torch.int32, device=dev)
for (h, w) in ((512, 512), (520, 520), (512, 576), (480, 512), (512, 640), (448, 512)):
    N = h * w; F = int(2000 * 512 * 512 / N) // 8 * 8
    px = workloads.poisson_u16(3.0, 0, F, N, device=dev)
    enc = codec.encode(px, workspace=ws_e, index=True); torch.cuda.synchronize(); enc.check()
    back = torch.empty_like(px)
    t_i = timed(lambda: codec.decode(enc.data, enc.frame_offsets, N, F, np.uint16, out=back, status=st, index=enc.index))
    ok = int(st[0]) == 0 and torch.equal(back, px)
    t_f = timed(lambda: codec.decode(enc.data, enc.frame_offsets, N, F, np.uint16, out=back, status=st, workspace=ws_d))
    ok = ok and int(st[0]) == 0 and torch.equal(back, px)
    gb = F * N * 2 / 1e9
    print(f"{F} x ({h} x {w}) = {gb:.3f} GB of pixels, frame {N * 2} B: with index {t_i:.4f} ms = {gb / t_i:.2f} TB/s, without {t_f:.4f} ms, exact={ok}", flush=True)
    del px, enc, back
