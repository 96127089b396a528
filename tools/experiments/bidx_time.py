"""trpx_build_index on stacks of small frames (per-frame walker with index writers; position-parallel walk for listed frames)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trpx_amd import codec, _lib
dev = torch.device("cuda"); n = 512 * 512
g = torch.Generator(device=dev); g.manual_seed(1)
def noisy(frames, signed):
    bg = torch.poisson(torch.full((frames, n), 1.5, device=dev), generator=g).clamp_(0, 6).to(torch.int32)
    hot = torch.rand((frames, n), device=dev, generator=g) < (1.0 / 4096)
    px = torch.where(hot, torch.randint(0, 4000, (frames, n), device=dev, generator=g, dtype=torch.int32), bg)
    if signed: px = px - 3
    return px.to(torch.int16) if signed else px.to(torch.int16).view(torch.uint16)
sets = [("synth-v1 u16", lambda: codec.synth(np.uint16, 0, 2000, n), np.uint16),
        ("noisy u16", lambda: noisy(2000, False), np.uint16),
        ("noisy i16 (const width)", lambda: noisy(2000, True), np.int16),
        ("wide u16 (10..12 bit bg)", lambda: torch.randint(0, 3000, (1000, n), device=dev, generator=g, dtype=torch.int32).to(torch.int16).view(torch.uint16), np.uint16)]
for name, make, dt in sets:
    px = make(); f = px.shape[0]
    enc = codec.encode(px, index=True); torch.cuda.synchronize(); enc.check()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    walked = codec.build_index(enc.stack(), enc.frame_offsets, n, f, dt); torch.cuda.synchronize()
    e0.record()
    for _ in range(5): walked = codec.build_index(enc.stack(), enc.frame_offsets, n, f, dt)
    e1.record(); torch.cuda.synchronize()
    nb = (n + 11) // 12; ng = (nb + 255) // 256; w_off = (8 * f * ng + 15) // 16 * 16
    same = torch.equal(enc.index[: 8 * f * ng], walked[: 8 * f * ng]) and torch.equal(enc.index[w_off: w_off + f * nb], walked[w_off: w_off + f * nb])
    print(f"{name:26s} frames {f:5d} build_index ms {e0.elapsed_time(e1) / 5:.3f} equal to the encoder's index: {same}", flush=True)
    del px, enc, walked
