"""Decode with the decode index (k_decode_frames_indexed for stacks of small frames, k_unpack_tiles otherwise) and noisy-stack decode: quick timing."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trpx_amd import codec
frames, n = (int(sys.argv[1]) if len(sys.argv) > 1 else 2000), 512 * 512
px = codec.synth(np.uint16, 0, frames, n)
ws = codec.Workspace("cuda")
enc = codec.encode(px, workspace=ws, index=True); torch.cuda.synchronize()
back = torch.empty((frames, n), dtype=torch.uint16, device="cuda"); st = torch.empty(8, dtype=torch.int32, device="cuda")
def timed(fn, reps=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps
t = timed(lambda: codec.decode(enc.data, enc.frame_offsets, n, frames, np.uint16, out=back, status=st, index=enc.index))
print("decode with index ms", round(t, 4), "exact", bool(torch.equal(back.view(torch.int16), px.view(torch.int16))))
g = torch.Generator(device="cuda"); g.manual_seed(1)
bg = torch.poisson(torch.full((frames, n), 1.5, device="cuda"), generator=g).clamp_(0, 6).to(torch.int32)
hot = torch.rand((frames, n), device="cuda", generator=g) < (1.0 / 4096)
pxn = torch.where(hot, torch.randint(0, 4000, (frames, n), device="cuda", generator=g, dtype=torch.int32), bg).to(torch.int16).view(torch.uint16)
del bg, hot
en = codec.encode(pxn, workspace=ws); torch.cuda.synchronize()
from trpx_amd import _lib
if os.environ.get("ROUTE"): _lib.lib().trpx_set_decode_path(int(os.environ["ROUTE"]))
t = timed(lambda: codec.decode(en.data, en.frame_offsets, n, frames, np.uint16, out=back, workspace=ws, status=st))
print("noisy decode ms", round(t, 4), "exact", bool(torch.equal(back.view(torch.int16), pxn.view(torch.int16))), "status", int(st[0]))
