"""Decode time as a function of what ran just before it (bench.py alternates encode and decode)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trpx_amd import codec, _lib
L = _lib.lib()
frames, n = 2000, 512 * 512
px = codec.synth(np.uint16, 0, frames, n)
ws = codec.Workspace("cuda")
enc = codec.encode(px, workspace=ws); torch.cuda.synchronize()
back, st = codec.decode(enc.data, enc.frame_offsets, n, frames, np.uint16, workspace=ws); torch.cuda.synchronize()
junk = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
L.trpx_profile_enable(1)
buf = (C.c_float * 8)()
def dec():
    codec.decode(enc.data, enc.frame_offsets, n, frames, np.uint16, out=back, workspace=ws, status=st)
    k = L.trpx_profile_read(buf, 8); return sum(buf[i] for i in range(k))
for name, before in (("nothing", lambda: None), ("encode", lambda: codec.encode(px, out=enc.data, workspace=ws, frame_offsets=enc.frame_offsets, status=enc.status)),
                     ("1 GB memset", lambda: junk.zero_()), ("read 1 GB (sum)", lambda: junk.view(torch.int64).sum()),
                     ("sleep 5 ms", lambda: (torch.cuda.synchronize(), __import__("time").sleep(0.005))),
                     ("encode, then read the stream", lambda: (codec.encode(px, out=enc.data, workspace=ws, frame_offsets=enc.frame_offsets, status=enc.status),
                                                               enc.data[: 203596120].view(torch.int64).sum())),
                     ("encode, then zero the output", lambda: (codec.encode(px, out=enc.data, workspace=ws, frame_offsets=enc.frame_offsets, status=enc.status),
                                                               back.zero_()))):
    t = []
    for _ in range(8):
        before(); torch.cuda.synchronize()
        t.append(dec())
    print(f"before = {name:30s}: decode {np.median(t):.4f} ms (min {min(t):.4f})")
