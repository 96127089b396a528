"""Quick decode-only timing (kernel experiments)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trpx_amd import codec, _lib
L = _lib.lib()
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 512 * 512
px = codec.synth(np.uint16, 0, frames, n)
enc = codec.encode(px); torch.cuda.synchronize()
ws = codec.Workspace("cuda")
back, st = codec.decode(enc.data, enc.frame_offsets, n, frames, np.uint16, workspace=ws); torch.cuda.synchronize()
ok = bool(torch.equal(back.view(torch.int16), px.view(torch.int16)))
L.trpx_profile_enable(1)
buf = (C.c_float * 8)(); acc = []
for _ in range(10):
    codec.decode(enc.data, enc.frame_offsets, n, frames, np.uint16, out=back, workspace=ws, status=st)
    k = L.trpx_profile_read(buf, 8); acc.append([buf[i] for i in range(k)])
m = np.median(np.array(acc), axis=0)
print("decode stages ms", m, "roundtrip ok", ok, "status", int(st[0]), "Mfps", frames / m.sum() / 1e3)
