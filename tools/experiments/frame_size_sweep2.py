"""Index-free decode of Poisson(3) stacks of ~1 GB against the frame size: per-kernel times through the profiling hook."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from trpx_amd import codec, workloads, _lib
dev = torch.device("cuda:0"); L = _lib.lib()
ws_e, ws_d = codec.Workspace(dev), codec.Workspace(dev)
st = torch.empty(8, dtype=torch.int32, device=dev)
L.trpx_profile_enable(1)
buf = (C.c_float * 8)()
for (h, w, F) in ((512, 512, 2000), (640, 640, 1280), (768, 768, 888), (1024, 1024, 500), (1448, 1448, 250), (2048, 2048, 125), (2048, 2048, 128)):
    N = h * w
    px = workloads.poisson_u16(3.0, 0, F, N, device=dev, chunk_frames=max(1, 125 * 262144 // N))
    enc = codec.encode(px, workspace=ws_e); torch.cuda.synchronize(); enc.check()
    back = torch.empty_like(px)
    ts = []
    for _ in range(6):
        codec.decode(enc.data, enc.frame_offsets, N, F, np.uint16, out=back, status=st, workspace=ws_d)
        k = L.trpx_profile_read(buf, 8); ts.append([buf[i] for i in range(k)])
    ok = int(st[0]) == 0 and torch.equal(back, px)
    t = np.median(np.array(ts[1:]), 0)
    P = L.trpx_decode_parts_per_frame(codec.dtype_code(np.uint16), N, F, 12)
    print(f"{F} x ({h} x {w}): parts/frame {P}, stages ms {np.round(t, 4)} sum {t.sum():.4f}, pixels {F * N * 2 / 1e9:.3f} GB, exact={ok}", flush=True)
    del px, enc, back
