"""debug: fuzz case 5 (int8, 1475 x 1679, 2 frames, widths flip every other block) -- which group offsets differ between the encoder's index and the walked one"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from trpx_amd import codec, _lib
L = _lib.lib()
rng = np.random.RandomState(int(os.environ.get("SEED", "3")))
n, frames, dt = 1475 * 1679, 2, np.dtype(np.int8)
nblk = (n + 11) // 12
hi = np.where(rng.rand(frames, nblk) < 0.5, 2, 3)
mag = (rng.rand(frames, nblk * 12) * (2.0 ** np.repeat(hi, 12, axis=1))).astype(np.int64)[:, :n]
if os.environ.get("HALF"): mag[:, : n // 2] = 0
mag = mag * rng.choice([-1, 1], size=mag.shape)
px = torch.from_numpy(mag.astype(dt)).cuda()
for route in (0, 4, 2):
    L.trpx_set_decode_path(route)
    enc = codec.encode(px, index=True); torch.cuda.synchronize(); enc.check()
    back, st = codec.decode(enc.stack(), enc.frame_offsets, n, frames, dt); torch.cuda.synchronize()
    walked = codec.build_index(enc.stack(), enc.frame_offsets, n, frames, dt); torch.cuda.synchronize()
    ng = (nblk + 255) // 256
    a = enc.index[: 8 * frames * ng].view(torch.int64).cpu().numpy(); b = walked[: 8 * frames * ng].view(torch.int64).cpu().numpy()
    bad = np.nonzero(a != b)[0]
    w_off = (8 * frames * ng + 15) // 16 * 16
    wa = enc.index[w_off: w_off + frames * nblk].cpu().numpy(); wb = walked[w_off: w_off + frames * nblk].cpu().numpy()
    print("route", route, "status", st.tolist(), "pixels", bool(torch.equal(back, px)), "groups differing", len(bad), bad[:10], a[bad[:6]], b[bad[:6]], "widths differing", int((wa != wb).sum()), flush=True)
