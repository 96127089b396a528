"""round 6: the dense walk's index against the encoder's own, on the route-matrix test's stacks (tests/test_gpu_parity.py:553)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trpx_amd import codec, _lib
dev = torch.device("cuda:0")
rng = np.random.RandomState(99)
for dtype in [getattr(np, a) for a in (sys.argv[1:] or ["uint8", "int8", "uint16", "int16", "uint32", "int32"])]:
    dt = np.dtype(dtype)
    top = 8 * dt.itemsize - (1 if dt.kind == "i" else 0)
    tdt = {1: torch.uint8 if dt.kind == "u" else torch.int8, 2: torch.uint16 if dt.kind == "u" else torch.int16,
           4: torch.uint32 if dt.kind == "u" else torch.int32}[dt.itemsize]
    for frames, n in ((140, 3000), (6, 40000)):
        nblk = (n + 11) // 12
        hi = rng.choice([0, 1, 2, 3, 5, min(9, top), top], size=(frames, nblk), p=[0.1, 0.2, 0.3, 0.2, 0.1, 0.07, 0.03])
        mag = (rng.rand(frames, nblk * 12) * (2.0 ** np.repeat(hi, 12, axis=1))).astype(np.int64)[:, :n]
        if dt.kind == "i":
            mag = np.clip(mag * rng.choice([-1, 1], size=mag.shape), np.iinfo(dt).min, np.iinfo(dt).max)
        px = mag.astype(dt)
        dpx = torch.from_numpy(px.view(np.dtype(f"i{dt.itemsize}"))).to(dev).view(tdt)
        enc = codec.encode(dpx, index=True)
        torch.cuda.synchronize(); enc.check()
        ntiles = (nblk + 255) // 256
        woff = (8 * frames * ntiles + 15) // 16 * 16
        ref = enc.index.cpu().numpy()
        for route in (5, 0):
            _lib.lib().trpx_set_decode_path(route)
            back, st = codec.decode(enc.stack(), enc.frame_offsets, n, frames, dt)
            torch.cuda.synchronize()
            s = st.cpu().numpy()
            ok = (back.cpu().numpy().reshape(frames, n).view(dt) == px).all()
            idx = codec.build_index(enc.stack(), enc.frame_offsets, n, frames, dt).cpu().numpy()
            wr, wg = ref[woff:woff + frames * nblk].reshape(frames, nblk), idx[woff:woff + frames * nblk].reshape(frames, nblk)
            tr, tg = ref[:8 * frames * ntiles].view(np.uint64).reshape(frames, ntiles), idx[:8 * frames * ntiles].view(np.uint64).reshape(frames, ntiles)
            badw = np.nonzero((wr != wg).any(axis=1))[0]
            badt = np.nonzero((tr != tg).any(axis=1))[0]
            print(dt.name, frames, n, "route", route, "status", s[:3].tolist(), "pixels ok", bool(ok), "frames with bad widths", badw[:8].tolist(), len(badw), "bad tile_off", badt[:8].tolist(), len(badt))
            if len(badw) and route == 0:
                f = badw[0]; b = np.nonzero(wr[f] != wg[f])[0]
                print("   frame", f, "first bad block", b[:10].tolist(), "of", nblk, "want", wr[f][b[:10]].tolist(), "got", wg[f][b[:10]].tolist(), "bytes", int(enc.frame_offsets[f + 1] - enc.frame_offsets[f]))
        _lib.lib().trpx_set_decode_path(0)
