import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from trpx_amd import codec
from oracle import oracle as O
rng = np.random.RandomState(5)
for dt, tdt in ((np.uint8, torch.uint8), (np.int8, torch.int8), (np.uint16, torch.uint16), (np.int16, torch.int16), (np.uint32, torch.uint32), (np.int32, torch.int32)):
    for n, frames in ((513 * 7, 5), (1030 * 53 + 1, 3), (12 * 300 + 7, 200), (2463 * 11, 130), (33, 140), (12 * 34000 + 3, 3)):
        top = 8 * np.dtype(dt).itemsize - (2 if np.dtype(dt).kind == "i" else (1 if np.dtype(dt).itemsize == 4 else 0))
        nblk = (n + 11) // 12
        hi = np.where(rng.rand(frames, nblk) < 0.05, rng.randint(0, top + 1, size=(frames, nblk)), 3)
        mag = (rng.rand(frames, nblk * 12) * (2.0 ** np.repeat(hi, 12, axis=1))).astype(np.int64)[:, :n]
        if np.dtype(dt).kind == "i": mag = mag * rng.choice([-1, 1], size=mag.shape)
        px = mag.astype(dt)
        want, sizes, pb = O.encode_stack(px)
        dpx = torch.from_numpy(px.view(np.dtype(f"i{np.dtype(dt).itemsize}"))).cuda().view(tdt)
        enc = codec.encode(dpx); torch.cuda.synchronize(); enc.check()
        assert enc.stack().cpu().numpy().tobytes() == want.tobytes() and enc.prolix_bits() == pb, ("encode", dt, n, frames)
        back, st = codec.decode(enc.stack(), enc.frame_offsets, n, frames, dt); torch.cuda.synchronize()
        assert int(st[0]) == 0 and (back.cpu().numpy().reshape(frames, n).view(dt) == px).all(), ("decode", dt, n, frames, int(st[0]))
    print(np.dtype(dt).name, "ok", flush=True)
print("ODD OK")
