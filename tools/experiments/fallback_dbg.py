"""Why do these frames take the fallback route?  `TRPX_LIB=tools/variants/libtrpx_partstats.so python3 tools/experiments/fallback_dbg.py`"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from trpx_amd import codec
rng = np.random.RandomState(5)
def run(name, a):
    px = torch.from_numpy(a).cuda()
    enc = codec.encode(px); torch.cuda.synchronize(); enc.check()
    back, st = codec.decode(enc.stack(), enc.frame_offsets, a.shape[1], a.shape[0], a.dtype)
    torch.cuda.synchronize()
    print(f"== {name}: status {st.cpu().numpy()} exact {torch.equal(back, px)} bytes/frame {enc.total_bytes() // a.shape[0]}", flush=True)
n = 840003
run("int16 all zero 2048 x 2048", np.zeros((2, 1 << 22), dtype=np.int16))
z = np.zeros((3, 1030 * 1065), dtype=np.uint16); z[1, 500000:500500] = 5; z[2, ::7001] = 300
run("u16 1030 x 1065: zero, zero with a stripe, zero with sparse peaks", z)
run("u8 pedestal 96 + 0..7", (96 + (rng.rand(3, n) * 8)).astype(np.uint8))
a = (96 + (rng.rand(3, n) * 8)).astype(np.uint8); a[:, : n // 2] = 0
run("u8 pedestal, first half empty", a)
a = (40 + (rng.rand(9, n) * 2)).astype(np.uint8)
run("u8 pedestal 40 + 0..1", a)
for w in (5, 9, 13):
    run(f"int16 one width {w}", ((rng.rand(2, 1 << 22) * (1 << w)).astype(np.int64) * rng.choice([-1, 1], size=(2, 1 << 22))).astype(np.int16))
