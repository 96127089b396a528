import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, os.getcwd())
os.environ["TRPX_FUSED_DEBUG"] = "8"
import numpy as np, torch
from trpx_amd import codec
frames = 2000; TB = 1024; n = 512 * 512
px = codec.synth(np.uint16, 0, frames, n)
ws = codec.Workspace("cuda")
for _ in range(3): enc = codec.encode(px, workspace=ws)
torch.cuda.synchronize()
tpf = (21846 + TB - 1) // TB; tiles = frames * tpf
t256 = frames * ((21846 + 255) // 256)
fused_off = ((((8 * frames + 15) // 16 * 16) + 8 * t256 + 15) // 16 * 16 + 4 * t256 + 255) // 256 * 256
stamp_off = fused_off + (8 * (3 * tiles + 18 * frames) + 255) // 256 * 256
st = ws.buf[stamp_off: stamp_off + 64 * tiles].view(torch.int64).cpu().numpy()[: 4096 * 32].reshape(4096, 4, 8)
t0 = st[:1024, :, 0].min()
us = (st[:, :, :7] - t0) / 100.0
np.set_printoptions(linewidth=200, precision=1, suppress=True)
for t in (0, 1, 15, 256, 512, 2000, 3000):
    print("tile", t); print(us[t])
d = np.diff(us, axis=2)
print("mean round time first 256 tiles", d[:256].mean(axis=(0, 1)), " tiles 2048..4096", d[2048:].mean(axis=(0, 1)))
