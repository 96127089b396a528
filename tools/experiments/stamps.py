"""Diagnostic: per-tile s_memrealtime stamps of the single-pass encoder (TRPX_FUSED_DEBUG=4)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["TRPX_FUSED_DEBUG"] = os.environ.get("TRPX_FUSED_DEBUG", "4")
import numpy as np, torch
from trpx_amd import codec, _lib
L = _lib.lib()
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
TB = int(os.environ.get("TRPX_TILE_BLOCKS", "1024"))
n = 512 * 512
px = codec.synth(np.uint16, 0, frames, n)
ws = codec.Workspace("cuda")
for _ in range(3):
    enc = codec.encode(px, workspace=ws)
torch.cuda.synchronize()
tpf = (21846 + TB - 1) // TB
tiles = frames * tpf
# workspace layout (api.hip enc_ws): [two-pass arrays][fused descriptors][stamps]
t256 = frames * ((21846 + 255) // 256)
fused_off = ((((8 * frames + 15) // 16 * 16) + 8 * t256 + 15) // 16 * 16 + 4 * t256 + 255) // 256 * 256
stamp_off = fused_off + (8 * (3 * tiles + 18 * frames) + 255) // 256 * 256
st = ws.buf[stamp_off: stamp_off + 64 * tiles].view(torch.int64).cpu().numpy().reshape(tiles, 8)
t0 = st[:, 0].min()
us = (st[:, :5] - t0) / 100.0      # slots: 0 start, 1 barrier#1, 2 wave-0 look-back done, 3 barrier#2, 4 end; 7 loads issued, 5 wave 0 at barrier#1
print("kernel span us", us[:, 4].max())
d = np.diff(us, axis=1)
names = ["load+scan(->b1)", "lookback(b1->lb)", "lb->b2", "flush(b2->end)"]
print("start->loads issued mean %.2f, ->wave0 scans done mean %.2f" % (((st[:, 7] - st[:, 0]) / 100.0).mean(), ((st[:, 5] - st[:, 0]) / 100.0).mean()))
for k, nme in enumerate(names):
    print(f"{nme:18s} mean {d[:, k].mean():8.2f} p50 {np.percentile(d[:, k], 50):8.2f} p99 {np.percentile(d[:, k], 99):8.2f} max {d[:, k].max():8.2f}")
print("tile total       mean", (us[:, 4] - us[:, 0]).mean())
# dispatch order: start time vs tile index
start = us[:, 0]
inv = (np.diff(start) < -0.5).mean()
print("fraction of consecutive tiles starting >0.5us out of order:", inv)
for x in range(8):
    m = st[:, 6] == x
    print("xcc", x, "tiles", m.sum(), "first idx%8 seen", np.unique(np.arange(tiles)[m] % 8)[:8], "mean start", start[m].mean())
# how far ahead of its predecessor does a tile start / finish its load phase?
lag = us[1:, 1] - us[:-1, 1]
print("b1 time minus predecessor's b1: mean %.2f p1 %.2f p50 %.2f p99 %.2f" % (lag.mean(), np.percentile(lag, 1), np.percentile(lag, 50), np.percentile(lag, 99)))
np.save("gpurun_out/stamps.npy", st)
