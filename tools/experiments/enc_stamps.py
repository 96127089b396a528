"""Per-tile phase times of the single-pass encoder for a stack of any shape (u16 synth-v1): `TRPX_LIB=tools/variants/libtrpx_diag.so
python3 tools/experiments/enc_stamps.py <h> <w> <frames>` (make -C trpx_amd/csrc diag; TRPX_FUSED_DEBUG=4)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["TRPX_FUSED_DEBUG"] = "4"
import numpy as np, torch
from trpx_amd import codec
h, w, F = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
n = h * w; nb = (n + 11) // 12; t256 = (nb + 255) // 256; tpf = (nb + 1023) // 1024
px = codec.synth(np.uint16, 0, F, n)
ws = codec.Workspace("cuda")
for _ in range(3): enc = codec.encode(px, workspace=ws)
torch.cuda.synchronize()
au = lambda x, a: (x + a - 1) // a * a
tiles = F * tpf
fused = au(au(au(8 * F, 16) + 8 * F * t256, 16) + 4 * F * t256, 256)
stamp_off = fused + au(8 * (3 * tiles + 18 * F + 1), 256)
st = ws.buf[stamp_off: stamp_off + 64 * tiles].view(torch.int64).cpu().numpy().reshape(tiles, 8)
t0 = st[:, 0].min()
us = (st[:, :5] - t0) / 100.0          # 0 start, 1 barrier #1 (sizes known), 2 wave 0's look-back done, 3 barrier #2 (packed, placed), 4 end
print(f"{F} x ({h} x {w}): {tpf} tiles per frame, {tiles} tiles, kernel span {us[:, 4].max():.1f} us, tile life mean {(us[:, 4] - us[:, 0]).mean():.2f} us")
d = np.diff(us, axis=1)
for k, nme in enumerate(["load + widths + scan", "look-back (wave 0)", "packing -> barrier 2", "flush"]):
    print(f"  {nme:22s} mean {d[:, k].mean():7.2f}  p50 {np.percentile(d[:, k], 50):7.2f}  p99 {np.percentile(d[:, k], 99):7.2f}  max {d[:, k].max():7.2f}")
t_in_frame = np.arange(tiles) % tpf
for lo, hi in ((0, 1), (1, 8), (8, 32), (32, 128), (128, 100000)):
    m = (t_in_frame >= lo) & (t_in_frame < hi)
    if m.any(): print(f"  tiles {lo:3d}..{min(hi, tpf) - 1:3d} of their frame: look-back mean {d[m, 1].mean():6.2f} us, life {(us[m, 4] - us[m, 0]).mean():6.2f} us")
