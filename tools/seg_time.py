"""Decode timing per data set (decode path from $TRPX_DECODE_PATH) + position-parallel walk statistics when
$TRPX_LIB points at a -DTRPX_SEG_STATS build (status[2] = fix-point rounds, [3] = wavefront steps, [4] = lane walks)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trpx_amd import codec, _lib
L = _lib.lib()
dev = torch.device("cuda")
n = 512 * 512
g = torch.Generator(device=dev); g.manual_seed(1)

def noisy(name, frames, signed):
    bg = torch.poisson(torch.full((frames, n), 1.5, device=dev), generator=g).clamp_(0, 6).to(torch.int32)
    hot = torch.rand((frames, n), device=dev, generator=g) < (1.0 / 4096)
    px = torch.where(hot, torch.randint(0, 4000, (frames, n), device=dev, generator=g, dtype=torch.int32), bg)
    if signed: px = px - 3
    return px.to(torch.int16) if signed else px.to(torch.int16).view(torch.uint16)

def flip(frames):   # block width alternates 2 / 3 / 5 every block
    b = torch.arange(n, device=dev) // 12
    hi = torch.tensor([3, 7, 31], device=dev)[(b % 3)]
    px = (torch.randint(0, 1 << 30, (frames, n), device=dev, generator=g) % (hi + 1)).to(torch.int32)
    px[:, ::12] = hi[::12]
    return px.to(torch.int16).view(torch.uint16)

from trpx_amd import workloads
NF = int(os.environ.get("SEG_FRAMES", "2000"))   # (frames of the Poisson(3) set: 1000 = one walking wavefront per SIMD)
sets = [("poisson3 u16", lambda: workloads.poisson_u16(3.0, 0, NF, n, device=dev), np.uint16),
        ("synth-v1 u16", lambda: codec.synth(np.uint16, 0, 2000, n), np.uint16),
        ("noisy u16", lambda: noisy("u16", 2000, False), np.uint16),
        ("noisy i16 (const width)", lambda: noisy("i16", 2000, True), np.int16),
        ("flip-every-block u16", lambda: flip(1000), np.uint16),
        ("zeros u16", lambda: torch.zeros((2000, n), device=dev, dtype=torch.int16).view(torch.uint16), np.uint16),
        ("wide u16 (10..12 bit bg)", lambda: torch.randint(0, 3000, (1000, n), device=dev, generator=g, dtype=torch.int32).to(torch.int16).view(torch.uint16), np.uint16)]
only = sys.argv[1] if len(sys.argv) > 1 else ""
for name, make, dt in sets:
    if only and only not in name: continue
    px = make(); f = px.shape[0]
    ws = codec.Workspace(dev)
    enc = codec.encode(px, workspace=ws); torch.cuda.synchronize(); enc.check()
    big = torch.zeros(16 + 8 * f, dtype=torch.int32, device=dev)
    back, st = codec.decode(enc.data, enc.frame_offsets, n, f, dt, workspace=ws, status=big); torch.cuda.synchronize()
    stat = st.cpu().numpy().copy()
    if os.environ.get("SEG_PER_FRAME"):
        per = stat[16:].reshape(f, 8).astype(np.int64) & 0xFFFFFFFF
        if per[:, 1].any():
            t0 = per[:, 0].min()
            q = [0, 10, 50, 90, 99, 100]
            print("   per frame (us): start", np.percentile((per[:, 0] - t0) / 100, q).round(1), "rounds", np.percentile(per[:, 1] / 100, q).round(1),
                  "write", np.percentile(per[:, 2] / 100, q).round(1), "end", np.percentile((per[:, 0] - t0 + per[:, 1] + per[:, 2]) / 100, q).round(1))
            print("   medians (us): count passes fill", np.median(per[:, 3]) / 100, "step", np.median(per[:, 4]) / 100, "guess", np.median(per[:, 5]) / 100,
                  "| write pass fill", np.median(per[:, 6]) / 100, "step", np.median(per[:, 7]) / 100)
    st = st[:8].clone()
    ok = torch.equal(back.view(torch.uint8), px.contiguous().view(torch.uint8))
    L.trpx_profile_enable(1)
    buf = (C.c_float * 8)(); td = []
    for _ in range(5):
        codec.decode(enc.data, enc.frame_offsets, n, f, dt, out=back, workspace=ws, status=st)
        k = L.trpx_profile_read(buf, 8); td.append([buf[i] for i in range(k)])
    L.trpx_profile_enable(0)
    d = np.median(np.array(td), 0)
    print(f"{name:26s} frames {f:5d} ratio {enc.total_bytes() / (f * n * 2):.3f} decode stages ms {np.round(d, 3)} exact {ok} status {stat[0]} "
          f"rounds/frame {stat[2] / f:.1f} wave-steps/frame {stat[3] / f:.0f} lane-walks/frame {stat[4] / f:.0f}", flush=True)
    del px, enc, back
