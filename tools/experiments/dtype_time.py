"""Encode / decode time per pixel type on a synthetic 2000 x 512 x 512 stack (background 0..6, sparse peaks)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trpx_amd import codec, _lib
L = _lib.lib()
frames, n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000, 512 * 512
dev = torch.device("cuda")
g = torch.Generator(device=dev); g.manual_seed(1)
for name, tdt, peak in (("uint8", torch.uint8, 200), ("int8", torch.int8, 100), ("uint16", torch.uint16, 4000), ("int16", torch.int16, 4000),
                        ("uint32", torch.uint32, 1 << 24), ("int32", torch.int32, 1 << 24)):
    f = frames if tdt.itemsize <= 2 else frames // 2
    bg = torch.poisson(torch.full((f, n), 1.5, device=dev), generator=g).clamp_(0, 6).to(torch.int32)
    hot = torch.rand((f, n), device=dev, generator=g) < (1.0 / 4096)
    px = torch.where(hot, torch.randint(0, peak, (f, n), device=dev, generator=g, dtype=torch.int32), bg)
    if name.startswith("int"):
        px = px - 3
    px = px.to(torch.int64).to({"uint8": torch.uint8, "int8": torch.int8, "uint16": torch.int32, "int16": torch.int16,
                                "uint32": torch.int64, "int32": torch.int32}[name])
    if name == "uint16": px = px.to(torch.int32).to(torch.int16).view(torch.uint16)
    if name == "uint32": px = px.to(torch.int32).view(torch.uint32)
    ws = codec.Workspace(dev)
    enc = codec.encode(px, workspace=ws); torch.cuda.synchronize(); enc.check()
    back, st = codec.decode(enc.data, enc.frame_offsets, n, f, np.dtype(name), workspace=ws); torch.cuda.synchronize()
    ok = torch.equal(back.view(torch.uint8), px.contiguous().view(torch.uint8))
    L.trpx_profile_enable(1)
    buf = (C.c_float * 8)(); te, td = [], []
    for _ in range(7):
        codec.encode(px, out=enc.data, workspace=ws, frame_offsets=enc.frame_offsets, status=enc.status)
        k = L.trpx_profile_read(buf, 8); te.append(sum(buf[i] for i in range(k)))
        codec.decode(enc.data, enc.frame_offsets, n, f, np.dtype(name), out=back, workspace=ws, status=st)
        k = L.trpx_profile_read(buf, 8); td.append(sum(buf[i] for i in range(k)))
    L.trpx_profile_enable(0)
    e, d = float(np.median(te)), float(np.median(td))
    pix = f * n * tdt.itemsize
    print(f"{name:7s} frames {f:5d} ratio {enc.total_bytes() / pix:.3f} encode {e:.3f} ms ({pix / e / 1e6:7.0f} GB/s pixels) decode {d:.3f} ms ({pix / d / 1e6:7.0f} GB/s) exact {ok}")
    del px, enc, back, bg, hot
