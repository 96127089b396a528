"""Encode throughput against the stack's shape (synth-v1, u16): frames x pixels per frame."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from trpx_amd import codec
dev = torch.device("cuda:0")
def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (F, h, w) in ((2000, 512, 512), (200, 1030, 1065), (200, 1024, 1024), (200, 1024, 1072), (400, 724, 724), (800, 512, 520), (50, 2048, 2048), (1000, 512, 512), (500, 512, 512)):
    N = h * w
    px = codec.synth(np.uint16, 0, F, N, device=dev)
    ws = codec.Workspace(dev)
    enc = codec.encode(px, workspace=ws); torch.cuda.synchronize(); enc.check()
    t = timed(lambda: codec.encode(px, out=enc.data, frame_offsets=enc.frame_offsets, status=enc.status, workspace=ws))
    gb = F * N * 2 / 1e9
    print(f"{F} x ({h} x {w}): {gb:.3f} GB, {(N + 11) // 12} blocks = {((N + 11) // 12) / 1024:.1f} tiles per frame: encode {t:.4f} ms = {gb / t:.2f} TB/s of pixels", flush=True)
    del px, enc
