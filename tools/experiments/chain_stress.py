"""The one-launch walk under disturbed timing: the large-frame decode on one stream while another stream keeps the GPU busy with
unrelated kernels (torch elementwise / copies), and two decodes of different stacks on two streams at once.  Every result is
compared with the pixels; status[2] (frames handed to the fallback route) is reported."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from trpx_amd import codec, workloads
dev = torch.device("cuda:0")
def stack(kind):
    if kind == "p3": return workloads.poisson_u16(3.0, 0, 200, 1030 * 1065, device=dev), np.uint16
    if kind == "synth": return codec.synth(np.uint16, 0, 200, 1030 * 1065, device=dev), np.uint16
    return codec.synth(np.int32, 0, 8, 4096 * 4096, device=dev), np.int32
sets = {}
for kind in ("p3", "synth", "c4"):
    px, dt = stack(kind)
    enc = codec.encode(px); torch.cuda.synchronize(); enc.check()
    sets[kind] = (px, dt, enc, torch.empty_like(px), torch.empty(8, dtype=torch.int32, device=dev), codec.Workspace(dev))
junk = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
s_a, s_b = torch.cuda.Stream(), torch.cuda.Stream()
def dec(kind):
    px, dt, enc, back, st, ws = sets[kind]
    codec.decode(enc.data, enc.frame_offsets, px[0].numel(), px.shape[0], dt, out=back, status=st, workspace=ws)
bad = fb = 0
for it in range(30):
    for kind in ("p3", "synth", "c4"): sets[kind][3].zero_()
    torch.cuda.synchronize()
    with torch.cuda.stream(s_b):
        for _ in range(6): junk.add_(1)                      # unrelated memory-bound kernels beside the decode
        dec("synth" if it % 2 else "c4")                     # ... and another large-frame decode on the second stream
    with torch.cuda.stream(s_a):
        dec("p3")
    torch.cuda.synchronize()
    for kind in ("p3", "synth" if it % 2 else "c4"):
        px, dt, enc, back, st, ws = sets[kind]
        ok = int(st[0]) == 0 and torch.equal(back, px)
        bad += 0 if ok else 1
        fb += int(st[2])
print(f"30 rounds of concurrent decodes: mismatches {bad}, frames handed to the fallback route {fb}")
