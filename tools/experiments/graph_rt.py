"""experiment: the headline round trip (2000 x 512^2 u16 synth-v1, encode + decode) eager against a replayed HIP graph of the same two calls"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from trpx_amd import codec, _lib
dev = torch.device("cuda:0")
n, frames = 512 * 512, 2000
px = codec.synth(np.uint16, 0, frames, n, device=dev)
ws_e, ws_d = codec.Workspace(dev), codec.Workspace(dev)
cap = (frames * codec.worst_case_bytes(np.uint16, n) + 15) // 16 * 16
out = torch.empty(cap, dtype=torch.uint8, device=dev); offs = torch.empty(frames + 1, dtype=torch.int64, device=dev)
st_e = torch.empty(8, dtype=torch.int32, device=dev); st_d = torch.empty(8, dtype=torch.int32, device=dev)
back = torch.empty((frames, n), dtype=torch.uint16, device=dev)
def step():
    codec.encode(px, out=out, workspace=ws_e, frame_offsets=offs, status=st_e)
    codec.decode(out, offs, n, frames, np.uint16, out=back, workspace=ws_d, status=st_d)
for _ in range(3): step()
torch.cuda.synchronize()
def timeit(fn, k=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k
print("eager   ms per round trip", [round(timeit(step), 4) for _ in range(3)])
g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
with torch.cuda.stream(s):
    with torch.cuda.graph(g, stream=s): step()
for _ in range(3): g.replay()
torch.cuda.synchronize()
print("graph   ms per round trip", [round(timeit(g.replay), 4) for _ in range(3)])
print("exact", bool(torch.equal(back, px)), st_e[0].item(), st_d[0].item())
