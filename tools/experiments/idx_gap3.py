"""Third part of tools/experiments/idx_gap.py: only the address of the decode index varies -- one underlying block, one stack, one
process state (behind a noisy leg, like bench.py) -- and, for comparison, only the address of the pixel output."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from trpx_amd import codec, workloads
dev = torch.device("cuda:0")
F, N = 2000, 512 * 512
def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
cap = (F * codec.worst_case_bytes(torch.uint16, N) + 15) // 16 * 16
out = torch.empty(cap, dtype=torch.uint8, device=dev); offs = torch.empty(F + 1, dtype=torch.int64, device=dev)
st_e = torch.empty(8, dtype=torch.int32, device=dev); st_d = torch.empty(8, dtype=torch.int32, device=dev)
ws = codec.Workspace(dev)
ib = codec.index_bytes(torch.uint16, N, F)
raw_i = torch.empty(ib + (1 << 23), dtype=torch.uint8, device=dev)
raw_o = torch.empty(F * N * 2 + (1 << 23), dtype=torch.uint8, device=dev)
def at(raw, nbytes, off):
    a = (-raw.data_ptr()) % (1 << 21) + off
    return raw[a: a + nbytes]
for name, px in (("poisson3", workloads.poisson_u16(3.0, 0, F, N, device=dev)), ("synth", codec.synth(np.uint16, 0, F, N, device=dev))):
    bk0 = at(raw_o, F * N * 2, 0).view(torch.uint16).view(F, N)
    print(f"{name}: index offset behind a 2 MiB boundary -> ms (pixel output 2 MiB-aligned)")
    row = []
    for off in [0, 128, 4096, 65536, 1 << 18, 1 << 19, 3 << 18, 1 << 20, 5 << 18, 3 << 19, 7 << 18, 2067456, (1 << 21) + 0, (1 << 21) + (1 << 19), (1 << 22)]:
        ix = at(raw_i, ib, off)
        codec.encode(px, out=out, workspace=ws, frame_offsets=offs, status=st_e, index=ix)
        t = timed(lambda: codec.decode(out, offs, N, F, np.uint16, out=bk0, status=st_d, index=ix))
        assert int(st_d[0].item()) == 0 and torch.equal(bk0.view(torch.int16), px.view(torch.int16))
        row.append(f"{off}:{t:.3f}")
    print("   ", "  ".join(row))
    print(f"{name}: pixel-output offset behind a 2 MiB boundary -> ms (index 2 MiB-aligned)")
    ix = at(raw_i, ib, 0)
    codec.encode(px, out=out, workspace=ws, frame_offsets=offs, status=st_e, index=ix)
    row = []
    for off in [0, 128, 4096, 65536, 1 << 18, 1 << 19, 1 << 20, (1 << 21) + (1 << 19), (1 << 22)]:
        bk = at(raw_o, F * N * 2, off).view(torch.uint16).view(F, N)
        t = timed(lambda: codec.decode(out, offs, N, F, np.uint16, out=bk, status=st_d, index=ix))
        assert int(st_d[0].item()) == 0 and torch.equal(bk.view(torch.int16), px.view(torch.int16))
        row.append(f"{off}:{t:.3f}")
    print("   ", "  ".join(row))
    del px
