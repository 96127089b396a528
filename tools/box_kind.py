"""Which kind of box is this?  `python3 tools/box_kind.py`: the aligned and the 2-byte-misaligned write stream of trpx_bench_stream
(modes 1 and 3) next to the decode times of the stack whose frames start inside a cache line (2000 x (513 x 511) u16), walker and
index route, and of the line-aligned stack -- one line per run, to be compared ACROSS gpurun calls (every call lands on another
box of the pool).  Round 4, nine boxes: profiles/r04_box_kind.txt, DESIGN.md section 8."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trpx_amd import codec, _lib
dev = torch.device("cuda:0"); L = _lib.lib()
def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
nbytes = 2000 * 512 * 512 * 2
buf = torch.empty(nbytes + 4096, dtype=torch.uint8, device=dev)
s_ = torch.cuda.current_stream().cuda_stream
w_al = nbytes / timed(lambda: _lib.check(L.trpx_bench_stream(1, None, buf.data_ptr(), nbytes, s_)), 10) / 1e6
w_mis = nbytes / timed(lambda: _lib.check(L.trpx_bench_stream(3, None, buf.data_ptr() + 2, nbytes, s_)), 10) / 1e6
del buf
res = {}
for name, n in (("odd", 513 * 511), ("aligned", 512 * 512)):
    px = codec.synth(np.uint16, 0, 2000, n, device=dev)
    ws_e, ws_d = codec.Workspace(dev), codec.Workspace(dev)
    enc = codec.encode(px, workspace=ws_e, index=True); torch.cuda.synchronize(); enc.check()
    back = torch.empty_like(px); st = torch.empty(8, dtype=torch.int32, device=dev)
    res[name + "_walker"] = timed(lambda: codec.decode(enc.data, enc.frame_offsets, n, 2000, np.uint16, out=back, status=st, workspace=ws_d))
    res[name + "_index"] = timed(lambda: codec.decode(enc.data, enc.frame_offsets, n, 2000, np.uint16, out=back, status=st, index=enc.index))
    assert int(st[0].item()) == 0 and torch.equal(back.view(torch.int16), px.view(torch.int16))
    del px, enc, back
print(f"BOX write {w_al:.0f} GB/s  misaligned {w_mis:.0f} GB/s | odd-size: walker {res['odd_walker']:.4f} index {res['odd_index']:.4f} ms"
      f" | aligned: walker {res['aligned_walker']:.4f} index {res['aligned_index']:.4f} ms | {torch.cuda.get_device_name(0)}")
