"""Fifth part of tools/experiments/idx_gap.py: is it the MEMORY?  idx_gap2's sequence (a Poisson(3) leg, a noisy leg, a Poisson(3)
leg: the third came out slow twice out of twice), and next to every leg's indexed decode the plain streams of trpx_bench_stream --
nothing of the codec in them -- over the very buffers that decode used: write into its pixel output, read of its stream and of its
index; the same over buffers of the same sizes allocated first thing in the process."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from trpx_amd import codec, workloads, _lib
dev = torch.device("cuda:0"); L = _lib.lib()
F, N = 2000, 512 * 512
def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
sink = torch.zeros(8, dtype=torch.int32, device=dev)
def wr(t):                                   # GB/s of a plain write stream over the tensor's bytes
    b = t.view(torch.uint8).reshape(-1); nb = b.numel() // 16 * 16
    return nb / timed(lambda: _lib.check(L.trpx_bench_stream(1, None, b.data_ptr(), nb, torch.cuda.current_stream().cuda_stream)), 10) / 1e6
def rd(t, nbytes=None):
    b = t.view(torch.uint8).reshape(-1); nb = (nbytes or b.numel()) // 16 * 16
    return nb / timed(lambda: _lib.check(L.trpx_bench_stream(0, b.data_ptr(), sink.data_ptr(), nb, torch.cuda.current_stream().cuda_stream)), 10) / 1e6
# first thing in the process: one buffer of every size the legs will use
early_px = torch.empty((F, N), dtype=torch.uint16, device=dev)
early_ix = torch.empty(codec.index_bytes(torch.uint16, N, F), dtype=torch.uint8, device=dev)
print(f"early buffers: write {wr(early_px):.0f} GB/s, read {rd(early_px):.0f} GB/s, index-sized read {rd(early_ix):.0f} GB/s", flush=True)
cap = (F * codec.worst_case_bytes(torch.uint16, N) + 15) // 16 * 16
out = torch.empty(cap, dtype=torch.uint8, device=dev); offs = torch.empty(F + 1, dtype=torch.int64, device=dev)
st_e = torch.empty(8, dtype=torch.int32, device=dev); st_d = torch.empty(8, dtype=torch.int32, device=dev)
ws, ws_d = codec.Workspace(dev), codec.Workspace(dev)
def noisy():
    g = torch.Generator(device=dev); g.manual_seed(1)
    bg = torch.poisson(torch.full((F, N), 1.5, device=dev), generator=g).clamp_(0, 6).to(torch.int32)
    hot = torch.rand((F, N), device=dev, generator=g) < (1.0 / 4096)
    return torch.where(hot, torch.randint(0, 4000, (F, N), device=dev, generator=g, dtype=torch.int32), bg).to(torch.int16).view(torch.uint16)
def leg(tag, pxl, with_free, into_early=False):
    segs = torch.cuda.memory_stats()["segment.all.allocated"]
    bk = early_px if into_early else torch.empty_like(pxl)
    en = codec.encode(pxl, out=out, workspace=ws, frame_offsets=offs, status=st_e); torch.cuda.synchronize(); total = en.total_bytes()
    if with_free: timed(lambda: codec.decode(out, offs, N, F, np.uint16, out=bk, workspace=ws_d, status=st_d))
    ix = early_ix if into_early else True
    en_i = codec.encode(pxl, out=out, workspace=ws, frame_offsets=offs, status=st_e, index=ix)
    new_segs = torch.cuda.memory_stats()["segment.all.allocated"] - segs
    t_i = timed(lambda: codec.decode(out, offs, N, F, np.uint16, out=bk, status=st_d, index=en_i.index))
    ok = int(st_d[0].item()) == 0 and torch.equal(bk.view(torch.int16), pxl.view(torch.int16))
    t_i2 = timed(lambda: codec.decode(out, offs, N, F, np.uint16, out=bk, status=st_d, index=en_i.index))
    print(f"{tag}: with index {t_i:.4f} / again {t_i2:.4f} ms exact={ok} | new segments for output+index: {new_segs} | streams over ITS buffers: "
          f"write output {wr(bk):.0f}, read pixels {rd(pxl):.0f}, read stream {rd(out, total):.0f}, read index {rd(en_i.index):.0f} GB/s", flush=True)
p = workloads.poisson_u16(3.0, 0, F, N, device=dev); leg("poisson3 first           ", p, False); del p
p = noisy();                                         leg("noisy                    ", p, True);  del p
p = workloads.poisson_u16(3.0, 0, F, N, device=dev); leg("poisson3 behind noisy    ", p, True)
leg("same pixels, output+index in the EARLY buffers", p, False, into_early=True)
leg("same pixels, allocator's buffers again        ", p, False); del p
print(f"early buffers again: write {wr(early_px):.0f} GB/s, read {rd(early_px):.0f} GB/s | reserved {torch.cuda.memory_reserved() / 2**30:.1f} GiB")
