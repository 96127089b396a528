"""Second half of tools/experiments/idx_gap.py: the Poisson(3) leg timed the way bench.py reaches it -- behind another leg of the
same shape, in buffers torch's caching allocator has recycled -- with the addresses of everything the indexed decode touches."""
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
def where(name, t): print(f"    {name:6s} ptr mod 2 MiB = {t.data_ptr() % (1 << 21):8d}  mod 128 = {t.data_ptr() % 128:3d}  bytes = {t.numel() * t.element_size()}")
# bench.py's shared buffers: one output buffer and one offsets tensor for every leg, sized for the headline
cap = (F * codec.worst_case_bytes(torch.uint16, N) + 15) // 16 * 16
out = torch.empty(cap, dtype=torch.uint8, device=dev); offs = torch.empty(F + 1, dtype=torch.int64, device=dev)
st_e = torch.empty(8, dtype=torch.int32, device=dev); st_d = torch.empty(8, dtype=torch.int32, device=dev)
ws, ws_d = codec.Workspace(dev), codec.Workspace(dev)
def placed(nbytes, off):                 # a uint8 tensor of nbytes whose address is `off` behind a 2 MiB boundary
    raw = torch.empty(nbytes + (1 << 22), dtype=torch.uint8, device=dev)
    a = (-raw.data_ptr()) % (1 << 21) + off
    return raw[a: a + nbytes]
def leg(tag, pxl, with_free, idx_off=None):
    bk = torch.empty_like(pxl)
    codec.encode(pxl, out=out, workspace=ws, frame_offsets=offs, status=st_e); torch.cuda.synchronize()
    t_d = timed(lambda: codec.decode(out, offs, N, F, np.uint16, out=bk, workspace=ws_d, status=st_d)) if with_free else float("nan")
    ix = True if idx_off is None else placed(codec.index_bytes(torch.uint16, N, F), idx_off)
    en_i = codec.encode(pxl, out=out, workspace=ws, frame_offsets=offs, status=st_e, index=ix)
    t_i = timed(lambda: codec.decode(out, offs, N, F, np.uint16, out=bk, status=st_d, index=en_i.index))
    ok = int(st_d[0].item()) == 0 and torch.equal(bk.view(torch.int16), pxl.view(torch.int16))
    print(f"{tag}: decode {t_d:.4f} ms, with index {t_i:.4f} ms, exact={ok}")
    where("pxl", pxl); where("bk", bk); where("index", en_i.index); where("out", out)
def noisy():
    g = torch.Generator(device=dev); g.manual_seed(1)
    bg = torch.poisson(torch.full((F, N), 1.5, device=dev), generator=g).clamp_(0, 6).to(torch.int32)
    hot = torch.rand((F, N), device=dev, generator=g) < (1.0 / 4096)
    return torch.where(hot, torch.randint(0, 4000, (F, N), device=dev, generator=g, dtype=torch.int32), bg).to(torch.int16).view(torch.uint16)
p = workloads.poisson_u16(3.0, 0, F, N, device=dev); leg("poisson3 first in the process     ", p, False); del p
p = noisy(); leg("noisy                            ", p, True); del p
p = workloads.poisson_u16(3.0, 0, F, N, device=dev); leg("poisson3 behind noisy (bench order)", p, True); del p
p = workloads.poisson_u16(3.0, 0, F, N, device=dev); leg("poisson3 once more               ", p, False); del p
p = workloads.poisson_u16(3.0, 0, F, N, device=dev)
leg("poisson3, index forced to 2 MiB + 0    ", p, False, 0)
leg("poisson3, index forced to 2 MiB + 2067456", p, False, 2067456)
leg("poisson3, index forced to 2 MiB + 4096 ", p, False, 4096)
leg("poisson3, allocator's index again      ", p, False)
del p
p = noisy()
leg("noisy, index forced to 2 MiB + 0       ", p, False, 0)
leg("noisy, index forced to 2 MiB + 2067456 ", p, False, 2067456)
del p
print("allocated GiB:", torch.cuda.memory_allocated() / 2**30, "reserved GiB:", torch.cuda.memory_reserved() / 2**30)
