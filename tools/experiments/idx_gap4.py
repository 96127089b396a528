"""(INCONCLUSIVE as run in round 4: the allocator split cached 2 GB blocks instead of calling hipMalloc, so the "fresh" arm never
engaged, and the slow state did not appear -- profiles/r04_idx_gap.txt.  A real test needs torch.cuda.empty_cache() + raw hipMalloc.)
Fourth part of tools/experiments/idx_gap.py: in the slow process state (behind a noisy leg, torch's caching allocator holding
recycled 1 GB blocks) -- is it the block of the decode INDEX or the block of the pixel OUTPUT that decides?  2 x 2: each of the two
either in a block the allocator recycles or in one it has to hipMalloc now (a size no cached block can serve)."""
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
ws, ws_d = codec.Workspace(dev), codec.Workspace(dev)
ib = codec.index_bytes(torch.uint16, N, F)
def noisy():
    g = torch.Generator(device=dev); g.manual_seed(1)
    bg = torch.poisson(torch.full((F, N), 1.5, device=dev), generator=g).clamp_(0, 6).to(torch.int32)
    hot = torch.rand((F, N), device=dev, generator=g) < (1.0 / 4096)
    return torch.where(hot, torch.randint(0, 4000, (F, N), device=dev, generator=g, dtype=torch.int32), bg).to(torch.int16).view(torch.uint16)
# the state bench.py is in when it reaches the Poisson(3) leg: a noisy leg has run and freed its buffers
p = noisy(); bk = torch.empty_like(p)
codec.encode(p, out=out, workspace=ws, frame_offsets=offs, status=st_e, index=True)
timed(lambda: codec.decode(out, offs, N, F, np.uint16, out=bk, workspace=ws_d, status=st_d), 5)
del p, bk
torch.cuda.synchronize()
px = workloads.poisson_u16(3.0, 0, F, N, device=dev)
seg0 = torch.cuda.memory_stats()["segment.all.allocated"]
def block(nbytes, fresh, salt):
    """fresh: a size just above anything the allocator can have cached -> a new hipMalloc; else: served from the cache."""
    before = torch.cuda.memory_stats()["segment.all.allocated"]
    t = torch.empty(nbytes + ((96 << 20) + salt * (8 << 20) if fresh else 0), dtype=torch.uint8, device=dev)
    new = torch.cuda.memory_stats()["segment.all.allocated"] - before
    return t, new
rows = []
for rep in range(2):
    for i_fresh in (False, True):
        for o_fresh in (False, True):
            it, i_new = block(ib, i_fresh, 2 * rep)
            ot, o_new = block(F * N * 2, o_fresh, 2 * rep + 1)
            ix = it[:ib]; bk = ot[: F * N * 2].view(torch.uint16).view(F, N)
            codec.encode(px, out=out, workspace=ws, frame_offsets=offs, status=st_e, index=ix)
            t = timed(lambda: codec.decode(out, offs, N, F, np.uint16, out=bk, status=st_d, index=ix))
            ok = int(st_d[0].item()) == 0 and torch.equal(bk.view(torch.int16), px.view(torch.int16))
            rows.append((rep, i_fresh, o_fresh, t))
            print(f"rep {rep}: index {'NEW malloc' if i_new else 'recycled  '} ({i_new} new segment)  output {'NEW malloc' if o_new else 'recycled  '} ({o_new} new segment)  -> {t:.4f} ms  exact={ok}", flush=True)
            del it, ot, ix, bk
print("reserved GiB:", torch.cuda.memory_reserved() / 2**30)
