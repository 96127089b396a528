"""Upper bound for overlapping the listed frames' walk with their extraction (first item): apply
tools/experiments/overlap_two_streams.patch, build decode_seg.hip with -DTRPX_OVERLAP_EXPERIMENT into a variant library, then
`TRPX_LIB=<that library> python3 tools/experiments/overlap_time.py <noisy|poisson3>`.  With TRPX_OVL set (after the first,
ordinary decode) the patched library starts k_seg_listed and k_decode_frames_indexed side by side on two streams with no
dependency between them: the extraction reads the widths the call before left, the walk writes a scratch index.
Measurement only -- the patch is not part of the product.  Result: profiles/r04_overlap_experiment.txt."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
from trpx_amd import codec
import leg_prof
leg = sys.argv[1]
dev = torch.device("cuda:0")
px, dt = leg_prof.make(leg, dev)
nf, nv = px.shape[0], px[0].numel()
ws_e, ws_d = codec.Workspace(dev), codec.Workspace(dev)
enc = codec.encode(px, workspace=ws_e); torch.cuda.synchronize(); enc.check()
back = torch.empty_like(px); st = torch.empty(8, dtype=torch.int32, device=dev)
fn = lambda: codec.decode(enc.data, enc.frame_offsets, nv, nf, dt, out=back, status=st, workspace=ws_d)
def timeit(tag):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    ok = int(st[0].item()) == 0 and torch.equal(back.view(torch.uint8), px.view(torch.uint8))
    print(f"{leg} {tag}: {e0.elapsed_time(e1)/20:.4f} ms exact={ok}", flush=True)
timeit("sequential")
os.environ["TRPX_OVL"] = "1"; timeit("overlap(extract queued first)")
os.environ["TRPX_OVL"] = "2"; timeit("overlap(walk queued first)")
del os.environ["TRPX_OVL"]; timeit("sequential again")
