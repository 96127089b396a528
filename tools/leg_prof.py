"""One benchmark leg under a profiler: `python3 tools/leg_prof.py <leg> <free|idx|enc> [reps]` encodes the leg's stack once and
then runs the chosen call `reps` times (index-free decode, decode with the decode index, or encode), so that a
`rocprofv3 --kernel-trace --stats` / `--pmc` pass of this command sees that route's kernels only.  Legs: synth, noisy, poisson3,
poisson1.5, poisson10, midsize, midsize_p3, oddsize, c4 (eight 4096^2 int32), mid2048 (128 x 2048^2 u16 synth), mid2048_p3."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trpx_amd import codec, workloads

def make(leg, dev):
    F, N = 2000, 512 * 512
    if leg == "synth": return codec.synth(np.uint16, 0, F, N, device=dev), np.uint16
    if leg == "noisy":
        g = torch.Generator(device=dev); g.manual_seed(1)
        bg = torch.poisson(torch.full((F, N), 1.5, device=dev), generator=g).clamp_(0, 6).to(torch.int32)
        hot = torch.rand((F, N), device=dev, generator=g) < (1.0 / 4096)
        px = torch.where(hot, torch.randint(0, 4000, (F, N), device=dev, generator=g, dtype=torch.int32), bg)
        return px.to(torch.int16).view(torch.uint16), np.uint16
    if leg.startswith("poisson"): return workloads.poisson_u16(float(leg[7:]), 0, F, N, device=dev), np.uint16
    if leg == "midsize": return codec.synth(np.uint16, 0, 200, 1030 * 1065, device=dev), np.uint16
    if leg == "midsize_p3": return workloads.poisson_u16(3.0, 0, 200, 1030 * 1065, device=dev), np.uint16
    if leg == "oddsize": return codec.synth(np.uint16, 0, F, 513 * 511, device=dev), np.uint16
    if leg == "c4": return codec.synth(np.int32, 0, 8, 4096 * 4096, device=dev), np.int32
    if leg == "mid2048": return codec.synth(np.uint16, 0, 128, 2048 * 2048, device=dev), np.uint16
    if leg == "mid2048_p3": return workloads.poisson_u16(3.0, 0, 128, 2048 * 2048, device=dev, chunk_frames=8), np.uint16
    raise SystemExit(f"unknown leg {leg}")

def main():
    leg, mode = sys.argv[1], sys.argv[2]
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    dev = torch.device("cuda:0")
    px, dt = make(leg, dev)
    nf, nv = px.shape[0], px[0].numel()
    ws_e, ws_d = codec.Workspace(dev), codec.Workspace(dev)
    enc = codec.encode(px, workspace=ws_e, index=True if mode == "idx" else None)
    torch.cuda.synchronize(); enc.check()
    back = torch.empty_like(px)
    st = torch.empty(8, dtype=torch.int32, device=dev)
    if mode == "enc":
        fn = lambda: codec.encode(px, out=enc.data, frame_offsets=enc.frame_offsets, status=enc.status, workspace=ws_e)
    elif mode == "idx":
        fn = lambda: codec.decode(enc.data, enc.frame_offsets, nv, nf, dt, out=back, status=st, index=enc.index)
    else:
        fn = lambda: codec.decode(enc.data, enc.frame_offsets, nv, nf, dt, out=back, status=st, workspace=ws_d)
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    ok = mode == "enc" or (int(st[0].item()) == 0 and torch.equal(back.view(torch.uint8), px.view(torch.uint8)))
    alg = nf * nv * px.element_size() + enc.total_bytes()
    ms = e0.elapsed_time(e1) / reps
    print(f"{leg} {mode}: {ms:.4f} ms per call, {alg / ms / 1e6 / 8000:.3f} of 8 TB/s on {alg} algorithmic bytes, exact={ok}, fallback frames={int(st[2].item()) if mode == 'free' else 0} status={st.tolist()}")

if __name__ == "__main__":
    main()
