"""Sixth part of tools/experiments/idx_gap.py: is it the SHADER CLOCK?  One Poisson(3) stack, one set of buffers, one process; the
only thing varied is what the GPU did just before the indexed decode is timed: nothing (fresh), two seconds of arithmetic (torch
elementwise on int64: what workloads.poisson_u16 and bench.py's generators do), three seconds of idling.  Behind every timed loop,
in-stream, tools/libclockprobe.so: a fixed chain of dependent VALU operations in one wavefront, stamped with the 100 MHz counter
(its time scales with 1 / shader clock) and with s_memtime.  Needs: hipcc ... tools/clockprobe.hip -o tools/libclockprobe.so."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from trpx_amd import codec, workloads
dev = torch.device("cuda:0")
P = ctypes.CDLL(os.path.join(ROOT, "tools", "libclockprobe.so"))
P.clock_probe.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
F, N, OPS = 2000, 512 * 512, 16000
probes = torch.zeros((8, 3), dtype=torch.int64, device=dev)
def probe(k):
    assert P.clock_probe(probes[k].data_ptr(), OPS, torch.cuda.current_stream().cuda_stream) == 0
def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    probe(0)                                     # before the loop (the GPU has just been idle for a synchronize)
    e0.record()
    for _ in range(n): fn()
    e1.record()
    probe(1); probe(2); probe(3)                 # right behind the loop, back to back
    torch.cuda.synchronize()
    pr = probes.cpu().numpy()
    chain = [pr[k][0] / 100.0 for k in range(4)]                       # us per chain (100 MHz ticks)
    ratio = [100.0 * pr[k][1] / max(1, pr[k][0]) for k in range(4)]    # s_memtime ticks per us
    return e0.elapsed_time(e1) / n, chain, ratio
px = workloads.poisson_u16(3.0, 0, F, N, device=dev)
sy = codec.synth(np.uint16, 0, F, N, device=dev)
cap = (F * codec.worst_case_bytes(torch.uint16, N) + 15) // 16 * 16
bufs = {}
for name, p in (("poisson3", px), ("synth", sy)):
    out = torch.empty(cap, dtype=torch.uint8, device=dev)
    enc = codec.encode(p, out=out, index=True); torch.cuda.synchronize(); enc.check()
    bufs[name] = (p, out, enc.frame_offsets, enc.index, torch.empty_like(p))
st = torch.empty(8, dtype=torch.int32, device=dev)
def dec(name):
    p, out, fo, ix, bk = bufs[name]
    return lambda: codec.decode(out, fo, N, F, np.uint16, out=bk, status=st, index=ix)
def show(tag):
    row = []
    for name in ("poisson3", "synth"):
        ms, chain, ratio = timed(dec(name))
        row.append(f"{name} {ms:.4f} ms [chain us before {chain[0]:.1f}, behind {chain[1]:.1f} {chain[2]:.1f} {chain[3]:.1f}; s_memtime/us {ratio[0]:.0f} {ratio[1]:.0f}]")
    print(f"{tag:34s} " + " | ".join(row), flush=True)
def burn(seconds):                               # int64 elementwise arithmetic, like the generators
    a = torch.arange(1, 1 + (1 << 26), dtype=torch.int64, device=dev); t0 = time.time()
    while time.time() - t0 < seconds:
        for _ in range(20): a = (a ^ (a >> 30)) * -4658895280553007687 + 7
        torch.cuda.synchronize()
    return int(a[0].item())
torch.cuda.synchronize(); time.sleep(2.0)
show("fresh (after 2 s idle)")
show("again at once")
burn(2.0); show("right after 2 s of arithmetic")
show("again at once")
time.sleep(3.0); show("after 3 s of idling")
burn(4.0); show("right after 4 s of arithmetic")
time.sleep(0.5); show("0.5 s later")
time.sleep(6.0); show("after 6 s of idling")
p, out, fo, ix, bk = bufs["poisson3"]
assert int(st[0].item()) == 0
dec("poisson3")(); torch.cuda.synchronize()
print("exact:", torch.equal(bk.view(torch.int16), p.view(torch.int16)))
