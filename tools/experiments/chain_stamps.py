"""Per-part times of k_chain_walk (decode_part.hip, -DTRPX_CHAIN_STAMPS): `TRPX_LIB=tools/variants/libtrpx_chainstamps.so python3
tools/chain_stamps.py <leg>`.  Ticks are 10 ns (s_memrealtime)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trpx_amd import codec, _lib
from leg_prof import make
leg = sys.argv[1]
dev = torch.device("cuda:0")
px, dt = make(leg, dev)
nf, nv = px.shape[0], px[0].numel()
enc = codec.encode(px); torch.cuda.synchronize(); enc.check()
P = _lib.lib().trpx_decode_parts_per_frame(codec.dtype_code(dt), nv, nf, 12)
st = torch.zeros(16 + 8 * nf * P, dtype=torch.int32, device=dev)
back = torch.empty_like(px)
ws = codec.Workspace(dev)
for _ in range(3):
    codec.decode(enc.data, enc.frame_offsets, nv, nf, dt, out=back, status=st, workspace=ws)
torch.cuda.synchronize()
s = st.cpu().numpy().astype(np.uint32)[16:].reshape(nf, P, 8)[:, : P - 1].reshape(-1, 8)
t0 = (s[:, 0] - s[:, 0].min()).astype(np.int64)
setup, warm, walk, nfill, link, cnt, nck = (s[:, i].astype(np.int64) for i in range(1, 8))
end = t0 + setup + warm + walk
end2 = end + link
q = lambda a: "min %6.1f  p50 %6.1f  p90 %6.1f  max %6.1f" % tuple(np.percentile(a, [0, 50, 90, 100]) / 100.0)
print(f"{leg}: {nf} frames x {P} parts, exact {torch.equal(back.view(torch.uint8), px.view(torch.uint8))}, status {int(st[0])}")
print("  start  us:", q(t0))
print("  guess  us:", q(setup))
print("  wait   us:", q(warm), " (for the neighbour's start state)")
print("  walk   us:", q(walk))
print("  link   us:", q(link), " (the wait for the neighbour's record + the walk into its part; links with work: %d)" % int((link > 300).sum()))
print("  fills    : n p50 %d max %d" % (np.median(nfill), nfill.max()))
print("  done   us:", q(end2))
print("  walked us:", q(end), "  blocks per part p50 %d max %d, checkpoints p50 %d" % (np.median(cnt), cnt.max(), np.median(nck)))
order = np.argsort(-walk)[:8]
for i in order:
    print("  slow part: frame %d part %d  walk %.1f us  link %.1f us  blocks %d  checkpoints %d  fills %d" %
          (i // (P - 1), i % (P - 1), walk[i] / 100.0, link[i] / 100.0, cnt[i], nck[i], nfill[i]))
order = np.argsort(-end2)[:6]
for i in order:
    print("  last done: frame %d part %d  walked at %.1f us  link %.1f us  done %.1f us" % (i // (P - 1), i % (P - 1), end[i] / 100.0, link[i] / 100.0, end2[i] / 100.0))
