"""configs[3]: 4096x4096 int32 synth-v1 frames with sparse peaks -- encode / decode timing + parity anchors."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trpx_amd import codec, _lib
L = _lib.lib()
frames, n = 8, 4096 * 4096
px = codec.synth(np.int32, 0, frames, n)
ws = codec.Workspace("cuda")
enc = codec.encode(px, workspace=ws); torch.cuda.synchronize(); enc.check()
total = enc.total_bytes()
back, st = codec.decode(enc.data, enc.frame_offsets, n, frames, np.int32, workspace=ws); torch.cuda.synchronize()
assert int(st[0].item()) == 0 and torch.equal(back, px)
if os.environ.get("SEG_PER_WAVE"):      # a -DTRPX_SEG_STAMPS build: per wavefront of the first k_seg_round launch
    big = torch.zeros(16 + 8 * 4096, dtype=torch.int32, device="cuda")
    codec.decode(enc.data, enc.frame_offsets, n, frames, np.int32, out=back, workspace=ws, status=big); torch.cuda.synchronize()
    per = big.cpu().numpy()[16:].reshape(-1, 8).astype(np.int64) & 0xFFFFFFFF
    per = per[per[:, 1] > 0]
    q = [0, 10, 50, 90, 100]
    t0 = per[:, 0].min()
    print("waves", len(per), "start us", np.percentile((per[:, 0] - t0) / 100, q).round(1), "total us", np.percentile(per[:, 1] / 100, q).round(1),
          "fill", np.percentile(per[:, 3] / 100, q).round(1), "step", np.percentile(per[:, 4] / 100, q).round(1),
          "guess", np.percentile(per[:, 5] / 100, q).round(1), "rounds", np.percentile(per[:, 6], q))
L.trpx_profile_enable(1)
buf = (C.c_float * 8)(); e, d = [], []
for _ in range(5):
    codec.encode(px, out=enc.data, workspace=ws, frame_offsets=enc.frame_offsets, status=enc.status)
    k = L.trpx_profile_read(buf, 8); e.append([buf[i] for i in range(k)])
    codec.decode(enc.data, enc.frame_offsets, n, frames, np.int32, out=back, workspace=ws, status=st)
    k = L.trpx_profile_read(buf, 8); d.append([buf[i] for i in range(k)])
e, d = np.median(np.array(e), 0), np.median(np.array(d), 0)
pix = frames * n * 4
print("size frame0", int(enc.frame_offsets[1]), "total", total, "prolix_bits", enc.prolix_bits())
print("encode stages ms", e, "-> fps", frames / e.sum() * 1e3, "pixel GB/s", pix / e.sum() / 1e6, "algorithmic GB/s", (pix + total) / e.sum() / 1e6)
print("decode stages ms", d, "-> fps", frames / d.sum() * 1e3, "pixel GB/s", pix / d.sum() / 1e6)
# row f1 for files: chain states at every 256th block -> index -> walk-free decode (what a reader of a `terse -index` file does)
enc_i = codec.encode(px, workspace=ws, index=True); torch.cuda.synchronize()
ng = L.trpx_group_count(n, 12)
states = torch.zeros(frames * ng, dtype=torch.int64, device="cuda")
_lib.check(L.trpx_index_group_states(enc_i.index.data_ptr(), n, frames, 12, states.data_ptr(), None))
idx = torch.zeros_like(enc_i.index)
st8 = torch.zeros(8, dtype=torch.int32, device="cuda")
stack = enc_i.stack()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
tt = []
for _ in range(5):
    ev[0].record()
    _lib.check(L.trpx_index_from_group_states(codec.dtype_code(np.int32), stack.data_ptr(), stack.numel(), enc_i.frame_offsets.data_ptr(),
                                              states.data_ptr(), n, frames, 12, idx.data_ptr(), st8.data_ptr(), None))
    ev[1].record()
    back2, st2 = codec.decode(stack, enc_i.frame_offsets, n, frames, np.int32, index=idx, out=back, workspace=ws)
    ev[2].record(); torch.cuda.synchronize()
    tt.append((ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2])))
assert int(st8[0]) == 0 and int(st2[0]) == 0 and torch.equal(back2, px)
tt = np.median(np.array(tt), 0)
print("grouped decode: states->index ms", tt[0], "indexed unpack ms", tt[1], "states bytes", 8 * frames * ng, "of", total)
