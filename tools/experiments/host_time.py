"""Host-pointer path (trpx_encode_host / trpx_decode_host): PCIe- and malloc-inclusive time per 2000-frame stack."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trpx_amd import _lib, codec
from trpx_amd.terse import _code
L = _lib.lib()
frames, n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000, 512 * 512
px = codec.synth(np.uint16, 0, frames, n).cpu().numpy().view(np.uint16).reshape(frames, n)
cap = frames * L.trpx_worst_case_bytes(_lib.U16, n, 12)
out = np.empty(cap, np.uint8); offs = np.zeros(frames + 1, np.uint64); total = C.c_size_t(0); pb = C.c_uint(0)
back = np.empty((frames, n), np.uint16)
for rep in range(4):
    t0 = time.perf_counter()
    _lib.check(L.trpx_encode_host(_lib.U16, px.ctypes.data, n, frames, 12, out.ctypes.data, cap, C.byref(total), offs.ctypes.data, C.byref(pb), -1))
    t1 = time.perf_counter()
    _lib.check(L.trpx_decode_host(0, _lib.U16, out.ctypes.data, total.value, offs.ctypes.data, n, frames, 12, back.ctypes.data, -1))
    t2 = time.perf_counter()
    print(f"rep {rep}: encode_host {1e3 * (t1 - t0):7.1f} ms ({px.nbytes / (t1 - t0) / 1e9:5.1f} GB/s of pixels)  decode_host {1e3 * (t2 - t1):7.1f} ms ({px.nbytes / (t2 - t1) / 1e9:5.1f} GB/s)  exact {bool((back == px).all())}")

# per-frame expansion (src/prolix.cpp:69-92's loop): the stack stays on the device, a window of frames per device call
h = C.c_void_p()
_lib.check(L.trpx_stack_open(C.byref(h), 0, out.ctypes.data, total.value, offs.ctypes.data, None, n, frames, 12, 0, -1))
one = np.empty(n, np.uint16)
for rep in range(3):
    t0 = time.perf_counter()
    for f in range(frames):
        L.trpx_stack_read(h, f, _lib.U16, back[f].ctypes.data)
    t1 = time.perf_counter()
    print(f"rep {rep}: trpx_stack_read, {frames} frames one by one: {1e6 * (t1 - t0) / frames:7.1f} us per frame ({px.nbytes / (t1 - t0) / 1e9:5.1f} GB/s)  exact {bool((back == px).all())}")
L.trpx_stack_close(h)
# pure transfer time of the same bytes (pageable host memory, like the callers'), for the ratio
import torch
d_px = torch.empty(px.nbytes, dtype=torch.uint8, device="cuda"); d_out = torch.empty(total.value, dtype=torch.uint8, device="cuda")
hp = torch.from_numpy(px.view(np.uint8).reshape(-1)); ho = torch.from_numpy(out[: total.value])
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    d_px.copy_(hp); ho.copy_(d_out); torch.cuda.synchronize()
    t1 = time.perf_counter()
    print(f"rep {rep}: bare copies (pixels H2D + stream D2H, pageable): {1e3 * (t1 - t0):7.1f} ms")
# the same with PINNED caller memory: bare copies, and the host entry points called on pinned buffers
pp = torch.empty(px.nbytes, dtype=torch.uint8).pin_memory(); pp.copy_(hp)
po = torch.empty(total.value, dtype=torch.uint8).pin_memory()
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    d_px.copy_(pp, non_blocking=True); po.copy_(d_out, non_blocking=True); torch.cuda.synchronize()
    t1 = time.perf_counter()
    print(f"rep {rep}: bare copies (pinned): {1e3 * (t1 - t0):7.1f} ms ({px.nbytes / (t1 - t0) / 1e9:5.1f} GB/s of pixels)")
pout = torch.empty(cap, dtype=torch.uint8).pin_memory()
for rep in range(3):
    t0 = time.perf_counter()
    _lib.check(L.trpx_encode_host(_lib.U16, pp.data_ptr(), n, frames, 12, pout.data_ptr(), cap, C.byref(total), offs.ctypes.data, C.byref(pb), -1))
    t1 = time.perf_counter()
    print(f"rep {rep}: encode_host on pinned caller buffers {1e3 * (t1 - t0):7.1f} ms ({px.nbytes / (t1 - t0) / 1e9:5.1f} GB/s of pixels)")
