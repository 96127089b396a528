"""Device-resident entry points: torch tensors in HBM -> trpx_encode / trpx_decode.

torch is plumbing only (device memory + streams); all compute happens in libtrpx_hip.so.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import torch

from . import _lib
from ._lib import check, lib

_TORCH2DT = {torch.uint8: _lib.U8, torch.int8: _lib.I8, torch.uint16: _lib.U16, torch.int16: _lib.I16,
             torch.uint32: _lib.U32, torch.int32: _lib.I32, torch.uint64: _lib.U64, torch.int64: _lib.I64}
_NP2TORCH = {np.dtype(np.uint8): torch.uint8, np.dtype(np.int8): torch.int8, np.dtype(np.uint16): torch.uint16,
             np.dtype(np.int16): torch.int16, np.dtype(np.uint32): torch.uint32, np.dtype(np.int32): torch.int32,
             np.dtype(np.uint64): torch.uint64, np.dtype(np.int64): torch.int64}
BLOCK = 12


def dtype_code(dt) -> int:
    if isinstance(dt, torch.dtype):
        return _TORCH2DT[dt]
    return _TORCH2DT[_NP2TORCH[np.dtype(dt)]]


def torch_dtype(dt) -> torch.dtype:
    return dt if isinstance(dt, torch.dtype) else _NP2TORCH[np.dtype(dt)]


def _stream_ptr(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


def worst_case_bytes(dtype, n_values: int, block: int = BLOCK) -> int:
    return lib().trpx_worst_case_bytes(dtype_code(dtype), n_values, block)


@dataclass
class Encoded:
    """A compact TERSE stack in HBM (= the reference's d_terse_data, Terse.hpp:482)."""
    data: torch.Tensor           # uint8, capacity sized; the stack is data[:total_bytes]
    frame_offsets: torch.Tensor  # int64 [n_frames + 1], byte offset of every frame, [-1] = total
    status: torch.Tensor         # int32 [8]; [0] error code, [1] prolix_bits
    n_values: int
    n_frames: int
    dtype: torch.dtype
    index: torch.Tensor | None = None   # optional decode index (trpx_encode_indexed)
    _retry: tuple | None = None         # what check() needs to run the call again (see check)

    def total_bytes(self) -> int:
        return int(self.frame_offsets[-1].item())

    def prolix_bits(self) -> int:
        return int(self.status[1].item())

    def check(self) -> None:
        """Synchronises and raises on a device error.  A look-back timeout of the single-pass encoder (TRPX_ERR_TIMEOUT,
        see trpx_encode_checked in include/trpx_hip.h) is not an error of the data: the call is run again through the
        two-pass pipeline (trpx_encode_checked), which writes the identical stream -- from the pixel tensor and workspace
        the encode was given, which this object keeps alive: ``px`` must still hold the encoded frames when ``check()`` runs
        (a caller that recycles its pixel buffer calls ``check()`` first)."""
        code = int(self.status[0].item())
        if code == _lib.ERR_TIMEOUT and self._retry is not None:
            px, ws, block = self._retry
            with torch.cuda.device(px.device):
                check(lib().trpx_encode_checked(dtype_code(px.dtype), px.data_ptr(), self.n_values, self.n_frames, block,
                                                self.data.data_ptr(), self.data.numel(), self.frame_offsets.data_ptr(),
                                                self.status.data_ptr(), self.index.data_ptr() if self.index is not None else None,
                                                ws.data_ptr(), ws.numel(), _stream_ptr(px), None))
            code = int(self.status[0].item())
        if code:
            raise _lib.TrpxError(code, "device status after encode")

    def stack(self) -> torch.Tensor:
        return self.data[: self.total_bytes()]


class Workspace:
    """Caller-owned scratch so that the hot path never allocates."""

    def __init__(self, device):
        self.device = torch.device(device)
        self.buf = None

    def get(self, nbytes: int) -> torch.Tensor:
        if self.buf is None or self.buf.numel() < nbytes:
            self.release()
            self.buf = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=self.device)
        return self.buf

    def release(self) -> None:
        """The buffer goes back to torch's allocator, which may hand its address to anything: the library must forget what it
        remembers about it (trpx_workspace_invalidate: include/trpx_hip.h, "Workspaces between calls")."""
        if self.buf is not None:
            try:
                lib().trpx_workspace_invalidate(self.buf.data_ptr(), self.buf.numel())
            except Exception:          # (interpreter shutdown)
                pass
            self.buf = None

    def __del__(self):
        self.release()


def index_bytes(dtype, n_values: int, n_frames: int, block: int = BLOCK) -> int:
    return lib().trpx_index_bytes(dtype_code(dtype), n_values, n_frames, block)


def encode(pixels: torch.Tensor, out: torch.Tensor | None = None, workspace: Workspace | None = None,
           frame_offsets: torch.Tensor | None = None, status: torch.Tensor | None = None,
           block: int = BLOCK, index: torch.Tensor | bool | None = None) -> Encoded:
    """Encode a [n_frames, n_values] (or [n_frames, H, W]) stack resident on the GPU.

    Asynchronous on the current stream; call ``Encoded.check()`` (synchronises) to test status."""
    if not pixels.is_cuda:
        raise ValueError("pixels must live on the GPU (use trpx_amd.Terse for host data)")
    px = pixels.contiguous()
    n_frames = px.shape[0]
    n_values = px[0].numel()
    code = dtype_code(px.dtype)
    dev = px.device
    if out is None:
        cap = (n_frames * worst_case_bytes(px.dtype, n_values, block) + 15) // 16 * 16
        out = torch.empty(cap, dtype=torch.uint8, device=dev)
    if frame_offsets is None:
        frame_offsets = torch.empty(n_frames + 1, dtype=torch.int64, device=dev)
    if status is None:
        status = torch.empty(_lib.STATUS_WORDS, dtype=torch.int32, device=dev)
    ws_bytes = lib().trpx_encode_workspace_bytes(code, n_values, n_frames, block)
    # (no Workspace given: a plain tensor that nobody owns beyond Encoded._retry -- a throw-away Workspace object would
    # invalidate in its destructor BEFORE the call below registers the memory as clean)
    ws = workspace.get(ws_bytes) if workspace is not None else torch.empty(max(ws_bytes, 256), dtype=torch.uint8, device=dev)
    if index is True:     # also keep the decode index (not part of the bitstream)
        index = torch.empty(index_bytes(px.dtype, n_values, n_frames, block), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        check(lib().trpx_encode_indexed(code, px.data_ptr(), n_values, n_frames, block, out.data_ptr(), out.numel(),
                                        frame_offsets.data_ptr(), status.data_ptr(),
                                        index.data_ptr() if index is not None else None, ws.data_ptr(), ws.numel(),
                                        _stream_ptr(px)))
        if workspace is None:
            # The buffer dies with the Encoded object and goes back to torch's allocator, which may hand its address to
            # anything: the library must not remember it as a clean workspace (include/trpx_hip.h, "Workspaces between
            # calls"; the registry is host-side, so forgetting right behind the launch is safe).
            check(lib().trpx_workspace_invalidate(ws.data_ptr(), ws.numel()))
    return Encoded(out, frame_offsets, status, n_values, n_frames, px.dtype, index if index is not None else None, (px, ws, block))


def decode(terse: torch.Tensor, frame_offsets: torch.Tensor | None, n_values: int, n_frames: int, dtype,
           out: torch.Tensor | None = None, workspace: Workspace | None = None,
           status: torch.Tensor | None = None, stream_signed: bool | None = None,
           block: int = BLOCK, index: torch.Tensor | None = None):
    """Decode a stack resident on the GPU. Returns (pixels [n_frames, n_values], status)."""
    tdt = torch_dtype(dtype)
    code = dtype_code(tdt)
    dev = terse.device
    if stream_signed is None:
        stream_signed = bool(lib().trpx_dtype_is_signed(code))
    if out is None:
        out = torch.empty((n_frames, n_values), dtype=tdt, device=dev)
    if status is None:
        status = torch.empty(_lib.STATUS_WORDS, dtype=torch.int32, device=dev)
    if index is not None:   # walk-free decode with a previously kept index
        with torch.cuda.device(dev):
            check(lib().trpx_decode_indexed(int(stream_signed), code, terse.data_ptr(), terse.numel(),
                                            frame_offsets.data_ptr(), index.data_ptr(), n_values, n_frames, block,
                                            out.data_ptr(), status.data_ptr(), _stream_ptr(terse)))
        return out, status
    ws_bytes = lib().trpx_decode_workspace_bytes(code, n_values, n_frames, block)
    ws = (workspace or Workspace(dev)).get(ws_bytes)
    with torch.cuda.device(dev):
        check(lib().trpx_decode(int(stream_signed), code, terse.data_ptr(), terse.numel(),
                                frame_offsets.data_ptr() if frame_offsets is not None else None,
                                n_values, n_frames, block, out.data_ptr(), status.data_ptr(), ws.data_ptr(),
                                ws.numel(), _stream_ptr(terse)))
    return out, status


def build_index(terse: torch.Tensor, frame_offsets: torch.Tensor, n_values: int, n_frames: int, dtype,
                block: int = BLOCK) -> torch.Tensor:
    """Walk an existing stack once and keep the decode index (trpx_build_index)."""
    code = dtype_code(torch_dtype(dtype))
    dev = terse.device
    index = torch.empty(index_bytes(dtype, n_values, n_frames, block), dtype=torch.uint8, device=dev)
    status = torch.empty(_lib.STATUS_WORDS, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        check(lib().trpx_build_index(code, terse.data_ptr(), terse.numel(), frame_offsets.data_ptr(), n_values,
                                     n_frames, block, index.data_ptr(), status.data_ptr(), _stream_ptr(terse)))
    return index


def synth(dtype, frame0: int, n_frames: int, n_values: int, device="cuda", seed: int = 20240807) -> torch.Tensor:
    """synth-v1 frames generated on the GPU (identical to the oracle's CPU generator)."""
    tdt = torch_dtype(dtype)
    out = torch.empty((n_frames, n_values), dtype=tdt, device=device)
    with torch.cuda.device(out.device):
        check(lib().trpx_synth_fill(dtype_code(tdt), seed, frame0, n_frames, n_values, out.data_ptr(),
                                    _stream_ptr(out)))
    return out
