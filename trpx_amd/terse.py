"""Host-side mirror of the reference class ``jpa::Terse`` (include/Terse.hpp:228-475).

Same surface and argument meaning -- construct from data, ``push_back`` frames, ``prolix`` a
frame into caller memory, ``write`` / read the ``.trpx`` stream -- but every encode / decode runs
on the MI355X through the C ABI (``trpx_encode_host`` / ``trpx_decode_host``).  The compressed
bytes live on the host, as in the reference (``d_terse_data``, Terse.hpp:482).  The C++ mirror
of the same class is ``include/trpx/Terse.hpp``.

Differences, all deliberate:
  * frames are located by the running sum of their sizes (the *intended* semantics of
    Terse.hpp:562-585; the reference's cached offsets are wrong for frame >= 2, defect D1);
  * argument errors raise ``ValueError`` where the reference ``assert``s (compiled out in its
    Release build);
  * ``push_back_stack`` / ``prolix_stack`` move many frames in one GPU call (no O(F^2), D6).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, lib

_NP2DT = {np.dtype(np.uint8): _lib.U8, np.dtype(np.int8): _lib.I8, np.dtype(np.uint16): _lib.U16,
          np.dtype(np.int16): _lib.I16, np.dtype(np.uint32): _lib.U32, np.dtype(np.int32): _lib.I32,
          np.dtype(np.uint64): _lib.U64, np.dtype(np.int64): _lib.I64}


_NP2DT_OUT = dict(_NP2DT)
_NP2DT_OUT.update({np.dtype(np.float32): _lib.F32, np.dtype(np.float64): _lib.F64})


def _code(dt, decode: bool = False) -> int:
    try:
        return (_NP2DT_OUT if decode else _NP2DT)[np.dtype(dt)]
    except KeyError:
        raise TypeError(f"Terse: unsupported pixel type {dt} (encode: u8/i8/u16/i16/u32/i32/u64/i64; decode also "
                        f"float32/float64)") from None


class Terse:
    def __init__(self, data=None, block: int = 12, device: int = -1):
        """``Terse()`` (Terse.hpp:237) or ``Terse(container)`` (Terse.hpp:249-253, :263-270)."""
        self._block = int(block)
        self._device = device
        self._signed = False
        self._size = 0
        self._prolix_bits = 0
        self._dim: list[int] = []
        self._data = bytearray()
        self._stack, self._stack_len = None, -1      # device-resident copy of the stack (prolix), see _drop_stack
        self._group_states = None                    # chain state at every 256th block of every frame (row f1), or None
        self._frame_sizes: list[int] = []
        if data is not None:
            self.push_back(data)

    # ---- encode -----------------------------------------------------------------------------
    def push_back(self, data) -> None:
        """Append one frame (Terse.hpp:290-302, :312-322)."""
        a = np.ascontiguousarray(data)
        if a.ndim > 1:
            if not self._frame_sizes and not self._dim:
                self._dim = list(a.shape)                   # captures dim() (Terse.hpp:251-252, :314-317)
            elif self._dim and list(a.shape) != self._dim:
                raise ValueError("each frame of a multi-Terse object must have the same dimensions")  # :319
        self.push_back_stack(a.reshape(1, -1))

    def push_back_stack(self, frames) -> None:
        """Append a [n_frames, ...] stack in ONE GPU call (what a detector pipeline would do)."""
        a = np.ascontiguousarray(frames)
        a = a.reshape(a.shape[0], -1)
        if a.dtype in (np.dtype(np.int64), np.dtype(np.uint64)):
            # 64-bit integers (what src/terse.cpp:120-123 makes of float images): values that fit 32 bits give the same
            # stream whatever the container type, so they are narrowed here and take the tuned kernels; a stack with wider
            # values goes through as 64-bit pixels (generic kernels, fields of up to 64 bits)
            narrow = np.int32 if a.dtype.kind == "i" else np.uint32
            if not (a.size and (a.min() < np.iinfo(narrow).min or a.max() > np.iinfo(narrow).max)):
                a = a.astype(narrow)
        code = _code(a.dtype)
        n_frames, n = a.shape
        if n_frames == 0:
            return
        if self._frame_sizes:
            if n != self._size:
                raise ValueError("each frame of a multi-Terse object must have the same size")   # Terse.hpp:297
            if (a.dtype.kind == "i") != self._signed:
                raise ValueError("signedness differs from the first frame")                      # Terse.hpp:298
        cap = n_frames * lib().trpx_worst_case_bytes(code, n, self._block)
        out = np.empty(cap, np.uint8)
        total = C.c_size_t(0)
        offs = np.empty(n_frames + 1, np.uint64)
        pb = C.c_uint(0)
        check(lib().trpx_encode_host(code, a.ctypes.data, n, n_frames, self._block, out.ctypes.data, cap,
                                     C.byref(total), offs.ctypes.data, C.byref(pb), self._device))
        if not self._frame_sizes:
            self._size = n
            self._signed = a.dtype.kind == "i"
        self._data += out[: total.value].tobytes()
        self._group_states = None
        self._frame_sizes += [int(x) for x in np.diff(offs)]
        self._prolix_bits = max(self._prolix_bits, int(pb.value))   # Terse.hpp:516

    # ---- decode -----------------------------------------------------------------------------
    def prolix(self, out: np.ndarray, frame: int = 0) -> np.ndarray:
        """Unpack frame `frame` into `out` (Terse.hpp:333-341, :352-389).  `out` may have any supported type:
        narrower integers clamp (Bit_pointer.hpp:747-763), float / double are exact (Terse.hpp:379-383)."""
        if not 0 <= frame < self.number_of_frames():
            raise ValueError("frame index out of range")                                          # Terse.hpp:354
        if out.size != self._size:
            raise ValueError("output container has the wrong size")                               # Terse.hpp:335
        if self._signed and out.dtype.kind == "u":
            raise ValueError("signed data cannot be decompressed into unsigned data")            # Terse.hpp:356-357
        if not out.flags.c_contiguous:
            raise ValueError("output must be contiguous")
        # src/prolix.cpp:69-92 calls this once per frame: the stack is uploaded once and kept on the device (trpx_stack_*),
        # a window of frames is expanded per device call and a call normally only copies its frame back
        if self._stack is None or self._stack_len != len(self._data):
            self._drop_stack()
            buf = np.frombuffer(self._data, np.uint8)
            offs = np.concatenate([[0], np.cumsum(self._frame_sizes)]).astype(np.uint64)
            h = C.c_void_p()
            gs = self._states()
            check(lib().trpx_stack_open(C.byref(h), int(self._signed), buf.ctypes.data, buf.size, offs.ctypes.data,
                                        gs.ctypes.data if gs is not None else None, self._size,
                                        len(self._frame_sizes), self._block, 0, self._device))
            self._stack, self._stack_len = h, len(self._data)
        check(lib().trpx_stack_read(self._stack, frame, _code(out.dtype, True), out.ctypes.data))
        return out

    def _states(self):
        """The group states if they belong to the object's current frames, else None."""
        groups = lib().trpx_group_count(self._size, self._block) if self._size else 0
        gs = self._group_states
        return gs if gs is not None and groups and gs.size == groups * len(self._frame_sizes) else None

    def has_group_index(self) -> bool:
        return self._states() is not None

    def _drop_stack(self) -> None:
        if getattr(self, "_stack", None) is not None:
            lib().trpx_stack_close(self._stack)
        self._stack, self._stack_len = None, -1

    def __del__(self):
        try:
            self._drop_stack()
        except Exception:
            pass

    # The device-side copy of the stack (a raw trpx_stack*) belongs to ONE object: copies and pickles start without it.
    def __getstate__(self):
        st = dict(self.__dict__)
        st["_stack"], st["_stack_len"] = None, -1
        return st

    def __setstate__(self, st):
        self.__dict__.update(st)
        self._stack, self._stack_len = None, -1

    def __copy__(self):
        t = Terse.__new__(Terse)
        t.__setstate__(self.__getstate__())
        return t

    def __deepcopy__(self, memo):
        import copy
        t = Terse.__new__(Terse)
        t.__setstate__({k: copy.deepcopy(v, memo) for k, v in self.__getstate__().items()})
        return t

    def prolix_stack(self, dtype) -> np.ndarray:
        """Decode every frame in ONE GPU call; returns [n_frames, size]."""
        f = self.number_of_frames()
        out = np.empty((f, self._size), np.dtype(dtype))
        buf = np.frombuffer(self._data, np.uint8)
        offs = np.concatenate([[0], np.cumsum(self._frame_sizes)]).astype(np.uint64)
        gs = self._states()
        check(lib().trpx_decode_host_grouped(int(self._signed), _code(dtype, True), buf.ctypes.data, buf.size, offs.ctypes.data,
                                             gs.ctypes.data if gs is not None else None, self._size, f, self._block,
                                             out.ctypes.data, self._device))
        return out

    # ---- accessors (Terse.hpp:396-444) --------------------------------------------------------
    def size(self) -> int:
        return self._size

    def number_of_frames(self) -> int:
        return len(self._frame_sizes)

    def is_signed(self) -> bool:
        return self._signed

    def bits_per_val(self) -> int:
        return self._prolix_bits

    def terse_size(self) -> int:
        return len(self._data)

    def imagej_readable(self) -> bool:
        """True if the reference's ImageJ plugin accepts a file written from this object: it only reads unsigned data
        of at most 16 bits per value (ImageJ/TRPX_Reader.java:94-98) and the whole file must fit a Java byte array."""
        return (not self._signed) and self._prolix_bits <= 16 and len(self._data) < (1 << 31) - 4096

    def frame_sizes(self) -> list[int]:
        return list(self._frame_sizes)

    def data(self) -> bytes:
        return bytes(self._data)

    def dim(self, dim=None):
        if dim is not None:
            if self._dim:
                raise ValueError("you cannot overwrite the dimensionality of a frame")           # Terse.hpp:419
            self._dim = [int(d) for d in dim]
        return list(self._dim)

    # ---- stream-serialise surface (Terse.hpp:454-474, :279, :485-498) -----------------------
    def header(self, frame_index: bool = False) -> bytes:
        """Header text of ``write`` (Terse.hpp:454-470).  ``frame_index=True`` adds the ``frame_sizes`` and
        ``group_bit_offsets`` attributes: ignored by the reference reader, they let ``read``
        locate the frames and ``prolix`` expand them without any header walk."""
        h = _lib.trpx_header()
        h.prolix_bits, h.is_signed, h.block = self._prolix_bits, int(self._signed), self._block
        h.memory_size, h.number_of_values = len(self._data), self._size
        h.number_of_frames = self.number_of_frames()
        h.n_dims = len(self._dim)
        for i, d in enumerate(self._dim[:8]):
            h.dims[i] = d
        gs = None
        # (values of more than 32 bits -- 64-bit containers on the generic kernels -- have no decode index: frame sizes only)
        if frame_index and self._frame_sizes and self._prolix_bits <= 32 and lib().trpx_group_count(self._size, self._block):
            gs = self._states()
            if gs is None:                                   # compute them once: one walk of the stack on the device
                data = np.frombuffer(self._data, np.uint8)
                offs = np.concatenate([[0], np.cumsum(self._frame_sizes)]).astype(np.uint64)
                gs = np.zeros(lib().trpx_group_count(self._size, self._block) * len(self._frame_sizes), np.uint64)
                max_bits = 8 if self._prolix_bits <= 8 else 16 if self._prolix_bits <= 16 else 32
                rc = lib().trpx_group_states_host(data.ctypes.data, data.size, offs.ctypes.data, self._size, len(self._frame_sizes),
                                                  self._block, max_bits, gs.ctypes.data, self._device)
                del data
                if rc == 0:
                    self._group_states = gs
                elif rc == _lib.ERR_UNSUPPORTED:
                    gs = None                                # no group index for this stack: the file still gets its frame sizes
                else:
                    check(rc)                                # device fault, no device, corrupt stack: not "no index"
        cap = 512 + (21 * len(self._frame_sizes) + (16 * gs.size if gs is not None else 0) if frame_index else 0)
        buf = C.create_string_buffer(cap)
        if frame_index and gs is not None:
            sizes = np.asarray(self._frame_sizes, np.uint64)
            n = lib().trpx_header_format_grouped(C.byref(h), sizes.ctypes.data, sizes.size, gs.ctypes.data, gs.size, buf, cap)
        elif frame_index:
            sizes = np.asarray(self._frame_sizes, np.uint64)
            n = lib().trpx_header_format_indexed(C.byref(h), sizes.ctypes.data, sizes.size, buf, cap)
        else:
            n = lib().trpx_header_format(C.byref(h), buf, cap)
        if n == 0:
            raise RuntimeError("header does not fit")
        return buf.raw[:n]

    def write(self, ostream, frame_index: bool = False) -> None:
        ostream.write(self.header(frame_index))
        ostream.write(bytes(self._data))
        if hasattr(ostream, "flush"):
            ostream.flush()

    @classmethod
    def read(cls, istream, device: int = -1) -> "Terse":
        """``Terse(std::ifstream&)``: scan for the header, read the payload, leave the stream
        positioned on the byte after it (Terse.hpp:275-279)."""
        pos = istream.tell()
        h = _lib.trpx_header()
        off = C.c_size_t(0)
        want = 4096
        while True:                                          # an indexed header (frame_sizes) can be long: read more
            istream.seek(pos)
            blob = istream.read(want)
            rc = lib().trpx_header_parse(blob, len(blob), C.byref(h), C.byref(off))
            if rc == _lib.OK or len(blob) < want or want >= 1 << 28:
                break
            want <<= 4
        if rc != _lib.OK:
            raise ValueError("no valid <Terse .../> header found")   # the reference's stoul throws
        t = cls(block=h.block, device=device)
        t._prolix_bits, t._signed, t._size = h.prolix_bits, bool(h.is_signed), int(h.number_of_values)
        t._dim = [int(h.dims[i]) for i in range(h.n_dims)]
        istream.seek(pos + off.value)
        t._data = bytearray(istream.read(int(h.memory_size)))
        if len(t._data) != h.memory_size:
            raise ValueError("truncated .trpx payload")
        n_frames = int(h.number_of_frames)
        sizes = np.empty(max(n_frames, 1), np.uint64)
        have = lib().trpx_header_frame_sizes(blob, len(blob), sizes.ctypes.data, sizes.size)
        if n_frames == 1:
            t._frame_sizes = [len(t._data)]
        elif n_frames > 1 and have == n_frames and int(sizes.sum()) == len(t._data) and (sizes > 0).all():
            t._frame_sizes = [int(x) for x in sizes]          # row f1: the file carries its own frame index
        elif n_frames > 1:
            # The file stores no frame index: locate the frames with the device's header walk.
            buf = np.frombuffer(t._data, np.uint8)
            offs = np.empty(n_frames + 1, np.uint64)
            max_bits = 8 if h.prolix_bits <= 8 else 16 if h.prolix_bits <= 16 else 32
            check(lib().trpx_frame_offsets_host(buf.ctypes.data, buf.size, t._size, n_frames, t._block,
                                                max_bits, offs.ctypes.data, device))
            if int(offs[-1]) != len(t._data):
                raise ValueError("frame chain does not cover the payload (corrupt .trpx)")
            t._frame_sizes = [int(x) for x in np.diff(offs)]
        groups = lib().trpx_group_count(t._size, t._block) if t._size else 0
        if groups and t._frame_sizes:                        # row f1: group states, if the file has them (checked on the device when used)
            gs = np.zeros(groups * len(t._frame_sizes), np.uint64)
            if lib().trpx_header_group_states(blob, off.value, gs.ctypes.data, gs.size) == gs.size:
                t._group_states = gs
        return t
