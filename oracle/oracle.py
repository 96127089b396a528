"""ctypes front-end of the CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module; the product package ``trpx_amd`` never does.

Two libraries are wrapped:

* ``liboracle.so``   -- the plain-C restatement (``terse_oracle.c``), always available.
* ``_ref/libtrpx_ref.so`` -- the REAL reference (``jpa::Terse``, /root/reference/include) behind
  ``ref_shim.cpp``; present wherever ``make -C oracle ref`` has been run (it travels to the GPU
  box as a built artefact, the reference sources do not).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

U8, I8, U16, I16, U32, I32, U64, I64 = range(8)
_NP2DT = {np.dtype(np.uint8): U8, np.dtype(np.int8): I8, np.dtype(np.uint16): U16,
          np.dtype(np.int16): I16, np.dtype(np.uint32): U32, np.dtype(np.int32): I32,
          np.dtype(np.uint64): U64, np.dtype(np.int64): I64,
          np.dtype(np.float32): 8, np.dtype(np.float64): 9}     # decode() output only (Terse.hpp:379-383)
_SFX = {U8: "u8", I8: "i8", U16: "u16", I16: "i16", U32: "u32", I32: "i32", U64: "u64", I64: "i64"}
SEED = 20240807


def dtype_code(dt) -> int:
    return _NP2DT[np.dtype(dt)]


def build(ref: bool = True) -> None:
    """Compile liboracle.so (and _ref when the reference tree is present)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    if ref and os.path.isdir("/root/reference/include"):
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build(ref=False)
        L = C.CDLL(path)
        L.trpx_oracle_worst_case_bytes.restype = C.c_size_t
        L.trpx_oracle_worst_case_bytes.argtypes = [C.c_int, C.c_size_t, C.c_uint]
        L.trpx_oracle_widths.restype = C.c_uint
        L.trpx_oracle_widths.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_uint, C.c_void_p]
        L.trpx_oracle_encode.restype = C.c_long
        L.trpx_oracle_encode.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_uint, C.c_void_p,
                                         C.c_size_t, C.POINTER(C.c_uint)]
        L.trpx_oracle_decode.restype = C.c_long
        L.trpx_oracle_decode.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_size_t,
                                         C.c_uint, C.c_void_p]
        L.trpx_oracle_frame_bytes.restype = C.c_long
        L.trpx_oracle_frame_bytes.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_uint]
        L.trpx_oracle_synth_u16.restype = None
        L.trpx_oracle_synth_u16.argtypes = [C.c_uint64, C.c_uint64, C.c_size_t, C.c_size_t, C.c_void_p]
        L.trpx_oracle_synth_i32.restype = None
        L.trpx_oracle_synth_i32.argtypes = [C.c_uint64, C.c_uint64, C.c_size_t, C.c_size_t, C.c_void_p]
        L.trpx_oracle_fnv1a64.restype = C.c_uint64
        L.trpx_oracle_fnv1a64.argtypes = [C.c_void_p, C.c_size_t]
        L.trpx_oracle_time.restype = C.c_int
        L.trpx_oracle_time.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_size_t, C.c_uint, C.c_int,
                                       C.POINTER(C.c_double), C.POINTER(C.c_double),
                                       C.POINTER(C.c_size_t), C.POINTER(C.c_int)]
        _lib = L
    return _lib


def have_ref() -> bool:
    return os.path.exists(os.path.join(_HERE, "_ref", "libtrpx_ref.so"))


def ref():
    """The real reference behind ref_shim.cpp (raises if it has not been built)."""
    global _ref
    if _ref is None:
        path = os.path.join(_HERE, "_ref", "libtrpx_ref.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle/_ref/libtrpx_ref.so not built (run `make -C oracle ref` "
                               "in a container that has /root/reference)")
        R = C.CDLL(path)
        for sfx in _SFX.values():
            enc = getattr(R, f"trpx_ref_encode_{sfx}")
            enc.restype = C.c_long
            enc.argtypes = [C.c_void_p, C.c_size_t, C.c_uint, C.c_void_p, C.c_size_t,
                            C.POINTER(C.c_uint), C.c_char_p, C.c_size_t]
            dec = getattr(R, f"trpx_ref_decode_{sfx}")
            dec.restype = C.c_int
            dec.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_uint, C.c_int, C.c_uint, C.c_void_p]
        R.trpx_ref_encode_stack_u16.restype = C.c_long
        R.trpx_ref_encode_stack_u16.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p,
                                                C.c_size_t, C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t]
        for name in ("trpx_ref_time_u16", "trpx_ref_time_i32"):
            fn = getattr(R, name)
            fn.restype = C.c_int
            fn.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.POINTER(C.c_double),
                           C.POINTER(C.c_double), C.POINTER(C.c_size_t), C.POINTER(C.c_int)]
        _ref = R
    return _ref


# ------------------------------------------------------------------------------------------------
# restatement
# ------------------------------------------------------------------------------------------------
def worst_case_bytes(dtype, n: int, block: int = 12) -> int:
    return lib().trpx_oracle_worst_case_bytes(dtype_code(dtype), n, block)


def widths(px: np.ndarray, block: int = 12) -> np.ndarray:
    px = np.ascontiguousarray(px).reshape(-1)
    nb = (px.size + block - 1) // block
    w = np.zeros(nb, np.uint8)
    lib().trpx_oracle_widths(dtype_code(px.dtype), px.ctypes.data, px.size, block, w.ctypes.data)
    return w


def group_states(px: np.ndarray, group_blocks: int = 256, block: int = 12) -> np.ndarray:
    """Chain state of ONE frame at every `group_blocks`-th block, from the widths alone: bit offset inside the frame |
    width of the block before << 40.  Header lengths as Terse.hpp:520-541 writes them (1 bit when the width repeats, else
    4 / 6 / 12 bits), payload = block values x width (Terse.hpp:543)."""
    px = np.ascontiguousarray(px).reshape(-1)
    w = widths(px, block).astype(np.int64)
    prev = np.concatenate([[0], w[:-1]])
    hl = np.where(w == prev, 1, np.where(w < 7, 4, np.where(w < 10, 6, 12)))
    nv = np.full(w.size, block, np.int64)
    nv[-1] = px.size - (w.size - 1) * block
    pos = np.concatenate([[0], np.cumsum(hl + nv * w)])
    k = np.arange(0, w.size, group_blocks)
    return (pos[k] | (prev[k] << 40)).astype(np.uint64)


def encode(px: np.ndarray, block: int = 12):
    """Encode ONE frame. Returns (stream bytes as np.uint8 array, prolix_bits)."""
    px = np.ascontiguousarray(px).reshape(-1)
    cap = worst_case_bytes(px.dtype, px.size, block)
    out = np.zeros(cap, np.uint8)
    pb = C.c_uint(0)
    s = lib().trpx_oracle_encode(dtype_code(px.dtype), px.ctypes.data, px.size, block,
                                 out.ctypes.data, cap, C.byref(pb))
    if s < 0:
        raise RuntimeError("oracle encode failed")
    return out[:s].copy(), int(pb.value)


def encode_stack(px: np.ndarray, block: int = 12):
    """Encode a [frames, N] stack: returns (concatenated bytes, per-frame sizes, prolix_bits).

    The stack layout is the plain concatenation of single-frame encodes (Terse.hpp:502-504)."""
    px = np.ascontiguousarray(px)
    px = px.reshape(px.shape[0], -1)
    parts, sizes, pb = [], [], 0
    for f in range(px.shape[0]):
        s, p = encode(px[f], block)
        parts.append(s)
        sizes.append(s.size)
        pb = max(pb, p)
    return (np.concatenate(parts) if parts else np.zeros(0, np.uint8)), np.array(sizes, np.uint64), pb


def decode(stream: np.ndarray, n: int, dtype, stream_signed=None, block: int = 12) -> np.ndarray:
    """Decode ONE frame into a fresh array of `dtype`."""
    stream = np.ascontiguousarray(stream, dtype=np.uint8)
    dt = np.dtype(dtype)
    if stream_signed is None:
        stream_signed = dt.kind == "i"
    out = np.zeros(n, dt)
    s = lib().trpx_oracle_decode(dtype_code(dt), int(bool(stream_signed)), stream.ctypes.data,
                                 stream.size, n, block, out.ctypes.data)
    if s < 0:
        raise RuntimeError("oracle decode failed (truncated stream?)")
    return out


def frame_bytes(stream: np.ndarray, n: int, block: int = 12) -> int:
    stream = np.ascontiguousarray(stream, dtype=np.uint8)
    return lib().trpx_oracle_frame_bytes(stream.ctypes.data, stream.size, n, block)


def synth(dtype, frame0: int, frames: int, n: int, seed: int = SEED) -> np.ndarray:
    dt = np.dtype(dtype)
    out = np.empty((frames, n), dt)
    if dt == np.uint16:
        lib().trpx_oracle_synth_u16(seed, frame0, frames, n, out.ctypes.data)
    elif dt == np.int32:
        lib().trpx_oracle_synth_i32(seed, frame0, frames, n, out.ctypes.data)
    else:
        raise ValueError("synth-v1 is defined for uint16 and int32")
    return out


def fnv1a64(a) -> int:
    a = np.ascontiguousarray(a)
    return lib().trpx_oracle_fnv1a64(a.ctypes.data, a.nbytes)


def time_port(px: np.ndarray, threads: int, block: int = 12):
    """Time the restatement on a [frames, N] stack; returns dict(enc_s, dec_s, bytes, ok)."""
    px = np.ascontiguousarray(px)
    frames, n = px.shape
    e, d, tb, ok = C.c_double(), C.c_double(), C.c_size_t(), C.c_int()
    rc = lib().trpx_oracle_time(dtype_code(px.dtype), px.ctypes.data, n, frames, block, threads,
                                C.byref(e), C.byref(d), C.byref(tb), C.byref(ok))
    if rc:
        raise RuntimeError("oracle timing failed")
    return dict(enc_s=e.value, dec_s=d.value, bytes=tb.value, ok=bool(ok.value))


# ------------------------------------------------------------------------------------------------
# real reference (oracle/_ref)
# ------------------------------------------------------------------------------------------------
def ref_encode(px: np.ndarray, block: int = 12):
    """Encode ONE frame with the real reference. Returns (stream, prolix_bits, header text)."""
    px = np.ascontiguousarray(px).reshape(-1)
    code = dtype_code(px.dtype)
    cap = worst_case_bytes(px.dtype, px.size, block) + 16
    out = np.zeros(cap, np.uint8)
    pb = C.c_uint(0)
    hdr = C.create_string_buffer(512)
    fn = getattr(ref(), f"trpx_ref_encode_{_SFX[code]}")
    s = fn(px.ctypes.data, px.size, block, out.ctypes.data, cap, C.byref(pb), hdr, 512)
    if s < 0:
        raise RuntimeError(f"reference encode failed ({s})")
    return out[:s].copy(), int(pb.value), hdr.value.decode()


def ref_decode(stream: np.ndarray, n: int, dtype, prolix_bits: int, block: int = 12) -> np.ndarray:
    stream = np.ascontiguousarray(stream, dtype=np.uint8)
    dt = np.dtype(dtype)
    out = np.zeros(n, dt)
    fn = getattr(ref(), f"trpx_ref_decode_{_SFX[dtype_code(dt)]}")
    rc = fn(stream.ctypes.data, stream.size, n, block, int(dt.kind == "i"), prolix_bits, out.ctypes.data)
    if rc:
        raise RuntimeError(f"reference decode failed ({rc})")
    return out


def ref_encode_stack_u16(px: np.ndarray, dims=()):
    px = np.ascontiguousarray(px, dtype=np.uint16)
    frames, n = px.shape
    cap = (worst_case_bytes(np.uint16, n) + 16) * frames
    out = np.zeros(cap, np.uint8)
    hdr = C.create_string_buffer(512)
    d = np.array(list(dims), dtype=np.uintp)
    s = ref().trpx_ref_encode_stack_u16(px.ctypes.data, n, frames, out.ctypes.data, cap, hdr, 512,
                                        d.ctypes.data if d.size else None, d.size)
    if s < 0:
        raise RuntimeError("reference stack encode failed")
    return out[:s].copy(), hdr.value.decode()


def time_ref(px: np.ndarray):
    """Single-thread timing of the real reference (one jpa::Terse per frame)."""
    px = np.ascontiguousarray(px)
    frames, n = px.shape
    name = {np.dtype(np.uint16): "trpx_ref_time_u16", np.dtype(np.int32): "trpx_ref_time_i32"}[px.dtype]
    e, d, tb, ok = C.c_double(), C.c_double(), C.c_size_t(), C.c_int()
    getattr(ref(), name)(px.ctypes.data, n, frames, C.byref(e), C.byref(d), C.byref(tb), C.byref(ok))
    return dict(enc_s=e.value, dec_s=d.value, bytes=tb.value, ok=bool(ok.value))
