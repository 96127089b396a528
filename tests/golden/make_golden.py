#!/usr/bin/env python3
"""Generate tests/golden/terse_golden.json with the REAL reference (oracle/_ref).

Run in the build container (needs /root/reference):  python tests/golden/make_golden.py

Every case stores the input pixels, and the reference's outputs for it: the encoded stream
(hex), prolix_bits, the header text written by jpa::Terse::write (Terse.hpp:454-474) and -- as a
cross-check of the reference decoder -- whether jpa::Terse::prolix reproduced the input.  The
cases stay inside the reference's validity domain and cover the list of
SURVEY.md section 8 row c(ii).  The fixture is DATA (inputs + expected outputs) only.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

rng = np.random.RandomState(20240807)
cases = []


def add(name, px, block=12):
    px = np.ascontiguousarray(px)
    stream, pb, hdr = O.ref_encode(px, block)
    back = O.ref_decode(stream, px.size, px.dtype, pb, block)
    assert (back == px).all(), f"reference does not round-trip case {name} (outside D3 domain?)"
    cases.append(dict(name=name, dtype=px.dtype.name, block=block, pixels=px.tolist(),
                      stream=stream.tobytes().hex(), prolix_bits=pb, header=hdr,
                      total_bits_mod8_zero=None))


# --- doc-comment known answers (Terse.hpp:53-57, :127-154) ---------------------------------
add("doc_u8_342_block3", np.array([3, 4, 2], np.uint8), block=3)
add("doc_i8_m342_block3", np.array([-3, 4, 2], np.int8), block=3)
add("doc_iota_m500_499_i32", np.arange(-500, 500, dtype=np.int32))
add("zeros24_u16", np.zeros(24, np.uint16))
add("survey_kat50_u16", np.array(list(range(12)) + [0] * 12 + [1000] + [0] * 11 + [1023] + [1] * 11
                                 + [65535, 2], np.uint16))

# --- partial last block -------------------------------------------------------------------
for rem in (1, 4, 11):
    add(f"partial_rem{rem}_u16", rng.randint(0, 40, size=12 * 7 + rem).astype(np.uint16))
    add(f"partial_rem{rem}_i32", rng.randint(-40, 40, size=12 * 5 + rem).astype(np.int32))
add("single_value_u16", np.array([5], np.uint16))
add("single_zero_u16", np.array([0], np.uint16))
add("all_zero_1000_u16", np.zeros(1000, np.uint16))
add("all_zero_1000_i32", np.zeros(1000, np.int32))

# --- header boundaries: one block per width, explicit <-> same transitions -----------------
def width_ladder(dtype, widths, signed):
    out = []
    for w in widths:
        blk = np.zeros(12, np.int64)
        if w:
            top = (1 << (w - 1)) if not signed else (1 << (w - 2)) if w >= 2 else 0
            if signed and w == 1:
                continue  # signed widths are 0 or >= 2
            blk[:] = rng.randint(0, max(top, 1), size=12)
            blk[rng.randint(12)] = top if not signed else top  # force the top bit
            if signed:
                blk[::2] *= -1
                if w >= 2 and rng.randint(2):
                    blk[3] = -top  # -2^k needs exactly k+2 bits incl. sign? (|v| = 2^(w-2))
        out.append(blk)
        out.append(blk.copy())  # repeated width -> "same" header bit
    return np.concatenate(out).astype(dtype)

add("ladder_u8", width_ladder(np.uint8, range(0, 9), False))
add("ladder_u16", width_ladder(np.uint16, [0, 1, 2, 5, 6, 7, 8, 9, 10, 11, 12, 15, 16, 0, 16, 3], False))
add("ladder_u32", width_ladder(np.uint32, [0, 6, 7, 9, 10, 11, 16, 17, 24, 25, 30, 31, 1], False))
add("ladder_i8", width_ladder(np.int8, range(0, 8), True))
add("ladder_i16", width_ladder(np.int16, [0, 2, 3, 6, 7, 8, 9, 10, 11, 14, 15, 2], True))
add("ladder_i32", width_ladder(np.int32, [0, 2, 6, 7, 9, 10, 11, 16, 17, 24, 25, 30, 31, 2], True))

# --- signed negatives incl. -2^k --------------------------------------------------------------
add("neg_pow2_i16", np.array([-(1 << k) for k in range(0, 14)] + [(1 << k) for k in range(0, 14)], np.int16))
add("neg_pow2_i32", np.array([-(1 << k) for k in range(0, 30)] + [(1 << k) - 1 for k in range(0, 30)], np.int32))
add("neg_small_i8", np.array([-1, 0, 1, -2, 2, -3, 3, -31, 31, -32, 0, 0, -1] * 3, np.int8))

# --- random frames of various statistics --------------------------------------------------
add("poisson3_u16_1500", rng.poisson(3, 1500).astype(np.uint16))
p = rng.poisson(3, 2000).astype(np.uint16)
p[rng.randint(0, 2000, 6)] = rng.randint(0, 4096, 6)
add("poisson3_peaks_u16_2000", p)
add("uniform16_u16_600", rng.randint(0, 65536, 600).astype(np.uint16))
add("uniform8_u8_1000", rng.randint(0, 256, 1000).astype(np.uint8))
add("small_i32_bg_peaks", np.where(rng.rand(1800) < 0.004, rng.randint(0, 1 << 24, 1800),
                                   rng.randint(-3, 4, 1800)).astype(np.int32))
add("u32_wide", np.where(rng.rand(900) < 0.02, rng.randint(0, 1 << 31, 900, dtype=np.int64),
                         rng.randint(0, 9, 900)).astype(np.uint32))
add("i16_mixed", (rng.randn(1300) * 40).astype(np.int16))
add("u16_runs_of_zero_blocks", np.concatenate([np.zeros(12 * 9, np.uint16), np.full(12, 3, np.uint16),
                                               np.zeros(12 * 33, np.uint16), np.array([1], np.uint16)]))

# --- a frame with total_bits % 8 == 0 (full extra pad byte, Terse.hpp:547) ------------------
found = False
for n in range(12, 400):
    px = (np.arange(n) % 5).astype(np.uint16)
    s, pb, _ = O.ref_encode(px)
    # total bits: recompute with the restatement's widths
    w = O.widths(px)
    prev, bits = 0, 0
    for b, wb in enumerate(w):
        nb = min(12, n - 12 * b)
        bits += (1 if wb == prev else (4 if wb < 7 else 6 if wb < 10 else 12)) + int(wb) * nb
        prev = wb
    if bits % 8 == 0:
        add(f"bits_mod8_zero_n{n}_u16", px)
        cases[-1]["total_bits_mod8_zero"] = True
        assert s[-1] == 0 and len(s) == bits // 8 + 1
        found = True
        break
assert found

# --- block != 12 (generic block path, Terse.hpp:264 `block` ctor argument) -----------------
add("block3_u16", rng.randint(0, 300, 100).astype(np.uint16), block=3)
add("block16_u16", rng.randint(0, 300, 100).astype(np.uint16), block=16)
add("block1_u8", rng.randint(0, 30, 20).astype(np.uint8), block=1)

# --- a 3-frame stack through the reference's own push_back (stack layout = concatenation) ----
stack = np.stack([rng.poisson(3, 700), rng.poisson(20, 700), np.zeros(700)]).astype(np.uint16)
sbytes, shdr = O.ref_encode_stack_u16(stack, dims=(35, 20))
sizes = [len(O.ref_encode(stack[f])[0]) for f in range(3)]
assert sum(sizes) == len(sbytes)
stack_case = dict(name="stack3_u16_700", dtype="uint16", block=12, pixels=stack.tolist(),
                  stream=sbytes.tobytes().hex(), header=shdr, frame_sizes=sizes, dims=[35, 20])

# --- anchors for the large synthetic frames (regenerated on both boxes from synth-v1) -------
anchors = []
for dt, n, frames in ((np.uint16, 512 * 512, 3), (np.int32, 4096 * 4096, 1)):
    px = O.synth(dt, 0, frames, n)
    for f in range(frames):
        s, pb, _ = O.ref_encode(px[f])
        anchors.append(dict(dtype=np.dtype(dt).name, n=n, frame=f, seed=O.SEED,
                            pixels_fnv=f"{O.fnv1a64(px[f]):016x}", size=len(s), prolix_bits=pb,
                            stream_fnv=f"{O.fnv1a64(s):016x}", first16=s[:16].tobytes().hex()))
# sum of sizes of the first 16 u16 frames (cheap stack anchor)
px = O.synth(np.uint16, 0, 16, 512 * 512)
tot = sum(len(O.ref_encode(px[f])[0]) for f in range(16))

out = dict(generator="tests/golden/make_golden.py", reference="senikm/trpx @ 2024_08_07 (oracle/_ref)",
           cases=cases, stack=stack_case, anchors=anchors, synth_u16_first16_total_bytes=tot)
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "terse_golden.json")
with open(path, "w") as f:
    json.dump(out, f, separators=(",", ":"))
print(f"wrote {path}: {len(cases)} cases, {os.path.getsize(path)} bytes")
for a in anchors:
    print(a)
print("first16 total", tot)
