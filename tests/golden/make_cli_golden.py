#!/usr/bin/env python3
"""Generates tests/golden/cli/*: small grey TIFF stacks, the .trpx files the REFERENCE `terse` CLI makes of them and
the TIFF files the REFERENCE `prolix` CLI makes of those.

Runs only where /root/reference exists: builds the reference CLIs from their sources in place into oracle/_ref/
(git-ignored) and runs them on copies in a temporary directory (they delete their inputs).  The committed outputs are
data fixtures: inputs and expected outputs.  The TIFF inputs are written here, in the layout of the reference's own
writer (Grey_tif.hpp:477-557: pixel data, pad to even, 7-entry IFD, next-IFD offset).
"""
import json, os, shutil, struct, subprocess, sys, tempfile
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden", "cli")


def write_tiff(path, frames, big_endian=False):
    """frames: [n, h, w] integer array.  dim() of the reference is {width, height} (tags 0x100, 0x101)."""
    e = ">" if big_endian else "<"
    dt = frames.dtype
    buf = bytearray((b"MM" if big_endian else b"II") + struct.pack(e + "HI", 42, 0))
    last = 4
    for img in frames:
        start = len(buf)
        buf += img.astype(dt.newbyteorder(e)).tobytes()
        if len(buf) & 1:
            buf += b"\0"
        struct.pack_into(e + "I", buf, last, len(buf))
        sample = 1 if dt.kind == "u" else 2 if dt.kind == "i" else 3
        ifd = struct.pack(e + "H", 7)
        for tag, typ, val in ((0x100, 3, img.shape[1]), (0x101, 3, img.shape[0]), (0x102, 3, 8 * dt.itemsize), (0x103, 3, 1),
                              (0x106, 3, 1), (0x111, 4, start), (0x153, 3, sample)):
            ifd += struct.pack(e + "HHI", tag, typ, 1) + (struct.pack(e + "HH", val, 0) if typ == 3 else struct.pack(e + "I", val))
        buf += ifd
        last = len(buf)
        buf += struct.pack(e + "I", 0)
    open(path, "wb").write(bytes(buf))


def build():
    os.makedirs(os.path.join(ROOT, "oracle", "_ref"), exist_ok=True)
    for name in ("terse", "prolix"):
        subprocess.check_call(["g++", "-std=c++20", "-O2", "-w", "-I" + REF + "/include", f"{REF}/src/{name}.cpp", "-o",
                               os.path.join(ROOT, "oracle", "_ref", name + "_cli")])


def main():
    build()
    rng = np.random.RandomState(20240807)
    def bg(shape, hi=7):
        a = rng.poisson(1.5, size=shape).clip(0, hi)
        a[rng.rand(*shape) < 0.01] = hi * 40
        return a
    cases = {
        "u16_stack3_35x20": (bg((3, 20, 35)).astype(np.uint16), False),
        "i16_single_17x9": ((bg((1, 9, 17)) - 3).astype(np.int16), False),
        "u8_stack2_16x12": (bg((2, 12, 16), 5).astype(np.uint8), False),
        "u16_bigendian_24x24": (bg((2, 24, 24)).astype(np.uint16), True),
        "u32_single_32x8": ((bg((1, 8, 32)).astype(np.uint32) * 70001), False),
        "i32_stack2_12x12": (((bg((2, 12, 12)) - 2) * 100003).astype(np.int32), False),
        # float / double pixels: the reference converts them to 64-bit integers first (terse.cpp:120-123)
        # -- as a plain vector, so the file carries no dimensions and `prolix` assumes a square image (prolix.cpp:64-65)
        "f32_single_12x12": ((bg((1, 12, 12)) - 2).astype(np.float32) + np.float32(0.75), False),
        "f64_stack2_8x8": ((bg((2, 8, 8)) * 3 - 4).astype(np.float64) - 0.5, False),
    }
    os.makedirs(OUT, exist_ok=True)
    index = {}
    with tempfile.TemporaryDirectory() as tmp:
        for name, (frames, be) in cases.items():
            tif = os.path.join(OUT, name + ".tif")
            write_tiff(tif, frames, be)
            work = os.path.join(tmp, name + ".tif")
            shutil.copy(tif, work)
            subprocess.check_call([os.path.join(ROOT, "oracle", "_ref", "terse_cli"), work], stdout=subprocess.DEVNULL)
            trpx = os.path.join(tmp, name + ".trpx")
            shutil.copy(trpx, os.path.join(OUT, name + ".trpx"))
            # What `prolix` must write (prolix.cpp:69-92): 16-bit pixels when prolix_bits <= 16, else 32-bit, same
            # signedness, little endian, the reference writer's layout.
            if frames.dtype.kind == "f":                     # truncation towards zero, then signed data of <= 16 bits here
                ints, out_dt = np.trunc(frames).astype(np.int64), np.dtype("i2")
            else:
                ints = frames
                out_dt = np.dtype(("i" if frames.dtype.kind == "i" else "u") + ("2" if frames.dtype.itemsize <= 2 else "4"))
            expect = os.path.join(OUT, name + ".expect.tif")
            write_tiff(expect, ints.astype(out_dt))
            entry = {"dtype": str(frames.dtype), "shape": list(frames.shape), "big_endian": be, "expect_tif": name + ".expect.tif",
                     "reference_prolix_matches": None}
            if frames.dtype.itemsize <= 2 or frames.dtype.kind == "f":   # the reference prolix writes wrong pixels for 32-bit output
                subprocess.check_call([os.path.join(ROOT, "oracle", "_ref", "prolix_cli"), trpx], stdout=subprocess.DEVNULL)
                same = open(os.path.join(tmp, name + ".tif"), "rb").read() == open(expect, "rb").read()
                entry["reference_prolix_matches"] = bool(same)   # False only for >= 3 frames: reference defects D1/D2
            index[name] = entry
    json.dump(index, open(os.path.join(OUT, "index.json"), "w"), indent=1)
    print("wrote", len(index), "cases to", OUT)


if __name__ == "__main__":
    sys.exit(main())
