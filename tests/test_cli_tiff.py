"""Rows f2 / f3: the GPU-backed `terse` / `prolix` tools and the minimal grey TIFF reader / writer.

Fixtures in tests/golden/cli (made by tests/golden/make_cli_golden.py with the REFERENCE tools, where /root/reference
exists): small TIFF stacks, the .trpx the reference `terse` makes of each, and the TIFF `prolix` must write
(`*.expect.tif`; index.json records for which of them the reference `prolix` wrote exactly these bytes -- all but the
3-frame stack, where the reference mislocates frames >= 2, defects D1/D2).
"""
import json
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "tests", "golden", "cli")
INDEX = json.load(open(os.path.join(CLI, "index.json")))


def _parse_tiff(path):
    """Independent little/big-endian parser of the reference writer's layout -> [n, h, w] array."""
    b = open(path, "rb").read()
    e = "<" if b[:2] == b"II" else ">"
    ifd = struct.unpack_from(e + "I", b, 4)[0]
    frames = []
    while ifd:
        n = struct.unpack_from(e + "H", b, ifd)[0]
        tags = {}
        for i in range(n):
            tag, typ, cnt = struct.unpack_from(e + "HHI", b, ifd + 2 + 12 * i)
            tags[tag] = struct.unpack_from(e + ("H" if typ == 3 else "I"), b, ifd + 2 + 12 * i + 8)[0]
        kind = {1: "u", 2: "i", 3: "f"}[tags.get(0x153, 1)]
        dt = np.dtype(f"{e}{kind}{tags[0x102] // 8}")
        w, h = tags[0x100], tags[0x101]
        frames.append(np.frombuffer(b, dt, w * h, tags[0x111]).reshape(h, w))
        ifd = struct.unpack_from(e + "I", b, ifd + 2 + 12 * n)[0]
    return np.stack(frames)


@pytest.mark.parametrize("name", sorted(INDEX))
def test_tiff_reader_writer_match_reference_layout(name, tmp_path):
    """CPU: Grey_tif.hpp reads every fixture (little and big endian) and re-writes it, as `prolix` would, byte for byte
    like the reference writer."""
    exe = os.path.join(ROOT, "tests", "cpp", "tiff_roundtrip")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpp"), "tiff_roundtrip"])
    info = INDEX[name]
    if np.dtype(info["dtype"]).kind == "f":
        pytest.skip("float pixels go through the tools only (converted to 64-bit integers first, terse.cpp:120-123)")
    out = tmp_path / "out.tif"
    bits = "16" if np.dtype(info["dtype"]).itemsize <= 2 else "32"
    r = subprocess.run([exe, os.path.join(CLI, name + ".tif"), str(out), bits], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    n, w, h, bpp, sgn, integral = (int(x) for x in r.stdout.split())
    assert [n, h, w] == info["shape"] and bpp == np.dtype(info["dtype"]).itemsize
    assert bool(sgn) == (np.dtype(info["dtype"]).kind == "i") and integral == 1
    assert out.read_bytes() == open(os.path.join(CLI, info["expect_tif"]), "rb").read()
    want = _parse_tiff(os.path.join(CLI, name + ".tif"))
    assert (_parse_tiff(str(out)).astype(np.int64) == want.astype(np.int64)).all()
    if info["reference_prolix_matches"] is not None and name != "u16_stack3_35x20":
        assert info["reference_prolix_matches"]            # the expected files ARE what the reference writer produced


def test_tiff_reader_rejects_corrupt_files(tmp_path):
    exe = os.path.join(ROOT, "tests", "cpp", "tiff_roundtrip")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpp"), "tiff_roundtrip"])
    good = open(os.path.join(CLI, "i16_single_17x9.tif"), "rb").read()
    for bad in (good[:40], b"XX" + good[2:], good[:4] + struct.pack("<I", len(good) + 100) + good[8:]):
        p = tmp_path / "bad.tif"
        p.write_bytes(bad)
        r = subprocess.run([exe, str(p), str(tmp_path / "o.tif")], capture_output=True, text=True)
        assert r.returncode == 1 and "error" in r.stderr     # the reference reads out of bounds instead


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(INDEX))
def test_terse_cli_writes_the_reference_trpx(name, tmp_path):
    """GPU: `terse file.tif` -> file.trpx, byte-identical (header text and payload) to the reference tool's output."""
    exe = os.path.join(ROOT, "trpx_amd", "bin", "terse")
    assert os.path.exists(exe), "build the tools first (__graft_entry__.build / make -C trpx_amd/cli)"
    work = tmp_path / (name + ".tif")
    shutil.copy(os.path.join(CLI, name + ".tif"), work)
    junk = tmp_path / "notes.txt"
    junk.write_text("not a tiff")                            # extension filter (terse.cpp:44-47)
    r = subprocess.run([exe, "-verbose", str(work), str(junk)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "Terse compressed: 1 files" in r.stdout and "Compression rate:" in r.stdout
    dtn = np.dtype(INDEX[name]["dtype"])                     # ImageJ/TRPX_Reader.java:94-98: unsigned, <= 16 bits only
    assert ("ImageJ TRPX reader only opens" in r.stdout) == (dtn.kind in "if" or dtn.itemsize == 4)
    assert work.exists()                                     # kept (the reference deletes it; -delete does that here)
    assert (tmp_path / (name + ".trpx")).read_bytes() == open(os.path.join(CLI, name + ".trpx"), "rb").read()
    r = subprocess.run([exe, "-delete", str(work)], capture_output=True, text=True)
    assert r.returncode == 0 and not work.exists()


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(INDEX))
def test_prolix_cli_expands_the_reference_trpx(name, tmp_path):
    """GPU: `prolix file.trpx` (a file written by the REFERENCE `terse`) -> file.tif with the expected bytes; for 32-bit
    stacks and for stacks of >= 3 frames this is what the reference intends but does not do (D5, D1/D2)."""
    exe = os.path.join(ROOT, "trpx_amd", "bin", "prolix")
    assert os.path.exists(exe)
    work = tmp_path / (name + ".trpx")
    shutil.copy(os.path.join(CLI, name + ".trpx"), work)
    r = subprocess.run([exe, "-verbose", str(work)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "Prolix expanded : 1 files" in r.stdout
    info = INDEX[name]
    got = (tmp_path / (name + ".tif")).read_bytes()
    assert got == open(os.path.join(CLI, info["expect_tif"]), "rb").read()
    want = _parse_tiff(os.path.join(CLI, name + ".tif"))
    assert (_parse_tiff(str(tmp_path / (name + ".tif"))).astype(np.int64) == np.trunc(want).astype(np.int64)).all()   # (float fixtures: truncated)


def _indexed_trpx(name):
    """The reference `terse`'s file of a fixture with the frame_sizes and group_bit_offsets attributes added (row f1),
    built on the CPU: frame sizes and group states from the oracle, header text from the library's host-only formatter."""
    import ctypes as C
    import sys
    sys.path.insert(0, ROOT)
    from oracle import oracle as O
    from trpx_amd import _lib
    info = INDEX[name]
    frames = _parse_tiff(os.path.join(CLI, name + ".tif"))
    px = np.ascontiguousarray(frames.reshape(frames.shape[0], -1).astype(np.dtype(info["dtype"])))
    stream, sizes, pb = O.encode_stack(px)
    blob = open(os.path.join(CLI, name + ".trpx"), "rb").read()
    h = _lib.trpx_header()
    off = C.c_size_t(0)
    assert _lib.lib().trpx_header_parse(blob, len(blob), C.byref(h), C.byref(off)) == _lib.OK
    assert blob[off.value:] == stream.tobytes()              # the oracle agrees with the reference tool's payload
    buf = C.create_string_buffer(1 << 16)
    s64 = np.asarray(sizes, np.uint64)
    gs = np.concatenate([O.group_states(f) for f in px])
    assert gs.size == px.shape[0] * _lib.lib().trpx_group_count(px.shape[1], 12)
    n = _lib.lib().trpx_header_format_grouped(C.byref(h), s64.ctypes.data, s64.size, gs.ctypes.data, gs.size, buf, 1 << 16)
    assert n > 0 and b' frame_sizes="' in buf.raw[:n] and b' group_bit_offsets="0:0' in buf.raw[:n]
    back = np.zeros(s64.size, np.uint64)
    assert _lib.lib().trpx_header_frame_sizes(buf.raw[:n], n, back.ctypes.data, back.size) == s64.size and (back == s64).all()
    assert _lib.lib().trpx_header_frame_sizes(blob, len(blob), back.ctypes.data, back.size) == 0   # plain header: none
    gback = np.zeros(gs.size, np.uint64)
    assert _lib.lib().trpx_header_group_states(buf.raw[:n], n, gback.ctypes.data, gback.size) == gs.size and (gback == gs).all()
    assert _lib.lib().trpx_header_group_states(blob, len(blob), gback.ctypes.data, gback.size) == 0
    h2, off2 = _lib.trpx_header(), C.c_size_t(0)
    assert _lib.lib().trpx_header_parse(buf.raw[:n], n, C.byref(h2), C.byref(off2)) == _lib.OK and off2.value == n
    return buf.raw[:n] + stream.tobytes()


@pytest.mark.parametrize("name", ["u8_stack2_16x12", "u16_bigendian_24x24", "i16_single_17x9"])
def test_reference_prolix_ignores_the_frame_index_attribute(name, tmp_path):
    """CPU, only where the reference tools can be built (/root/reference): a .trpx file with the extra frame_sizes
    and group_bit_offsets attributes expands with the REFERENCE `prolix` to the same TIFF as the plain file (stacks of <= 2 frames: the
    reference mislocates later frames either way, D1/D2)."""
    indexed = _indexed_trpx(name)
    ref = _reference_prolix()
    work = tmp_path / (name + ".trpx")
    work.write_bytes(indexed)
    subprocess.check_call([ref, str(work)], stdout=subprocess.DEVNULL)
    assert (tmp_path / (name + ".tif")).read_bytes() == open(os.path.join(CLI, INDEX[name]["expect_tif"]), "rb").read()


def _reference_prolix():
    ref = os.path.join(ROOT, "oracle", "_ref", "prolix_cli")
    if not os.path.exists(ref):
        if not os.path.isdir("/root/reference"):
            pytest.skip("reference sources not available here")
        os.makedirs(os.path.dirname(ref), exist_ok=True)
        subprocess.check_call(["g++", "-std=c++20", "-O2", "-w", "-I/root/reference/include", "/root/reference/src/prolix.cpp",
                               "-o", ref])
    return ref


def test_reference_prolix_reads_a_file_with_many_group_states(tmp_path):
    """CPU, where the reference tool can be built: two 200x160 u16 frames (2667 blocks = 11 groups each, a 22-token
    group_bit_offsets attribute of several hundred characters) written with both index attributes expand with the
    REFERENCE `prolix` to the original pixels -- the attributes are skipped like any unknown one (XML_element.hpp:296-307)."""
    import ctypes as C
    import sys
    sys.path.insert(0, ROOT)
    from oracle import oracle as O
    from trpx_amd import _lib
    ref = _reference_prolix()
    rng = np.random.RandomState(5)
    px = (rng.poisson(3.0, size=(2, 160 * 200)) * (rng.rand(2, 160 * 200) < 0.7)).astype(np.uint16)
    px[:, ::97] = 40000
    stream, sizes, pb = O.encode_stack(px)
    h = _lib.trpx_header()
    h.prolix_bits, h.is_signed, h.block, h.memory_size, h.number_of_values = pb, 0, 12, stream.size, px.shape[1]
    h.number_of_frames, h.n_dims = 2, 2
    h.dims[0], h.dims[1] = 200, 160
    gs = np.concatenate([O.group_states(f) for f in px])
    assert gs.size == 22 and gs[1] > 0 and gs[11] == 0
    s64 = np.asarray(sizes, np.uint64)
    buf = C.create_string_buffer(1 << 14)
    n = _lib.lib().trpx_header_format_grouped(C.byref(h), s64.ctypes.data, 2, gs.ctypes.data, gs.size, buf, 1 << 14)
    assert n > 300
    work = tmp_path / "many.trpx"
    work.write_bytes(buf.raw[:n] + stream.tobytes())
    subprocess.check_call([ref, str(work)], stdout=subprocess.DEVNULL)
    assert (_parse_tiff(str(tmp_path / "many.tif")).reshape(2, -1) == px).all()


@pytest.mark.gpu
def test_indexed_files_round_trip_through_tools_and_classes(tmp_path):
    """GPU: `terse -index` writes the attribute, `prolix` and both Terse classes read such files (and plain ones)."""
    import sys
    sys.path.insert(0, ROOT)
    from trpx_amd import Terse
    name = "u16_stack3_35x20"
    work = tmp_path / (name + ".tif")
    shutil.copy(os.path.join(CLI, name + ".tif"), work)
    terse, prolix = (os.path.join(ROOT, "trpx_amd", "bin", x) for x in ("terse", "prolix"))
    assert subprocess.run([terse, "-index", str(work)], capture_output=True).returncode == 0
    made = (tmp_path / (name + ".trpx")).read_bytes()
    assert made == _indexed_trpx(name)
    want = _parse_tiff(os.path.join(CLI, name + ".tif"))
    with open(tmp_path / (name + ".trpx"), "rb") as f:
        t = Terse.read(f)
    assert t.number_of_frames() == 3 and sum(t.frame_sizes()) == t.terse_size()
    assert (t.prolix_stack(np.uint16).reshape(want.shape) == want).all()
    import io
    out = io.BytesIO()
    t.write(out, frame_index=True)
    assert out.getvalue() == made
    out = io.BytesIO()
    t.write(out)
    assert out.getvalue() == open(os.path.join(CLI, name + ".trpx"), "rb").read()        # and the plain header again
    os.remove(work)
    assert subprocess.run([prolix, str(tmp_path / (name + ".trpx"))], capture_output=True).returncode == 0
    assert work.read_bytes() == open(os.path.join(CLI, INDEX[name]["expect_tif"]), "rb").read()
