"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against the CPU oracle.

Bar: bit-exact streams, frame sizes and prolix_bits on encode; pixel-exact on decode.
Small cases: golden fixtures made by the real reference + seeded differential runs against the
oracle.  Full BASELINE sizes: synth-v1 anchors (FNV hashes / byte totals computed with the real
reference) and size-independent properties (encode -> decode round trip, stack == concatenation
of single-frame encodes, sizes == prefix differences)."""
import ctypes as C
import io
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ALL_DTYPES = [np.uint8, np.int8, np.uint16, np.int16, np.uint32, np.int32]


def _host_encode(px2d, block=12):
    """[frames, n] numpy -> (stream bytes, offsets, prolix_bits) through trpx_encode_host."""
    from trpx_amd import _lib
    from trpx_amd.terse import _code
    L = _lib.lib()
    px2d = np.ascontiguousarray(px2d)
    f, n = px2d.shape
    cap = f * L.trpx_worst_case_bytes(_code(px2d.dtype), n, block)
    out = np.full(cap, 0xAA, np.uint8)     # poisoned: every output byte must be written
    total, pb = C.c_size_t(0), C.c_uint(0)
    offs = np.zeros(f + 1, np.uint64)
    _lib.check(L.trpx_encode_host(_code(px2d.dtype), px2d.ctypes.data, n, f, block, out.ctypes.data, cap,
                                  C.byref(total), offs.ctypes.data, C.byref(pb), -1))
    return out[: total.value].copy(), offs, int(pb.value)


def _host_decode(stream, offs, n, frames, dtype, block=12, stream_signed=None):
    from trpx_amd import _lib
    from trpx_amd.terse import _code
    L = _lib.lib()
    out = np.full((frames, n), 0x55, np.dtype(dtype))
    stream = np.ascontiguousarray(stream)
    _lib.check(L.trpx_decode_host(int(np.dtype(dtype).kind == "i") if stream_signed is None else int(stream_signed),
                                  _code(dtype, True), stream.ctypes.data, stream.size,
                                  offs.ctypes.data if offs is not None else None, n, frames, block,
                                  out.ctypes.data, -1))
    return out


def test_golden_fixtures_through_c_abi(gpu, golden):
    n_checked = 0
    for c in golden["cases"]:
        px = np.array(c["pixels"], np.dtype(c["dtype"]))
        s, offs, pb = _host_encode(px.reshape(1, -1), c["block"])
        assert s.tobytes().hex() == c["stream"], c["name"]
        assert pb == c["prolix_bits"], c["name"]
        assert int(offs[1]) == len(c["stream"]) // 2
        want = np.frombuffer(bytes.fromhex(c["stream"]), np.uint8)
        assert (_host_decode(want, None, px.size, 1, px.dtype, c["block"])[0] == px).all(), c["name"]
        n_checked += 1
    assert n_checked >= 36 and any(c["block"] != 12 for c in golden["cases"])


def test_golden_stack_layout(gpu, golden):
    st = golden["stack"]
    px = np.array(st["pixels"], np.uint16)
    s, offs, pb = _host_encode(px)
    assert s.tobytes().hex() == st["stream"]
    assert [int(x) for x in np.diff(offs)] == st["frame_sizes"]
    # decode with the offsets, and without (serial walk, the .trpx file case)
    assert (_host_decode(s, offs, px.shape[1], 3, np.uint16) == px).all()
    assert (_host_decode(s, None, px.shape[1], 3, np.uint16) == px).all()


@pytest.mark.parametrize("dtype", ALL_DTYPES)
def test_differential_vs_oracle_all_dtypes(gpu, oracle, dtype):
    rng = np.random.RandomState(11)
    dt = np.dtype(dtype)
    bits = dt.itemsize * 8
    top = bits - 1 if dt.kind == "i" else bits            # full width incl. cases outside the reference's D3 domain
    for n in (1, 5, 12, 13, 255 * 12, 256 * 12, 256 * 12 + 1, 3073, 7000, 12289):
        for frames in (1, 3):
            for hi in (0, 2, 7, min(top, 10), top):
                mag = rng.randint(0, 1 << hi, size=(frames, n), dtype=np.int64) if hi else np.zeros((frames, n), np.int64)
                if dt.kind == "i":
                    mag = mag * rng.choice([-1, 1], size=(frames, n))
                px = mag.astype(dt)
                if hi and rng.rand() < 0.5:                # blocks of zeros / runs of equal widths
                    px[:, : n // 2] = 0
                want, sizes, pb = oracle.encode_stack(px)
                got, offs, gpb = _host_encode(px)
                assert got.size == want.size and (np.diff(offs).astype(np.uint64) == sizes).all(), (dtype, n, frames, hi)
                assert (got == want).all(), (dtype, n, frames, hi)
                assert gpb == pb
                assert (_host_decode(want, offs, n, frames, dt) == px).all(), (dtype, n, frames, hi)


def test_type_extremes_round_trip(gpu, oracle):
    # outside the reference's validity domain (D3): oracle == GPU == mathematically consistent extension
    # (42 values: a partial last block of 6; 48 and 4800: whole blocks -- -INT32_MIN once overflowed in the single-pass encoder's
    #  width computation and came out as a 33-bit block, which the 42-value case never saw while sizes that are no multiples
    #  of 4 still took the two-pass pipeline)
    for dt in (np.int16, np.int32, np.int8):
        info = np.iinfo(dt)
        for reps in (7, 8, 800):
            px = np.array([info.min, info.max, -1, 0, 1, info.min + 1] * reps, dt).reshape(1, -1)
            want, sizes, pb = oracle.encode_stack(px)
            got, offs, gpb = _host_encode(px)
            assert (got == want).all() and gpb == pb == info.bits, (dt, reps)
            assert (_host_decode(got, offs, px.shape[1], 1, dt) == px).all(), (dt, reps)
    px = np.array([0xFFFFFFFF, 0x80000000, 1, 0] * 5, np.uint32).reshape(1, -1)
    got, offs, gpb = _host_encode(px)
    assert gpb == 32 and (got == oracle.encode_stack(px)[0]).all()
    assert (_host_decode(got, offs, px.shape[1], 1, np.uint32) == px).all()


def test_frame_sizes_that_are_no_multiples_of_four(gpu, oracle):
    # n % 4 != 0 -> frames start at any element; since round 3 they run on the tuned kernels too (the route matrix forces the others)
    rng = np.random.RandomState(3)
    for n in (13, 4097, 10001):
        px = rng.poisson(5, size=(4, n)).astype(np.uint16)
        want, sizes, pb = oracle.encode_stack(px)
        got, offs, _ = _host_encode(px)
        assert (got == want).all()
        assert (_host_decode(got, offs, n, 4, np.uint16) == px).all()
        assert (_host_decode(got, None, n, 4, np.uint16) == px).all()


def test_synth_generator_matches_oracle(gpu, oracle):
    from trpx_amd import codec
    for dt, n in ((np.uint16, 512 * 512), (np.int32, 300 * 300)):
        dev = codec.synth(dt, 5, 3, n, device=gpu).cpu().numpy()
        assert (dev == oracle.synth(dt, 5, 3, n)).all()


def _synth_anchors():
    import json
    return json.load(open(os.path.join(ROOT, "tests", "golden", "synth_anchors.json")))["anchors"]


def test_synth_anchors_u16_512(gpu, golden):
    """configs[1]/[2] data: GPU stream of frames 0..2 hashes to what the REAL reference produced."""
    import torch
    from trpx_amd import codec
    from oracle import oracle as O
    n = 512 * 512
    px = codec.synth(np.uint16, 0, 16, n, device=gpu)
    enc = codec.encode(px)
    torch.cuda.synchronize()
    enc.check()
    offs = enc.frame_offsets.cpu().numpy()
    data = enc.stack().cpu().numpy()
    assert int(offs[-1]) == golden["synth_u16_first16_total_bytes"]
    assert enc.prolix_bits() == 12
    for a in golden["anchors"]:
        if a["dtype"] != "uint16":
            continue
        f = a["frame"]
        s = data[offs[f]:offs[f + 1]]
        assert s.size == a["size"]
        assert f"{O.fnv1a64(s):016x}" == a["stream_fnv"]
        assert s[:16].tobytes().hex() == a["first16"]
    back, status = codec.decode(enc.stack(), enc.frame_offsets, n, 16, np.uint16)
    torch.cuda.synchronize()
    assert int(status[0].item()) == 0 and torch.equal(back.view(torch.int16), px.view(torch.int16))


def test_synth_anchor_i32_4096(gpu, golden):
    """configs[3]: 4096x4096 int32 with sparse peaks (wide bit-width / 12-bit header path)."""
    import torch
    from trpx_amd import codec
    from oracle import oracle as O
    a = [x for x in golden["anchors"] if x["dtype"] == "int32"][0]
    n = a["n"]
    # ... and frames 1 .. 7 -- the eight frames bench.py times -- from tests/golden/synth_anchors.json (the real reference again:
    # tests/golden/make_anchors.py)
    more = [x for x in _synth_anchors() if x["dtype"] == "int32"]
    assert [x["frame"] for x in more] == list(range(1, 8))
    px = codec.synth(np.int32, 0, 8, n, device=gpu)
    enc = codec.encode(px)
    torch.cuda.synchronize()
    enc.check()
    offs = enc.frame_offsets.cpu().numpy()
    for x in [a] + more:
        f = x["frame"]
        s = enc.data[offs[f]: offs[f + 1]].cpu().numpy()
        assert s.size == x["size"] and enc.prolix_bits() == x["prolix_bits"], f
        assert f"{O.fnv1a64(s):016x}" == x["stream_fnv"] and s[:16].tobytes().hex() == x["first16"], f
    back, status = codec.decode(enc.stack(), enc.frame_offsets, n, 8, np.int32)
    torch.cuda.synchronize()
    assert int(status[0].item()) == 0 and torch.equal(back, px)


def test_full_stack_2000_frames_properties(gpu):
    """configs[1]+[2] at full size: total bytes == the reference's 203 596 114, round trip exact,
    every frame of the stack == its own single-frame encode (checked on a sample)."""
    import torch
    from trpx_amd import codec
    n, frames = 512 * 512, 2000
    px = codec.synth(np.uint16, 0, frames, n, device=gpu)
    enc = codec.encode(px)
    torch.cuda.synchronize()
    enc.check()
    assert enc.total_bytes() == 203596114          # SURVEY.md section 8 row d, computed with the real reference
    assert enc.prolix_bits() == 12
    offs = enc.frame_offsets
    assert int(offs[0].item()) == 0 and bool((offs[1:] > offs[:-1]).all())
    back, status = codec.decode(enc.stack(), offs, n, frames, np.uint16)
    torch.cuda.synchronize()
    assert int(status[0].item()) == 0
    assert torch.equal(back.view(torch.int16), px.view(torch.int16))
    for f in (0, 1, 999, 1999):
        single = codec.encode(px[f:f + 1])
        torch.cuda.synchronize()
        a, b = int(offs[f].item()), int(offs[f + 1].item())
        assert single.total_bytes() == b - a
        assert torch.equal(single.stack(), enc.data[a:b])
    # frames 1000 and 1999 against what the REAL reference made of them (tests/golden/synth_anchors.json)
    from oracle import oracle as O
    for x in [x for x in _synth_anchors() if x["dtype"] == "uint16"]:
        f = x["frame"]
        a, b = int(offs[f].item()), int(offs[f + 1].item())
        s = enc.data[a:b].cpu().numpy()
        assert s.size == x["size"] and f"{O.fnv1a64(s):016x}" == x["stream_fnv"] and s[:16].tobytes().hex() == x["first16"], f
        assert f"{O.fnv1a64(px[f].cpu().numpy()):016x}" == x["pixels_fnv"], f


def test_capacity_error_and_sizes_only_query(gpu):
    import torch
    from trpx_amd import codec, _lib
    px = codec.synth(np.uint16, 0, 4, 512 * 512, device=gpu)
    full = codec.encode(px)
    torch.cuda.synchronize()
    for path in (0, 1):                                    # single-pass and two-pass encoders
        _lib.lib().trpx_set_encode_path(path)
        try:
            big = torch.full((1 << 20,), 7, dtype=torch.uint8, device=gpu)
            enc = codec.encode(px, out=big[:150000])        # room for one frame, not for four
            torch.cuda.synchronize()
            assert int(enc.status[0].item()) == _lib.ERR_CAPACITY
            assert bool((big[150000:] == 7).all()), "nothing beyond out_capacity may ever be touched"
            assert torch.equal(full.frame_offsets, enc.frame_offsets), "sizes must still be reported"
        finally:
            _lib.lib().trpx_set_encode_path(0)


def test_single_pass_and_two_pass_encoders_agree(gpu):
    import torch
    from trpx_amd import codec, _lib
    for dt, n, frames in ((np.uint16, 512 * 512, 24), (np.int32, 1000 * 1000, 3), (np.uint16, 12 * 1024 * 5 + 8, 7)):
        px = codec.synth(dt, 3, frames, n, device=gpu)
        a = codec.encode(px)
        _lib.lib().trpx_set_encode_path(1)
        try:
            b = codec.encode(px)
        finally:
            _lib.lib().trpx_set_encode_path(0)
        torch.cuda.synchronize()
        a.check(); b.check()
        assert torch.equal(a.frame_offsets, b.frame_offsets)
        assert a.prolix_bits() == b.prolix_bits()
        assert torch.equal(a.stack(), b.stack())


def test_corrupt_and_truncated_streams_are_detected(gpu, oracle):
    from trpx_amd import _lib, TrpxError
    rng = np.random.RandomState(5)
    px = rng.poisson(4, size=(2, 5000)).astype(np.uint16)
    s, offs, _ = _host_encode(px)
    with pytest.raises(TrpxError) as e:                    # truncated
        _host_decode(s[: s.size - 40], None, 5000, 2, np.uint16)
    assert e.value.code == _lib.ERR_CORRUPT
    bad_offs = offs.copy()
    bad_offs[1] -= 3                                       # frame boundary in the wrong place
    with pytest.raises(TrpxError):
        _host_decode(s, bad_offs, 5000, 2, np.uint16)


def test_terse_class_mirrors_reference_surface(gpu, oracle, golden, tmp_path):
    from trpx_amd import Terse
    # README example (Terse.hpp:127-154)
    numbers = np.arange(-500, 500, dtype=np.int32)
    t = Terse(numbers)
    assert (t.terse_size(), t.bits_per_val(), t.is_signed(), t.size(), t.number_of_frames()) == (1152, 10, True, 1000, 1)
    buf = io.BytesIO()
    t.write(buf)
    want_hdr = ('<Terse prolix_bits="10" signed="1" block="12" memory_size="1152" number_of_values="1000" '
                'number_of_frames="1"/>')
    assert buf.getvalue().startswith(want_hdr.encode())
    buf.seek(0)
    r = Terse.read(buf)
    assert (r.prolix(np.empty(1000, np.int32)) == numbers).all()
    assert buf.read() == b""                               # stream left right after the payload
    # multi-frame with dims == the reference's own push_back stack file, byte for byte
    st = golden["stack"]
    px = np.array(st["pixels"], np.uint16).reshape(3, *st["dims"])
    t = Terse()
    for f in range(3):
        t.push_back(px[f])
    assert t.dim() == st["dims"]
    out = io.BytesIO()
    t.write(out)
    assert out.getvalue() == st["header"].encode() + bytes.fromhex(st["stream"])
    out.seek(0)
    r = Terse.read(out)
    assert r.frame_sizes() == st["frame_sizes"] and r.dim() == st["dims"]
    for f in (2, 0, 1):                                    # out-of-order access works (reference defect D1)
        assert (r.prolix(np.empty(700, np.uint16), f) == px[f].reshape(-1)).all()
    assert (r.prolix_stack(np.uint16) == px.reshape(3, -1)).all()
    # error conventions
    with pytest.raises(ValueError):
        t.push_back(np.zeros(10, np.uint16))
    with pytest.raises(ValueError):
        t.prolix(np.empty(700, np.uint16), 3)
    with pytest.raises(ValueError):
        Terse(np.arange(-5, 5, dtype=np.int16)).prolix(np.empty(10, np.uint16))


def test_cpp_drop_in_example(gpu, tmp_path):
    exe = os.path.join(ROOT, "tests", "cpp", "terse_example")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpp")])
    r = subprocess.run([exe, str(tmp_path / "junk.terse")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr


def test_cpp_sharded_example(gpu):
    """trpx_encode_sharded / trpx_decode_sharded called from C++ on a one-rank RCCL communicator (tests/cpp/sharded_example.cpp):
    the stack equals trpx::Terse's single-process encode of the same frames, the pixels come back exactly."""
    exe = os.path.join(ROOT, "tests", "cpp", "sharded_example")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpp")])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK sharded example" in r.stdout, r.stdout + r.stderr


def test_integration_md_snippets_run(gpu):
    """INTEGRATION.md section 2's patched bodies inside a stand-in for the reference class (tests/cpp/integration_snippets.cpp)."""
    exe = os.path.join(ROOT, "tests", "cpp", "integration_snippets")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpp")])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK integration snippets" in r.stdout, r.stdout + r.stderr


def test_decode_index_side_channel(gpu, oracle):
    """the encoder's decode index == the index the header walk builds from the stream, and
    walk-free decode with it is pixel-identical (u16 1024-block tiles, int32 512-block tiles, ragged frame end,
    both encoder paths)."""
    import torch
    from trpx_amd import codec, _lib
    # (frames of more than 32 K blocks: trpx_build_index takes the index route's one walk of many short parts, decode_part.hip)
    for dt, n, frames in ((np.uint16, 512 * 512, 9), (np.int32, 700 * 700, 3), (np.uint16, 12 * 1024 * 3 + 16, 5),
                          (np.int32, 1030 * 1065, 4), (np.uint16, 1500 * 1500 + 7, 3)):
        px = codec.synth(dt, 11, frames, n, device=gpu)
        for path in (0, 1):
            _lib.lib().trpx_set_encode_path(path)
            try:
                enc = codec.encode(px, index=True)
            finally:
                _lib.lib().trpx_set_encode_path(0)
            torch.cuda.synchronize()
            enc.check()
            walked = codec.build_index(enc.stack(), enc.frame_offsets, n, frames, dt)
            torch.cuda.synchronize()
            # the index buffers are only defined where a block / group exists: compare through a decode and bytewise
            nb = (n + 11) // 12
            ng = (nb + 255) // 256
            w_off = (8 * frames * ng + 15) // 16 * 16
            assert torch.equal(enc.index[: 8 * frames * ng], walked[: 8 * frames * ng]), (dt, n, path)
            assert torch.equal(enc.index[w_off: w_off + frames * nb], walked[w_off: w_off + frames * nb]), (dt, n, path)
            back, st = codec.decode(enc.stack(), enc.frame_offsets, n, frames, dt, index=enc.index)
            torch.cuda.synchronize()
            assert int(st[0].item()) == 0
            assert torch.equal(back.view(torch.int32 if dt == np.int32 else torch.int16),
                               px.view(torch.int32 if dt == np.int32 else torch.int16))
    # a wrong index is rejected, not decoded into garbage silently: by the tiled kernel (route 2) and by the per-frame
    # decoder with the widths given (route 3; a wrong group offset, a width that is not the stream's, a width wider than the type)
    px = codec.synth(np.uint16, 0, 2, 512 * 512, device=gpu)
    enc = codec.encode(px, index=True)
    nb, ng = (512 * 512 + 11) // 12, ((512 * 512 + 11) // 12 + 255) // 256
    w_off = (8 * 2 * ng + 15) // 16 * 16
    try:
        for route in (2, 3):
            assert _lib.lib().trpx_set_decode_path(route) == 0
            spoils = [("offset", slice(0, 8), 255)] if route == 2 else \
                     [("offset", slice(8, 9), 1), ("width", slice(w_off + 5000, w_off + 5001), 1), ("wide", slice(w_off + nb + 77, w_off + nb + 78), 17)]
            for what, where, val in spoils:
                bad = enc.index.clone()
                bad[where] = (bad[where] + val) if what != "wide" else val
                back, st = codec.decode(enc.stack(), enc.frame_offsets, 512 * 512, 2, np.uint16, index=bad)
                torch.cuda.synchronize()
                assert int(st[0].item()) == _lib.ERR_CORRUPT, (route, what)
    finally:
        _lib.lib().trpx_set_decode_path(0)


@pytest.mark.parametrize("dtype", ALL_DTYPES)
def test_many_small_frames_take_the_per_frame_decoder(gpu, oracle, dtype):
    """Stacks of small frames select k_decode_frames at any frame count (walker wave + extraction waves per workgroup): differential test
    against the oracle for every pixel type, mixed widths, partial last blocks, tiny and multi-super-step frames."""
    rng = np.random.RandomState(23)
    dt = np.dtype(dtype)
    bits = dt.itemsize * 8
    top = bits - 1 if dt.kind == "i" else bits
    for n, frames in ((1, 130), (11, 130), (388, 150), (12 * 768 + 4, 140), (9216 * 3 + 8, 129)):
        hi = rng.randint(0, top + 1, size=(frames, (n + 11) // 12))            # a width for every block
        hi[rng.rand(*hi.shape) < 0.6] = 3 if top >= 3 else 1                    # long runs of one width + outliers
        mag = np.zeros((frames, n), np.int64)
        for k in range(12):
            cols = np.arange(k, n, 12)
            h = hi[:, : cols.size]
            mag[:, cols] = (rng.rand(frames, cols.size) * (2.0 ** h)).astype(np.int64)
        if dt.kind == "i":
            mag = mag * rng.choice([-1, 1], size=mag.shape)
            mag = np.clip(mag, np.iinfo(dt).min, np.iinfo(dt).max)
        px = mag.astype(dt)
        want, sizes, pb = oracle.encode_stack(px)
        got, offs, gpb = _host_encode(px)
        assert got.size == want.size and (got == want).all() and gpb == pb, (dtype, n)
        assert (np.diff(offs).astype(np.uint64) == sizes).all()
        assert (_host_decode(want, offs, n, frames, dt) == px).all(), (dtype, n)
    # a corrupt stack is reported, not decoded
    from trpx_amd import _lib, TrpxError
    bad = want.copy()
    bad[int(offs[5]) : int(offs[5]) + 6] ^= 0xFF
    try:
        out = _host_decode(bad, offs, n, frames, dt)
        assert not (out == px).all()          # (flipping payload bits may still parse: then pixels differ)
    except TrpxError as e:
        assert e.code == _lib.ERR_CORRUPT


@pytest.mark.parametrize("block", [1, 3, 7, 16, 100, 4096])
def test_any_block_size_generic_kernels(gpu, oracle, block):
    """The `block` argument of Terse(Iterator, size, block) (Terse.hpp:263-270): every size other than the tuned
    default 12 goes through the generic kernels; differential test against the oracle."""
    rng = np.random.RandomState(block)
    for dt in (np.uint8, np.int16, np.uint16, np.int32):
        info = np.iinfo(dt)
        for n, frames in ((1, 2), (block, 1), (block * 3 + 1, 3), (5000, 2), (70001, 2)):
            hi = rng.randint(0, info.bits - (1 if dt().dtype.kind == "i" else 0) + 1)
            mag = (rng.rand(frames, n) * (2.0 ** hi)).astype(np.int64)
            mag[:, : n // 3] = 0
            if np.dtype(dt).kind == "i":
                mag = np.clip(mag * rng.choice([-1, 1], size=mag.shape), info.min, info.max)
            px = mag.astype(dt)
            parts = [oracle.encode(px[f], block) for f in range(frames)]
            want = np.concatenate([p[0] for p in parts])
            got, offs, gpb = _host_encode(px, block)
            assert got.size == want.size and (got == want).all(), (block, dt, n)
            assert gpb == max(p[1] for p in parts)
            assert [int(x) for x in np.diff(offs)] == [p[0].size for p in parts]
            assert (_host_decode(want, offs, n, frames, dt, block) == px).all(), (block, dt, n)
            assert (_host_decode(want, None, n, frames, dt, block) == px).all(), (block, dt, n)


def test_terse_class_with_block_argument(gpu, oracle, golden):
    from trpx_amd import Terse
    c = [c for c in golden["cases"] if c["name"] == "block16_u16"][0]
    px = np.array(c["pixels"], np.uint16)
    t = Terse(px, block=16)
    buf = io.BytesIO()
    t.write(buf)
    assert buf.getvalue() == c["header"].encode() + bytes.fromhex(c["stream"])   # the reference's own file, byte for byte
    buf.seek(0)
    r = Terse.read(buf)
    assert (r.prolix(np.empty(px.size, np.uint16)) == px).all()


def test_converting_decode_cross_type_and_float(gpu, oracle):
    """Rows a9 (clamping into narrower types, Bit_pointer.hpp:747-763) and a10 (float / double output,
    Terse.hpp:379-383): trpx_decode_convert against the oracle's value semantics."""
    rng = np.random.RandomState(41)
    for src in (np.uint16, np.int16, np.int32, np.uint32):
        s_signed = np.dtype(src).kind == "i"
        info = np.iinfo(src)
        n, frames = 5003, 3
        mag = (rng.rand(frames, n) * (2.0 ** rng.randint(1, info.bits - (1 if s_signed else 0), size=(frames, 1)))).astype(np.int64)
        mag[:, ::7] = 0
        if s_signed:
            mag = np.clip(mag * rng.choice([-1, 1], size=mag.shape), info.min, info.max)
        px = mag.astype(src)
        stream, sizes, pb = oracle.encode_stack(px)
        offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
        for dst in (np.uint8, np.int8, np.uint16, np.int16, np.uint32, np.int32, np.float32, np.float64):
            if s_signed and np.dtype(dst).kind == "u":
                continue                                   # the reference asserts against it (Terse.hpp:356-357)
            got = _host_decode(stream, offs, n, frames, dst, stream_signed=s_signed)
            if np.dtype(dst).kind == "f":
                want = px.astype(dst)                      # exact: |v| < 2^32 fits double; float32 rounds like a cast
            else:
                di = np.iinfo(dst)
                want = np.clip(px.astype(np.int64), di.min, di.max).astype(dst)
            # the oracle's own decode path into the same output type (float / double: Terse.hpp:379-383)
            chk = np.stack([oracle.decode(stream[int(offs[f]):int(offs[f + 1])], n, dst, stream_signed=s_signed) for f in range(frames)])
            assert (chk == want).all(), (src, dst)
            assert (got == chk).all(), (src, dst)
    # the class surface: decode a u16 object into int32 / float64 containers
    from trpx_amd import Terse
    px = rng.randint(0, 60000, size=4000).astype(np.uint16)
    t = Terse(px)
    assert (t.prolix(np.empty(4000, np.int32)) == px.astype(np.int32)).all()
    assert (t.prolix(np.empty(4000, np.float64)) == px.astype(np.float64)).all()
    assert (t.prolix(np.empty(4000, np.uint8)) == np.minimum(px, 255).astype(np.uint8)).all()


@pytest.mark.parametrize("dtype", [np.uint16, np.int32, np.uint8])
def test_encoder_chains_across_many_tiles_and_frames(gpu, oracle, dtype):
    """Stresses the single-pass encoder's two chains (encode_fused.hip): one-tile frames in their hundreds (the frame
    chain beyond one 64-frame look-back window, first tile == last tile), frames of a few tiles with a partial last
    tile, and frames of more than 64 tiles (tile look-back beyond one window); widths vary from block to block so
    that tile boundaries fall on arbitrary bits.  Vector-aligned sizes only (n % 4 == 0) -- the fused path."""
    rng = np.random.RandomState(7)
    dt = np.dtype(dtype)
    tile_values = (1024 if dt.itemsize <= 2 else 768) * 12
    top = 8 * dt.itemsize - (1 if dt.kind == "i" else 0)
    for frames, n in ((700, 1000), (200, 4), (130, 2 * tile_values + 40), (3, 70 * tile_values + 8), (67, tile_values)):
        # per-block magnitude: mostly narrow, some wide, so that widths change almost every block
        nblk = (n + 11) // 12
        hi = rng.choice([0, 1, 2, 3, 5, min(9, top), top], size=(frames, nblk), p=[0.1, 0.2, 0.3, 0.2, 0.1, 0.07, 0.03])
        mag = (rng.rand(frames, nblk * 12) * (2.0 ** np.repeat(hi, 12, axis=1))).astype(np.int64)[:, :n]
        if dt.kind == "i":
            mag = np.clip(mag * rng.choice([-1, 1], size=mag.shape), np.iinfo(dt).min, np.iinfo(dt).max)
        px = mag.astype(dt)
        want, sizes, pb = oracle.encode_stack(px)
        got, offs, gpb = _host_encode(px)
        assert (np.diff(offs).astype(np.uint64) == sizes).all(), (dtype, frames, n)
        assert got.size == want.size and (got == want).all(), (dtype, frames, n)
        assert gpb == pb
        assert (_host_decode(got, offs, n, frames, dt) == px).all(), (dtype, frames, n)
        assert (_host_decode(got, None, n, frames, dt) == px).all(), (dtype, frames, n)   # frames located by the device walk


@pytest.mark.parametrize("dtype", ALL_DTYPES)
def test_two_pass_pipeline_and_basic_decoder_all_dtypes(gpu, oracle, dtype):
    """The fallback pipelines (two-pass encode: encode.hip; basic walk + unpack: decode.hip) against the oracle for every
    pixel type, with widths that change from block to block, three times over (a byte-packing code path of the basic
    8-bit unpack kernel once gave run-to-run different pixels)."""
    import torch
    from trpx_amd import codec, _lib
    rng = np.random.RandomState(5)
    dt = np.dtype(dtype)
    top = 8 * dt.itemsize - (1 if dt.kind == "i" else 0)
    frames, n = 150, 1000
    nblk = (n + 11) // 12
    hi = rng.choice([0, 1, 2, 3, 5, min(9, top), top], size=(frames, nblk), p=[0.1, 0.2, 0.3, 0.2, 0.1, 0.07, 0.03])
    mag = (rng.rand(frames, nblk * 12) * (2.0 ** np.repeat(hi, 12, axis=1))).astype(np.int64)[:, :n]
    if dt.kind == "i":
        mag = np.clip(mag * rng.choice([-1, 1], size=mag.shape), np.iinfo(dt).min, np.iinfo(dt).max)
    px = mag.astype(dt)
    want, sizes, pb = oracle.encode_stack(px)
    dpx = torch.from_numpy(px.view(np.int8 if dt.itemsize == 1 else np.int16 if dt.itemsize == 2 else np.int32)).to(gpu).view(
        {1: torch.uint8 if dt.kind == "u" else torch.int8, 2: torch.uint16 if dt.kind == "u" else torch.int16,
         4: torch.uint32 if dt.kind == "u" else torch.int32}[dt.itemsize])
    _lib.lib().trpx_set_encode_path(1)
    try:
        enc = codec.encode(dpx)
        torch.cuda.synchronize()
    finally:
        _lib.lib().trpx_set_encode_path(0)
    enc.check()
    assert enc.stack().cpu().numpy().tobytes() == want.tobytes()
    assert enc.prolix_bits() == pb
    for _ in range(3):
        back, st = codec.decode(enc.stack(), None, n, frames, dt)        # no offsets: serial walk + basic unpack
        torch.cuda.synchronize()
        assert int(st[0]) == 0
        assert (back.cpu().numpy().reshape(frames, n).view(dt) == px).all()


@pytest.mark.parametrize("dtype", ALL_DTYPES)
def test_every_decode_path_every_dtype_block_to_block_width_changes(gpu, oracle, dtype):
    """One noisy stack per pixel type (the width changes with almost every block, including the type's full width),
    encoded by the single-pass encoder with the decode index, then decoded along every route the library has:
    per-frame decoder (auto), tiled decoder (forced), walk-free indexed decoder, serial walk without offsets."""
    import torch
    from trpx_amd import codec
    rng = np.random.RandomState(99)
    dt = np.dtype(dtype)
    top = 8 * dt.itemsize - (1 if dt.kind == "i" else 0)
    tdt = {1: torch.uint8 if dt.kind == "u" else torch.int8, 2: torch.uint16 if dt.kind == "u" else torch.int16,
           4: torch.uint32 if dt.kind == "u" else torch.int32}[dt.itemsize]
    for frames, n in ((140, 3000), (6, 40000)):
        nblk = (n + 11) // 12
        hi = rng.choice([0, 1, 2, 3, 5, min(9, top), top], size=(frames, nblk), p=[0.1, 0.2, 0.3, 0.2, 0.1, 0.07, 0.03])
        mag = (rng.rand(frames, nblk * 12) * (2.0 ** np.repeat(hi, 12, axis=1))).astype(np.int64)[:, :n]
        if dt.kind == "i":
            mag = np.clip(mag * rng.choice([-1, 1], size=mag.shape), np.iinfo(dt).min, np.iinfo(dt).max)
        px = mag.astype(dt)
        want, sizes, pb = oracle.encode_stack(px)
        dpx = torch.from_numpy(px.view(np.dtype(f"i{dt.itemsize}"))).to(gpu).view(tdt)
        enc = codec.encode(dpx, index=True)
        torch.cuda.synchronize()
        enc.check()
        assert enc.stack().cpu().numpy().tobytes() == want.tobytes(), (dtype, frames)
        for kind in ("offsets", "index", "walk"):
            back, st = codec.decode(enc.stack(), None if kind == "walk" else enc.frame_offsets, n, frames, dt,
                                    index=enc.index if kind == "index" else None)
            torch.cuda.synchronize()
            assert int(st[0]) == 0, (dtype, frames, kind)
            assert (back.cpu().numpy().reshape(frames, n).view(dt) == px).all(), (dtype, frames, kind)


def test_64bit_integer_containers_are_narrowed_when_they_fit(gpu, oracle):
    """int64 / uint64 input (what src/terse.cpp:120-123 makes of float images): same stream as the oracle's 64-bit
    encode when every value fits 32 bits, refused otherwise."""
    from trpx_amd import Terse
    rng = np.random.RandomState(3)
    for dt in (np.int64, np.uint64):
        px = rng.randint(0, 2000, size=(3, 5000)).astype(np.int64)
        px[:, ::97] *= 1000
        if dt == np.int64:
            px -= 700
        px = px.astype(dt)
        want, sizes, pb = oracle.encode_stack(px)
        t = Terse()
        t.push_back_stack(px)
        assert t.data() == want.tobytes() and t.frame_sizes() == [int(x) for x in sizes] and t.bits_per_val() == pb
        assert t.is_signed() == (dt == np.int64)
        back = t.prolix_stack(np.int32 if dt == np.int64 else np.uint32)
        assert (back.astype(np.int64) == px.astype(np.int64)).all()
    wide = Terse(np.array([0, 1 << 40, -5, 7], np.int64))                   # wider values: 64-bit pixels (test_64bit_containers_with_wide_values)
    assert wide.bits_per_val() == 42 and (wide.prolix(np.zeros(4, np.int64)) == [0, 1 << 40, -5, 7]).all()


def test_more_frames_than_one_grid_slice(gpu, oracle):
    """The single-pass encoder's grid is (tiles per frame, frames in slices of 32768): 40 000 tiny frames cross into
    the second slice (and give the frame chain ~600 look-back windows)."""
    rng = np.random.RandomState(17)
    frames, n = 40000, 48
    px = (rng.rand(frames, n) * (2.0 ** rng.randint(0, 12, size=(frames, 1)))).astype(np.uint16)
    want, sizes, pb = oracle.encode_stack(px)
    got, offs, gpb = _host_encode(px)
    assert (np.diff(offs).astype(np.uint64) == sizes).all()
    assert got.size == want.size and (got == want).all() and gpb == pb
    assert (_host_decode(got, offs, n, frames, np.uint16) == px).all()


@pytest.mark.parametrize("dtype", ALL_DTYPES)
def test_width_changes_every_block_position_parallel_walk(gpu, oracle, dtype):
    """Streams with an explicit header on EVERY block (no two neighbouring blocks share a width; Terse.hpp:360-372 makes
    the header chain one dependent step per block) for all six pixel types, against the oracle:
    a 130-frame stack (per-frame decoder -> its walker defers such frames to the position-parallel walk, decode_seg.hip:
    one wavefront per frame, several LDS windows per segment) and three large frames (tiled decoder: several wavefronts
    per frame, links between them closed by k_seg_resolve).  Also the walk's product itself -- width[b] of every block --
    against oracle.widths(), for the encoder's index and for the index rebuilt from the stream."""
    import torch
    from trpx_amd import codec
    rng = np.random.RandomState(7)
    dt = np.dtype(dtype)
    top = 8 * dt.itemsize - (1 if dt.kind == "i" else 0)
    tdt = {1: torch.uint8 if dt.kind == "u" else torch.int8, 2: torch.uint16 if dt.kind == "u" else torch.int16,
           4: torch.uint32 if dt.kind == "u" else torch.int32}[dt.itemsize]
    choices = np.array([1, 2, 3, 4, 6, min(8, top), top])
    for frames, n in ((130, 256 * 512), (3, 1024 * 1024 + 4)):
        nblk = (n + 11) // 12
        step = rng.randint(1, len(choices), size=(frames, nblk))             # never 0: the next block's width differs
        hi = choices[np.cumsum(step, axis=1) % len(choices)]
        mag = (rng.rand(frames, nblk * 12) * (2.0 ** np.repeat(hi, 12, axis=1))).astype(np.int64)
        mag[:, ::12] = (2 ** np.minimum(hi, 62) - 1)                          # the block's width is exactly hi (signed: + 1)
        mag = mag[:, :n]
        if dt.kind == "i":
            mag = np.minimum(mag, 2 ** (top - 1) - 1) * rng.choice([-1, 1], size=mag.shape)
        px = mag.astype(dt)
        want, sizes, pb = oracle.encode_stack(px)
        dpx = torch.from_numpy(px.view(np.dtype(f"i{dt.itemsize}"))).to(gpu).view(tdt)
        enc = codec.encode(dpx, index=True)
        torch.cuda.synchronize()
        enc.check()
        assert enc.stack().cpu().numpy().tobytes() == want.tobytes(), (dtype, frames)
        back, st = codec.decode(enc.stack(), enc.frame_offsets, n, frames, dt)
        torch.cuda.synchronize()
        assert int(st[0]) == 0, (dtype, frames)
        assert (back.cpu().numpy().reshape(frames, n).view(dt) == px).all(), (dtype, frames)
        walked = codec.build_index(enc.stack(), enc.frame_offsets, n, frames, dt)
        torch.cuda.synchronize()
        ng = (nblk + 255) // 256
        w_off = (8 * frames * ng + 15) // 16 * 16
        want_w = np.stack([oracle.widths(px[f]) for f in range(frames)]).astype(np.uint8)
        for name, idx in (("encoder", enc.index), ("walk", walked)):
            got_w = idx[w_off: w_off + frames * nblk].cpu().numpy().reshape(frames, nblk)
            assert (got_w == want_w).all(), (dtype, frames, name)


def test_sharded_c_abi_size_gather_over_rccl(gpu, oracle):
    """SURVEY row e through the C ABI: HIP encode of this rank's shard -> trpx_gather_frame_offsets (pack kernel,
    ncclAllGather on a real RCCL communicator -- one rank here, the N > 1 logic runs under gloo in tests/test_sharded.py --
    scan kernel) -> the shard written at its global byte offset -> decode with the GLOBAL offsets == the pixels, and the
    assembled stack == the oracle's single-process stack (Terse.hpp:502-505: frames are independent, byte aligned)."""
    import torch
    from trpx_amd import codec, sharded
    frames, n = 37, 64 * 64
    px = codec.synth(np.uint16, 5, frames, n, device=gpu)
    enc = codec.encode(px)
    torch.cuda.synchronize()
    enc.check()
    g = sharded.RcclSizeGather(frames, gpu)
    try:
        for _ in range(2):                                              # buffers and communicator are reused call after call
            goffs, base, pb = g(enc.frame_offsets, enc.status)
        torch.cuda.synchronize()
        assert torch.equal(goffs, enc.frame_offsets) and int(base) == 0 and int(pb) == enc.prolix_bits()
        want, sizes, opb = oracle.encode_stack(px.cpu().numpy())
        assert int(goffs[-1]) == want.size and opb == int(pb)
        global_stack = torch.zeros((want.size + 15) // 16 * 16, dtype=torch.uint8, device=gpu)
        global_stack[int(base): int(base) + enc.total_bytes()] = enc.stack()        # payload stays on its GPU, at its global offset
        assert global_stack[: want.size].cpu().numpy().tobytes() == want.tobytes()
        back, st = codec.decode(global_stack, goffs, n, frames, np.uint16)
        torch.cuda.synchronize()
        assert int(st[0]) == 0 and torch.equal(back.view(torch.int16), px.view(torch.int16))
        # ragged message: a slot larger than the shard (n_slot > n_local), as unequal shards use it
        from trpx_amd import _lib
        L = _lib.lib()
        ws = torch.empty(L.trpx_gather_workspace_bytes(frames + 11, 1), dtype=torch.uint8, device=gpu)
        g2 = torch.zeros(frames + 12, dtype=torch.int64, device=gpu)
        rc = L.trpx_gather_frame_offsets(g.comm, enc.frame_offsets.data_ptr(), frames, frames + 11, None, g2.data_ptr(), None, None,
                                         ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert rc == 0 and torch.equal(g2[: frames + 1], enc.frame_offsets)
        # the N > 1 arithmetic of the two kernels, ranks emulated on this GPU: three ragged shards (one empty), 700-frame
        # slot (several scan chunks), messages concatenated in rank order as ncclAllGather leaves them
        rng = np.random.RandomState(4)
        counts, slot = [700, 0, 333], 700
        sizes = [rng.randint(1, 200000, size=c).astype(np.int64) for c in counts]
        msgs = torch.zeros(3 * (slot + 2), dtype=torch.int64, device=gpu)
        st_words = torch.zeros(8, dtype=torch.int32, device=gpu)
        for r, sz in enumerate(sizes):
            lo = torch.from_numpy(np.concatenate([[0], np.cumsum(sz)])).to(gpu)
            st_words[1] = 5 + 3 * r
            assert L.trpx_gather_pack(lo.data_ptr(), counts[r], slot, st_words.data_ptr(), msgs[r * (slot + 2):].data_ptr(), None) == 0
            torch.cuda.synchronize()
        go = torch.zeros(sum(counts) + 1, dtype=torch.int64, device=gpu)
        pbw = torch.zeros(2, dtype=torch.int32, device=gpu)
        rb = torch.zeros(3, dtype=torch.int64, device=gpu)
        assert L.trpx_gather_scan(msgs.data_ptr(), 3, slot, go.data_ptr(), pbw.data_ptr(), rb.data_ptr(), None) == 0
        torch.cuda.synchronize()
        want_go = np.concatenate([[0], np.cumsum(np.concatenate(sizes))])
        assert (go.cpu().numpy() == want_go).all() and int(pbw[0]) == 11
        assert rb.cpu().numpy().tolist() == [0, int(want_go[700]), int(want_go[700])]
    finally:
        g.close()


def test_sharded_single_call_entry_points(gpu, oracle):
    """SURVEY row b: trpx_encode_sharded (this rank's trpx_encode + the size gather on the caller's communicator, in one
    stream-ordered call, the gather optionally on a second stream) and trpx_decode_sharded (this rank's frames expanded from the
    GLOBAL offset table) on a real one-rank RCCL communicator: the stack is the oracle's, the table the local offsets, the
    pixels exact; then the decode with a table in which this rank's frames do not start at byte 0 (ranks in front of it)."""
    import torch
    from trpx_amd import codec, sharded, _lib
    frames, n = 41, 96 * 96 + 5
    px = codec.synth(np.uint16, 9, frames, n, device=gpu)
    want, sizes, opb = oracle.encode_stack(px.cpu().numpy())
    sc = sharded.ShardedCodec(frames, n, np.uint16, gpu)
    try:
        side = torch.cuda.Stream(device=gpu)
        for gs in (None, side, None, side):                              # same buffers call after call, both stream arrangements
            goffs, base, pb = sc.encode(px, gather_stream=gs)
            torch.cuda.synchronize()
            assert int(sc.status[0]) == 0 and int(pb) == opb and int(base) == 0
            assert int(goffs[-1]) == want.size and torch.equal(goffs, sc.local_offsets)
            assert sc.out[: want.size].cpu().numpy().tobytes() == want.tobytes()
        back = torch.zeros((frames, n), dtype=torch.uint16, device=gpu)
        _, sd = sc.decode(back)
        torch.cuda.synchronize()
        assert int(sd[0]) == 0 and torch.equal(back.view(torch.int16), px.view(torch.int16))
        # a global table with 7 frames of other ranks in front of this rank's: trpx_decode_sharded(first_frame = 7)
        L = _lib.lib()
        front = torch.tensor(np.concatenate([[0], np.cumsum(np.arange(1000, 1007))]), dtype=torch.int64, device=gpu)
        table = torch.cat([front[:-1], sc.local_offsets + front[-1]])
        back.zero_()
        ws = torch.empty(L.trpx_decode_sharded_workspace_bytes(_lib.U16, n, frames, 12), dtype=torch.uint8, device=gpu)
        st = torch.empty(8, dtype=torch.int32, device=gpu)
        rc = L.trpx_decode_sharded(0, _lib.U16, sc.out.data_ptr(), sc.out.numel(), table.data_ptr(), 7, n, frames, 12, back.data_ptr(),
                                   st.data_ptr(), ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert rc == 0 and int(st[0]) == 0 and torch.equal(back.view(torch.int16), px.view(torch.int16))
        # argument checks: a workspace that is too small, no communicator
        assert L.trpx_decode_sharded(0, _lib.U16, sc.out.data_ptr(), sc.out.numel(), table.data_ptr(), 7, n, frames, 12, back.data_ptr(),
                                     st.data_ptr(), ws.data_ptr(), 64, None) == _lib.ERR_CAPACITY
        assert L.trpx_encode_sharded(None, _lib.U16, px.data_ptr(), n, frames, frames, 12, sc.out.data_ptr(), sc.out.numel(),
                                     sc.local_offsets.data_ptr(), sc.status.data_ptr(), sc.global_offsets.data_ptr(), None, None,
                                     sc.ws_e.data_ptr(), sc.ws_e.numel(), None, None) == _lib.ERR_INVALID_ARG
    finally:
        sc.close()


def test_lookback_timeout_falls_back_to_two_pass(gpu, oracle, tmp_path):
    """A look-back wait of the single-pass encoder that gives up (TRPX_ERR_TIMEOUT) must not cost the caller the stack:
    in a test build of the library whose waits give up at once (make force_timeout: -DTRPX_FORCE_TIMEOUT) the plain
    stream-ordered call reports status 7, and trpx_encode_checked / Encoded.check() / trpx_encode_host end with status 0
    and the oracle's bytes (re-run through the two-pass pipeline)."""
    variant = os.path.join(ROOT, "tools", "variants", "libtrpx_force_timeout.so")
    if not os.path.exists(variant):
        pytest.skip("test variant not built (make -C trpx_amd/csrc force_timeout)")
    script = tmp_path / "t.py"
    script.write_text(f"""
import sys, ctypes as C
sys.path.insert(0, {ROOT!r})
import numpy as np, torch
from trpx_amd import codec, _lib
from oracle import oracle as O
L = _lib.lib()
frames, n = 40, 512 * 512
px = codec.synth(np.uint16, 0, frames, n)
want, sizes, pb = O.encode_stack(px.cpu().numpy())
enc = codec.encode(px)
torch.cuda.synchronize()
assert int(enc.status[0]) == _lib.ERR_TIMEOUT, int(enc.status[0])       # the forced failure is really there
enc.check()                                                              # re-runs through trpx_encode_checked
assert int(enc.status[0]) == 0 and enc.prolix_bits() == pb
assert enc.stack().cpu().numpy().tobytes() == want.tobytes()
host = np.zeros(8, np.uint32)
ws = codec.Workspace("cuda")
enc2 = codec.encode(px, workspace=ws)
torch.cuda.synchronize()
w = ws.buf
rc = L.trpx_encode_checked(_lib.U16, px.data_ptr(), n, frames, 12, enc2.data.data_ptr(), enc2.data.numel(), enc2.frame_offsets.data_ptr(),
                           enc2.status.data_ptr(), None, w.data_ptr(), w.numel(), None, host.ctypes.data)
assert rc == 0 and host[0] == 0 and host[1] == pb and enc2.stack().cpu().numpy().tobytes() == want.tobytes()
out = np.zeros(want.size + 64, np.uint8); total = C.c_size_t(0); gpb = C.c_uint(0); offs = np.zeros(frames + 1, np.uint64)
hp = px.cpu().numpy()
_lib.check(L.trpx_encode_host(_lib.U16, hp.ctypes.data, n, frames, 12, out.ctypes.data, out.size, C.byref(total), offs.ctypes.data, C.byref(gpb), -1))
assert total.value == want.size and (out[: want.size] == want).all() and gpb.value == pb
print("OK")
""")
    env = dict(os.environ, TRPX_LIB=variant)
    r = subprocess.run([os.sys.executable, str(script)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_device_resident_stack_read_frame_by_frame(gpu, oracle):
    """Row f3 / src/prolix.cpp:69-92: trpx_stack_open uploads a stack once, trpx_stack_read expands a window of frames
    per device call and serves single frames from it -- in order, out of order (window misses), into other output types
    (clamping / float, as trpx_decode_host), and for a stack opened without frame offsets (serial frame location)."""
    from trpx_amd import _lib
    L = _lib.lib()
    frames, n = 40, 512 * 512                                  # window = 64 MB / (8 * n) = 32 frames: two windows
    px = oracle.synth(np.uint16, 3, frames, n)
    stream, sizes, pb = oracle.encode_stack(px)
    offs = np.concatenate([[0], np.cumsum(sizes.astype(np.uint64))]).astype(np.uint64)
    for with_offsets in (True, False):
        h = C.c_void_p()
        _lib.check(L.trpx_stack_open(C.byref(h), 0, stream.ctypes.data, stream.size, offs.ctypes.data if with_offsets else None,
                                     None, n, frames, 12, 16, -1))
        try:
            out = np.zeros(n, np.uint16)
            for f in list(range(frames)) + [5, 39, 0, 33, 31, 32]:
                out[:] = 0xAAAA
                _lib.check(L.trpx_stack_read(h, f, _lib.U16, out.ctypes.data))
                assert (out == px[f]).all(), (with_offsets, f)
            o8 = np.zeros(n, np.uint8)
            _lib.check(L.trpx_stack_read(h, 7, _lib.U8, o8.ctypes.data))                      # narrower: clamps (Bit_pointer.hpp:747-763)
            assert (o8 == np.minimum(px[7], 255)).all()
            o64 = np.zeros(n, np.float64)
            _lib.check(L.trpx_stack_read(h, 38, _lib.F64, o64.ctypes.data))                   # Terse.hpp:379-383
            assert (o64 == px[38].astype(np.float64)).all()
            assert L.trpx_stack_read(h, frames, _lib.U16, out.ctypes.data) == _lib.ERR_INVALID_ARG
        finally:
            L.trpx_stack_close(h)
    L.trpx_host_release()
    bad = stream.copy()
    bad[int(offs[3]) + 2: int(offs[3]) + 9] ^= 0xFF
    h = C.c_void_p()
    _lib.check(L.trpx_stack_open(C.byref(h), 0, bad.ctypes.data, bad.size, offs.ctypes.data, None, n, frames, 12, 16, -1))
    try:
        out = np.zeros(n, np.uint16)
        rc = L.trpx_stack_read(h, 0, _lib.U16, out.ctypes.data)
        assert rc == _lib.ERR_CORRUPT or not (out == px[0]).all() or True    # (payload flips may still parse)
        assert rc in (_lib.OK, _lib.ERR_CORRUPT)
    finally:
        L.trpx_stack_close(h)


@pytest.mark.parametrize("dtype,n,frames", [(np.uint16, 512 * 512, 5), (np.int32, 700 * 700, 3), (np.uint8, 12 * 256 * 3 + 8, 4),
                                            (np.int16, 12 * 256, 3), (np.uint16, 40, 2)])
def test_group_states_for_walk_free_decode_of_files(gpu, oracle, dtype, n, frames, tmp_path):
    """Row f1 (second half): the chain state at every 256th block.  States read off the encoder's index == states computed
    from oracle.widths(); the index rebuilt from the states == the encoder's index; host decode with the states is
    pixel-identical; states that do not fit the stream are detected (device route) or ignored (host route, which falls
    back to the walk); a stack object opened with states serves frames; both Terse classes carry them through a file."""
    import io
    import torch
    from trpx_amd import codec, _lib, Terse
    L = _lib.lib()
    dt = np.dtype(dtype)
    rng = np.random.RandomState(n % 1000 + frames)
    hi = rng.randint(0, 8 * dt.itemsize - 2, size=(frames, (n + 11) // 12))
    px = (rng.rand(frames, ((n + 11) // 12) * 12) * 2.0 ** np.repeat(hi, 12, axis=1)).astype(np.int64)[:, :n]
    if dt.kind == "i":
        px = px * rng.choice([-1, 1], size=px.shape)
    px = px.astype(dt)
    want_states = np.concatenate([oracle.group_states(f) for f in px])
    ng = L.trpx_group_count(n, 12)
    assert ng == ((n + 11) // 12 + 255) // 256 and want_states.size == frames * ng and L.trpx_group_count(n, 7) == 0
    dpx = torch.from_numpy(px.view(np.dtype(f"i{dt.itemsize}"))).to(gpu).view(codec.torch_dtype(dt))
    enc = codec.encode(dpx, index=True)
    torch.cuda.synchronize()
    enc.check()
    # index -> states
    d_states = torch.zeros(frames * ng, dtype=torch.int64, device=gpu)
    _lib.check(L.trpx_index_group_states(enc.index.data_ptr(), n, frames, 12, d_states.data_ptr(), None))
    torch.cuda.synchronize()
    assert (d_states.cpu().numpy().view(np.uint64) == want_states).all()
    # states -> index
    rebuilt = torch.full_like(enc.index, 0x5A)
    st = torch.zeros(8, dtype=torch.int32, device=gpu)
    stack = enc.stack()
    _lib.check(L.trpx_index_from_group_states(codec.dtype_code(dt), stack.data_ptr(), stack.numel(), enc.frame_offsets.data_ptr(),
                                              d_states.data_ptr(), n, frames, 12, rebuilt.data_ptr(), st.data_ptr(), None))
    torch.cuda.synchronize()
    assert int(st[0].item()) == 0
    nb = (n + 11) // 12
    w_off = (8 * frames * ng + 15) // 16 * 16
    assert torch.equal(enc.index[: 8 * frames * ng], rebuilt[: 8 * frames * ng])
    assert torch.equal(enc.index[w_off: w_off + frames * nb], rebuilt[w_off: w_off + frames * nb])
    back, st2 = codec.decode(stack, enc.frame_offsets, n, frames, dt, index=rebuilt)
    torch.cuda.synchronize()
    assert int(st2[0].item()) == 0 and (back.cpu().numpy().view(dt).reshape(frames, n) == px).all()
    # wrong states are detected on the device: offset off by one bit / previous width changed / first state not zero
    for which, delta in ((frames * ng - 1, 1), (ng - 1, 1 << 40), (0, 1)):
        bad = d_states.clone()
        bad[which] += delta
        _lib.check(L.trpx_index_from_group_states(codec.dtype_code(dt), stack.data_ptr(), stack.numel(), enc.frame_offsets.data_ptr(),
                                                  bad.data_ptr(), n, frames, 12, rebuilt.data_ptr(), st.data_ptr(), None))
        torch.cuda.synchronize()
        assert int(st[0].item()) == _lib.ERR_CORRUPT, (which, delta)
    # host route: states of a host stack, decode with them, decode with wrong ones (falls back to the walking decoder)
    h_stack = stack.cpu().numpy()
    offs = enc.frame_offsets.cpu().numpy().astype(np.uint64)
    got_states = np.zeros(frames * ng, np.uint64)
    _lib.check(L.trpx_group_states_host(h_stack.ctypes.data, h_stack.size, offs.ctypes.data, n, frames, 12, 8 * dt.itemsize,
                                        got_states.ctypes.data, -1))
    assert (got_states == want_states).all()
    for states in (want_states, want_states + np.uint64(3), None):
        out = np.zeros((frames, n), dt)
        _lib.check(L.trpx_decode_host_grouped(int(dt.kind == "i"), codec.dtype_code(dt), h_stack.ctypes.data, h_stack.size, offs.ctypes.data,
                                              states.ctypes.data if states is not None else None, n, frames, 12, out.ctypes.data, -1))
        assert (out == px).all()
    wide = np.zeros((frames, n), np.float64)                                   # converting output: general route
    _lib.check(L.trpx_decode_host_grouped(int(dt.kind == "i"), _lib.F64, h_stack.ctypes.data, h_stack.size, offs.ctypes.data,
                                          want_states.ctypes.data, n, frames, 12, wide.ctypes.data, -1))
    assert (wide == px.astype(np.float64)).all()
    for states in (want_states, want_states ^ np.uint64(1)):                   # the stack object, right and wrong states
        h = C.c_void_p()
        _lib.check(L.trpx_stack_open(C.byref(h), int(dt.kind == "i"), h_stack.ctypes.data, h_stack.size, offs.ctypes.data,
                                     states.ctypes.data, n, frames, 12, 0, -1))
        try:
            one = np.zeros(n, dt)
            for f in (frames - 1, 0, 1):
                _lib.check(L.trpx_stack_read(h, f, codec.dtype_code(dt), one.ctypes.data))
                assert (one == px[f]).all()
        finally:
            L.trpx_stack_close(h)
    # the classes: write(frame_index=True) computes and writes the states, read() brings them back, prolix uses them
    t = Terse()
    for f in px:
        t.push_back(f)
    assert not t.has_group_index()
    out = io.BytesIO()
    t.write(out, frame_index=True)
    assert t.has_group_index() and b' group_bit_offsets="0:0' in out.getvalue()[:out.getvalue().index(b"/>")]
    t2 = Terse.read(io.BytesIO(out.getvalue()))
    assert t2.has_group_index() and (t2._group_states == want_states).all()
    assert (t2.prolix_stack(dt) == px).all() and (t2.prolix(np.zeros(n, dt), 1) == px[1]).all()
    t2.push_back(px[0])
    assert not t2.has_group_index() and (t2.prolix(np.zeros(n, dt), frames) == px[0]).all()
    L.trpx_host_release()


@pytest.mark.parametrize("dtype", [np.uint64, np.int64])
@pytest.mark.parametrize("block", [12, 7])
def test_64bit_containers_with_wide_values(gpu, oracle, dtype, block):
    """Row f4: values that need more than 32 bits (what src/terse.cpp:120-123 makes of float / double images with a wide
    range).  64-bit containers go through generic kernels with fields of up to 64 bits: stream bytes == oracle, exact round
    trip into 64-bit containers and double, clamped into 32-bit containers (Bit_pointer.hpp:747-763), through the C ABI
    on device memory and through both host-side Terse routes.  Inside the reference's validity domain (D3): unsigned
    values < 2^63, |signed values| < 2^62."""
    import torch
    from trpx_amd import codec, _lib, Terse
    dt = np.dtype(dtype)
    rng = np.random.RandomState(block)
    frames, n = 3, 1000 + block                                             # ragged last block
    nblk = (n + block - 1) // block
    top = 62 if dt.kind == "u" else 61
    hi = rng.randint(0, top + 1, size=(frames, nblk))
    hi[:, ::5] = rng.randint(33, top + 1, size=hi[:, ::5].shape)            # plenty of blocks wider than 32 bits
    hi[1, : nblk // 2] = 0                                                  # and an empty half frame
    mag = (rng.rand(frames, nblk * block) * 2.0 ** np.repeat(hi, block, axis=1)).astype(np.uint64)[:, :n]
    px = mag.astype(dt)
    if dt.kind == "i":
        px = px * rng.choice([-1, 1], size=px.shape).astype(np.int64)
    want, sizes, pb = oracle.encode_stack(px, block)
    assert pb > 32
    # device route
    dpx = torch.from_numpy(px.view(np.int64)).to(gpu).view(codec.torch_dtype(dt))
    enc = codec.encode(dpx, block=block)
    torch.cuda.synchronize()
    enc.check()
    assert enc.stack().cpu().numpy().tobytes() == want.tobytes() and enc.prolix_bits() == pb
    assert (np.diff(enc.frame_offsets.cpu().numpy()) == sizes.astype(np.int64)).all()
    back, st = codec.decode(enc.stack(), enc.frame_offsets, n, frames, dt, block=block)
    torch.cuda.synchronize()
    assert int(st[0].item()) == 0 and (back.cpu().numpy().view(dt).reshape(frames, n) == px).all()
    nooffs, st = codec.decode(enc.stack(), None, n, frames, dt, block=block)            # frames located by the serial walk
    torch.cuda.synchronize()
    assert int(st[0].item()) == 0 and (nooffs.cpu().numpy().view(dt).reshape(frames, n) == px).all()
    # host routes: converting decode into double (exact below 2^53: compare with numpy's cast) and into 32 bits (clamped)
    stack = enc.stack().cpu().numpy()
    offs = enc.frame_offsets.cpu().numpy().astype(np.uint64)
    for out_dt, code in ((np.float64, _lib.F64), (np.int32 if dt.kind == "i" else np.uint32, None)):
        out = np.zeros((frames, n), out_dt)
        _lib.check(_lib.lib().trpx_decode_host(int(dt.kind == "i"), code if code is not None else codec.dtype_code(out_dt),
                                               stack.ctypes.data, stack.size, offs.ctypes.data, n, frames, block, out.ctypes.data, -1))
        if out_dt is np.float64:
            assert (out == px.astype(np.float64)).all()
        else:
            info = np.iinfo(out_dt)
            assert (out == np.clip(px.astype(object), info.min, info.max).astype(out_dt)).all()
    # the oracle decodes the GPU's stream to the same pixels
    assert (oracle.decode(stack[: int(sizes[0])], n, dt, block=block) == px[0]).all()
    # the class: frames pushed one by one, prolix into 64-bit containers and double
    t = Terse(block=block)
    for f in px:
        t.push_back(f)
    assert t.bits_per_val() == pb and bytes(t._data) == want.tobytes()
    assert (t.prolix(np.zeros(n, dt), 2) == px[2]).all() and (t.prolix_stack(np.float64) == px.astype(np.float64)).all()
    narrow = Terse(block=block)
    narrow.push_back((px[0] % 1000).astype(dt))                             # values that fit 32 bits: narrowed, same stream as int32 pixels
    ref32, _, pb32 = oracle.encode_stack((px[0:1] % 1000).astype(np.int32 if dt.kind == "i" else np.uint32), block)
    assert bytes(narrow._data) == ref32.tobytes() and narrow.bits_per_val() == pb32
    _lib.lib().trpx_host_release()


def test_large_frames_in_a_long_stack_take_the_tiled_path(gpu, oracle):
    """The per-frame decoder packs a block's bit position and width into 32 bits (units of < 2^26 bits worst case).  A
    stack whose frames' worst case is larger (1500 x 1500 int32: 74 Mbit) is cut into parts (decode_part.hip) or routed to
    the tiled kernels -- also when the per-frame path is forced -- and decodes exactly."""
    import torch
    from trpx_amd import codec
    frames, n = 130, 1500 * 1500
    px = codec.synth(np.int32, 5, frames, n, device=gpu)
    enc = codec.encode(px)
    torch.cuda.synchronize()
    enc.check()
    for f in (0, 64, 129):
        want, pb = oracle.encode(px[f].cpu().numpy())
        o = enc.frame_offsets.cpu().numpy()
        assert enc.stack()[int(o[f]): int(o[f + 1])].cpu().numpy().tobytes() == want.tobytes()
    back, st = codec.decode(enc.stack(), enc.frame_offsets, n, frames, np.int32)
    torch.cuda.synchronize()
    assert int(st[0].item()) == 0 and torch.equal(back, px)
    script = (
        "import sys, numpy as np, torch\n"
        "sys.path.insert(0, %r)\n"
        "from trpx_amd import codec\n"
        "px = codec.synth(np.int32, 5, 130, 1500 * 1500)\n"
        "enc = codec.encode(px); torch.cuda.synchronize(); enc.check()\n"
        "back, st = codec.decode(enc.stack(), enc.frame_offsets, 1500 * 1500, 130, np.int32); torch.cuda.synchronize()\n"
        "assert int(st[0].item()) == 0 and torch.equal(back, px)\n"
        "print('OK')\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([os.sys.executable, "-c", script], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, TRPX_DECODE_PATH="frames"))
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_config5_16000_frame_stream_in_eight_shards(gpu, oracle):
    """configs[4] (16 000 frames 512x512 u16, GPU g <- frames [2000 g, 2000 g + 2000)) on the one GPU of the test box:
    the eight shards are encoded one after the other exactly as eight ranks would (same frame numbers, same calls), their
    size messages go through the gather's scan kernel (trpx_gather_pack / trpx_gather_scan: ncclAllGather only moves
    them), and the properties that do not need the CPU at full size are checked: every shard's stack size is what the
    global offsets say, global offsets == running sum, sampled frames byte-identical to the oracle's single-frame encode,
    every shard decodes pixel-identically with its slice of the GLOBAL offset table rebased to its shard."""
    import torch
    from trpx_amd import codec, _lib
    L = _lib.lib()
    n, per, world = 512 * 512, 2000, 8
    ws = codec.Workspace(gpu)
    msgs = torch.zeros(world * (per + 2), dtype=torch.int64, device=gpu)
    totals, shards = [], []
    px = None
    for r in range(world):
        px = codec.synth(np.uint16, r * per, per, n, device=gpu)
        enc = codec.encode(px, workspace=ws)
        torch.cuda.synchronize()
        enc.check()
        assert L.trpx_gather_pack(enc.frame_offsets.data_ptr(), per, per, enc.status.data_ptr(), msgs[r * (per + 2):].data_ptr(), None) == 0
        torch.cuda.synchronize()
        totals.append(enc.total_bytes())
        for f in (0, per - 1):                                               # the shard's first and last frame against the oracle
            want = oracle.encode(px[f].cpu().numpy())[0]
            a, b = int(enc.frame_offsets[f]), int(enc.frame_offsets[f + 1])
            assert b - a == want.size and (enc.data[a:b].cpu().numpy() == want).all(), (r, f)
        back, st = codec.decode(enc.stack(), enc.frame_offsets, n, per, np.uint16, workspace=ws)
        torch.cuda.synchronize()
        assert int(st[0]) == 0 and torch.equal(back.view(torch.int16), px.view(torch.int16)), r
        shards.append((enc.frame_offsets.clone(), enc.total_bytes()))
        del enc, back
    assert totals[0] == 203596114                                            # SURVEY.md 8 row d (frames 0..1999)
    go = torch.zeros(world * per + 1, dtype=torch.int64, device=gpu)
    pb = torch.zeros(2, dtype=torch.int32, device=gpu)
    rb = torch.zeros(world, dtype=torch.int64, device=gpu)
    assert L.trpx_gather_scan(msgs.data_ptr(), world, per, go.data_ptr(), pb.data_ptr(), rb.data_ptr(), None) == 0
    torch.cuda.synchronize()
    assert int(pb[0]) == 12 and int(go[0]) == 0 and int(go[-1]) == sum(totals)
    bases = np.concatenate([[0], np.cumsum(totals)])
    assert rb.cpu().numpy().tolist() == bases[:-1].tolist()
    for r, (loc, tot) in enumerate(shards):
        assert torch.equal(go[r * per: (r + 1) * per + 1] - int(bases[r]), loc), r


# ---- the decode route matrix: every route x every pixel type on the fuzz generator ---------------------
_ROUTES = {"basic": 1, "tiles": 2, "frames": 3, "parts": 4, "dense": 5}   # (4: large frames by round 4's parts route instead of the index route; = auto for small frames; 5: auto with the listed frames' index by the dense walk, decode_dense.hip)


def _fuzz_stack(rng, dt, kind, n, frames):
    """tools/fuzz_paths.py's generator: width patterns that exercise runs, outliers, flips and long runs (inside D3)."""
    top = 8 * dt.itemsize - (2 if dt.kind == "i" else (1 if dt.itemsize == 4 else 0))
    nblk = (n + 11) // 12
    if kind == 0:
        hi = np.full((frames, nblk), rng.randint(0, top + 1))                                          # one width
    elif kind == 1:
        hi = rng.randint(0, top + 1, size=(frames, nblk))                                              # every block its own width
    elif kind == 2:
        hi = np.where(rng.rand(frames, nblk) < 0.02, rng.randint(0, top + 1, size=(frames, nblk)), 3 if top >= 3 else 1)   # runs + outliers
    elif kind == 3:
        hi = np.where(rng.rand(frames, nblk) < 0.5, 2, 3 if top >= 3 else 1)                          # flips every other block
    elif kind == 5:                                                                                    # payloads of one bits: a header position
        hi = np.repeat(rng.randint(0, min(top, 12) + 1, size=(frames, (nblk + 39) // 40)), 40, axis=1)[:, :nblk]   # may read 0xFFFFFFFF
        sat = np.repeat(hi, 12, axis=1)[:, :n]
        mag = np.where(rng.rand(frames, n) < 0.97, (1 << sat) - 1, 0).astype(np.int64)
        mag[:, n // 2: n // 2 + n // 5] = 0                                                             # and a run of empty blocks
        if dt.kind == "i":
            return np.where(mag > 0, -1, 0).astype(dt) * (rng.rand(frames, n) < 0.9) + (rng.randint(-40, 40, size=(frames, n)) * (rng.rand(frames, n) < 0.01)).astype(dt)
        return mag.astype(dt)
    else:
        hi = np.repeat(rng.randint(0, top + 1, size=(frames, (nblk + 299) // 300)), 300, axis=1)[:, :nblk]   # long runs of changing widths
    mag = (rng.rand(frames, nblk * 12) * (2.0 ** np.repeat(hi, 12, axis=1))).astype(np.int64)[:, :n]
    if kind % 2 == 0:
        mag[:, : n // 3] = 0                                                                          # empty stretches
    if dt.kind == "i":
        mag = mag * rng.choice([-1, 1], size=mag.shape)
    return mag.astype(dt)


@pytest.mark.parametrize("dtype", ALL_DTYPES)
def test_build_index_of_many_small_frames(gpu, oracle, dtype):
    """trpx_build_index on a stack of frames of < 2^26 bits runs the per-frame walker with index writers in place of the extraction
    (k_index_frames; header-dense frames go to the position-parallel walk): the index must equal the one the encoder writes as a
    by-product, block for block and group for group, and decode the stack (Terse.hpp:360-372 has one chain per frame)."""
    import torch
    from trpx_amd import codec
    dt = np.dtype(dtype)
    tdt = {1: torch.uint8 if dt.kind == "u" else torch.int8, 2: torch.uint16 if dt.kind == "u" else torch.int16,
           4: torch.uint32 if dt.kind == "u" else torch.int32}[dt.itemsize]
    rng = np.random.RandomState(4242 + ALL_DTYPES.index(dtype))
    for kind, n, frames in ((2, 40000, 130), (3, 12 * 1024 + 8, 140), (1, 3000, 200), (5, 50000, 131), (0, 52, 129),
                            (3, 12 * 34000, 4), (2, 12 * 34000 + 4, 3),           # (> 32 K blocks: dense frames go through several wavefronts each)
                            (2, 513 * 7, 131), (3, 1030 * 53 + 1, 3)):            # (no multiples of 4)
        px = _fuzz_stack(rng, dt, kind, n, frames)
        dpx = torch.from_numpy(px.view(np.dtype(f"i{dt.itemsize}"))).to(gpu).view(tdt)
        enc = codec.encode(dpx, index=True)
        torch.cuda.synchronize()
        enc.check()
        walked = codec.build_index(enc.stack(), enc.frame_offsets, n, frames, dt)
        torch.cuda.synchronize()
        nb = (n + 11) // 12
        ng = (nb + 255) // 256
        w_off = (8 * frames * ng + 15) // 16 * 16
        assert torch.equal(enc.index[: 8 * frames * ng], walked[: 8 * frames * ng]), (dtype, kind, n, frames, "group offsets")
        assert torch.equal(enc.index[w_off: w_off + frames * nb], walked[w_off: w_off + frames * nb]), (dtype, kind, n, frames, "widths")
        back, st = codec.decode(enc.stack(), enc.frame_offsets, n, frames, dt, index=walked)
        torch.cuda.synchronize()
        assert int(st[0]) == 0 and (back.cpu().numpy().reshape(frames, n).view(dt) == px).all(), (dtype, kind, n, frames)


@pytest.mark.parametrize("route", ["auto", "tiles", "frames", "parts", "dense", "basic"])
def test_fuzz_slice_every_route(gpu, oracle, route):
    """A bounded, seeded slice of tools/fuzz_paths.py inside the tier the driver runs: random pixel types, frame sizes (small,
    512 x 512, detector sizes beyond 32 K blocks), frame counts and width patterns (runs, flips, pedestals, Poisson counts, type
    extremes), encoded by the GPU (== the oracle's bytes), decoded along the forced route (== the pixels), the decode index
    three ways.  6 - 10 s per route; the long runs stay with the tool (profiles/rNN_fuzz.txt)."""
    import importlib.util
    from trpx_amd import _lib
    spec = importlib.util.spec_from_file_location("fuzz_paths", os.path.join(ROOT, "tools", "fuzz_paths.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    code = {"auto": 0, **_ROUTES}[route]
    assert _lib.lib().trpx_set_decode_path(code) == 0
    try:
        n_run, n_large, n_fb = fz.run(60, seed=600 + code, basic=route == "basic", budget_s=6.0, quiet=True,
                                      max_pixels=1 << (21 if route in ("basic", "frames") else 23))
    finally:
        _lib.lib().trpx_set_decode_path(0)
    assert n_run >= 8, n_run


@pytest.mark.parametrize("dtype", ALL_DTYPES)
@pytest.mark.parametrize("route", sorted(_ROUTES))
def test_decode_route_matrix(gpu, oracle, route, dtype):
    """Each decode route forced through trpx_set_decode_path (the basic kernels, the position-parallel walk + tiled
    extraction, the per-frame decoder for any number of frames) must give the oracle's pixels for every pixel type:
    Terse.hpp:352-389 has one answer whatever the route."""
    import torch
    from trpx_amd import codec, _lib
    L = _lib.lib()
    dt = np.dtype(dtype)
    tdt = {1: torch.uint8 if dt.kind == "u" else torch.int8, 2: torch.uint16 if dt.kind == "u" else torch.int16,
           4: torch.uint32 if dt.kind == "u" else torch.int32}[dt.itemsize]
    rng = np.random.RandomState(1000 * _ROUTES[route] + ALL_DTYPES.index(dtype))
    cases = [(0, 4096, 3), (1, 12 * 768 + 4, 17), (2, 40000, 130), (3, 3000, 140), (4, 131072, 3), (2, 388, 129), (1, 52, 2),
             (5, 50000, 131), (5, 262144, 4), (2, 12 * 40000 + 8, 3),      # (> 32 K blocks: several wavefronts per frame on the tiled route,
             (3, 12 * 34000, 5),                                            #  and the per-frame decoder's hand-over of dense frames through them)
             (2, 12 * 300 + 7, 130), (1, 2463 * 3, 4), (3, 1030 * 53 + 1, 3), (2, 12 * 34000 + 3, 3)]   # pixel counts that are no multiples of 4:
    #                                                                            frames start at any element (1030 x 1065, 2463 x 2527 detectors)
    assert L.trpx_set_decode_path(9) != 0                                    # out of range: refused
    try:
        assert L.trpx_set_decode_path(_ROUTES[route]) == 0
        for kind, n, frames in cases:
            px = _fuzz_stack(rng, dt, kind, n, frames)
            want, sizes, pb = oracle.encode_stack(px)
            dpx = torch.from_numpy(px.view(np.dtype(f"i{dt.itemsize}"))).to(gpu).view(tdt)
            enc = codec.encode(dpx, index=True)
            torch.cuda.synchronize()
            enc.check()
            assert enc.stack().cpu().numpy().tobytes() == want.tobytes() and enc.prolix_bits() == pb, ("encode", route, dtype, kind, n, frames)
            back, st = codec.decode(enc.stack(), enc.frame_offsets, n, frames, dt)
            torch.cuda.synchronize()
            assert int(st[0]) == 0, (route, dtype, kind, n, frames, int(st[0]))
            assert (back.cpu().numpy().reshape(frames, n).view(dt) == px).all(), (route, dtype, kind, n, frames)
            if enc.index is not None and route != "basic":       # walk-free: tiled kernel / per-frame decoder with the widths given
                back, st = codec.decode(enc.stack(), enc.frame_offsets, n, frames, dt, index=enc.index)
                torch.cuda.synchronize()
                assert int(st[0]) == 0, ("index", route, dtype, kind, n, frames, int(st[0]))
                assert (back.cpu().numpy().reshape(frames, n).view(dt) == px).all(), ("index", route, dtype, kind, n, frames)
    finally:
        L.trpx_set_decode_path(0)


@pytest.mark.parametrize("dtype,frames", [(np.uint16, 3000), (np.int32, 1537), (np.uint8, 4100)])
def test_stacks_larger_than_one_round_of_workgroups(gpu, oracle, dtype, frames):
    """The per-frame decoder runs one workgroup per frame, 2048 resident at a time for 8/16-bit pixels, 1536 for 32-bit:
    stacks past one round (prolix.cpp:69-92 loops over any number of frames).  Full size, so checked through properties:
    sizes == prefix differences, round trip pixel-identical, and a sample of frames -- the first, the last and the ones
    around the round's edge -- byte-identical to the oracle's single-frame encodes."""
    import torch
    from trpx_amd import codec
    n = 512 * 512
    dt = np.dtype(dtype)
    if dtype == np.uint8:                                                    # synth-v1 exists for u16 / i32: fold the u16 frames into bytes
        px = (codec.synth(np.uint16, 0, frames, n, device=gpu).view(torch.int16) & 0xFF).to(torch.uint8)
    else:
        px = codec.synth(dtype, 0, frames, n, device=gpu)
    enc = codec.encode(px)
    torch.cuda.synchronize()
    enc.check()
    offs = enc.frame_offsets.cpu().numpy()
    assert offs[0] == 0 and (np.diff(offs.astype(np.int64)) > 0).all() and int(offs[-1]) == enc.total_bytes()
    back, st = codec.decode(enc.stack(), enc.frame_offsets, n, frames, dt)
    torch.cuda.synchronize()
    assert int(st[0]) == 0
    same = back.view(torch.uint8) == px.view(torch.uint8).reshape(back.view(torch.uint8).shape)
    assert bool(same.all()), "round trip is not pixel-identical"
    edge = 2048 if dt.itemsize < 4 else 1536
    stream = enc.stack().cpu().numpy()
    for f in sorted(x for x in {0, 1, edge - 1, edge, edge + 1, frames - 1} if x < frames):
        want = oracle.encode(px[f].cpu().numpy().view(dt))[0]
        got = stream[int(offs[f]): int(offs[f + 1])]
        assert got.size == want.size and (got == want).all(), (dtype, f)


def test_host_entry_points_from_two_threads(gpu, oracle):
    """src/terse.cpp:63-69 / prolix.cpp:69-92 callers in two threads: the *_host entry points order their copies and kernels
    on a private stream per calling thread and wait for that stream alone, so two threads neither serialise each other
    through the null stream nor see each other's buffers; both get the oracle's bytes and their own pixels back."""
    import threading
    rng = np.random.RandomState(3)
    stacks = [rng.poisson(2.0 + 3 * k, size=(60, 50000)).astype(np.uint16) for k in range(2)]
    want = [oracle.encode_stack(p) for p in stacks]
    errors = []

    def work(k):
        try:
            for _ in range(6):
                s, offs, pb = _host_encode(stacks[k])
                assert s.tobytes() == want[k][0].tobytes() and pb == want[k][2]
                assert (_host_decode(s, offs, stacks[k].shape[1], stacks[k].shape[0], np.uint16) == stacks[k]).all()
                assert (_host_decode(s, None, stacks[k].shape[1], stacks[k].shape[0], np.uint16) == stacks[k]).all()
        except Exception as ex:                                              # noqa: BLE001 -- reported by the main thread
            errors.append((k, repr(ex)))

    threads = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors and not any(t.is_alive() for t in threads), errors


def test_frame_index_of_stacks_with_values_wider_than_32_bits(gpu, oracle, tmp_path):
    """write(frame_index=True) on a stack of 64-bit pixels (what src/terse.cpp:120-123 makes of float / double
    images) -- prolix_bits > 32, generic kernels, no decode index -- must write the frame sizes alone instead of failing, and
    the file must read back and expand; copies / pickles of a Terse object do not share its device-side stack."""
    import copy
    import pickle
    from trpx_amd import Terse
    rng = np.random.RandomState(8)
    px = (rng.randint(0, 1 << 20, size=(3, 600)).astype(np.int64) << 22) * rng.choice([-1, 1], size=(3, 600))
    t = Terse()
    t.push_back_stack(px)
    assert t.bits_per_val() > 32
    hdr = t.header(frame_index=True)
    assert b"frame_sizes=" in hdr and b"group_bit_offsets" not in hdr
    path = tmp_path / "wide.trpx"
    with open(path, "wb") as f:
        t.write(f, frame_index=True)
    with open(path, "rb") as f:
        r = Terse.read(f)
    assert r.number_of_frames() == 3 and (r.prolix_stack(np.int64) == px).all()
    out = np.zeros(600, np.int64)
    r.prolix(out, 1)                                                          # opens the device-side stack
    assert (out == px[1]).all()
    for clone in (copy.copy(r), copy.deepcopy(r), pickle.loads(pickle.dumps(r))):
        assert clone._stack is None
        clone.prolix(out, 2)
        assert (out == px[2]).all()
    del clone, r                                                              # two closes of one handle would crash here


def _banded(px, rows, width, period, band, rng):
    """Header-dense bands (Poisson(1.5) counts: the block width flips with every other block) in a run-dominated frame."""
    a = px.reshape(rows, width).copy()
    for r0 in range(period // 2, rows, period):
        a[r0:r0 + band] = np.minimum(rng.poisson(1.5, a[r0:r0 + band].shape), 6).astype(px.dtype)
    return a.reshape(-1)


@pytest.mark.parametrize("dtype,rows,width,frames", [(np.uint16, 1030, 1065, 6), (np.int32, 1211, 997, 3), (np.uint8, 2048, 2048, 2),
                                                       (np.int16, 613, 1999, 33)])
def test_large_frames_are_cut_into_parts(gpu, oracle, dtype, rows, width, frames):
    """Frames of more than 32 K blocks take the per-frame decoder part by part (decode_part.hip: start states inside runs of
    equal widths, a counting walk per part, repairs of the links that did not close, the part table).  Run-dominated frames,
    frames with header-dense bands (cuts without a run to start in, parts that are too dense: the frame falls back to the
    position-parallel walk), a frame of noise; pixel counts that are no multiple of 12, frames that start anywhere in the
    stack.  Every frame must decode exactly; the stream is the oracle's (Terse.hpp:352-389 against :500-549)."""
    import torch
    from trpx_amd import codec
    rng = np.random.RandomState(rows)
    n = rows * width
    dt = np.dtype(dtype)
    base = oracle.synth(np.uint16, 3, frames, n).astype(np.int64)
    if dt.kind == "i":
        base = base - 3
    base = np.clip(base, np.iinfo(dt).min, np.iinfo(dt).max).astype(dt)
    for f in range(frames):
        if f % 3 == 1:
            base[f] = _banded(base[f], rows, width, 64, 6, rng)           # thin bands: the guess search reaches past them
        elif f % 3 == 2:
            base[f] = _banded(base[f], rows, width, 200, 60, rng)         # wide bands: plain guesses, dense parts
    if frames > 4:
        base[4] = np.minimum(rng.poisson(1.5, n), 6).astype(dt)           # all noise
    px = torch.from_numpy(base).to(gpu)
    enc = codec.encode(px)
    torch.cuda.synchronize()
    enc.check()
    o = enc.frame_offsets.cpu().numpy()
    for f in (0, 1, frames - 1):
        want, _ = oracle.encode(base[f])
        assert enc.stack()[int(o[f]): int(o[f + 1])].cpu().numpy().tobytes() == want.tobytes()
    back, st = codec.decode(enc.stack(), enc.frame_offsets, n, frames, dtype)
    torch.cuda.synchronize()
    assert int(st[0].item()) == 0
    assert torch.equal(back.view(torch.uint8), px.view(torch.uint8).reshape(frames, -1))
    # round 4's parts route (two walks; its header-dense frames are listed for the position-parallel walk, whose launches for a
    # list are one persistent kernel since round 5: k_seg_fallback) must give the same pixels
    from trpx_amd import _lib
    try:
        assert _lib.lib().trpx_set_decode_path(4) == 0
        back4, st4 = codec.decode(enc.stack(), enc.frame_offsets, n, frames, dtype)
        torch.cuda.synchronize()
    finally:
        _lib.lib().trpx_set_decode_path(0)
    assert int(st4[0].item()) == 0 and torch.equal(back4.view(torch.uint8), px.view(torch.uint8).reshape(frames, -1))
    # a flipped bit in the middle of a large frame must not go unnoticed (every part checks the state it ends in)
    bad = enc.stack().clone()
    mid = int(o[1] + (o[2] - o[1]) // 2) if frames > 2 else int(o[0] + (o[1] - o[0]) // 2)
    bad[mid] ^= 0x10
    back2, st2 = codec.decode(bad, enc.frame_offsets, n, frames, dtype)
    torch.cuda.synchronize()
    got = back2.view(torch.uint8).reshape(frames, -1)
    assert int(st2[0].item()) == 5 or not torch.equal(got, px.view(torch.uint8).reshape(frames, -1))


def test_large_frames_every_part_link_repaired(gpu, oracle, tmp_path):
    """The same route in a test build whose cuts never find a run to start in (make weakparts: -DTRPX_PART_FORCE_WEAK):
    every part starts on a chain that is not the frame's, every link is open, and k_part_repair has to count every part
    again from the true state up to the checkpoint where the chains have merged.  The frames change their width every 16
    blocks (a chain that is not the frame's merges at an explicit header it meets on a block start; in a frame of long runs
    it may not meet one before the part ends, and the frame then takes the other route -- which is why real cuts start
    inside runs).  The build counts its verdicts in the status block: no frame may fall back, every link must have been
    repaired, the pixels must be exact."""
    variant = os.path.join(ROOT, "tools", "variants", "libtrpx_weakparts.so")
    if not os.path.exists(variant):
        pytest.skip("test variant not built (make -C trpx_amd/csrc weakparts)")
    script = tmp_path / "t.py"
    script.write_text(f"""
import sys
sys.path.insert(0, {ROOT!r})
import numpy as np, torch
from trpx_amd import codec, _lib
rng = np.random.RandomState(5)
for dtype, n, frames in ((np.uint16, 1030 * 1065, 12), (np.int32, 2048 * 2048 + 5, 2)):
    a = rng.randint(0, 8, (frames, n)).astype(dtype)
    a[:, 0::192] = 9                                      # every 16th block is 4 bits wide
    if np.dtype(dtype).kind == "i":
        a[:, 1::384] = -9
    px = torch.from_numpy(a).cuda()
    enc = codec.encode(px); torch.cuda.synchronize(); enc.check()
    # the index route (default): no cut finds a run, no part warms up in this build -- every link between two parts is open and
    # repaired, and every repaired part's entries are spliced from the repair's and its own walk's (none is walked again)
    P = _lib.lib().trpx_decode_parts_per_frame(codec.dtype_code(dtype), n, frames, 12)
    back, st = codec.decode(enc.stack(), enc.frame_offsets, n, frames, dtype)
    torch.cuda.synchronize()
    s = st.cpu().numpy()
    assert s[0] == 0 and torch.equal(back, px), s
    # (a part of ~1000 blocks on a chain that has not merged with the frame's by its end leaves the NEXT repair a false start:
    # such a frame -- a few of them here, none with real cuts, which warm up -- is listed for the other route and still exact)
    assert P > 3 and s[4] == 0 and s[5] + s[6] == frames * (P - 2) and s[5] > 9 * s[6] and s[2] <= frames, (P, s)
    # round 4's parts route: plain guesses == repaired links; no fallback, no failed repair
    assert _lib.lib().trpx_set_decode_path(4) == 0
    back, st = codec.decode(enc.stack(), enc.frame_offsets, n, frames, dtype)
    torch.cuda.synchronize()
    _lib.lib().trpx_set_decode_path(0)
    s = st.cpu().numpy()
    assert s[0] == 0 and torch.equal(back, px), s
    assert s[2] == 0 and s[3] > 0 and s[5] == s[3] and s[6] == 0, s
print("OK")
""")
    r = subprocess.run([os.sys.executable, str(script)], capture_output=True, text=True, timeout=600, env=dict(os.environ, TRPX_LIB=variant))
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_dense_walk_last_resort_and_stress(gpu, oracle, tmp_path):
    """The dense walk of listed frames (decode_dense.hip, route 5) in two test builds.  `denseserial` (-DTRPX_DENSE_FORCE_SERIAL):
    every listed frame skips its write pass and is walked by one lane from (0, 0) -- the last resort a frame takes when its
    speculative walks do not close.  `densetiny` (-DTRPX_DENSE_SEG_BLOCKS=1): regions of a few blocks, so that link walks cross
    dozens of regions, deep records collide and the write pass's checks (end state AND block count of every region) have to
    catch what the records got wrong -- frames that do not close end in the serial walk, none may be wrong (with 128-byte windows
    most frames did; a region is at least one 224-byte window now, and few do).  status[2] counts those frames."""
    script = tmp_path / "t.py"
    script.write_text(f"""
import sys
sys.path.insert(0, {ROOT!r})
import numpy as np, torch
from trpx_amd import codec, _lib, workloads
_lib.lib().trpx_set_decode_path(5)
want_serial = sys.argv[1] == "all"
rng = np.random.RandomState(11)
total = 0
hits = []
for dtype, n, frames in ((np.uint16, 512 * 512, 9), (np.int8, 3000, 140), (np.uint16, 3000, 140), (np.int32, 40000, 6), (np.uint8, 12 * 300 + 7, 130)):
    dt = np.dtype(dtype)
    if dtype == np.uint16 and n == 512 * 512:
        px = workloads.poisson_u16(3.0, 0, frames, n, device="cuda")
    else:
        nblk = (n + 11) // 12
        top = 8 * dt.itemsize - (1 if dt.kind == "i" else 0)
        hi = rng.choice([0, 1, 2, 3, 5, min(9, top), top], size=(frames, nblk), p=[0.1, 0.2, 0.3, 0.2, 0.1, 0.07, 0.03])
        mag = (rng.rand(frames, nblk * 12) * (2.0 ** np.repeat(hi, 12, axis=1))).astype(np.int64)[:, :n]
        if dt.kind == "i":
            mag = np.clip(mag * rng.choice([-1, 1], size=mag.shape), np.iinfo(dt).min, np.iinfo(dt).max)
        px = torch.from_numpy(mag.astype(dt)).cuda()
    enc = codec.encode(px, index=True); torch.cuda.synchronize(); enc.check()
    back, st = codec.decode(enc.stack(), enc.frame_offsets, n, frames, dtype)
    torch.cuda.synchronize()
    s = st.cpu().numpy()
    assert s[0] == 0 and torch.equal(back.view(torch.uint8), px.view(torch.uint8)), (dtype, n, frames, s)
    walked = codec.build_index(enc.stack(), enc.frame_offsets, n, frames, dtype)
    nb = (n + 11) // 12; ng = (nb + 255) // 256; w_off = (8 * frames * ng + 15) // 16 * 16
    assert torch.equal(enc.index[: 8 * frames * ng], walked[: 8 * frames * ng]) and torch.equal(enc.index[w_off: w_off + frames * nb], walked[w_off: w_off + frames * nb]), (dtype, n, frames)
    total += int(s[2])
    hits.append(int(s[2]))
assert not want_serial or (total > 0 and sum(1 for h in hits if h > 0) >= 3), hits     # (every LISTED frame of the serial build; which frames are listed is the per-frame decoder's call)
print("OK", total)
""")
    for name, arg in (("denseserial", "all"), ("densetiny", "some")):
        variant = os.path.join(ROOT, "tools", "variants", f"libtrpx_{name}.so")
        if not os.path.exists(variant):
            pytest.skip(f"test variant not built (make -C trpx_amd/csrc {name})")
        r = subprocess.run([os.sys.executable, str(script), arg], capture_output=True, text=True, timeout=600, env=dict(os.environ, TRPX_LIB=variant))
        assert r.returncode == 0 and "OK" in r.stdout, (name, r.stdout[-2000:], r.stderr[-2000:])


def test_listed_large_frames_fallback_on_a_small_device(gpu, oracle, tmp_path):
    """k_seg_fallback (decode_seg.hip) synchronises its persistent grid with device-wide barriers, which needs every workgroup
    resident at once: the grid is sized by what the device holds (hipOccupancyMaxActiveBlocksPerMultiprocessor x CUs, halved,
    at most 1024).  Test build `segfbgrid`: a device that holds 24 (and k_seg_wg hands every header-dense frame back after one
    round, so that the listed frames ARE k_seg_fallback's).  Header-dense frames of 1030 x 1065 and 2048 x 520 pixels are listed
    by the large-frame routes and must come out exact through the 24-workgroup grid (status 0, no timeout)."""
    variant = os.path.join(ROOT, "tools", "variants", "libtrpx_segfbgrid.so")
    if not os.path.exists(variant):
        pytest.skip("test variant not built (make -C trpx_amd/csrc segfbgrid)")
    script = tmp_path / "t.py"
    script.write_text(f"""
import sys
sys.path.insert(0, {ROOT!r})
import numpy as np, torch
from trpx_amd import codec, _lib, workloads
for route in (0, 2, 4):
    _lib.lib().trpx_set_decode_path(route)
    for n, frames in ((1030 * 1065, 5), (2048 * 520, 3)):
        px = workloads.poisson_u16(3.0, 0, frames, n, device="cuda")
        enc = codec.encode(px); torch.cuda.synchronize(); enc.check()
        back, st = codec.decode(enc.stack(), enc.frame_offsets, n, frames, np.uint16)
        torch.cuda.synchronize()
        assert int(st[0]) == 0 and torch.equal(back.view(torch.int16), px.view(torch.int16)), (route, n, frames, st.tolist())
print("OK")
""")
    r = subprocess.run([os.sys.executable, str(script)], capture_output=True, text=True, timeout=600, env=dict(os.environ, TRPX_LIB=variant))
    assert r.returncode == 0 and "OK" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


def test_header_dense_large_frames_one_workgroup_each(gpu, oracle, tmp_path):
    """Header-dense stacks of LARGE frames (decode_part.hip: ChainVote -- more than one block in six of the voting frames' heads
    starts with an explicit header) are left alone by the serial part walkers, listed, and walked by k_seg_wg (decode_seg.hip):
    one workgroup of 2 / 4 / 8 wavefronts per frame, lane per segment, links in LDS (Terse.hpp:352-389 all the same).  Product
    build: Poisson(3) frames of the three workgroup shapes, signed and 32-bit pixels, every frame listed (status[2]), pixels
    exact, the index built from the stream and used; a stack of run-dominated frames is NOT listed; frames 0 and last against
    the oracle.  Test builds: `alldense` -- run-dominated synth-v1 frames are called header-dense, their links do not close in
    twenty rounds, k_seg_wg hands them back and k_seg_fallback walks them; `segwgrounds` -- every header-dense frame is handed
    back after one round; `noclassify` -- the serial walkers and their repairs on header-dense data (what the product does for
    stacks whose frame heads are blank)."""
    import torch
    from trpx_amd import codec, _lib, workloads
    L = _lib.lib()
    cases = [(700 * 700, 5, np.uint16), (1030 * 1065, 6, np.uint16), (1300 * 1000, 3, np.uint16), (1030 * 1065, 3, np.int16), (1200 * 900, 3, np.int32)]
    for n, frames, dt in cases:
        px = workloads.poisson_u16(3.0, 0, frames, n, device=gpu)
        if dt == np.int16: px = (px.view(torch.int16) - 3)
        if dt == np.int32: px = px.view(torch.int16).to(torch.int32) * 37 - 90
        assert L.trpx_decode_parts_per_frame(codec.dtype_code(dt), n, frames, 12) > 1
        enc = codec.encode(px); torch.cuda.synchronize(); enc.check()
        offs = enc.frame_offsets.cpu().numpy()
        for f in (0, frames - 1):
            want = oracle.encode_stack(px[f:f + 1].cpu().numpy())[0]
            assert bytes(enc.data[int(offs[f]): int(offs[f + 1])].cpu().numpy()) == bytes(want), (n, f)
        back, st = codec.decode(enc.stack(), enc.frame_offsets, n, frames, dt)
        torch.cuda.synchronize()
        s = st.cpu().numpy()
        assert s[0] == 0 and torch.equal(back, px), (n, dt, s)
        assert s[2] == frames, (n, dt, s)                     # every frame listed for k_seg_wg
        idx = codec.build_index(enc.stack(), enc.frame_offsets, n, frames, dt)
        back2, st2 = codec.decode(enc.stack(), enc.frame_offsets, n, frames, dt, index=idx)
        torch.cuda.synchronize()
        assert int(st2[0]) == 0 and torch.equal(back2, px), (n, dt)
        del px, enc, back, back2, idx
    px = codec.synth(np.uint16, 3, 6, 1030 * 1065, device=gpu)
    enc = codec.encode(px); torch.cuda.synchronize(); enc.check()
    back, st = codec.decode(enc.stack(), enc.frame_offsets, 1030 * 1065, 6, np.uint16)
    torch.cuda.synchronize()
    assert int(st[0]) == 0 and int(st[2]) == 0 and torch.equal(back, px), st.tolist()
    del enc, back
    # mixed stacks: the verdict is the stack's -- a run-dominated frame in a header-dense stack goes to k_seg_wg with the others (and
    # back to k_seg_fallback if its links do not close), a header-dense frame in a run-dominated stack stays with the serial walkers
    dense = workloads.poisson_u16(3.0, 0, 6, 1030 * 1065, device=gpu)
    for majority, odd in ((dense, px), (px, dense)):
        mix = majority.clone()
        mix[3] = odd[3]
        enc = codec.encode(mix); torch.cuda.synchronize(); enc.check()
        back, st = codec.decode(enc.stack(), enc.frame_offsets, 1030 * 1065, 6, np.uint16)
        torch.cuda.synchronize()
        s = st.cpu().numpy()
        assert s[0] == 0 and torch.equal(back, mix), s
        assert (s[2] >= 5) if majority is dense else (s[2] <= 1), s    # (the odd frame's walkers may be through before the verdict is in)
        del mix, enc, back
    del px, dense
    script = tmp_path / "t.py"
    script.write_text(f"""
import sys
sys.path.insert(0, {ROOT!r})
import numpy as np, torch
from trpx_amd import codec, _lib, workloads
kind = sys.argv[1]
for n, frames in ((1030 * 1065, 5), (1300 * 1000, 3)):
    for data in ("poisson3", "synth"):
        px = workloads.poisson_u16(3.0, 0, frames, n, device="cuda") if data == "poisson3" else codec.synth(np.uint16, 3, frames, n, device="cuda")
        enc = codec.encode(px); torch.cuda.synchronize(); enc.check()
        back, st = codec.decode(enc.stack(), enc.frame_offsets, n, frames, np.uint16)
        torch.cuda.synchronize()
        s = st.cpu().numpy()
        assert s[0] == 0 and torch.equal(back, px), (kind, data, n, s)
        if kind == "alldense": assert s[2] == frames, (kind, data, s)
        if kind == "segwgrounds": assert s[2] == (frames if data == "poisson3" else 0), (kind, data, s)
        if kind == "noclassify" and data == "synth": assert s[2] == 0, (kind, data, s)
        idx = codec.build_index(enc.stack(), enc.frame_offsets, n, frames, np.uint16)
        back2, st2 = codec.decode(enc.stack(), enc.frame_offsets, n, frames, np.uint16, index=idx)
        torch.cuda.synchronize()
        assert int(st2[0]) == 0 and torch.equal(back2, px), (kind, data, n)
print("OK")
""")
    for name in ("alldense", "segwgrounds", "noclassify"):
        variant = os.path.join(ROOT, "tools", "variants", f"libtrpx_{name}.so")
        if not os.path.exists(variant):
            pytest.skip(f"test variant not built (make -C trpx_amd/csrc {name})")
        r = subprocess.run([os.sys.executable, str(script), name], capture_output=True, text=True, timeout=600, env=dict(os.environ, TRPX_LIB=variant))
        assert r.returncode == 0 and "OK" in r.stdout, (name, r.stdout[-2000:], r.stderr[-2000:])


def test_many_large_frames_stay_whole(gpu, oracle):
    """Stacks of 768 frames and more keep frames of more than 32 K blocks on the per-frame route (one workgroup per frame,
    header-dense frames handed to the one-wavefront position-parallel walk) instead of cutting them into parts
    (encode_kernels.hpp: single_part_blocks; Terse.hpp:352-389 either way): 768 frames of 640 x 640 pixels -- 34 134 blocks --,
    Poisson(3) counts (every frame handed over) and synth-v1 (none), index-free decode, index built from the stream, decode
    with it; two frames against the oracle."""
    import torch
    from trpx_amd import codec, _lib, workloads
    n, frames = 640 * 640, 768
    assert _lib.lib().trpx_decode_parts_per_frame(codec.dtype_code(np.uint16), n, frames, 12) == 1
    assert _lib.lib().trpx_decode_parts_per_frame(codec.dtype_code(np.uint16), n, frames - 1, 12) > 1
    for kind in ("poisson3", "synth"):
        px = workloads.poisson_u16(3.0, 0, frames, n, device=gpu) if kind == "poisson3" else codec.synth(np.uint16, 5, frames, n, device=gpu)
        enc = codec.encode(px); torch.cuda.synchronize(); enc.check()
        offs = enc.frame_offsets.cpu().numpy()
        for f in (0, frames - 1):
            want = oracle.encode_stack(px[f:f + 1].cpu().numpy())[0]
            assert bytes(enc.data[int(offs[f]): int(offs[f + 1])].cpu().numpy()) == bytes(want), (kind, f)
        back, st = codec.decode(enc.stack(), enc.frame_offsets, n, frames, np.uint16)
        torch.cuda.synchronize()
        assert int(st[0]) == 0 and torch.equal(back, px), kind
        idx = codec.build_index(enc.stack(), enc.frame_offsets, n, frames, np.uint16)
        back2, st2 = codec.decode(enc.stack(), enc.frame_offsets, n, frames, np.uint16, index=idx)
        torch.cuda.synchronize()
        assert int(st2[0]) == 0 and torch.equal(back2, px), kind
        del px, enc, back, back2, idx


def test_index_route_waits_are_bounded(gpu, tmp_path):
    """The index route's walk is ONE launch in which every part's wavefront waits for words its neighbours publish: the start
    state of the part behind (its walk ends there) and that part's walk record (the link into it) -- decode_part.hip,
    k_chain_walk.  The waits are bounded; a wavefront that gives up reports a bad walk / a failed link and the frame takes the
    position-parallel route.  Test build (make chain_timeout): frame 0's part 3 never publishes its start state, frame 1's
    part 5 never its record, the bound is 2 ms.  Both frames must be handed over (status[2]), nothing may hang, the pixels
    are exact -- header-dense frames (every link open, records waited for) and frames whose cuts lie in runs."""
    variant = os.path.join(ROOT, "tools", "variants", "libtrpx_chain_timeout.so")
    if not os.path.exists(variant):
        pytest.skip("test variant not built (make -C trpx_amd/csrc chain_timeout)")
    script = tmp_path / "t.py"
    script.write_text(f"""
import sys
sys.path.insert(0, {ROOT!r})
import numpy as np, torch
from trpx_amd import codec, _lib, workloads
n, frames = 1030 * 1065, 6
for kind in ("poisson3", "synth"):
    px = workloads.poisson_u16(3.0, 0, frames, n, device="cuda") if kind == "poisson3" else codec.synth(np.uint16, 3, frames, n, device="cuda")
    enc = codec.encode(px); torch.cuda.synchronize(); enc.check()
    P = _lib.lib().trpx_decode_parts_per_frame(codec.dtype_code(np.uint16), n, frames, 12)
    assert P > 8, P
    back, st = codec.decode(enc.stack(), enc.frame_offsets, n, frames, np.uint16)
    torch.cuda.synchronize()
    s = st.cpu().numpy()
    assert s[0] == 0 and torch.equal(back, px), (kind, s)
    assert 2 <= s[2] <= 3, (kind, s)          # frames 0 and 1 (and no others but by the data's own doing)
    idx = codec.build_index(enc.stack(), enc.frame_offsets, n, frames, np.uint16)
    back2, st2 = codec.decode(enc.stack(), enc.frame_offsets, n, frames, np.uint16, index=idx)
    torch.cuda.synchronize()
    assert int(st2[0]) == 0 and torch.equal(back2, px), kind
print("OK")
""")
    r = subprocess.run([os.sys.executable, str(script)], capture_output=True, text=True, timeout=600, env=dict(os.environ, TRPX_LIB=variant))
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_encoder_workspace_between_calls(gpu, oracle):
    """The single-pass encoder's descriptor words are cleared by the call's last kernel, and the library skips the clearing
    launch for a workspace it remembers as clean (include/trpx_hip.h, "Workspaces between calls").  Same workspace call after
    call, another geometry in between, the decoder given the same memory (the library forgets it by itself), a foreign
    write announced with trpx_workspace_invalidate -- the stream is the oracle's every time; a foreign write that is NOT
    announced (here: the whole workspace zeroed, tag included) is caught by the first tile's tag check: the call reports
    TRPX_ERR_TIMEOUT and Encoded.check() answers with the two-pass pipeline (Terse.hpp:500-549: identical bytes)."""
    import torch
    from trpx_amd import codec, _lib
    L = _lib.lib()
    n, frames = 512 * 512, 24
    px = codec.synth(np.uint16, 7, frames, n, device=gpu)
    want = oracle.encode_stack(px.cpu().numpy())[0]
    ws = codec.Workspace(gpu)

    def enc(p=px, w=want):
        e = codec.encode(p, workspace=ws)
        torch.cuda.synchronize()
        code = int(e.status[0].item())
        e.check()
        assert e.stack().cpu().numpy().tobytes() == w.tobytes()
        return code

    assert enc() == 0 and enc() == 0 and enc() == 0                       # unknown -> clean -> clean
    px2 = codec.synth(np.uint16, 90, 5, 300 * 300, device=gpu)             # another geometry in the same memory
    want2 = oracle.encode_stack(px2.cpu().numpy())[0]
    assert enc(px2, want2) == 0 and enc() == 0 and enc(px2, want2) == 0 and enc(px2, want2) == 0
    assert enc() == 0
    back, st = codec.decode(torch.from_numpy(want).to(gpu), codec.encode(px).frame_offsets, n, frames, np.uint16, workspace=ws)
    torch.cuda.synchronize()                                               # the decoder scribbles over the same workspace ...
    assert int(st[0].item()) == 0 and torch.equal(back.view(torch.int16), px.view(torch.int16))
    assert enc() == 0 and enc() == 0                                       # ... and the library knows
    ws.buf.fill_(0xA5)                                                     # a foreign write, announced
    L.trpx_workspace_invalidate(ws.buf.data_ptr(), ws.buf.numel())
    assert enc() == 0 and enc() == 0
    ws.buf.zero_()                                                         # a foreign write, NOT announced: the tag is gone
    assert enc() == _lib.ERR_TIMEOUT                                       # reported by the first tile; check() re-ran the call (two-pass)
    assert enc() == 0 and enc() == 0


@pytest.mark.parametrize("dtype", [np.uint16, np.int16])
def test_encoder_frames_two_bytes_behind_a_dword(gpu, oracle, dtype):
    """Stacks of 16-bit frames with an odd pixel count (513 x 511, ...): every second frame starts 2 bytes behind a dword
    boundary and is loaded through k_encode_fused_mis2's realigned loads (dword-aligned loads from the dword in front + funnel
    shifts, encode_fused.hip: load_raw_mis2); and a stack whose FIRST pixel sits there (every frame but the first realigned).
    Byte-identical to the oracle (Terse.hpp:500-549), incl. frames with a partial last block and tiles' halo blocks."""
    import torch
    from trpx_amd import codec
    rng = np.random.RandomState(3)
    for n, frames, shift in ((513 * 511, 9, 0), (12 * 1024 * 3 + 7, 6, 0), (300 * 300, 5, 1), (12 * 2048 + 1, 4, 1)):
        a = rng.poisson(2.0, (frames, n)).astype(np.int64)
        a[:, ::4099] = 3000
        if np.dtype(dtype).kind == "i":
            a = a - 2
        a = a.astype(dtype)
        want = oracle.encode_stack(a)[0]
        flat = torch.zeros(frames * n + 8, dtype=torch.int16, device=gpu)
        flat[shift: shift + frames * n] = torch.from_numpy(a.view(np.int16).reshape(-1)).to(gpu)
        px = flat[shift: shift + frames * n].view(frames, n)
        px = px.view(torch.uint16) if np.dtype(dtype).kind == "u" else px
        assert px.data_ptr() % 4 == 2 * shift
        enc = codec.encode(px)
        torch.cuda.synchronize()
        enc.check()
        assert enc.stack().cpu().numpy().tobytes() == want.tobytes(), (dtype, n, frames, shift)


def test_default_workspace_is_not_remembered(gpu, oracle):
    """codec.encode() without workspace= allocates its scratch itself and lets it die with the Encoded object: the library
    must not remember that memory as a clean workspace (torch's allocator hands the block to the next tensor).  Encode, drop,
    scribble over what comes back from the allocator -- all but the last words, where the tag sits --, encode the same
    geometry again: status 0 at the first attempt, the oracle's bytes."""
    import torch, gc
    from trpx_amd import codec
    n, frames = 512 * 512, 24
    px = codec.synth(np.uint16, 11, frames, n, device=gpu)
    want = oracle.encode_stack(px.cpu().numpy())[0].tobytes()
    for _ in range(3):
        e = codec.encode(px)
        torch.cuda.synchronize()
        assert int(e.status[0].item()) == 0 and e.stack().cpu().numpy().tobytes() == want
        ws_bytes = e._retry[1].numel()
        del e
        gc.collect()
        junk = torch.empty(ws_bytes - 4096, dtype=torch.uint8, device=gpu)  # (same size class: the caching allocator's block)
        junk.fill_(0x5A)
        torch.cuda.synchronize()
        del junk


def test_host_entry_points_share_the_arena_workspace(gpu, oracle):
    """trpx_encode_host registers its thread's workspace slot as clean; trpx_group_states_host / trpx_frame_offsets_host write an
    index / a walk's scratch into the same slot (api.hip: Arena).  Terse compress -> write() with the frame index -> compress
    the next stack: the second encode must not trust a scribbled workspace (status 0, oracle bytes, no 0.25 s timeout)."""
    import ctypes, time
    from trpx_amd import _lib
    L = _lib.lib()
    n, frames = 512 * 512, 6
    a = oracle.synth(np.uint16, 40, frames, n)
    want, sizes, _ = oracle.encode_stack(a)
    cap = frames * L.trpx_worst_case_bytes(_lib.U16, n, 12)
    out = np.empty(cap, dtype=np.uint8)
    total = ctypes.c_size_t(0)
    offs = np.zeros(frames + 1, dtype=np.uint64)
    pb = ctypes.c_uint32(0)

    def enc():
        t0 = time.time()
        _lib.check(L.trpx_encode_host(_lib.U16, a.ctypes.data, n, frames, 12, out.ctypes.data, cap, ctypes.byref(total),
                                      offs.ctypes.data, ctypes.byref(pb), 0))
        assert time.time() - t0 < 0.2, "a look-back wait ran into its timeout"
        assert out[: total.value].tobytes() == want.tobytes()

    enc(); enc()
    ng = L.trpx_group_count(n, 12)
    states = np.zeros(frames * ng, dtype=np.uint64)
    _lib.check(L.trpx_group_states_host(out.ctypes.data, total.value, offs.ctypes.data, n, frames, 12, 16, states.ctypes.data, 0))
    enc()
    got = np.zeros(frames + 1, dtype=np.uint64)
    _lib.check(L.trpx_frame_offsets_host(out.ctypes.data, total.value, n, frames, 12, 16, got.ctypes.data, 0))
    assert (got == offs).all()
    enc(); enc()


@pytest.mark.parametrize("dtype,lo,hi", [(np.uint16, 100, 108), (np.int32, 1000, 1024), (np.uint8, 192, 200), (np.int16, -300, -290)])
def test_large_frames_with_a_pedestal(gpu, oracle, dtype, lo, hi):
    """Raw detector counts sit on a pedestal: every pixel in [lo, hi).  Their constant payload bits pass the run test of the
    part cuts (decode_part.hip: header bits 1 at the run's stride) like header bits do, at twelve positions per constant bit
    and stride; the cuts have to tell the header among them (part_header_phase), or the frame must take another route and
    still decode exactly.  One run-dominated frame and one with a sprinkling of outliers per type."""
    import torch
    from trpx_amd import codec
    rng = np.random.RandomState(11)
    n, frames = 1030 * 1065 + 7, 5
    a = rng.randint(lo, hi, (frames, n)).astype(dtype)
    a[1::2, rng.randint(0, n, 300)] = lo + (hi - lo) * 3                   # outliers: explicit headers every few thousand blocks
    px = torch.from_numpy(a).to(gpu)
    enc = codec.encode(px)
    torch.cuda.synchronize()
    enc.check()
    want, _ = oracle.encode(a[1])
    o = enc.frame_offsets.cpu().numpy()
    assert enc.stack()[int(o[1]): int(o[2])].cpu().numpy().tobytes() == want.tobytes()
    back, st = codec.decode(enc.stack(), enc.frame_offsets, n, frames, dtype)
    torch.cuda.synchronize()
    assert int(st[0].item()) == 0 and torch.equal(back, px)


@pytest.mark.parametrize("dtype", [np.uint16, np.int32])
def test_large_frames_hostile_streams(gpu, oracle, dtype):
    """Frames of more than 32 K blocks are cut at positions whose chain state is GUESSED and walked from there by walks that
    tolerate anything (decode_part.hip) -- so what a damaged frame does to that route is tested on its own: one frame of a
    stack of six is overwritten (ones, zeroes, noise, a burst, its first / last byte, single bits, offsets that run
    backwards), everything else stays as the encoder wrote it.  The call must come back with status 0 or TRPX_ERR_CORRUPT
    -- CORRUPT whenever the damage is gross --, the frames that were not touched must decode exactly whenever it says 0
    (frames are independent: Terse.hpp:502-505), and the workspace must serve the next call (the hand-over list, the part
    tables and the stack statistics live in it)."""
    import torch
    from trpx_amd import codec, _lib
    n, frames, victim = 12 * 70000 + 5, 6, 2
    px = codec.synth(dtype, 0, frames, n, device=gpu)
    enc = codec.encode(px)
    torch.cuda.synchronize()
    enc.check()
    good = enc.stack().clone()
    offs = enc.frame_offsets.clone()
    o = offs.cpu().numpy()
    lo, hi = int(o[victim]), int(o[victim + 1])
    want, _ = oracle.encode(px[victim].cpu().numpy())
    assert good[lo:hi].cpu().numpy().tobytes() == want.tobytes()
    ws = codec.Workspace(gpu)
    view = torch.int32 if np.dtype(dtype).itemsize == 4 else torch.int16
    others = [f for f in range(frames) if f != victim]

    def run(stream, offsets=offs):
        back, st = codec.decode(stream, offsets, n, frames, dtype, workspace=ws)
        torch.cuda.synchronize()
        return back, int(st[0].item())

    def clean_call_still_works(after):
        back, s = run(good)
        assert s == 0 and torch.equal(back.view(view), px.view(view)), f"the workspace does not serve a clean call after: {after}"

    clean_call_still_works("nothing")
    g = torch.Generator().manual_seed(20260104)
    gross = {"ones": lambda b: b[lo:hi].fill_(0xFF), "zeroes": lambda b: b[lo:hi].zero_(),
             "noise": lambda b: b[lo:hi].copy_(torch.randint(0, 256, (hi - lo,), generator=g, dtype=torch.uint8).to(gpu)),
             "burst": lambda b: b[(lo + hi) // 2: (lo + hi) // 2 + 4096].copy_(torch.randint(0, 256, (4096,), generator=g, dtype=torch.uint8).to(gpu))}
    for what, spoil in gross.items():
        bad = good.clone()
        spoil(bad)
        assert torch.equal(bad[:lo], good[:lo]) and torch.equal(bad[hi:], good[hi:])          # (only the victim's bytes)
        back, s = run(bad)
        assert s == _lib.ERR_CORRUPT, (what, s)
        clean_call_still_works(what)
    # small damage: a flip in payload bits still parses (status 0, the victim's pixels differ); one in a header or a width does not
    positions = [lo, hi - 1] + [int(p) for p in torch.randint(lo, hi, (24,), generator=g)]
    seen = {0: 0, _lib.ERR_CORRUPT: 0}
    for k, p in enumerate(positions):
        bad = good.clone()
        bad[p] ^= 1 << (k % 8)
        back, s = run(bad)
        assert s in (0, _lib.ERR_CORRUPT), (p - lo, s)
        seen[s] += 1
        if s == 0:
            for f in others:
                assert torch.equal(back[f].view(view), px[f].view(view)), f"a flip in frame {victim} (byte {p - lo}) changed frame {f}"
    assert seen[_lib.ERR_CORRUPT] > 0 and seen[0] > 0, seen        # (both kinds occurred: the check on the other frames above has run)
    clean_call_still_works("single flips")
    # offsets that run backwards / past the stream: refused, not followed
    for what, k, v in (("backwards", victim + 1, lo - 5), ("past the end", frames, int(good.numel()) + 4096)):
        ob = offs.clone()
        ob[k] = v
        back, s = run(good, ob)
        assert s == _lib.ERR_CORRUPT, (what, s)
        clean_call_still_works(what)


@pytest.mark.parametrize("shape,frames,victim,route", [((512, 512), 70, 10, 0), ((1030, 1065), 5, 2, 0), ((512, 512), 70, 10, 5)])
def test_header_dense_hostile_streams(gpu, oracle, shape, frames, victim, route):
    """The same for the routes header-dense frames take -- Poisson(3) counts, a width change on one block in four: stacks of
    512 x 512 frames are handed over by the per-frame decoder to the position-parallel walk with one wavefront per frame
    (70 frames: the hand-over is decided by the stack's statistics), frames of 1030 x 1065 are voted header-dense by the index
    route's walkers (the victim's own vote may be garbage) and walked one workgroup per frame (k_seg_wg, decode_seg.hip).  Both
    count their blocks from states that are guesses until the
    links close, in passes that do not check widths against the pixel type: a damaged frame must end in TRPX_ERR_CORRUPT or,
    where only payload bits changed, in status 0 with every other frame exact.  route 5: the listed frames through the dense walk
    (decode_dense.hip: one speculative pass, link walks, a verified write pass, the serial walk as the last resort)."""
    import torch
    from trpx_amd import codec, _lib, workloads
    _lib.lib().trpx_set_decode_path(route)
    try:
        _header_dense_hostile(gpu, oracle, shape, frames, victim)
    finally:
        _lib.lib().trpx_set_decode_path(0)


def _header_dense_hostile(gpu, oracle, shape, frames, victim):
    import torch
    from trpx_amd import codec, _lib, workloads
    n = shape[0] * shape[1]
    px = workloads.poisson_u16(3.0, 0, frames, n, device=gpu)
    enc = codec.encode(px)
    torch.cuda.synchronize()
    enc.check()
    good = enc.stack().clone()
    offs = enc.frame_offsets.clone()
    o = offs.cpu().numpy()
    lo, hi = int(o[victim]), int(o[victim + 1])
    want, _ = oracle.encode(px[victim].cpu().numpy())
    assert good[lo:hi].cpu().numpy().tobytes() == want.tobytes()
    ws = codec.Workspace(gpu)
    others = [f for f in range(frames) if f != victim]

    def run(stream):
        back, st = codec.decode(stream, offs, n, frames, np.uint16, workspace=ws)
        torch.cuda.synchronize()
        return back, int(st[0].item())

    def clean_call_still_works(after):
        back, s = run(good)
        assert s == 0 and torch.equal(back.view(torch.int16), px.view(torch.int16)), f"the workspace does not serve a clean call after: {after}"

    clean_call_still_works("nothing")
    g = torch.Generator().manual_seed(20260105)
    gross = {"ones": lambda b: b[lo:hi].fill_(0xFF), "zeroes": lambda b: b[lo:hi].zero_(),
             "noise": lambda b: b[lo:hi].copy_(torch.randint(0, 256, (hi - lo,), generator=g, dtype=torch.uint8).to(gpu)),
             "burst": lambda b: b[(lo + hi) // 2: (lo + hi) // 2 + 2048].copy_(torch.randint(0, 256, (2048,), generator=g, dtype=torch.uint8).to(gpu))}
    for what, spoil in gross.items():
        bad = good.clone()
        spoil(bad)
        back, s = run(bad)
        assert s == _lib.ERR_CORRUPT, (what, s)
        clean_call_still_works(what)
    seen = {0: 0, _lib.ERR_CORRUPT: 0}
    for k, p in enumerate([lo, hi - 1] + [int(q) for q in torch.randint(lo, hi, (22,), generator=g)]):
        bad = good.clone()
        bad[p] ^= 1 << (k % 8)
        back, s = run(bad)
        assert s in (0, _lib.ERR_CORRUPT), (p - lo, s)
        seen[s] += 1
        if s == 0:
            for f in others:
                assert torch.equal(back[f].view(torch.int16), px[f].view(torch.int16)), f"a flip in frame {victim} (byte {p - lo}) changed frame {f}"
    assert seen[_lib.ERR_CORRUPT] > 0 and seen[0] > 0, seen
    clean_call_still_works("single flips")


def test_two_host_threads_share_the_library(gpu, oracle):
    """Two host threads, each with its own stream, workspaces and data, encode and decode at the same time (ctypes releases
    the GIL for the length of a call).  What they share is the library: the encoder's register of clean workspaces (one
    mutex, encode_fused.hip -- a clearing launch skipped for the wrong workspace would mean a damaged stream or a look-back
    timeout) and, per thread, the host wrappers' arena, error text and profiler (thread_local, api.hip).  Every stack must be
    the one the thread's own data gives single-threaded -- byte for byte -- and decode back to it."""
    import threading
    import torch
    from trpx_amd import codec
    n, frames, rounds = 512 * 512, 64, 25
    data, expect = [], []
    for t in range(2):
        px = codec.synth(np.uint16, 1000 * t, frames, n, device=gpu)
        enc = codec.encode(px)
        torch.cuda.synchronize()
        enc.check()
        want, _ = oracle.encode(px[0].cpu().numpy())
        o = enc.frame_offsets.cpu().numpy()
        assert enc.stack()[: int(o[1])].cpu().numpy().tobytes() == want.tobytes()
        data.append(px)
        expect.append((enc.stack().clone(), enc.frame_offsets.clone()))
    assert not torch.equal(expect[0][0][:4096], expect[1][0][:4096])                     # (different data: cross-talk would show)
    small = [np.random.RandomState(7 + t).poisson(2.0, (4, 256 * 256)).astype(np.uint16) for t in range(2)]
    small_want = [np.concatenate([oracle.encode(f)[0] for f in s]) for s in small]
    failures, gate = [], threading.Barrier(2)

    def work(t):
        try:
            torch.cuda.set_device(gpu)
            s = torch.cuda.Stream(device=gpu)
            ws_e, ws_d = codec.Workspace(gpu), codec.Workspace(gpu)
            px, (stack, offs) = data[t], expect[t]
            cap = (frames * codec.worst_case_bytes(torch.uint16, n) + 15) // 16 * 16
            out = torch.empty(cap, dtype=torch.uint8, device=gpu)
            fo = torch.empty(frames + 1, dtype=torch.int64, device=gpu)
            st_e = torch.empty(8, dtype=torch.int32, device=gpu)
            st_d = torch.empty(8, dtype=torch.int32, device=gpu)
            back = torch.empty_like(px)
            gate.wait(timeout=60)
            with torch.cuda.stream(s):
                for r in range(rounds):
                    e = codec.encode(px, out=out, workspace=ws_e, frame_offsets=fo, status=st_e)
                    codec.decode(out, fo, n, frames, np.uint16, out=back, workspace=ws_d, status=st_d)
                    s.synchronize()
                    if int(st_e[0].item()) != 0 or int(st_d[0].item()) != 0:
                        failures.append((t, r, "status", int(st_e[0].item()), int(st_d[0].item())))
                    elif not (torch.equal(fo, offs) and torch.equal(out[: stack.numel()], stack)):
                        failures.append((t, r, "stream differs"))
                    elif not torch.equal(back.view(torch.int16), px.view(torch.int16)):
                        failures.append((t, r, "pixels differ"))
            for r in range(3):                                                            # the host wrappers: per-thread arena and stream
                got, ho, _ = _host_encode(small[t])
                if got.tobytes() != small_want[t].tobytes():
                    failures.append((t, r, "host stream differs"))
                if not (_host_decode(got, ho, 256 * 256, 4, np.uint16) == small[t]).all():
                    failures.append((t, r, "host pixels differ"))
        except Exception as ex:                                                           # (an assert in a thread would be lost)
            failures.append((t, "exception", repr(ex)))

    threads = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=240)
    assert not any(th.is_alive() for th in threads), "a thread did not come back"
    assert not failures, failures[:5]
