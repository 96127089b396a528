"""CPU tests (no GPU): the C-ABI library loads, exports every symbol include/trpx_hip.h
declares, its pure-arithmetic / text entry points work, and compute entry points FAIL LOUDLY
without a device (no CPU fallback exists in the product)."""
import ctypes as C
import io
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "trpx_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(trpx_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from trpx_amd import _lib
    L = _lib.lib()
    names = _declared_symbols()
    assert len(names) >= 15
    for n in names:
        assert hasattr(L, n), f"libtrpx_hip.so does not export {n}"
    assert set(names) == set(_lib.SYMBOLS), "python binding table out of sync with the header"
    hdr = open(os.path.join(ROOT, "include", "trpx_hip.h")).read()
    assert f"#define TRPX_ABI_VERSION {_lib.ABI_VERSION}\n" in hdr and L.trpx_abi_version() == _lib.ABI_VERSION


def test_pure_arithmetic_entry_points(oracle):
    from trpx_amd import _lib
    L = _lib.lib()
    assert [L.trpx_dtype_size(d) for d in range(6)] == [1, 1, 2, 2, 4, 4]
    assert L.trpx_dtype_size(17) == 0
    assert [L.trpx_dtype_is_signed(d) for d in range(6)] == [0, 1, 0, 1, 0, 1]
    for dt, code in ((np.uint8, 0), (np.uint16, 2), (np.int32, 5)):
        for n in (1, 12, 13, 262144, 4096 * 4096):
            assert L.trpx_worst_case_bytes(code, n, 12) == oracle.worst_case_bytes(dt, n)
    assert L.trpx_worst_case_bytes(2, 262144, 12) == 524288 + (12 * 21846 + 7) // 8 + 1
    assert L.trpx_encode_workspace_bytes(2, 262144, 2000, 12) > 0
    assert L.trpx_decode_workspace_bytes(2, 262144, 2000, 12) > 2000 * 21846
    assert L.trpx_encode_workspace_bytes(2, 262144, 2000, 7) > 0       # any block size 1..4096 (generic kernels)
    assert L.trpx_encode_workspace_bytes(2, 262144, 2000, 0) == 0 and L.trpx_encode_workspace_bytes(2, 262144, 2000, 5000) == 0


def test_header_text_matches_reference_goldens(golden):
    from trpx_amd import _lib
    L = _lib.lib()
    for c in golden["cases"] + [golden["stack"]]:
        m = dict(re.findall(r'(\w+)="([^"]*)"', c["header"]))
        h = _lib.trpx_header()
        h.prolix_bits, h.is_signed, h.block = int(m["prolix_bits"]), int(m["signed"]), int(m["block"])
        h.memory_size, h.number_of_values = int(m["memory_size"]), int(m["number_of_values"])
        h.number_of_frames = int(m["number_of_frames"])
        dims = [int(x) for x in m.get("dimensions", "").split()]
        h.n_dims = len(dims)
        for i, d in enumerate(dims):
            h.dims[i] = d
        buf = C.create_string_buffer(512)
        n = L.trpx_header_format(C.byref(h), buf, 512)
        assert buf.raw[:n].decode() == c["header"], c["name"]
        # and back, with junk in front and a payload behind (XML_element.hpp:442-452)
        blob = b"junk <Ters <!-- x -->" + c["header"].encode() + b"\x00\x01payload"
        h2, off = _lib.trpx_header(), C.c_size_t(0)
        assert L.trpx_header_parse(blob, len(blob), C.byref(h2), C.byref(off)) == 0
        assert blob[off.value:] == b"\x00\x01payload"
        for f in ("prolix_bits", "is_signed", "block", "memory_size", "number_of_values", "number_of_frames", "n_dims"):
            assert getattr(h2, f) == getattr(h, f), (c["name"], f)
        assert list(h2.dims)[:h2.n_dims] == dims


def test_header_parse_tolerates_unknown_attributes_and_order():
    from trpx_amd import _lib
    L = _lib.lib()
    blob = (b"<Terse frame_sizes='5 6' number_of_frames=\"2\" number_of_values=\"7\" memory_size=\"11\" "
            b"block=\"12\" signed='0' prolix_bits=\"3\"/>XYZ")
    h, off = _lib.trpx_header(), C.c_size_t(0)
    assert L.trpx_header_parse(blob, len(blob), C.byref(h), C.byref(off)) == 0
    assert (h.number_of_frames, h.number_of_values, h.memory_size, h.block, h.is_signed, h.prolix_bits) == (2, 7, 11, 12, 0, 3)
    assert blob[off.value:] == b"XYZ"
    # missing number_of_frames: the reference's stoull("") throws -> error here
    bad = b'<Terse prolix_bits="3" signed="0" block="12" memory_size="11" number_of_values="7"/>'
    assert L.trpx_header_parse(bad, len(bad), C.byref(h), C.byref(off)) != 0
    assert L.trpx_header_parse(b"no header here", 14, C.byref(h), C.byref(off)) != 0


def test_compute_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from trpx_amd import Terse, TrpxError, _lib
    with pytest.raises(TrpxError) as e:
        Terse(np.arange(100, dtype=np.uint16))
    assert e.value.code in (_lib.ERR_NO_DEVICE, _lib.ERR_HIP)


def test_argument_validation_without_gpu():
    from trpx_amd import _lib
    L = _lib.lib()
    # an impossible block size is rejected before anything touches a device
    rc = L.trpx_encode(2, 16, 100, 1, 0, 16, 0, 16, 16, 16, 1 << 20, None)
    assert rc == _lib.ERR_UNSUPPORTED and b"block" in L.trpx_last_error_string()
    rc = L.trpx_encode_indexed(2, 16, 100, 1, 7, 16, 0, 16, 16, 16, 16, 1 << 20, None)   # the decode index needs block = 12
    assert rc == _lib.ERR_UNSUPPORTED
    rc = L.trpx_encode(11, 16, 100, 1, 12, 16, 0, 16, 16, 16, 1 << 20, None)       # (8 / 9 are the 64-bit containers)
    assert rc == _lib.ERR_INVALID_ARG
    rc = L.trpx_decode(1, 2, 16, 10, None, 100, 1, 12, 16, 16, 16, 1 << 20, None)   # signed stream -> u16
    assert rc == _lib.ERR_UNSUPPORTED
    rc = L.trpx_encode(2, 16, 100, 1, 12, 16, 0, 16, 16, 16, 8, None)                # workspace too small
    assert rc == _lib.ERR_CAPACITY


def test_host_code_survives_corrupt_input_under_asan_ubsan(tmp_path):
    """SURVEY section 5 (sanitizers on host code): tests/cpp/host_sanitize.cpp drives everything in the product that
    parses untrusted bytes on the host -- the .trpx header text (header_text.cpp), the TIFF reader (Grey_tif.hpp scan:
    including the size_t-overflow image and the 2^32-entry array of ADVICE r1) and Terse(std::ifstream&) (Terse.hpp
    f_read) -- with valid, truncated and randomly corrupted inputs in a g++ -fsanitize=address,undefined build (CPU only)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "tests", "cpp"), "host_sanitize"])
    scratch = "/dev/shm" if os.path.isdir("/dev/shm") else str(tmp_path)
    path = os.path.join(scratch, f"trpx_host_sanitize_{os.getpid()}.trpx")
    try:
        r = subprocess.run([os.path.join(root, "tests", "cpp", "host_sanitize"), path], capture_output=True, text=True, timeout=300)
    finally:
        if os.path.exists(path):
            os.remove(path)
    assert r.returncode == 0 and "OK host_sanitize" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_integration_md_code_blocks_are_the_compiled_snippets():
    """INTEGRATION.md sections 2 and 4 show a maintainer the calls to write; tests/cpp/integration_snippets.cpp is those
    blocks compiled against include/trpx_hip.h (make -C tests/cpp).  Every line of every marked snippet must appear in the
    document, in order: a signature change that is not carried into the document fails here."""
    import re
    cpp = open(os.path.join(ROOT, "tests", "cpp", "integration_snippets.cpp")).read()
    doc = [ln.strip() for ln in open(os.path.join(ROOT, "INTEGRATION.md")).read().splitlines()]
    snippets = re.findall(r"// \[snippet:(\w+)\]\n(.*?)// \[/snippet\]", cpp, flags=re.S)
    assert {n for n, _ in snippets} == {"include", "f_compress", "prolix", "sharded", "sharded1"}
    for name, body in snippets:
        at = 0
        for ln in (x.strip() for x in body.splitlines()):
            if not ln:
                continue
            assert ln in doc[at:], (name, ln)
            at = doc.index(ln, at) + 1
    assert os.path.exists(os.path.join(ROOT, "tests", "cpp", "integration_snippets")), "make -C tests/cpp did not build it"


def test_kernel_sources_compile_without_the_llvm_testing_option():
    """decode_frame.hip / encode_fused.hip are built with -mllvm -structurizecfg-skip-uniform-regions=1 (a speed matter: scalar
    branches stay scalar) around hand-written asm; the sources must stay valid C++ / asm for the compiler's default pipeline
    too, so that a toolchain without the option still builds a correct library (csrc/Makefile: noflag, toolchain_check)."""
    import subprocess
    r = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "trpx_amd", "csrc"), "noflag"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
