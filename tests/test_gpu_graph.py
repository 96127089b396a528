"""GPU test: the device entry points only enqueue work (no allocation, no host sync) -- they can be captured
into a HIP graph and replayed (include/trpx_hip.h conventions)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_encode_decode_capture_into_hip_graph(gpu):
    import torch
    from trpx_amd import codec, _lib
    n, frames = 512 * 512, 64
    px = codec.synth(np.uint16, 0, frames, n, device=gpu)
    ws_e, ws_d = codec.Workspace(gpu), codec.Workspace(gpu)
    cap = (frames * codec.worst_case_bytes(np.uint16, n) + 15) // 16 * 16
    out = torch.empty(cap, dtype=torch.uint8, device=gpu)
    offs = torch.empty(frames + 1, dtype=torch.int64, device=gpu)
    st_e = torch.empty(8, dtype=torch.int32, device=gpu)
    st_d = torch.empty(8, dtype=torch.int32, device=gpu)
    back = torch.empty((frames, n), dtype=torch.uint16, device=gpu)
    L = _lib.lib()
    ws_e.get(L.trpx_encode_workspace_bytes(_lib.U16, n, frames, 12))
    ws_d.get(L.trpx_decode_workspace_bytes(_lib.U16, n, frames, 12))
    # eager reference run
    codec.encode(px, out=out, workspace=ws_e, frame_offsets=offs, status=st_e)
    torch.cuda.synchronize()
    want = out[: int(offs[-1].item())].clone()
    want_offs = offs.clone()
    # capture encode + decode, replay on new pixels
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            codec.encode(px, out=out, workspace=ws_e, frame_offsets=offs, status=st_e)
            codec.decode(out, offs, n, frames, np.uint16, out=back, workspace=ws_d, status=st_d)
    out.zero_(); offs.zero_(); back.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert int(st_e[0].item()) == 0 and int(st_d[0].item()) == 0
    assert torch.equal(offs, want_offs) and torch.equal(out[: want.numel()], want)
    assert torch.equal(back.view(torch.int16), px.view(torch.int16))
    px.copy_(codec.synth(np.uint16, 500, frames, n, device=gpu))           # new data, same graph
    g.replay()
    torch.cuda.synchronize()
    fresh = codec.encode(px)
    torch.cuda.synchronize()
    assert torch.equal(offs, fresh.frame_offsets) and torch.equal(out[: fresh.total_bytes()], fresh.stack())
    assert torch.equal(back.view(torch.int16), px.view(torch.int16))


def test_large_frame_decode_replays_from_a_hip_graph(gpu):
    """The large-frame route's walk is one launch whose wavefronts wait for each other's words (decode_part.hip, k_chain_walk);
    the words are cleared by the call's first launch (k_chain_zero), so a captured decode replays like an eager one: replayed
    three times over new pixels each time -- 1030 x 1065 frames (parts extracted by the per-frame decoder), Poisson(3) counts of
    the same size (the index route's extraction) -- and compared with the pixels."""
    import torch
    from trpx_amd import codec, _lib, workloads
    n, frames = 1030 * 1065, 6
    L = _lib.lib()
    assert L.trpx_decode_parts_per_frame(_lib.U16, n, frames, 12) > 3
    ws_e, ws_d = codec.Workspace(gpu), codec.Workspace(gpu)
    ws_e.get(L.trpx_encode_workspace_bytes(_lib.U16, n, frames, 12))
    ws_d.get(L.trpx_decode_workspace_bytes(_lib.U16, n, frames, 12))
    cap = (frames * codec.worst_case_bytes(np.uint16, n) + 15) // 16 * 16
    out = torch.empty(cap, dtype=torch.uint8, device=gpu)
    offs = torch.empty(frames + 1, dtype=torch.int64, device=gpu)
    st_e = torch.empty(8, dtype=torch.int32, device=gpu)
    st_d = torch.empty(8, dtype=torch.int32, device=gpu)
    px = codec.synth(np.uint16, 0, frames, n, device=gpu)
    back = torch.empty_like(px)
    codec.encode(px, out=out, workspace=ws_e, frame_offsets=offs, status=st_e)           # (eager warm-up: the encoder's workspace is known clean)
    codec.decode(out, offs, n, frames, np.uint16, out=back, workspace=ws_d, status=st_d)
    torch.cuda.synchronize()
    assert torch.equal(back, px)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            codec.encode(px, out=out, workspace=ws_e, frame_offsets=offs, status=st_e)
            codec.decode(out, offs, n, frames, np.uint16, out=back, workspace=ws_d, status=st_d)
    for k, make in enumerate((lambda: codec.synth(np.uint16, 100, frames, n, device=gpu),
                              lambda: workloads.poisson_u16(3.0, 7, frames, n, device=gpu),
                              lambda: codec.synth(np.uint16, 300, frames, n, device=gpu))):
        px.copy_(make())
        back.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert int(st_e[0].item()) == 0 and int(st_d[0].item()) == 0, k
        assert torch.equal(back, px), k


def test_eager_and_replayed_encodes_alternate_on_one_workspace(gpu, oracle):
    """A captured call bakes its arguments in; the library's memory of clean workspaces is only updated by eager calls
    (encode_fused.hip: launch_fused_t).  Capture -> eager -> replay -> eager -> replay -> eager on ONE workspace: every call
    reports status 0 at its first attempt (no spurious TRPX_ERR_TIMEOUT from a tag the replay overwrote) and writes the
    oracle's bytes."""
    import torch
    from trpx_amd import codec, _lib
    n, frames = 512 * 512, 16
    px = codec.synth(np.uint16, 3, frames, n, device=gpu)
    want = oracle.encode_stack(px.cpu().numpy())[0].tobytes()
    ws = codec.Workspace(gpu)
    cap = (frames * codec.worst_case_bytes(np.uint16, n) + 15) // 16 * 16
    out = torch.empty(cap, dtype=torch.uint8, device=gpu)
    offs = torch.empty(frames + 1, dtype=torch.int64, device=gpu)
    st = torch.empty(8, dtype=torch.int32, device=gpu)
    ws.get(_lib.lib().trpx_encode_workspace_bytes(_lib.U16, n, frames, 12))

    def eager():
        out.zero_(); st.fill_(99)
        codec.encode(px, out=out, workspace=ws, frame_offsets=offs, status=st)
        torch.cuda.synchronize()
        assert int(st[0].item()) == 0, "eager call: device status %d at the first attempt" % int(st[0].item())
        assert out[: int(offs[-1].item())].cpu().numpy().tobytes() == want

    eager(); eager()                                                        # the workspace is remembered as clean
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            codec.encode(px, out=out, workspace=ws, frame_offsets=offs, status=st)

    def replay():
        out.zero_(); st.fill_(99)
        g.replay()
        torch.cuda.synchronize()
        assert int(st[0].item()) == 0
        assert out[: int(offs[-1].item())].cpu().numpy().tobytes() == want

    eager(); replay(); eager(); eager(); replay(); replay(); eager()


def test_indexed_decode_of_line_straddling_frames_eager_and_captured(gpu, oracle):
    """trpx_decode_indexed on a stack of >= 1024 small frames that start inside a cache line (31 x 33 u16 pixels: 2046 bytes per
    frame): eagerly it takes the walking decoder + the caller's index for handed-over frames, with a hand-over list in a buffer the
    library keeps per thread, device and stream (api.hip: indexed_scratch); a call that is being captured into a graph allocates
    nothing and takes the plain indexed route.  Both give the pixels -- run-dominated frames and header-dense ones --, the graph
    replays over new data, and an eager call with more frames (the buffer grows) in between does not disturb it."""
    import torch
    from trpx_amd import codec
    rng = np.random.RandomState(77)
    n, frames = 31 * 33, 1200
    px_a = rng.poisson(3.0, (frames, n)).astype(np.uint16)                    # header-dense: handed over
    px_b = np.minimum(rng.poisson(0.05, (frames, n)), 3).astype(np.uint16)   # long runs: kept by the walker
    px_b[:, ::97] = 900
    stacks = []
    for px in (px_a, px_b):
        d = torch.from_numpy(px.view(np.int16)).to(gpu).view(torch.uint16)
        enc = codec.encode(d, index=True)
        torch.cuda.synchronize()
        enc.check()
        assert enc.stack().cpu().numpy().tobytes() == oracle.encode_stack(px)[0].tobytes()
        back, st = codec.decode(enc.stack(), enc.frame_offsets, n, frames, np.uint16, index=enc.index)    # eager
        torch.cuda.synchronize()
        assert int(st[0]) == 0 and (back.cpu().numpy().view(np.uint16) == px).all()
        stacks.append((d, enc))
    d, enc = stacks[0]
    back = torch.empty_like(d)
    st = torch.empty(8, dtype=torch.int32, device=gpu)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            codec.decode(enc.data, enc.frame_offsets, n, frames, np.uint16, out=back, status=st, index=enc.index)
    for k in range(3):
        back.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert int(st[0].item()) == 0 and torch.equal(back.view(torch.int16), d.view(torch.int16)), k
        if k == 0:                                                            # an eager call with more frames: the library's buffer grows
            big = torch.from_numpy(rng.poisson(3.0, (2 * frames, n)).astype(np.int16)).to(gpu).view(torch.uint16)
            eb = codec.encode(big, index=True)
            bb, sb = codec.decode(eb.stack(), eb.frame_offsets, n, 2 * frames, np.uint16, index=eb.index)
            torch.cuda.synchronize()
            assert int(sb[0]) == 0 and torch.equal(bb.view(torch.int16), big.view(torch.int16))
