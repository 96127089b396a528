#!/usr/bin/env python3
"""bench.py -- headline benchmark of the TERSE/PROLIX hot path on MI355X.

A "step" = one pass of the hot path over one batch: Terse-encode the resident 2000-frame
512x512 uint16 synth-v1 stack (BASELINE.json configs[1]), gather the per-frame sizes across ranks
(RCCL, N>1 only), Prolix-decode it again (configs[2]).  Inputs are resident in HBM before the timed
region; nothing is cached between steps (outputs are re-produced every step).

    python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1: starts its own N ranks, one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (same, under a launcher)

Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` (dominant kernel,
HIP-event timed on the launch stream) and `cpu_baseline` (the reference's CPU path, N=1 only).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBPS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)
N_VALUES = 512 * 512
FRAMES_PER_GPU = 2000
ENC_STAGES_TWOPASS = ["tile_bits", "frame_scan", "stack_scan", "zero_edges", "pack"]
ENC_STAGES_FUSED = ["memset", "encode_fused", "stitch"]
PROFILE_TAG = "r06"                    # profiles/<tag>_traffic.json: PMC traffic + rocprofv3 averages of the round's final kernels
DEC_STAGES_FRAMES = ["decode_frames", "deferred_frames"]  # one workgroup per frame (walk + extraction fused) + the frames it defers
# frames of more than 32 K blocks (decode_part.hip, the index route): the one walk of short parts up to the part table / decode index
# (k_chain_walk -- start states, walk and links in one launch --, k_chain_resolve, k_chain_index), then the other route's launch for listed frames
# (k_seg_fallback, normally empty) + the extraction (k_unpack_tiles / k_decode_units_indexed / k_decode_parts)
DEC_STAGES_LARGE = ["chain_walk_to_index", "fallback_and_extract"]


def dec_stage_names(n_values):
    return DEC_STAGES_LARGE if (n_values + 11) // 12 > 32768 else DEC_STAGES_FRAMES


def kernel_sources_sha16() -> str:
    """Identity of the kernel sources the committed profiles/ numbers belong to."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "trpx_amd", "csrc", "*.h*"))) + [os.path.join(ROOT, "trpx_amd", "csrc", "Makefile")]:
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def host_cores() -> int:
    """CPUs this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(px_host: np.ndarray, cores: int):
    """The reference's CPU path timed on this box's host cores (bounded sample)."""
    from oracle import oracle as O
    frames = px_host.shape[0]
    kind = "reference" if O.have_ref() else "port"
    chunks = [px_host[i::cores] for i in range(cores)]
    chunks = [np.ascontiguousarray(c) for c in chunks if c.shape[0]]

    reps = 20         # 2000 frames x 20 passes x ~0.4 ms (enc+dec) = ~16-20 CPU-seconds in total

    def run(c, n=reps):
        tot = dict(enc_s=0.0, dec_s=0.0, bytes=0, ok=True)
        for _ in range(n):
            r = O.time_ref(c) if kind == "reference" else O.time_port(c, 1)
            tot["enc_s"] += r["enc_s"]; tot["dec_s"] += r["dec_s"]; tot["bytes"] = r["bytes"]; tot["ok"] &= r["ok"]
        return tot

    one = run(np.ascontiguousarray(px_host[:64]), 1)                    # 1-thread rate (and warm-up)
    t0 = time.perf_counter()
    with ThreadPoolExecutor(len(chunks)) as ex:
        res = list(ex.map(run, chunks))
    wall = time.perf_counter() - t0
    assert all(r["ok"] for r in res) and one["ok"], "CPU baseline failed to round-trip"
    enc_wall = max(r["enc_s"] for r in res) / reps
    dec_wall = max(r["dec_s"] for r in res) / reps
    return {
        "value": frames / (enc_wall + dec_wall), "unit": "frames/s", "cores": len(chunks), "kind": kind,
        "sample": f"{frames} frames 512x512 u16 synth-v1, encode+decode, one codec object per frame, "
                  f"{len(chunks)} threads (frames strided), {reps} passes, wall {wall:.2f}s",
        "encode_fps": frames / enc_wall, "decode_fps": frames / dec_wall,
        "one_thread_encode_fps": 64 / one["enc_s"], "one_thread_decode_fps": 64 / one["dec_s"],
        "compressed_bytes": int(sum(r["bytes"] for r in res)),
    }


def rank_commands(n: int, port: int):
    """One (argv, env additions) pair per rank: this script again, with the rendezvous a launcher would have set."""
    argv = [sys.executable, os.path.abspath(__file__)] + [a for a in sys.argv[1:] if a != "--spawn-dry-run"]
    return [(argv, {"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
                    "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")}) for r in range(n)]


def spawn_ranks(n: int, dry_run: bool) -> int:
    """`python bench.py --gpus N` started without torch.distributed.run: start the N ranks as child processes (one per
    GPU; frames are independent, Terse.hpp:502-505, so every rank codes its own 2000-frame shard), relay rank 0's JSON
    line and return the worst exit code.  This parent never initialises the GPU and nothing is exec'd after GPU init."""
    import socket
    import subprocess
    with socket.socket() as s:                      # a free rendezvous port
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmds = rank_commands(n, port)
    if dry_run:
        for argv, env in cmds:
            print(json.dumps({"argv": argv, "env": env}))
        return 0
    procs = []
    for r, (argv, env) in enumerate(cmds):
        procs.append(subprocess.Popen(argv, env={**os.environ, **env}, stdout=subprocess.PIPE if r == 0 else sys.stderr))
    worst, pending = 0, set(range(n))
    out0 = b""
    while pending:
        for r in sorted(pending):
            p = procs[r]
            try:
                if r == 0:
                    o, _ = p.communicate(timeout=0.5)
                    out0 += o or b""
                else:
                    p.wait(timeout=0.5)
            except subprocess.TimeoutExpired:
                continue
            pending.discard(r)
            if p.returncode != 0:
                worst = worst or (p.returncode if p.returncode > 0 else 1)
                for q in pending:                   # a rank that died would leave the others in the barrier: end them (exact PIDs)
                    procs[q].terminate()
    sys.stdout.write(out0.decode(errors="replace"))
    sys.stdout.flush()
    return worst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=FRAMES_PER_GPU, help="frames per GPU (default: configs[1])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--spawn-dry-run", action="store_true",
                    help="print the per-rank command lines / environments `--gpus N` would start, and exit (no GPU call)")
    ap.add_argument("--allow-gather-fallback", action="store_true",
                    help="N > 1: if the C-ABI RCCL size gather cannot be set up, exchange the sizes through torch.distributed instead of failing")
    ap.add_argument("--headline-only", action="store_true",
                    help="skip the informative legs (configs[3], noisy_u16, index decode): profiler passes see the headline kernels only")
    args = ap.parse_args()

    # `python bench.py --gpus N` without a launcher: this process becomes the launcher (it never touches the GPU)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, args.spawn_dry_run))
    if args.spawn_dry_run:
        sys.exit(spawn_ranks(args.gpus, True))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world} (launcher and flag disagree)")
    if "TRPX_BENCH_SPAWN_SELFTEST" in os.environ:       # tests/test_sharded.py: the launcher's relay / exit-code logic, no GPU
        if os.environ["TRPX_BENCH_SPAWN_SELFTEST"] == f"fail:{rank}":
            sys.exit(3)
        if rank == 0:
            print(json.dumps({"selftest": True, "n_gpus": world, "master": os.environ["MASTER_ADDR"]}))
        sys.exit(0)
    # stdout carries the ONE JSON line and nothing else: native libraries print banners to file descriptor 1 (RCCL: "RCCL
    # version : ..." at communicator creation), so fd 1 points at stderr while the run lasts and the line goes to the saved one
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # TRPX_BENCH_FORCE_DIST=1: run the RCCL code path (init, size gather, barrier) even with one rank (self test)
    use_dist = world > 1 or os.environ.get("TRPX_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)

    from trpx_amd import codec, sharded, _lib
    L = _lib.lib()

    frames = args.frames
    frame0 = rank * frames                                   # C5: GPU g <- frames [2000g, 2000g+2000)
    px = codec.synth(np.uint16, frame0, frames, N_VALUES, device=dev)
    # one workspace per direction: the encoder's stays the library's between calls (trpx_hip.h, "Workspaces between calls"),
    # so its descriptor words need no clearing launch in front of every encode
    ws = codec.Workspace(dev)
    ws_d = codec.Workspace(dev)
    cap = (frames * codec.worst_case_bytes(np.uint16, N_VALUES) + 15) // 16 * 16
    out = torch.empty(cap, dtype=torch.uint8, device=dev)
    offs = torch.empty(frames + 1, dtype=torch.int64, device=dev)
    st_e = torch.empty(8, dtype=torch.int32, device=dev)
    st_d = torch.empty(8, dtype=torch.int32, device=dev)
    back = torch.empty((frames, N_VALUES), dtype=torch.uint16, device=dev)
    # size the workspace once (never allocate inside the timed region)
    ws.get(L.trpx_encode_workspace_bytes(_lib.U16, N_VALUES, frames, 12))
    ws_d.get(L.trpx_decode_workspace_bytes(_lib.U16, N_VALUES, frames, 12))
    torch.cuda.synchronize()

    # The per-frame size gather (RCCL over xGMI -> global byte offset of every frame) depends only on the encode and
    # nothing in the decode depends on it: it runs on its own stream next to the decode and is joined at the step's end.
    # (C ABI: trpx_gather_frame_offsets = pack kernel + ncclAllGather + scan kernel on the communicator below)
    gather, gather_kind = None, None
    if use_dist:
        # the C-ABI gather on its own RCCL communicator; if ANY rank cannot set it up (agreed on collectively, so that no rank
        # waits in a collective the others never enter), every rank takes the same exchange through torch.distributed instead
        try:
            gather, ok = sharded.ShardedCodec(frames, N_VALUES, np.uint16, dev), 1
        except Exception as ex:
            print(f"bench.py rank {rank}: C-ABI RCCL gather unavailable ({ex!r})", file=sys.stderr)
            gather, ok = None, 0
        flag = torch.tensor([ok], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            # ONE C-ABI call per step and rank: trpx_encode_sharded = trpx_encode of the rank's frames + pack kernel + ncclAllGather +
            # scan kernel, the gather on the second stream; the codec object owns the rank's stack, offsets and status
            gather_kind = "trpx_encode_sharded (C ABI: trpx_encode + pack kernel + ncclAllGather + scan kernel in one call, the gather on a second stream)"
            out, offs, st_e = gather.out, gather.local_offsets, gather.status
        else:
            if gather is not None:
                gather.close()
            if not args.allow_gather_fallback:
                # the metric names RCCL's size gather: a run that silently measured another exchange would not be that metric
                sys.exit(f"bench.py rank {rank}: the C-ABI RCCL size gather could not be set up on every rank (see stderr); "
                         "--allow-gather-fallback exchanges the sizes through torch.distributed instead")
            tg = sharded.SizeGather(frames, dev)
            gather = lambda o, st: tg(o, st[1:2])                                     # noqa: E731
            gather.close = lambda: None
            gather_kind = "torch.distributed all_gather_into_tensor (fallback)"
    comm_stream = torch.cuda.Stream(device=dev) if use_dist else None

    one_call = use_dist and isinstance(gather, sharded.ShardedCodec)

    def step():
        if one_call:
            enc = gather.encode(px, gather_stream=comm_stream)
        else:
            enc = codec.encode(px, out=out, workspace=ws, frame_offsets=offs, status=st_e)
            if use_dist:
                cur = torch.cuda.current_stream()
                comm_stream.wait_stream(cur)
                with torch.cuda.stream(comm_stream):
                    gather(offs, st_e)
        # decode straight from the device-resident stack (bounded by its worst-case capacity; the
        # frame offsets tell the kernels where every frame ends -- no host sync inside the step)
        codec.decode(out, offs, N_VALUES, frames, np.uint16, out=back, workspace=ws_d, status=st_d)
        if use_dist:
            torch.cuda.current_stream().wait_stream(comm_stream)
        return enc

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        enc = step()
    barrier()
    elapsed = time.perf_counter() - t0
    rank_fps_min = rank_fps_max = rccl_ranks = None
    if use_dist:
        mine = args.steps * frames / elapsed                 # this rank's own frames/s over its own clock
        t = torch.tensor([elapsed, -mine, mine], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, rank_fps_min, rank_fps_max = float(t[0].item()), -float(t[1].item()), float(t[2].item())
        # how many ranks the size gather's communicator spans, as RCCL itself reports it (ncclCommCount), agreed over all ranks
        info = list(gather.rccl_info()) if one_call else [0, rank]
        ri = torch.tensor([info[0], -info[0], int(info[1] == rank)], dtype=torch.int32, device=dev)
        dist.all_reduce(ri, op=dist.ReduceOp.MIN)
        rccl_ranks = int(ri[0].item())
        if one_call:
            assert rccl_ranks == world and -int(ri[1].item()) == world and int(ri[2].item()) == 1, \
                f"RCCL communicator spans {info[0]} ranks (rank {info[1]}) but the job has {world} (rank {rank})"

    # ---- correctness of what was just timed (every rank) -----------------------------------
    assert int(st_e[0].item()) == 0 and int(st_d[0].item()) == 0, "device status reports an error"
    assert torch.equal(back.view(torch.int16), px.view(torch.int16)), "round trip is not pixel-identical"
    total_bytes = int(offs[-1].item())
    if rank == 0 and frames == FRAMES_PER_GPU:
        assert total_bytes == 203596114, "stack size differs from the reference's (SURVEY.md 8 row d)"
    if use_dist:                                             # what the step's gather computed: this rank's place in the global stack
        goffs, gbase, gpb = gather.encode(px) if one_call else gather(offs, st_e)
        torch.cuda.synchronize()
        assert int(goffs[rank * frames + frames] - goffs[rank * frames]) == total_bytes and int(gbase) == int(goffs[rank * frames])
        assert int(gpb) >= int(st_e[1].item())
        if one_call:                                         # ... and the rank's frames expanded from the GLOBAL table (trpx_decode_sharded)
            back.zero_()
            _, sd = gather.decode(back)
            torch.cuda.synchronize()
            assert int(sd[0].item()) == 0 and torch.equal(back.view(torch.int16), px.view(torch.int16)), "trpx_decode_sharded differs"

    # ---- the encoded stack against the CPU oracle (a sample of what was just timed; rank 0) ----
    oracle_check = None
    if rank == 0:
        from oracle import oracle as O
        sample = [0, 1, frames // 2, frames - 1] if frames >= 4 else list(range(frames))
        o_host, s_host = offs.cpu().numpy(), out[: total_bytes].cpu().numpy()
        for f in sample:
            want = O.encode(px[f].cpu().numpy())[0]
            got = s_host[int(o_host[f]): int(o_host[f + 1])]
            assert got.size == want.size and (got == want).all(), f"frame {f} differs from the CPU oracle"
        oracle_check = f"frames {sample} byte-identical to the CPU oracle's encode; all {frames} frames round-trip pixel-identical"

    # ---- separate encode-only / decode-only rates + per-kernel HIP-event timing (rank 0) ----
    detail = {}
    if rank == 0:
        reps = max(5, min(args.steps, 20))

        def timed(fn, n=reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            fn()                                              # (a kernel's first launch in a process loads its code)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(n):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / n

        enc_ms = timed(lambda: codec.encode(px, out=out, workspace=ws, frame_offsets=offs, status=st_e))
        dec_ms = timed(lambda: codec.decode(out, offs, N_VALUES, frames, np.uint16, out=back, workspace=ws_d, status=st_d))
        # walk-free decode with the encoder's optional decode index (not part of `value`)
        dec_idx_ms = float("nan")
        if not args.headline_only:
            enc_i = codec.encode(px, out=out, workspace=ws, frame_offsets=offs, status=st_e, index=True)
            dec_idx_ms = timed(lambda: codec.decode(out, offs, N_VALUES, frames, np.uint16, out=back, status=st_d, index=enc_i.index))
            assert int(st_d[0].item()) == 0 and torch.equal(back.view(torch.int16), px.view(torch.int16))

        # per-kernel durations: HIP events recorded by the library on the launch stream
        def stages(enc_fn, dec_fn, dec_names, n=reps):
            L.trpx_profile_enable(1)
            acc = {}
            buf = (C.c_float * 8)()
            for _ in range(n):
                enc_fn()
                k = L.trpx_profile_read(buf, 8)
                for i, name in enumerate((ENC_STAGES_FUSED if k == 3 else ENC_STAGES_TWOPASS)[:k]):
                    acc.setdefault(name, []).append(buf[i])
                dec_fn()
                k = L.trpx_profile_read(buf, 8)
                for i, name in enumerate(dec_names[:k]):
                    acc.setdefault(name, []).append(buf[i])
            L.trpx_profile_enable(0)
            stages.samples = acc                               # (roof() reports the spread of the launches next to their mean)
            return {k: float(np.mean(v)) for k, v in acc.items()}

        stage_ms = stages(lambda: codec.encode(px, out=out, workspace=ws, frame_offsets=offs, status=st_e),
                          lambda: codec.decode(out, offs, N_VALUES, frames, np.uint16, out=back, workspace=ws_d, status=st_d),
                          DEC_STAGES_FRAMES)
        pix_bytes = frames * N_VALUES * 2
        alg_bytes = pix_bytes + total_bytes                    # B_enc = B_dec = N*sizeof(T) + S_f per frame

        # HBM bytes per launch from the TCC counters (separate rocprofv3 --pmc passes, tools/pmc_traffic.sh, corrected as
        # MI355X_MICROARCH.md prescribes) and the rocprofv3 --kernel-trace --stats average of the same kernels: both
        # come from the committed profiles/ files and are only quoted while those were taken at this source state
        prof = {}
        try:
            prof = json.load(open(os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_traffic.json")))
            if prof.get("kernel_sources_sha16") != kernel_sources_sha16():
                prof = {"stale": f"profiles/{PROFILE_TAG}_traffic.json was taken at kernel sources {prof.get('kernel_sources_sha16')}"}
        except (OSError, ValueError):
            pass

        head_samples = dict(stages.samples)

        def roof(kernel, ms, stage=None):
            r = {"bound": "hbm", "kernel": kernel, "achieved": alg_bytes / ms / 1e6, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                 "frac": alg_bytes / ms / 1e6 / HBM_PEAK_GBPS, "traffic": None, "algorithmic_bytes_per_launch": alg_bytes,
                 "avg_launch_ms": ms}
            v = head_samples.get(stage)
            if v:                                              # the launches of the timed round trips: how far they spread
                r.update({"launches": len(v), "launch_ms_std": float(np.std(v)), "launch_ms_min": float(np.min(v)),
                          "launch_ms_max": float(np.max(v))})
            k = prof.get(kernel)
            if k and frames == FRAMES_PER_GPU:
                r["traffic"] = k.get("traffic_bytes")
                r["rocprof_avg_launch_ms"] = k.get("rocprof_avg_ms")
                r["profile_source"] = prof.get("source")
            elif "stale" in prof:
                r["profile_source"] = prof["stale"]
            return r

        ekey = "encode_fused" if "encode_fused" in stage_ms else "pack"
        roofs = {"encode": roof({"encode_fused": "k_encode_fused<uint16_t>", "pack": "k_pack<uint16_t>"}[ekey], stage_ms[ekey], ekey)}
        if "decode_frames" in stage_ms:
            roofs["decode"] = roof("k_decode_frames<uint16_t>", stage_ms["decode_frames"], "decode_frames")
        elif "unpack" in stage_ms:
            roofs["decode"] = roof("k_unpack_tiles<uint16_t>", stage_ms["unpack"], "unpack")

        # configs[3] (informative, not part of `value`): 4096x4096 int32 frames with sparse peaks, 1 GPU
        c4 = {}
        try:
            if args.headline_only:
                raise RuntimeError("skipped (--headline-only)")
            n4, f4 = 4096 * 4096, 8                          # C4 = frames 0..7 (537 MB)
            px4 = codec.synth(np.int32, 0, f4, n4, device=dev)
            e4 = codec.encode(px4, index=True)
            torch.cuda.synchronize()
            e4.check()
            b4 = torch.empty_like(px4)
            s4 = torch.empty(8, dtype=torch.int32, device=dev)
            t_e = timed(lambda: codec.encode(px4, out=e4.data, frame_offsets=e4.frame_offsets, status=e4.status, workspace=ws), 5)
            t_d = timed(lambda: codec.decode(e4.data, e4.frame_offsets, n4, f4, np.int32, out=b4, status=s4, index=e4.index), 5)
            assert int(s4[0].item()) == 0 and torch.equal(b4, px4)
            b4.zero_()
            t_p = timed(lambda: codec.decode(e4.data, e4.frame_offsets, n4, f4, np.int32, out=b4, status=s4, workspace=ws_d), 5)
            assert int(s4[0].item()) == 0 and torch.equal(b4, px4)
            st4 = stages(lambda: codec.encode(px4, out=e4.data, frame_offsets=e4.frame_offsets, status=e4.status, workspace=ws),
                         lambda: codec.decode(e4.data, e4.frame_offsets, n4, f4, np.int32, out=b4, status=s4, workspace=ws_d), dec_stage_names(n4), 5)
            alg4 = f4 * n4 * 4 + e4.total_bytes()
            c4 = {"workload": f"{f4} frames 4096x4096 int32 synth-v1 (bg -3..3 + sparse peaks < 2^24)",
                  "encode_fps": f4 / t_e * 1e3, "encode_pixel_GBps": f4 * n4 * 4 / t_e / 1e6,
                  "decode_fps": f4 / t_p * 1e3, "decode_pixel_GBps": f4 * n4 * 4 / t_p / 1e6,
                  "decode_with_index_fps": f4 / t_d * 1e3, "decode_with_index_pixel_GBps": f4 * n4 * 4 / t_d / 1e6,
                  "encode_ms": t_e, "decode_ms": t_p, "decode_with_index_ms": t_d, "algorithmic_bytes": alg4,
                  "decode_frac_of_hbm_peak": alg4 / t_p / 1e6 / HBM_PEAK_GBPS,
                  "decode_with_index_frac_of_hbm_peak": alg4 / t_d / 1e6 / HBM_PEAK_GBPS,
                  "kernel_ms": st4,
                  "roofline_encode": {"bound": "hbm", "kernel": "k_encode_fused<int32_t>", "achieved": alg4 / st4["encode_fused"] / 1e6,
                                      "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": alg4 / st4["encode_fused"] / 1e6 / HBM_PEAK_GBPS,
                                      "algorithmic_bytes_per_launch": alg4, "avg_launch_ms": st4["encode_fused"]} if "encode_fused" in st4 else None,
                  "compressed_bytes": e4.total_bytes(), "prolix_bits": e4.prolix_bits(), "roundtrip_exact": True}
            del px4, e4, b4
        except Exception as ex:      # informative leg only
            c4 = {"error": repr(ex)}

        # ---- measured memory ceilings of THIS box (SURVEY.md 8 row d: fraction of nominal AND of measured-achievable) ----
        def stream_GBps(mode, nbytes):
            src = px.view(torch.uint8).reshape(-1)[:nbytes]
            dst = back.view(torch.uint8).reshape(-1)[:nbytes]
            sink = st_d if mode == 0 else dst
            if mode == 3:                                     # from a 2-byte aligned address: the decoder's stores on frames that start inside a cache line
                sink = back.view(torch.uint8).reshape(-1)[2: 2 + nbytes]
            s_ = torch.cuda.current_stream().cuda_stream
            ms = timed(lambda: _lib.check(L.trpx_bench_stream(mode, src.data_ptr(), sink.data_ptr(), nbytes, s_)), 10)
            return (2 if mode == 2 else 1) * nbytes / ms / 1e6
        measured = {"read_GBps": stream_GBps(0, pix_bytes), "write_GBps": stream_GBps(1, pix_bytes), "copy_GBps": stream_GBps(2, pix_bytes),
                    "write_misaligned_GBps": stream_GBps(3, pix_bytes - 16),
                    "how": "trpx_bench_stream: grid-stride 16 B/lane non-temporal kernels over the stack's pixel bytes, HIP events, 10 launches; "
                           "write_misaligned: the same stores from a 2-byte aligned address"}
        # (The write rates differ between the pool's boxes -- 4.8-6.7 / 3.5-5.0 TB/s over nine of them -- and the kernels that keep
        # plain misaligned stores (`oddsize_u16` with the index) were fast only on the two fastest: DESIGN.md 8, tools/box_kind.py.
        # The RATIO of the two rates hardly moves, 1.34-1.47, and tells nothing: it is not reported.)
        # (what the decoder left in `back` was overwritten: restore the headline result for the checks below)
        codec.decode(out, offs, N_VALUES, frames, np.uint16, out=back, workspace=ws_d, status=st_d)
        for r_, kind in ((roofs.get("encode"), "read"), (roofs.get("decode"), "write")):
            if r_:
                r_["peak_measured"] = measured[kind + "_GBps"]
                r_["peak_measured_kind"] = f"{kind} stream ({'pixel loads' if kind == 'read' else 'pixel stores'} are 84 % of this kernel's bytes)"
                r_["frac_of_measured"] = r_["achieved"] / measured[kind + "_GBps"]

        # ---- informative legs (not part of `value`): the workloads the headline stack hides ---------------------------------
        # Each leg: encode, index-free decode, decode with the encoder's decode index, per-kernel HIP-event times, frame 0
        # byte-identical to the CPU oracle, every frame round-trips.  Fractions are of the 8 TB/s nominal peak on ALGORITHMIC
        # bytes (pixels + stream), SURVEY.md 8d.
        def leg(pxl, what):
            nf, nv = pxl.shape[0], pxl[0].numel()
            bk = torch.empty_like(pxl)
            en = codec.encode(pxl, out=out, workspace=ws, frame_offsets=offs[: nf + 1], status=st_e)
            torch.cuda.synchronize()
            en.check()
            nbytes = en.total_bytes()
            want = O.encode(pxl[0].cpu().numpy().reshape(-1))[0]
            assert (out[: want.size].cpu().numpy() == want).all(), f"{what}: frame 0 differs from the CPU oracle"
            fo = offs[: nf + 1]
            t_e = timed(lambda: codec.encode(pxl, out=out, workspace=ws, frame_offsets=fo, status=st_e))
            t_d = timed(lambda: codec.decode(out, fo, nv, nf, np.uint16, out=bk, workspace=ws_d, status=st_d))
            assert int(st_d[0].item()) == 0 and torch.equal(bk.view(torch.int16), pxl.view(torch.int16)), f"{what}: round trip differs"
            bk.zero_()
            en_i = codec.encode(pxl, out=out, workspace=ws, frame_offsets=fo, status=st_e, index=True)
            t_i = timed(lambda: codec.decode(out, fo, nv, nf, np.uint16, out=bk, status=st_d, index=en_i.index))
            assert int(st_d[0].item()) == 0 and torch.equal(bk.view(torch.int16), pxl.view(torch.int16)), f"{what}: indexed decode differs"
            km = stages(lambda: codec.encode(pxl, out=out, workspace=ws, frame_offsets=fo, status=st_e),
                        lambda: codec.decode(out, fo, nv, nf, np.uint16, out=bk, workspace=ws_d, status=st_d), dec_stage_names(nv), 5)
            pb = nf * nv * 2
            alg = pb + nbytes
            return {"workload": what, "frames": nf, "n_values": nv,
                    "encode_ms": t_e, "decode_ms": t_d, "decode_with_index_ms": t_i,
                    "encode_fps": nf / t_e * 1e3, "decode_fps": nf / t_d * 1e3, "decode_with_index_fps": nf / t_i * 1e3,
                    "algorithmic_bytes": alg,
                    "encode_frac_of_hbm_peak": alg / t_e / 1e6 / HBM_PEAK_GBPS,
                    "decode_algorithmic_GBps": alg / t_d / 1e6,
                    "decode_frac_of_hbm_peak": alg / t_d / 1e6 / HBM_PEAK_GBPS,
                    "decode_with_index_frac_of_hbm_peak": alg / t_i / 1e6 / HBM_PEAK_GBPS,
                    "decode_frac_of_measured_write": alg / t_d / 1e6 / measured["write_GBps"],
                    "kernel_ms": km, "compression_ratio": nbytes / pb, "roundtrip_exact": True}

        def run_leg(make, what):
            if args.headline_only:
                return {"error": "skipped (--headline-only)"}
            try:
                pxl = make()
                r = leg(pxl, what)
                del pxl
                return r
            except Exception as ex:      # informative legs only
                return {"error": repr(ex)}

        def make_noisy():
            # Poisson(1.5) background clamped to 0..6 + 1/4096 peaks < 4000 (the generator of tools/dtype_time.py): the block width
            # flips between 2 and 3 bits from block to block -- an explicit header every other block, the worst case for the
            # header chain (Terse.hpp:360-372)
            g = torch.Generator(device=dev)
            g.manual_seed(1)
            bg = torch.poisson(torch.full((frames, N_VALUES), 1.5, device=dev), generator=g).clamp_(0, 6).to(torch.int32)
            hot = torch.rand((frames, N_VALUES), device=dev, generator=g) < (1.0 / 4096)
            pxn = torch.where(hot, torch.randint(0, 4000, (frames, N_VALUES), device=dev, generator=g, dtype=torch.int32), bg)
            return pxn.to(torch.int16).view(torch.uint16)

        from trpx_amd import workloads
        noisy = run_leg(make_noisy, f"{frames} frames 512x512 uint16, Poisson(1.5) background 0..6 + 1/4096 peaks "
                                    f"(block width changes every ~2 blocks)")
        # BASELINE.md section 2's first anchor, the reference's own use case ("diffraction data", README.md:10): Poisson(3)
        # background + 1/4096 12-bit peaks; the width changes on ~25 % of the blocks.  Counter-based (trpx_amd/workloads.py).
        poisson3 = run_leg(lambda: workloads.poisson_u16(3.0, 0, frames, N_VALUES, device=dev),
                           f"{frames} frames 512x512 uint16, counter-based Poisson(3) + 1/4096 12-bit peaks (width changes on ~25 % of the blocks)")
        # a detector-shaped stack: 1030 x 1065 pixels (no multiple of 4 or of a cache line), 200 frames of 91 413 blocks each
        mid_f = max(1, frames // 10)
        midsize = run_leg(lambda: codec.synth(np.uint16, 0, mid_f, 1030 * 1065, device=dev),
                          f"{mid_f} frames 1030x1065 uint16 synth-v1 (mid-size frames, odd pixel count)")
        midsize_p3 = run_leg(lambda: workloads.poisson_u16(3.0, 0, mid_f, 1030 * 1065, device=dev),
                             f"{mid_f} frames 1030x1065 uint16, counter-based Poisson(3) + 1/4096 12-bit peaks")
        # frames whose byte size is no multiple of 128: every frame starts inside a cache line
        oddsize = run_leg(lambda: codec.synth(np.uint16, 0, frames, 513 * 511, device=dev),
                          f"{frames} frames 513x511 uint16 synth-v1 (frame size no multiple of a 128-byte line)")

        detail = {
            "encode_ms": enc_ms, "decode_ms": dec_ms,
            "encode_fps": frames / enc_ms * 1e3, "decode_fps": frames / dec_ms * 1e3,
            "encode_pixel_GBps": pix_bytes / enc_ms / 1e6, "decode_pixel_GBps": pix_bytes / dec_ms / 1e6,
            "encode_algorithmic_GBps": alg_bytes / enc_ms / 1e6,
            "decode_algorithmic_GBps": alg_bytes / dec_ms / 1e6,
            "encode_pixel_frac_of_hbm_peak": pix_bytes / enc_ms / 1e6 / HBM_PEAK_GBPS,
            "decode_pixel_frac_of_hbm_peak": pix_bytes / dec_ms / 1e6 / HBM_PEAK_GBPS,
            "decode_with_index_ms": dec_idx_ms, "decode_with_index_fps": frames / dec_idx_ms * 1e3,
            "decode_with_index_frac_of_hbm_peak": alg_bytes / dec_idx_ms / 1e6 / HBM_PEAK_GBPS,
            "config3_4096x4096_int32": c4, "noisy_u16": noisy, "poisson3_u16": poisson3, "midsize_u16": midsize,
            "midsize_poisson3_u16": midsize_p3, "oddsize_u16": oddsize, "peak_measured": measured,
            "kernel_ms": stage_ms, "compressed_bytes_per_gpu": total_bytes,
            "compression_ratio": total_bytes / pix_bytes, "oracle_check": oracle_check,
        }
        # `roofline` describes the kernel with the largest per-step time; the other side of the step rides along
        dominant = max(roofs, key=lambda k: roofs[k]["avg_launch_ms"])
        roofline = roofs[dominant]
        for k, v in roofs.items():
            if k != dominant:
                detail[f"roofline_{k}"] = v

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        result = {
            "metric": "frames/s (Terse encode + Prolix decode round trip, 512x512 uint16 stack, bit-exact vs CPU ref)",   # see oracle_check
            "value": world * frames * args.steps / elapsed, "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u16",
            "data": "synthetic (synth-v1 counter-based frames, SURVEY.md 8 row d)",
            "config": {"workload": f"{frames}-frame 512x512 uint16 synth-v1 stack per GPU: Terse encode "
                                   f"(configs[1]) + Prolix decode (configs[2])",
                       "frames_per_gpu": frames, "n_values": N_VALUES, "block": 12,
                       "parallelism": f"frames sharded {world}-way, RCCL all-gather of per-frame sizes",
                       "size_gather": gather_kind,
                       # self-verification of a sharded run (N > 1): which exchange was timed, how many ranks RCCL's communicator
                       # spans (ncclCommCount through trpx_comm_info, agreed over all ranks; 0: not the C-ABI path), and the
                       # slowest / fastest rank's own frames/s
                       "size_gather_path": (None if not use_dist else "c_abi_rccl" if one_call else "torch_distributed_fallback"),
                       "rccl_ranks": rccl_ranks, "rank_frames_per_s_min": rank_fps_min, "rank_frames_per_s_max": rank_fps_max},
            "roundtrip_GBps_pixels": world * frames * N_VALUES * 2 * 2 * args.steps / elapsed / 1e9,
            "roofline": roofline,
        }
        result.update(detail)
        if world == 1 and not args.no_cpu_baseline:
            cores = host_cores()
            result["cpu_baseline"] = cpu_baseline(px.cpu().numpy(), cores)
        os.write(json_fd, (json.dumps(result) + "\n").encode())
    if use_dist:
        gather.close()
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
