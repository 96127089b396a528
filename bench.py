#!/usr/bin/env python3
"""bench.py -- headline benchmark of the TERSE/PROLIX hot path on MI355X.

A "step" = one pass of the hot path over one batch: Terse-encode the resident 2000-frame
512x512 uint16 synth-v1 stack (BASELINE.json configs[1]), gather the per-frame sizes across ranks
(RCCL, N>1 only), Prolix-decode it again (configs[2]).  Inputs are resident in HBM before the timed
region; nothing is cached between steps (outputs are re-produced every step).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

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
DEC_STAGES = ["walk", "unpack"]        # tiled decode (two kernels)
DEC_STAGES_FRAMES = ["decode_frames"]  # one workgroup per frame (walk + extraction fused)


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=FRAMES_PER_GPU, help="frames per GPU (default: configs[1])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N "
                     "--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")
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
    ws = codec.Workspace(dev)
    cap = (frames * codec.worst_case_bytes(np.uint16, N_VALUES) + 15) // 16 * 16
    out = torch.empty(cap, dtype=torch.uint8, device=dev)
    offs = torch.empty(frames + 1, dtype=torch.int64, device=dev)
    st_e = torch.empty(8, dtype=torch.int32, device=dev)
    st_d = torch.empty(8, dtype=torch.int32, device=dev)
    back = torch.empty((frames, N_VALUES), dtype=torch.uint16, device=dev)
    # size the workspace once (never allocate inside the timed region)
    ws.get(max(L.trpx_encode_workspace_bytes(_lib.U16, N_VALUES, frames, 12),
               L.trpx_decode_workspace_bytes(_lib.U16, N_VALUES, frames, 12)))
    torch.cuda.synchronize()

    # The per-frame size gather (RCCL over xGMI -> global byte offset of every frame) depends only on the encode and
    # nothing in the decode depends on it: it runs on its own stream next to the decode and is joined at the step's end.
    gather = sharded.SizeGather(frames, dev) if use_dist else None
    comm_stream = torch.cuda.Stream(device=dev) if use_dist else None

    def step():
        enc = codec.encode(px, out=out, workspace=ws, frame_offsets=offs, status=st_e)
        if use_dist:
            cur = torch.cuda.current_stream()
            comm_stream.wait_stream(cur)
            with torch.cuda.stream(comm_stream):
                gather(offs, st_e[1:2])
        # decode straight from the device-resident stack (bounded by its worst-case capacity; the
        # frame offsets tell the kernels where every frame ends -- no host sync inside the step)
        codec.decode(out, offs, N_VALUES, frames, np.uint16, out=back, workspace=ws, status=st_d)
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
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- correctness of what was just timed (every rank) -----------------------------------
    assert int(st_e[0].item()) == 0 and int(st_d[0].item()) == 0, "device status reports an error"
    assert torch.equal(back.view(torch.int16), px.view(torch.int16)), "round trip is not pixel-identical"
    total_bytes = int(offs[-1].item())
    if rank == 0 and frames == FRAMES_PER_GPU:
        assert total_bytes == 203596114, "stack size differs from the reference's (SURVEY.md 8 row d)"

    # ---- separate encode-only / decode-only rates + per-kernel HIP-event timing (rank 0) ----
    detail = {}
    if rank == 0:
        reps = max(5, min(args.steps, 20))
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        torch.cuda.synchronize()
        ev[0].record()
        for _ in range(reps):
            codec.encode(px, out=out, workspace=ws, frame_offsets=offs, status=st_e)
        ev[1].record()
        for _ in range(reps):
            codec.decode(out, offs, N_VALUES, frames, np.uint16, out=back, workspace=ws, status=st_d)
        ev[2].record()
        torch.cuda.synchronize()
        enc_ms = ev[0].elapsed_time(ev[1]) / reps
        dec_ms = ev[1].elapsed_time(ev[2]) / reps
        # walk-free decode with the encoder's optional decode index (not part of `value`)
        enc_i = codec.encode(px, out=out, workspace=ws, frame_offsets=offs, status=st_e, index=True)
        torch.cuda.synchronize()
        ev2 = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev2[0].record()
        for _ in range(reps):
            codec.decode(out, offs, N_VALUES, frames, np.uint16, out=back, status=st_d, index=enc_i.index)
        ev2[1].record()
        torch.cuda.synchronize()
        dec_idx_ms = ev2[0].elapsed_time(ev2[1]) / reps
        assert int(st_d[0].item()) == 0 and torch.equal(back.view(torch.int16), px.view(torch.int16))
        # configs[3] (informative, not part of `value`): 4096x4096 int32 frames with sparse peaks, 1 GPU
        c4 = {}
        try:
            n4, f4 = 4096 * 4096, 4
            px4 = codec.synth(np.int32, 0, f4, n4, device=dev)
            e4 = codec.encode(px4, index=True)
            torch.cuda.synchronize()
            e4.check()
            ev4 = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            ev4[0].record()
            for _ in range(5):
                codec.encode(px4, out=e4.data, frame_offsets=e4.frame_offsets, status=e4.status, workspace=ws)
            ev4[1].record()
            for _ in range(5):
                b4, s4 = codec.decode(e4.data, e4.frame_offsets, n4, f4, np.int32, index=e4.index)
            ev4[2].record()
            torch.cuda.synchronize()
            assert int(s4[0].item()) == 0 and torch.equal(b4, px4)
            t_e, t_d = ev4[0].elapsed_time(ev4[1]) / 5, ev4[1].elapsed_time(ev4[2]) / 5
            c4 = {"workload": f"{f4} frames 4096x4096 int32 synth-v1 (bg -3..3 + sparse peaks < 2^24)",
                  "encode_fps": f4 / t_e * 1e3, "encode_pixel_GBps": f4 * n4 * 4 / t_e / 1e6,
                  "decode_with_index_fps": f4 / t_d * 1e3, "decode_with_index_pixel_GBps": f4 * n4 * 4 / t_d / 1e6,
                  "compressed_bytes": e4.total_bytes(), "prolix_bits": e4.prolix_bits(), "roundtrip_exact": True}
            del px4, e4, b4
        except Exception as ex:      # informative leg only
            c4 = {"error": repr(ex)}
        # per-kernel durations: HIP events recorded by the library on the launch stream
        L.trpx_profile_enable(1)
        stage = {n: [] for n in ENC_STAGES_TWOPASS + ENC_STAGES_FUSED + DEC_STAGES + DEC_STAGES_FRAMES}
        buf = (C.c_float * 8)()
        for _ in range(reps):
            codec.encode(px, out=out, workspace=ws, frame_offsets=offs, status=st_e)
            n = L.trpx_profile_read(buf, 8)
            names = ENC_STAGES_FUSED if n == 3 else ENC_STAGES_TWOPASS
            for k in range(n):
                stage[names[k]].append(buf[k])
            codec.decode(out, offs, N_VALUES, frames, np.uint16, out=back, workspace=ws, status=st_d)
            n = L.trpx_profile_read(buf, 8)
            dnames = DEC_STAGES_FRAMES if n == 1 else DEC_STAGES
            for k in range(n):
                stage[dnames[k]].append(buf[k])
        L.trpx_profile_enable(0)
        stage_ms = {k: float(np.mean(v)) for k, v in stage.items() if v}
        pix_bytes = frames * N_VALUES * 2
        detail = {
            "encode_ms": enc_ms, "decode_ms": dec_ms,
            "encode_fps": frames / enc_ms * 1e3, "decode_fps": frames / dec_ms * 1e3,
            "encode_pixel_GBps": pix_bytes / enc_ms / 1e6, "decode_pixel_GBps": pix_bytes / dec_ms / 1e6,
            "encode_algorithmic_GBps": (pix_bytes + total_bytes) / enc_ms / 1e6,
            "decode_algorithmic_GBps": (pix_bytes + total_bytes) / dec_ms / 1e6,
            "encode_pixel_frac_of_hbm_peak": pix_bytes / enc_ms / 1e6 / HBM_PEAK_GBPS,
            "decode_pixel_frac_of_hbm_peak": pix_bytes / dec_ms / 1e6 / HBM_PEAK_GBPS,
            "decode_with_index_ms": dec_idx_ms, "decode_with_index_fps": frames / dec_idx_ms * 1e3,
            "config3_4096x4096_int32": c4,
            "kernel_ms": stage_ms, "compressed_bytes_per_gpu": total_bytes,
            "compression_ratio": total_bytes / pix_bytes,
        }
        # dominant kernel of the encode: reads every pixel once, writes every stream byte once
        kname = "encode_fused" if "encode_fused" in stage_ms else "pack"
        pack_ms = stage_ms[kname]
        alg_bytes = pix_bytes + total_bytes                    # B_enc = N*sizeof(T) + S_f per frame
        roofline = {"bound": "hbm", "kernel": {"encode_fused": "k_encode_fused<uint16_t>", "pack": "k_pack<uint16_t>"}[kname],
                    "achieved": alg_bytes / pack_ms / 1e6,
                    "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": alg_bytes / pack_ms / 1e6 / HBM_PEAK_GBPS,
                    "traffic": None, "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": pack_ms}
        # HBM bytes of that kernel from the TCC counters (collected in separate rocprofv3 --pmc passes with
        # tools/pmc_traffic.sh and corrected as MI355X_MICROARCH.md prescribes; see profiles/r01_traffic.json)
        try:
            tr = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
            if frames == FRAMES_PER_GPU and roofline["kernel"] in tr:
                roofline["traffic"] = tr[roofline["kernel"]]["traffic_bytes"]
                roofline["traffic_source"] = tr["source"]
        except (OSError, ValueError, KeyError):
            pass

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        result = {
            "metric": "frames/s (Terse encode + Prolix decode round trip, 512x512 uint16 stack, bit-exact vs CPU ref)",
            "value": world * frames * args.steps / elapsed, "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u16",
            "data": "synthetic (synth-v1 counter-based frames, SURVEY.md 8 row d)",
            "config": {"workload": f"{frames}-frame 512x512 uint16 synth-v1 stack per GPU: Terse encode "
                                   f"(configs[1]) + Prolix decode (configs[2])",
                       "frames_per_gpu": frames, "n_values": N_VALUES, "block": 12,
                       "parallelism": f"frames sharded {world}-way, RCCL all-gather of per-frame sizes"},
            "roundtrip_GBps_pixels": world * frames * N_VALUES * 2 * 2 * args.steps / elapsed / 1e9,
            "roofline": roofline,
        }
        result.update(detail)
        if world == 1 and not args.no_cpu_baseline:
            cores = host_cores()
            result["cpu_baseline"] = cpu_baseline(px.cpu().numpy(), cores)
        print(json.dumps(result))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
