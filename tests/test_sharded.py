"""CPU test of the N>1 path: world_size-2 gloo processes run the size gather / global offset
logic of trpx_amd.sharded on (oracle-encoded) shards and the result equals the single-process stack."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total_frames, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    from trpx_amd import sharded
    n = 3000
    lo, hi = sharded.frame_range(total_frames, rank, world)
    px = O.synth(np.uint16, lo, hi - lo, n)
    data, sizes, pb = O.encode_stack(px)                    # stands in for the GPU encode of this shard
    local_offsets = torch.zeros(hi - lo + 1, dtype=torch.int64)
    local_offsets[1:] = torch.cumsum(torch.from_numpy(sizes.astype(np.int64)), 0)
    goffs, base, gpb = sharded.gather_global_offsets(local_offsets, torch.tensor([pb]))
    counts = [sharded.frame_range(total_frames, r, world)[1] - sharded.frame_range(total_frames, r, world)[0] for r in range(world)]
    goffs2, base2, gpb2 = sharded.gather_global_offsets(local_offsets, torch.tensor([pb]), counts=counts)
    assert torch.equal(goffs, goffs2) and int(base) == int(base2) and int(gpb) == int(gpb2)
    # trpx_decode_sharded's first step (shard.hip: k_rebase_offsets), host-side: this rank's offsets back out of the GLOBAL table --
    # ragged shards included -- are the offsets its own encode wrote, and its frames decode from them
    mine = sharded.rebase_offsets(goffs, lo, hi - lo)
    assert torch.equal(mine, local_offsets) and int(goffs[lo]) == int(base)
    for f in (0, hi - lo - 1):
        got = O.decode(data[int(mine[f]): int(mine[f + 1])], n, np.uint16)
        assert (got == px[f]).all()
    if total_frames % world == 0:                           # equal shards: the preallocated gather bench.py overlaps with decode
        sg = sharded.SizeGather(hi - lo, "cpu")
        for _ in range(2):                                   # buffers are reused call after call
            goffs3, base3, gpb3 = sg(local_offsets, torch.tensor([pb], dtype=torch.int32))
            assert torch.equal(goffs, goffs3) and int(base) == int(base3) and int(gpb) == int(gpb3)
    q.put((rank, goffs.numpy().copy(), int(base), int(gpb), data.tobytes()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_size_gather_matches_single_process_stack():
    from oracle import oracle as O
    world, total_frames, n = 2, 7, 3000                     # ragged: 3 + 4 frames
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total_frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want, sizes, pb = O.encode_stack(O.synth(np.uint16, 0, total_frames, n))
    want_offs = np.concatenate([[0], np.cumsum(sizes.astype(np.int64))])
    assembled = bytearray(want.size)
    for rank, goffs, base, gpb, data in res:
        assert (goffs == want_offs).all()
        assert gpb == pb
        assembled[base:base + len(data)] = data             # each rank writes its shard at its global offset
    assert bytes(assembled) == want.tobytes()


def test_two_rank_equal_shards_preallocated_gather():
    world, total_frames = 2, 6
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total_frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert len(res) == world


def test_frame_range_partitions_exactly():
    from trpx_amd import sharded
    for total in (1, 7, 2000, 16000):
        for world in (1, 2, 4, 8):
            r = [sharded.frame_range(total, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == total
            assert all(r[k][1] == r[k + 1][0] for k in range(world - 1))
    assert sharded.frame_range(16000, 3, 8) == (6000, 8000)


def _bench(*args, env=None):
    import subprocess
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=e, capture_output=True, text=True, timeout=300)


def test_bench_gpus_n_starts_its_own_ranks_dry_run():
    """`python bench.py --gpus 2` (no launcher) becomes the launcher: one child per GPU with the rendezvous in its environment."""
    import json
    r = _bench("--gpus", "2", "--steps", "3", "--spawn-dry-run")
    assert r.returncode == 0, r.stderr
    lines = [json.loads(x) for x in r.stdout.splitlines() if x.strip()]
    assert len(lines) == 2
    for rank, ln in enumerate(lines):
        assert ln["argv"][1].endswith("bench.py") and ln["argv"][2:] == ["--gpus", "2", "--steps", "3"]
        assert ln["env"]["RANK"] == ln["env"]["LOCAL_RANK"] == str(rank)
        assert ln["env"]["WORLD_SIZE"] == "2" and ln["env"]["MASTER_ADDR"] == "127.0.0.1"
    assert lines[0]["env"]["MASTER_PORT"] == lines[1]["env"]["MASTER_PORT"]


def test_bench_launcher_relays_rank0_line_and_worst_exit_code():
    import json
    ok = _bench("--gpus", "3", env={"TRPX_BENCH_SPAWN_SELFTEST": "1"})
    assert ok.returncode == 0, ok.stderr
    lines = [x for x in ok.stdout.splitlines() if x.strip()]
    assert len(lines) == 1 and json.loads(lines[0]) == {"selftest": True, "n_gpus": 3, "master": "127.0.0.1"}
    bad = _bench("--gpus", "3", env={"TRPX_BENCH_SPAWN_SELFTEST": "fail:2"})
    assert bad.returncode == 3
    mismatch = _bench("--gpus", "2", env={"WORLD_SIZE": "4", "RANK": "0", "TRPX_BENCH_SPAWN_SELFTEST": "1"})
    assert mismatch.returncode != 0 and "disagree" in mismatch.stderr
