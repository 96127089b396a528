"""Frames shard across GPUs; the only exchange is the per-frame size gather (SURVEY.md 8 row e).

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU
tests).  Rank r owns the contiguous frame range [r*F/G, (r+1)*F/G) of the global stack
(Terse.hpp:502-504: frames are independent and byte aligned, so the global stack is the
concatenation of the ranks' local stacks).  The compressed payload never crosses xGMI: every rank
keeps / writes its frames at the global byte offsets computed here.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.distributed as dist


def frame_range(n_frames_total: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous partition: GPU g <- frames [g*F/G, (g+1)*F/G)."""
    return n_frames_total * rank // world, n_frames_total * (rank + 1) // world


def gather_global_offsets(local_offsets: torch.Tensor, local_prolix_bits: torch.Tensor | None = None,
                          group=None, counts: list[int] | None = None, force: bool = False):
    """All-gather the per-frame sizes of every rank and prefix-sum them.

    local_offsets: int64 [f_local + 1] (byte offsets of the local stack, as trpx_encode writes them).
    Returns (global_offsets int64 [F_total + 1], my_base: 0-dim int64 tensor = first byte of this rank's
    stack in the global stack, prolix_bits: max over ranks or None).  Works with ragged shards: sizes are
    padded to the largest shard for the fixed-size collective."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    sizes = local_offsets[1:] - local_offsets[:-1]
    if world == 1 and not force:   # (force: run the collective even alone -- self test of the RCCL path)
        pb = local_prolix_bits.max() if local_prolix_bits is not None else None
        return local_offsets.clone(), torch.zeros((), dtype=torch.int64, device=local_offsets.device), pb
    if counts is not None:                                  # shard sizes known up front: no host round trip
        counts_host = [int(c) for c in counts]
    else:
        my_count = torch.tensor([sizes.numel()], dtype=torch.int64, device=sizes.device)
        all_counts = torch.empty(world, dtype=torch.int64, device=sizes.device)
        dist.all_gather_into_tensor(all_counts, my_count, group=group)
        counts_host = [int(c) for c in all_counts.tolist()]
    fmax = max(counts_host)
    padded = torch.zeros(fmax + 1, dtype=torch.int64, device=sizes.device)
    padded[: sizes.numel()] = sizes
    if local_prolix_bits is not None:                       # fold the max-reduce into the same gather
        padded[fmax] = local_prolix_bits.to(torch.int64).max()
    gathered = torch.empty(world * (fmax + 1), dtype=torch.int64, device=sizes.device)
    dist.all_gather_into_tensor(gathered, padded, group=group)
    gathered = gathered.view(world, fmax + 1)
    parts = [gathered[r, : counts_host[r]] for r in range(world)]
    all_sizes = torch.cat(parts)
    global_offsets = torch.zeros(all_sizes.numel() + 1, dtype=torch.int64, device=sizes.device)
    torch.cumsum(all_sizes, 0, out=global_offsets[1:])
    my_base = global_offsets[sum(counts_host[:rank])]
    pb = gathered[:, fmax].max() if local_prolix_bits is not None else None
    return global_offsets, my_base, pb


def rebase_offsets(global_offsets: torch.Tensor, first_frame: int, n_local: int) -> torch.Tensor:
    """A rank's frame offsets relative to its own first byte, from the global table: what trpx_decode_sharded computes on the
    device in front of its decode (shard.hip: k_rebase_offsets -- local[i] = global[first + i] - global[first], i = 0 .. n_local).
    Host-side statement of the same arithmetic, for ragged shards too (tests/test_sharded.py drives it over gloo)."""
    g = global_offsets[first_frame: first_frame + n_local + 1]
    assert g.numel() == n_local + 1, "the global table does not hold this rank's frames"
    return g - g[0]


class SizeGather:
    """The same exchange for a fixed, equal shard size, with every buffer allocated once: five small launches and one
    collective per call, meant to run on its own stream next to the decode (which does not need the global offsets)."""

    def __init__(self, frames_per_rank: int, device, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.f = int(frames_per_rank)
        self.padded = torch.zeros(self.f + 1, dtype=torch.int64, device=device)
        self.gathered = torch.empty(self.world * (self.f + 1), dtype=torch.int64, device=device)
        self.global_offsets = torch.zeros(self.world * self.f + 1, dtype=torch.int64, device=device)

    def __call__(self, local_offsets: torch.Tensor, local_prolix_bits: torch.Tensor):
        """Returns (global_offsets, my_base, prolix_bits) like gather_global_offsets (views of internal buffers)."""
        torch.sub(local_offsets[1:], local_offsets[:-1], out=self.padded[: self.f])
        self.padded[self.f:].copy_(local_prolix_bits.reshape(-1)[:1])
        if dist.is_initialized():
            dist.all_gather_into_tensor(self.gathered, self.padded, group=self.group)
        else:
            self.gathered.copy_(self.padded)
        g = self.gathered.view(self.world, self.f + 1)
        torch.cumsum(g[:, : self.f].reshape(-1), 0, out=self.global_offsets[1:])
        return self.global_offsets, self.global_offsets[self.rank * self.f], g[:, self.f].max()


class RcclSizeGather:
    """The size gather through the C ABI (`trpx_gather_frame_offsets`, include/trpx_hip.h): one pack kernel, one
    ncclAllGather on the calling stream, one scan kernel -- what a C++ `jpa::Terse`-style caller uses.  The RCCL
    communicator is made here from an id that rank 0 creates and torch.distributed carries to the other ranks (any
    transport would do); with no process group it is a one-rank communicator.  GPU only."""

    def __init__(self, frames_per_rank: int, device, group=None):
        import ctypes as C
        from . import _lib
        self._L = _lib.lib()
        self._lib = _lib
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.f = int(frames_per_rank)
        device = torch.device(device)
        ident = torch.zeros(128, dtype=torch.uint8)
        if self.rank == 0:
            buf = (C.c_char * 128)()
            self._check(self._L.trpx_comm_unique_id(C.cast(buf, C.c_void_p)))
            ident = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
        if self.world > 1:
            on = ident.to(device) if dist.get_backend(group) == "nccl" else ident
            dist.broadcast(on, src=0, group=group)
            ident = on.cpu()
        raw = bytes(ident.numpy().tobytes())
        comm = C.c_void_p()
        with torch.cuda.device(device):
            self._check(self._L.trpx_comm_init(C.byref(comm), self.world, self.rank, raw))
        self.comm = comm
        self.global_offsets = torch.zeros(self.world * self.f + 1, dtype=torch.int64, device=device)
        self.prolix = torch.zeros(2, dtype=torch.int32, device=device)
        self.rank_base = torch.zeros(self.world, dtype=torch.int64, device=device)
        self.ws = torch.empty(self._L.trpx_gather_workspace_bytes(self.f, self.world), dtype=torch.uint8, device=device)

    def _check(self, rc: int) -> None:
        if rc != 0:
            raise self._lib.TrpxError(rc, self._L.trpx_shard_last_error().decode(errors="replace"))

    def __call__(self, local_offsets: torch.Tensor, encode_status: torch.Tensor | None = None):
        """local_offsets: int64 [f + 1] on the GPU (what trpx_encode wrote); encode_status: the encode's status block
        (word 1 = prolix_bits) or None.  Returns (global_offsets, my_base, prolix_bits) as views of internal buffers;
        stream-ordered on the current stream."""
        assert local_offsets.is_cuda and local_offsets.numel() == self.f + 1 and local_offsets.dtype == torch.int64
        st = torch.cuda.current_stream(local_offsets.device).cuda_stream
        self._check(self._L.trpx_gather_frame_offsets(self.comm, local_offsets.data_ptr(), self.f, self.f,
                                                      encode_status.data_ptr() if encode_status is not None else None,
                                                      self.global_offsets.data_ptr(), self.prolix.data_ptr(),
                                                      self.rank_base.data_ptr(), self.ws.data_ptr(), self.ws.numel(), st))
        return self.global_offsets, self.rank_base[self.rank], self.prolix[0]

    def rccl_info(self):
        """(ranks, rank) of the communicator as RCCL reports them (ncclCommCount / ncclCommUserRank through trpx_comm_info)."""
        w, r = C.c_int(-1), C.c_int(-1)
        self._check(self._L.trpx_comm_info(self.comm, C.byref(w), C.byref(r)))
        return int(w.value), int(r.value)

    def close(self) -> None:
        if self.comm:
            self._L.trpx_comm_destroy(self.comm)
            self.comm = None


class ShardedCodec(RcclSizeGather):
    """One rank's share of a sharded stack through the single-call C ABI: `encode()` = `trpx_encode_sharded` (this rank's frames
    encoded + the size gather over RCCL, optionally on a second stream so that the caller's next work overlaps the
    collective), `decode()` = `trpx_decode_sharded` (this rank's frames expanded from the GLOBAL offset table).  Equal shards
    of `frames_per_rank` frames; buffers allocated once.  GPU only."""

    def __init__(self, frames_per_rank: int, n_values: int, dtype, device, group=None, block: int = 12):
        super().__init__(frames_per_rank, device, group)
        from . import codec
        self._codec = codec
        self.n_values, self.block = int(n_values), int(block)
        self.tdt = codec.torch_dtype(dtype)
        self.code = codec.dtype_code(self.tdt)
        device = torch.device(device)
        L = self._L
        cap = (self.f * L.trpx_worst_case_bytes(self.code, self.n_values, self.block) + 15) // 16 * 16
        self.out = torch.empty(cap, dtype=torch.uint8, device=device)
        self.local_offsets = torch.empty(self.f + 1, dtype=torch.int64, device=device)
        self.status = torch.empty(self._lib.STATUS_WORDS, dtype=torch.int32, device=device)
        self.dec_status = torch.empty(self._lib.STATUS_WORDS, dtype=torch.int32, device=device)
        self.ws_e = torch.empty(L.trpx_encode_sharded_workspace_bytes(self.code, self.n_values, self.f, self.f, self.block, self.world),
                                dtype=torch.uint8, device=device)
        self.ws_d = torch.empty(L.trpx_decode_sharded_workspace_bytes(self.code, self.n_values, self.f, self.block), dtype=torch.uint8, device=device)

    def encode(self, pixels: torch.Tensor, gather_stream: "torch.cuda.Stream | None" = None):
        """pixels: [frames_per_rank, n_values] on this rank's GPU.  Stream-ordered on the current stream (the gather on
        `gather_stream` if given).  Returns (global_offsets, my_base, prolix_bits) as views of internal buffers.

        A gather on a side stream writes global_offsets / prolix / rank_base there while it still reads local_offsets and
        status: the object remembers that stream, and the next encode() or decode() -- or join() -- makes the current stream
        wait for it first (include/trpx_hip.h: the caller joins the two streams where it reads the table)."""
        assert pixels.is_cuda and pixels.shape[0] == self.f and pixels[0].numel() == self.n_values and pixels.dtype == self.tdt
        self.join(pixels.device)                               # (a gather of the call before may still be reading what this call overwrites)
        st = torch.cuda.current_stream(pixels.device).cuda_stream
        with torch.cuda.device(pixels.device):
            self._check(self._L.trpx_encode_sharded(self.comm, self.code, pixels.data_ptr(), self.n_values, self.f, self.f, self.block,
                                                    self.out.data_ptr(), self.out.numel(), self.local_offsets.data_ptr(), self.status.data_ptr(),
                                                    self.global_offsets.data_ptr(), self.prolix.data_ptr(), self.rank_base.data_ptr(),
                                                    self.ws_e.data_ptr(), self.ws_e.numel(), st,
                                                    gather_stream.cuda_stream if gather_stream is not None else None))
        self._gather_stream = gather_stream
        return self.global_offsets, self.rank_base[self.rank], self.prolix[0]

    def join(self, device=None):
        """Makes the current stream wait for the last encode()'s gather if that ran on a side stream."""
        gs = getattr(self, "_gather_stream", None)
        if gs is not None:
            torch.cuda.current_stream(device if device is not None else gs.device).wait_stream(gs)
            self._gather_stream = None

    def decode(self, out: torch.Tensor, stream_signed: bool | None = None):
        """Expands this rank's frames (the bytes `encode` left in self.out) from the global offset table into `out`."""
        assert out.is_cuda and out.numel() == self.f * self.n_values and out.dtype == self.tdt
        if stream_signed is None:
            stream_signed = bool(self._L.trpx_dtype_is_signed(self.code))
        self.join(out.device)                                  # (the global table may come from a side stream)
        st = torch.cuda.current_stream(out.device).cuda_stream
        with torch.cuda.device(out.device):
            self._check(self._L.trpx_decode_sharded(int(stream_signed), self.code, self.out.data_ptr(), self.out.numel(),
                                                    self.global_offsets.data_ptr(), self.rank * self.f, self.n_values, self.f, self.block,
                                                    out.data_ptr(), self.dec_status.data_ptr(), self.ws_d.data_ptr(), self.ws_d.numel(), st))
        return out, self.dec_status
