"""Randomised differential run on the GPU box: random pixel types, frame sizes, frame counts and width patterns, encoded by
the GPU (compared with the oracle's bytes) and decoded along the path selected by $TRPX_DECODE_PATH (compared with the
pixels).  `python tools/fuzz_paths.py [cases] [seed]`; run it once per decode path.  tests/test_gpu_parity.py imports `run` for a
bounded, seeded slice per route (trpx_set_decode_path) inside the GPU tier."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trpx_amd import codec
from oracle import oracle as O
DT = [np.uint8, np.int8, np.uint16, np.int16, np.uint32, np.int32]
TDT = {np.uint8: torch.uint8, np.int8: torch.int8, np.uint16: torch.uint16, np.int16: torch.int16, np.uint32: torch.uint32, np.int32: torch.int32}


def run(cases, seed=1, basic=False, budget_s=None, quiet=False, max_pixels=None):
    """`cases` random stacks from `seed` through the route the library is set to.  basic: skip the decode-index checks (the basic
    kernels have none); budget_s: stop starting cases after this many seconds; max_pixels: skip stacks larger than this (the
    random sequence stays the same).  Returns (cases run, large frames seen, of them handed to the fallback route)."""
    rng = np.random.RandomState(seed)
    t0 = time.time()
    n_large = n_fallback = n_run = 0
    for c in range(cases):
        dt = np.dtype(DT[rng.randint(6)])
        top = 8 * dt.itemsize - (2 if dt.kind == "i" else (1 if dt.itemsize == 4 else 0))      # inside the reference's validity domain (D3)
        n = int(rng.choice([4, 12, 52, 388, 4096, 12 * 768 + 4, 40000, 131072, 262144 + 8 * rng.randint(0, 3), 12 * 34000 + 8 * rng.randint(0, 3),
                            513 * 511, 12 * 70000 + rng.randint(0, 12), 1030 * 1065, 1030 * 1065, 1475 * 1679, 2048 * 2048 + rng.randint(0, 3)]))   # (> 32768 blocks: cut into parts, decode_part.hip)
        n = max(1, n - rng.randint(0, 4) * rng.randint(0, 2))                                     # (half of the cases: no multiple of 4)
        frames = int(rng.choice([1, 2, 3, 17, 129, 140])) if n <= 40000 else (int(rng.choice([1, 3, 130, 130, 130, 800])) if n <= 12 * 34000 + 16 else int(rng.choice([1, 2, 9])))
        nblk = (n + 11) // 12
        kind = rng.randint(8)
        if kind == 0:   hi = np.full((frames, nblk), rng.randint(0, top + 1))                    # one width
        elif kind == 1: hi = rng.randint(0, top + 1, size=(frames, nblk))                        # every block its own width
        elif kind == 2: hi = np.where(rng.rand(frames, nblk) < 0.02, rng.randint(0, top + 1, size=(frames, nblk)), 3 if top >= 3 else 1)   # runs + outliers
        elif kind == 3: hi = np.where(rng.rand(frames, nblk) < 0.5, 2, 3 if top >= 3 else 1)    # flips every other block
        elif kind == 4: hi = np.repeat(rng.randint(0, top + 1, size=(frames, (nblk + 299) // 300)), 300, axis=1)[:, :nblk]   # long runs of changing widths
        elif kind == 5:                                                                            # every frame its own rate of width changes (the hand-over's stack statistics)
            rate = rng.choice([0.01, 0.1, 0.18, 0.25, 0.4], size=(frames, 1))
            hi = np.where(rng.rand(frames, nblk) < rate, 2, 3 if top >= 3 else 1)
        else:           hi = np.full((frames, nblk), min(top, 3))                                   # (6, 7: values set below)
        mag = (rng.rand(frames, nblk * 12) * (2.0 ** np.repeat(hi, 12, axis=1))).astype(np.int64)[:, :n]
        if kind == 6:                                                                              # a pedestal: constant top bits in every field
            base = int(rng.choice([96, 100, 1000, 4000, 30000])) if dt.itemsize > 1 else int(rng.choice([16, 40, 96]))
            base = min(base, (1 << top) - 8)
            mag = base + (rng.rand(frames, n) * rng.choice([2, 7, 8])).astype(np.int64)
            if rng.rand() < 0.5: mag = np.where(rng.rand(frames, n) < 1.0 / 4096, (rng.rand(frames, n) * (1 << top)).astype(np.int64), mag)
        if kind == 7:                                                                              # Poisson counts + rare peaks: header-dense
            mag = rng.poisson(rng.choice([0.3, 1.5, 3.0, 10.0]), size=(frames, n)).astype(np.int64)
            mag = np.minimum(np.where(rng.rand(frames, n) < 1.0 / 4096, (rng.rand(frames, n) * (1 << min(top, 12))).astype(np.int64), mag), (1 << top) - 1)
        if rng.rand() < 0.3: mag[:, : n // 2] = 0                                                 # empty half frames
        if dt.kind == "i": mag = mag * rng.choice([-1, 1], size=mag.shape)
        px = mag.astype(dt)
        if rng.rand() < 0.25:                                                                      # the type's extremes, outside the reference's validity domain
            info = np.iinfo(dt)
            hit = rng.rand(*px.shape) < 0.002
            px = np.where(hit, rng.choice([info.min, info.max], size=px.shape), px).astype(dt)
        if os.environ.get("TRPX_FUZZ_ONLY") and c not in [int(x) for x in os.environ["TRPX_FUZZ_ONLY"].split(",")]: continue   # (same random sequence, only these cases run)
        if max_pixels is not None and frames * n > max_pixels: continue
        if budget_s is not None and time.time() - t0 > budget_s: break
        n_run += 1
        if os.environ.get("TRPX_FUZZ_ONLY"): print(f"case {c}: kind {kind} {dt} n {n} frames {frames} widths {np.unique(hi)[:8]} pixels {np.unique(px)[:12]}", flush=True)
        want, sizes, pb = O.encode_stack(px)
        dpx = torch.from_numpy(px.view(np.dtype(f"i{dt.itemsize}"))).cuda().view(TDT[dt.type])
        enc = codec.encode(dpx); torch.cuda.synchronize(); enc.check()
        assert enc.stack().cpu().numpy().tobytes() == want.tobytes() and enc.prolix_bits() == pb, ("encode", c, dt, n, frames, kind)
        back, st = codec.decode(enc.stack(), enc.frame_offsets, n, frames, dt); torch.cuda.synchronize()
        assert int(st[0]) == 0 and (back.cpu().numpy().reshape(frames, n).view(dt) == px).all(), ("decode", c, dt, n, frames, kind, int(st[0]))
        if nblk > 32768:                                                                           # the index route's frames / those it handed to the other route
            n_large += frames; n_fallback += int(st[2])
            if int(st[2]) and os.environ.get("TRPX_FUZZ_VERBOSE"): print(f"  fallback: case {c} kind {kind} {dt} n {n} frames {frames}: {int(st[2])} frames", flush=True)
        if not basic:
            # the decode index three ways: the encoder's by-product, trpx_build_index, rebuilt from the group states -- all equal, all decode
            from trpx_amd import _lib
            L = _lib.lib()
            enc_i = codec.encode(dpx, index=True); torch.cuda.synchronize(); enc_i.check()
            walked = codec.build_index(enc_i.stack(), enc_i.frame_offsets, n, frames, dt); torch.cuda.synchronize()
            ng = (nblk + 255) // 256
            w_off = (8 * frames * ng + 15) // 16 * 16
            for what, other in (("build_index", walked),):
                assert torch.equal(enc_i.index[: 8 * frames * ng], other[: 8 * frames * ng]), (what, "group offsets", c, dt, n, frames, kind)
                assert torch.equal(enc_i.index[w_off: w_off + frames * nblk], other[w_off: w_off + frames * nblk]), (what, "widths", c, dt, n, frames, kind)
            states = torch.zeros(frames * ng, dtype=torch.int64, device="cuda")
            _lib.check(L.trpx_index_group_states(enc_i.index.data_ptr(), n, frames, 12, states.data_ptr(), None))
            rebuilt = torch.zeros_like(enc_i.index)
            st8 = torch.zeros(8, dtype=torch.int32, device="cuda")
            stack = enc_i.stack()
            _lib.check(L.trpx_index_from_group_states(codec.dtype_code(TDT[dt.type]), stack.data_ptr(), stack.numel(), enc_i.frame_offsets.data_ptr(),
                                                      states.data_ptr(), n, frames, 12, rebuilt.data_ptr(), st8.data_ptr(), None))
            torch.cuda.synchronize()
            assert int(st8[0]) == 0, ("group states", c, dt, n, frames, kind, int(st8[0]))
            assert torch.equal(enc_i.index[: 8 * frames * ng], rebuilt[: 8 * frames * ng]) and \
                torch.equal(enc_i.index[w_off: w_off + frames * nblk], rebuilt[w_off: w_off + frames * nblk]), ("group states", c, dt, n, frames, kind)
            back, st = codec.decode(stack, enc_i.frame_offsets, n, frames, dt, index=rebuilt); torch.cuda.synchronize()
            assert int(st[0]) == 0 and (back.cpu().numpy().reshape(frames, n).view(dt) == px).all(), ("indexed decode", c, dt, n, frames, kind, int(st[0]))
        if c % 20 == 0 and not quiet: print(f"case {c} ok ({time.time() - t0:.0f} s)", flush=True)

    return n_run, n_large, n_fallback


if __name__ == "__main__":
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    n_run, n_large, n_fallback = run(cases, int(sys.argv[2]) if len(sys.argv) > 2 else 1, basic=os.environ.get("TRPX_DECODE_PATH", "") == "basic")
    print(f"OK {n_run} cases, path {os.environ.get('TRPX_DECODE_PATH', 'default')}; large frames {n_large}, of them handed to the fallback route {n_fallback}")
