"""Counter-based synthetic detector frames for the benchmark legs and the GPU tests.

Every pixel is a pure function of (seed, frame, pixel index), built from the same 64-bit mixer as synth-v1
(SURVEY.md section 8 row d), so the GPU box, this container and the CPU oracle regenerate identical stacks without an
RNG library: ``numpy`` (host) and ``torch`` (device) versions of the same integer arithmetic.  torch is plumbing here
(device tensors); nothing in this file is part of the codec.

poisson_u16(lam): Poisson(lam) background by inverse CDF on a 53-bit uniform + sparse peaks (one pixel in 4096 gets a
uniform 12-bit value) -- BASELINE.md section 2's "diffraction-like" anchor (Poisson(3) + 1/4096 12-bit peaks) is
``poisson_u16(3.0, ...)``.  The thresholds are computed with ``decimal`` (50 digits), i.e. identically on every box.
"""
from __future__ import annotations

from decimal import Decimal, getcontext

import numpy as np

SEED_POISSON = 20260104
_GOLD = 0x9E3779B97F4A7C15
_M1 = 0xBF58476D1CE4E5B9
_M2 = 0x94D049BB133111EB
_MASK64 = (1 << 64) - 1


def poisson_thresholds(lam: float, tail: float = 1e-17) -> list[int]:
    """t_k = floor(CDF(k) * 2^53): a 53-bit uniform u maps to #{k : u >= t_k}."""
    getcontext().prec = 50
    L = Decimal(repr(float(lam)))
    p = (-L).exp()
    cdf, k, out = Decimal(0), 0, []
    two53 = Decimal(1 << 53)
    while True:
        cdf += p
        out.append(int((cdf * two53).to_integral_value(rounding="ROUND_FLOOR")))
        if Decimal(1) - cdf < Decimal(repr(tail)) or k > 200:
            break
        k += 1
        p = p * L / k
    return out


# ---- numpy (host; tests and the oracle check of the bench) ----------------------------------------------------------------
def _mix_np(z: np.ndarray) -> np.ndarray:
    z = (z ^ (z >> np.uint64(30))) * np.uint64(_M1)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(_M2)
    return z ^ (z >> np.uint64(31))


def poisson_u16_np(lam: float, frame0: int, n_frames: int, n_values: int, seed: int = SEED_POISSON) -> np.ndarray:
    thr = poisson_thresholds(lam)
    out = np.empty((n_frames, n_values), np.uint16)
    i = np.arange(1, n_values + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        for f in range(n_frames):
            ctr = np.uint64(((frame0 + f) * n_values) & _MASK64) + i
            r = _mix_np(np.uint64(seed) + np.uint64(_GOLD) * ctr)
            u = r >> np.uint64(11)
            v = np.zeros(n_values, np.uint16)
            for t in thr:
                v += (u >= np.uint64(t)).astype(np.uint16)
            r2 = _mix_np(r)
            peak = ((r2 >> np.uint64(40)) & np.uint64(0xFFF)) == 0
            v[peak] = ((r2[peak] >> np.uint64(24)) & np.uint64(0xFFF)).astype(np.uint16)
            out[f] = v
    return out


# ---- torch (device) ---------------------------------------------------------------------------------------------------------
def _s64(x: int) -> int:
    """The two's-complement int64 with the bit pattern of the unsigned 64-bit x."""
    x &= _MASK64
    return x - (1 << 64) if x >> 63 else x


def _lsr(z, n: int):
    return (z >> n) & ((1 << (64 - n)) - 1)           # logical shift right on int64 tensors


def _mix_t(z):
    z = (z ^ _lsr(z, 30)) * _s64(_M1)
    z = (z ^ _lsr(z, 27)) * _s64(_M2)
    return z ^ _lsr(z, 31)


def poisson_u16(lam: float, frame0: int, n_frames: int, n_values: int, device="cuda", seed: int = SEED_POISSON,
                chunk_frames: int = 125):
    """The same stack on the device: uint16 [n_frames, n_values].  int64 tensor arithmetic wraps like uint64."""
    import torch
    thr = poisson_thresholds(lam)
    out = torch.empty((n_frames, n_values), dtype=torch.uint16, device=device)
    i = torch.arange(1, n_values + 1, dtype=torch.int64, device=device)
    for f0 in range(0, n_frames, chunk_frames):
        nf = min(chunk_frames, n_frames - f0)
        fr = torch.arange(frame0 + f0, frame0 + f0 + nf, dtype=torch.int64, device=device)
        ctr = (fr * n_values)[:, None] + i[None, :]
        r = _mix_t(ctr * _s64(_GOLD) + _s64(seed))
        u = _lsr(r, 11)                                # 53 bits: non-negative, ordinary signed compares
        v = torch.zeros((nf, n_values), dtype=torch.int16, device=device)
        for t in thr:
            v += (u >= t).to(torch.int16)
        r2 = _mix_t(r)
        peak = (_lsr(r2, 40) & 0xFFF) == 0
        v = torch.where(peak, (_lsr(r2, 24) & 0xFFF).to(torch.int16), v)
        out[f0:f0 + nf] = v.view(torch.uint16)
        del ctr, r, u, v, r2, peak
    return out
