"""CPU model of the position-parallel header walk (decode_seg.hip): how many fix-point rounds does a frame need
under a given guess heuristic?  Uses the oracle encoder for the streams; numpy over the 64 lanes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as O

def bits_of(stream):
    return np.unpackbits(stream, bitorder="little")

def peek(bits, pos, nb):
    idx = pos.astype(np.int64)[:, None] + np.arange(nb, dtype=np.int64)[None, :]
    idx = np.minimum(idx, bits.size - 1)
    return (bits[idx].astype(np.uint32) << np.arange(nb, dtype=np.uint32)[None, :]).sum(1)

def step(bits, pos, w):
    b = peek(bits, pos, 12)
    same = (b & 1) == 1
    w3 = (b >> 1) & 7; wa = 7 + ((b >> 4) & 3); wb = 10 + ((b >> 6) & 63)
    wx = np.where(w3 != 7, w3, np.where(wa != 10, wa, wb))
    hx = np.where(w3 != 7, 4, np.where(wa != 10, 6, 12))
    wn = np.where(same, w, wx)
    wn = np.where(wn > 16, 0, wn)
    ln = np.where(same, 1, hx) + 12 * wn
    return pos + ln, wn

def walk(bits, pos, w, end, active):
    pos, w = pos.copy(), w.copy()
    cnt = np.zeros_like(pos)
    act = active & (pos < end)
    steps = 0
    while act.any():
        np_, nw = step(bits, pos, w)
        pos = np.where(act, np_, pos); w = np.where(act, nw, w); cnt += act
        act = act & (pos < end)
        steps += 1
    return pos, w, cnt, steps

def comb_guess(bits, X, R=32, wmax=8):
    out_p = np.full(X.size, -1, np.int64); out_w = np.zeros(X.size, np.int64)
    for j, x in enumerate(X):
        for w in range(wmax + 1):
            s = 1 + 12 * w
            fit = min(R, 1 + (864 - 32 * ((s + 31) // 32)) // s)
            hit = -1
            for p in range(s):
                if all(bits[min(x + p + k * s, bits.size - 1)] for k in range(fit)):
                    hit = p; break
            if hit >= 0:
                out_p[j] = x + hit; out_w[j] = w; break
    return out_p, out_w

def comb_flex(bits, X, R=16, wmax=4, span=864):
    out_p = np.full(X.size, -1, np.int64); out_w = np.zeros(X.size, np.int64)
    n = bits.size
    for j, x in enumerate(X):
        seg = bits[x:x + span + 64].astype(bool)
        if seg.size < span + 64: seg = np.concatenate([seg, np.zeros(span + 64 - seg.size, bool)])
        for w in range(1, wmax + 1):
            s_ = 1 + 12 * w
            R = max(12, min(32, 560 // s_))
            rng_ = span - (R - 1) * s_ - 12
            if rng_ <= 0: continue
            a = np.ones(rng_, bool)
            for k in range(R): a &= seg[k * s_: k * s_ + rng_]
            nz = np.flatnonzero(a)
            if nz.size:
                out_p[j] = x + nz[0]; out_w[j] = w; break
    return out_p, out_w

def simulate(stream, G=64, guess="trivial", verbose=False, trust=False):
    bits = bits_of(stream)
    limit = 8 * stream.size
    L = ((limit + G - 1) // G + 127) // 128 * 128
    jl = min(G - 1, (limit - 8 - 400 - 864) // L)
    X = np.arange(G, dtype=np.int64) * L
    ipos = X.copy(); iw = np.zeros(G, np.int64)
    B = X.copy()
    if guess in ("comb", "flex"):
        gp, gw = comb_guess(bits, X) if guess == "comb" else comb_flex(bits, X)
        ok = gp >= 0
        if guess == "flex": B = np.where(ok, gp, X); B[0] = 0
        ipos = np.where(ok, gp, ipos); iw = np.where(ok, gw, iw)
    strong = np.zeros(G, bool)
    if guess in ("comb", "flex"): strong = ok.copy()
    ipos[0] = 0; iw[0] = 0
    walks = np.arange(G) < jl
    dirty = walks.copy()
    opos = np.zeros(G, np.int64); ow = np.zeros(G, np.int64)
    rounds = 0; total_steps = 0; hist = []
    tent = np.zeros(G, bool); spos = ipos.copy(); sw = iw.copy(); sopos = opos.copy(); sow = ow.copy()
    while dirty.any():
        p, w, c, st = walk(bits, ipos, iw, np.roll(B, -1), dirty)
        # tentative walks of strong lanes: keep the adopted state only if the chain ends where the strong one did
        merged = tent & (p == sopos) & (w == sow)
        revert = tent & ~merged
        opos = np.where(dirty & ~revert, p, opos); ow = np.where(dirty & ~revert, w, ow)
        ipos = np.where(revert, spos, ipos); iw = np.where(revert, sw, iw)
        strong = strong & ~merged
        rounds += 1; total_steps += st; hist.append(int(dirty.sum()))
        npos = np.roll(opos, 1); nw = np.roll(ow, 1)
        ch = ((npos != ipos) | (nw != iw)) & (np.arange(G) > 0) & (np.arange(G) <= jl)
        tent = np.zeros(G, bool)
        if trust:   # a lane with a strong guess only gives it up for a trusted predecessor -- or tentatively
            link_ok = ~ch; link_ok[0] = True
            ver = np.cumprod(link_ok).astype(bool)
            pred_ver = np.roll(ver, 1); pred_ver[0] = True
            pred_link = np.roll(link_ok, 1); pred_link[0] = True; pred_link[1] = True
            trusted = pred_ver | pred_link | ~strong
            tent = ch & ~trusted & ~revert          # (a reverted lane does not retry the same candidate)
            spos = np.where(tent, ipos, spos); sw = np.where(tent, iw, sw); sopos = np.where(tent, opos, sopos); sow = np.where(tent, ow, sow)
            strong = strong & ~(ch & trusted)
            ch = ch & (trusted | tent)
        ipos = np.where(ch, npos, ipos); iw = np.where(ch, nw, iw)
        dirty = ch & walks
    return rounds, total_steps, hist

if __name__ == "__main__":
    n = 512 * 512
    rng = np.random.default_rng(1)
    sets = {}
    sets["synth-v1"] = O.synth(np.uint16, 0, 1, n)[0]
    bg = np.minimum(rng.poisson(1.5, n), 6)
    hot = rng.random(n) < 1 / 4096
    noisy = np.where(hot, rng.integers(0, 4000, n), bg)
    sets["noisy u16"] = noisy.astype(np.uint16)
    sets["const i16"] = (noisy - 3).astype(np.int16)
    sets["wide u16"] = rng.integers(0, 3000, n).astype(np.uint16)
    for name, px in sets.items():
        stream = O.encode(px)[0]
        for G in (64, 128, 256):
            for guess, trust in (("trivial", False), ("flex", True)):
                r, s, h = simulate(np.asarray(stream), G, guess, trust=trust)
                print(f"{name:10s} G={G:3d} guess={guess:7s} trust={trust} rounds {r:2d} wave-steps {s:6d} dirty/round {h}")
