"""How long does a chain that starts beside the frame's chain (a plain guess at a random bit) take to merge with it?  CPU model on
Poisson(lambda) frames: blocks until the false chain passes through a state of the true one, for 2000 random starts per lambda
(median / p90 / p99 / never inside a 341-block segment) -- the number that sets the position-parallel walk's rounds."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O
from trpx_amd import workloads as W

def true_chain(bits, n_blocks, maxw=16):
    pos, w = 0, 0
    starts = {}
    P = []
    for b in range(n_blocks):
        starts[pos] = w
        P.append(pos)
        if bits[pos]:
            pos += 1 + 12 * w
        else:
            w3 = bits[pos+1] | bits[pos+2] << 1 | bits[pos+3] << 2
            if w3 != 7: w = w3; hl = 4
            else:
                wa = 7 + (bits[pos+4] | bits[pos+5] << 1)
                if wa != 10: w = wa; hl = 6
                else:
                    wb = 10 + sum(int(bits[pos+6+i]) << i for i in range(6)); w = wb; hl = 12
            pos += hl + 12 * w
    return starts, np.array(P)

def step1(bits, pos, w, wmask=7):
    if bits[pos]:
        return pos + 1 + 12 * w, w
    w3 = bits[pos+1] | bits[pos+2] << 1 | bits[pos+3] << 2
    if w3 != 7: return pos + 4 + 12 * w3, w3
    wa = 7 + (bits[pos+4] | bits[pos+5] << 1)
    if wa != 10: return pos + 6 + 12 * wa, wa
    wb = 10 + (sum(int(bits[pos+6+i]) << i for i in range(6)) & wmask)
    return pos + 12 + 12 * wb, wb

for lam in (1.5, 3.0, 10.0):
    px = W.poisson_u16_np(lam, 0, 1, 512*512)[0]
    s = O.encode(px)[0]
    bits = np.unpackbits(s, bitorder='little').astype(np.int64)
    bits = np.concatenate([bits, np.zeros(4096, np.int64)])
    nb = (512*512 + 11)//12
    starts, P = true_chain(bits, nb)
    limit = 8 * s.size
    rng = np.random.RandomState(1)
    dist_blocks = []; never = 0
    N = 2000
    seglen_bits = limit // 64
    for trial in range(N):
        p0 = int(rng.randint(0, limit - 2*seglen_bits)); pos, w = p0, 0
        end = p0 + seglen_bits
        merged = None
        while pos < end:
            if pos in starts and starts[pos] == w:
                merged = pos; break
            # explicit header at true start also merges (state after is the same) -> check after stepping
            if pos in starts and bits[pos] == 0:
                merged = pos; break
            pos, w = step1(bits, pos, w)
        if merged is None: never += 1
        else:
            dist_blocks.append(np.searchsorted(P, merged) - np.searchsorted(P, p0))
    d = np.array(dist_blocks)
    print(f"lam={lam}: bits/frame={limit} blocks/seg={nb/64:.0f} merged {len(d)}/{N} never={never} median={np.median(d):.0f} p90={np.percentile(d,90):.0f} p99={np.percentile(d,99):.0f} max={d.max()}")
