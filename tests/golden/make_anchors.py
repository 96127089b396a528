"""More anchors for the large synthetic stacks, made by the REAL reference (oracle/_ref: the reference's headers compiled in
place, oracle/Makefile `ref`): size, FNV-1a64 of the stream, prolix_bits and first bytes of configs[3]'s frames 1..7 (frame 0 is
in terse_golden.json) and of frames 1000 and 1999 of the 2000-frame u16 stack -- what bench.py times is then pinned by the
reference beyond frame 0.  Run in the build container (needs /root/reference): python tests/golden/make_anchors.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import oracle as O

anchors = []
for dt, n, frames in ((np.int32, 4096 * 4096, range(1, 8)), (np.uint16, 512 * 512, (1000, 1999))):
    for f in frames:
        px = O.synth(dt, f, 1, n)[0]
        s, pb, _ = O.ref_encode(px)
        anchors.append(dict(dtype=np.dtype(dt).name, n=n, frame=f, seed=O.SEED, pixels_fnv=f"{O.fnv1a64(px):016x}", size=len(s),
                            prolix_bits=pb, stream_fnv=f"{O.fnv1a64(s):016x}", first16=s[:16].tobytes().hex()))
        print(anchors[-1], flush=True)
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "synth_anchors.json")
json.dump(dict(generator="tests/golden/make_anchors.py", reference="senikm/trpx @ 2024_08_07 (oracle/_ref)", anchors=anchors), open(path, "w"), indent=1)
print("wrote", path)
