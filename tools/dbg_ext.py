import sys; sys.path.insert(0, "/root/repo")
import numpy as np, torch
from trpx_amd import codec
from oracle import oracle as O
for dt, tdt in ((np.int16, torch.int16), (np.int32, torch.int32), (np.int8, torch.int8)):
    info = np.iinfo(dt)
    px = np.array([info.min, info.max, -1, 0, 1, info.min + 1] * 7, dt).reshape(1, -1)
    want, sizes, pb = O.encode_stack(px)
    dpx = torch.from_numpy(px).cuda()
    out = torch.zeros(4096, dtype=torch.uint8, device="cuda")
    enc = codec.encode(dpx, out=out); torch.cuda.synchronize()
    print(np.dtype(dt).name, "oracle size", sizes, "gpu offsets", enc.frame_offsets.cpu().numpy(), "status", enc.status.cpu().numpy()[:2], "equal", enc.stack().cpu().numpy().tobytes() == want.tobytes())
