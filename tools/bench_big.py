"""One 32768x32768 4:2:0 real-dct image on one GPU, whole: wall time per encode (the first three calls
also allocate the context's three sets of output buffers)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import jpeg_encoder_amd as J
w = h = 32768
base = J.host.synth_bgr(8192, 4096, seed=1)[:, :3 * 8192]
rows = np.tile(base, (h // 4096, w // 8192))
t = torch.from_numpy(rows).cuda(); torch.cuda.synchronize()
enc = J.Encoder((4, 2, 0), "real-dct")
for i in range(4):
    t0 = time.perf_counter()
    enc.encode_device_async(t.data_ptr(), w, h, rows.shape[1], True)
    print(enc.finish(), (time.perf_counter() - t0) * 1e6, flush=True)
