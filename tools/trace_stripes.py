"""One 32768x8192 image resident in HBM, coded as 4 stripes by 4 contexts driven by one host thread (every step queued
on all contexts before any is waited for): run under `rocprofv3 --kernel-trace` to see the stripes' kernels side by side."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import jpeg_encoder_amd as J  # noqa: E402

w, h, n, ratio = 32768, 8192, 4, (4, 2, 0)
base = J.host.synth_bgr(8192, 2048, seed=3)[:, :3 * 8192]
rows = np.tile(base, (h // 2048, w // 8192))
stride = rows.shape[1]
dev = torch.from_numpy(rows).cuda()
torch.cuda.synchronize()
S = J.stripes
plan = S.plan_stripes(h, 2, n)
encs = [J.Encoder(ratio, "real-dct") for _ in range(n)]
for rep in range(3):
    for e, (y0, y1) in zip(encs, plan):
        e.stripe_device(dev.data_ptr() + (h - y1) * stride, w, y1 - y0, stride, True)
    for e in encs:
        e.last_dc_async()
    dcs = [e.last_dc() for e in encs]
    for k, e in enumerate(encs):
        e.entropy_stripe_async(dcs[k - 1] if k else (0, 0, 0))
    codes = [e.entropy_stripe_finish() for e in encs]
    offsets, heads = S.stitch_plan([c[0] for c in codes], [c[1] for c in codes])
    for k, e in enumerate(encs):
        e.stitch_async(offsets[k], heads[k], k == n - 1)
    lens = [e.stitch_finish()[0] for e in encs]
print("stripes of %d rows, %d bytes in all" % (plan[0][1], sum(lens)))
