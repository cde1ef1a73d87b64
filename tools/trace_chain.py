"""One 32768x32768 image resident in HBM, coded as 8 stripes by 8 contexts of one process through the stripe chain
(a whole pass queued on every context, in rank order, before any is waited for): run under `rocprofv3 --kernel-trace`
to see that nothing but kernels lies between the first and the last kernel of a pass (tools/trace_chain.sh)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import jpeg_encoder_amd as J  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
w, h, ratio = 32768, 32768, (4, 2, 0)
base = J.host.synth_bgr(8192, 2048, seed=3)[:, :3 * 8192]
rows = np.tile(base, (h // 2048, w // 8192))
stride = rows.shape[1]
dev = torch.from_numpy(rows).cuda()
del rows
torch.cuda.synchronize()
S, L = J.stripes, J._native.lib()
plan = S.plan_stripes(h, 2, n)
mask = (1 << n) - 1
cap = 2 * (w * h // 2)                        # two halves: one per pass in flight
mailbox, gather = L.jpg_chain_mailbox_alloc(), L.jpg_device_alloc(0, cap)
encs = [J.Encoder(ratio, "real-dct") for _ in range(n)]
for k, e in enumerate(encs):
    e.chain_setup(k, n, mailbox, gather, cap)
for rep in range(4):
    t0 = time.perf_counter()
    for e, (y0, y1) in zip(encs, plan):
        e.stripe_device(dev.data_ptr() + (h - y1) * stride, w, y1 - y0, stride, True)
        e.chain_pass_async(mask)
    res = [e.chain_pass_finish() for e in reversed(encs)][::-1]
    dt = time.perf_counter() - t0
print("%d stripes of %d rows: %d bytes in all, last pass %.3f ms on the host's clock" % (n, plan[0][1], res[0][4], dt * 1e3))
