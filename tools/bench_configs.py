"""Device-resident encode rates of the named configurations on one GPU (inputs in HBM, stuffed scan
in HBM, results taken for every image; two encodes in flight).  The headline figure is bench.py's;
this is the companion table."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import jpeg_encoder_amd as J  # noqa: E402


def run(name, w, h, ratio, algo, nimg, reps, tile=None):
    if tile:
        base = J.host.synth_bgr(tile[0], tile[1], seed=1)[:, :3 * tile[0]]
        rows = [np.tile(base, (h // tile[1], w // tile[0]))]
    else:
        rows = [J.host.synth_bgr(w, h, seed=k) for k in range(nimg)]
    stride = rows[0].shape[1]
    dev = [torch.from_numpy(r).cuda() for r in rows]
    torch.cuda.synchronize()
    enc = J.Encoder(ratio, algo)

    def one_pass():
        inflight, last = 0, None
        for t in dev:
            if inflight == 2:
                last = enc.finish_next()
                inflight -= 1
            enc.encode_device_async(t.data_ptr(), w, h, stride, True)
            inflight += 1
        while inflight:
            last = enc.finish_next()
            inflight -= 1
        return last

    for _ in range(3):          # every set of output buffers of the context has been used once
        one_pass()
    t0 = time.perf_counter()
    for _ in range(reps):
        nbits, n = one_pass()
    dt = (time.perf_counter() - t0) / (reps * len(dev))
    print("%-44s %9.1f us/image %10.0f MP/s   scan %d B" % (name, dt * 1e6, w * h / dt / 1e6, n), flush=True)
    enc.close()


def run_batched(name, w, h, ratio, algo, nimg, per_launch, reps):
    rows = [J.host.synth_bgr(w, h, seed=k) for k in range(nimg)]
    stride = rows[0].shape[1]
    dev = [torch.from_numpy(r).cuda() for r in rows]
    torch.cuda.synchronize()
    enc = J.Encoder(ratio, algo)
    groups = [[t.data_ptr() for t in dev[i:i + per_launch]] for i in range(0, nimg, per_launch)]

    def one_pass():
        inflight, total = 0, 0
        for g in groups:
            if inflight == 2:
                enc.finish_next()
                total += enc.batch_result(0)[1]
                inflight -= 1
            enc.encode_batch_device_async(g, w, h, stride, True)
            inflight += 1
        while inflight:
            enc.finish_next()
            total += enc.batch_result(0)[1]
            inflight -= 1
        return total

    one_pass()
    t0 = time.perf_counter()
    for _ in range(reps):
        one_pass()
    dt = (time.perf_counter() - t0) / (reps * nimg)
    print("%-44s %9.1f us/image %10.0f MP/s   (%d images per launch)" % (name, dt * 1e6, w * h / dt / 1e6, per_launch),
          flush=True)
    enc.close()


run("512x512 4:2:0 real-dct", 512, 512, (4, 2, 0), "real-dct", 16, 20)
run_batched("512x512 4:2:0 real-dct, batched", 512, 512, (4, 2, 0), "real-dct", 256, 64, 5)
run("3840x2160 4:4:4 bin-dct", 3840, 2160, (4, 4, 4), "bin-dct", 4, 10)
run("7680x4320 4:2:0 real-dct", 7680, 4320, (4, 2, 0), "real-dct", 4, 10)
run("1920x1080 4:2:2 real-dct, batch of 64", 1920, 1080, (4, 2, 2), "real-dct", 64, 5)
run("1920x1080 4:2:2 bin-dct, batch of 64", 1920, 1080, (4, 2, 2), "bin-dct", 64, 5)
for per in (8, 16, 32, 64):
    run_batched("1920x1080 4:2:2 real-dct, batched", 1920, 1080, (4, 2, 2), "real-dct", 128, per, 5)
run("32768x32768 4:2:0 real-dct (one GPU, whole)", 32768, 32768, (4, 2, 0), "real-dct", 1, 3, tile=(8192, 4096))
