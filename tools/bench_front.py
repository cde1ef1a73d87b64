"""Times the fused front kernel alone (HIP events on the context's stream) for each ratio/algorithm."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import jpeg_encoder_amd as J  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--width", type=int, default=7680)
ap.add_argument("--height", type=int, default=4320)
ap.add_argument("--iters", type=int, default=40)
ap.add_argument("--configs", default="4:2:0/real-dct")
ap.add_argument("--images", type=int, default=4)
args = ap.parse_args()
W, H = args.width, args.height
stride = J.host.bmp_stride(W)
imgs = [torch.from_numpy(J.host.synth_bgr(W, H, seed=i)).cuda() for i in range(args.images)]
for cfg in args.configs.split(","):
    ratio, algo = cfg.split("/")
    enc = J.Encoder(tuple(int(x) for x in ratio.split(":")), algo)
    s = torch.cuda.ExternalStream(enc.stream)
    for k in range(4):
        enc.front_device(imgs[k % len(imgs)].data_ptr(), W, H, stride, True)
    enc.sync()
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for k in range(args.iters):
            enc.front_device(imgs[k % len(imgs)].data_ptr(), W, H, stride, True)
        e1.record(s)
        enc.sync()
        best = min(best, e0.elapsed_time(e1) / args.iters)
    b = enc.front_bytes()
    print("%-28s %8.2f us  %7.1f GB/s  %5.1f%% of 8 TB/s  %9.0f MP/s" %
          (cfg, best * 1e3, b / best / 1e6, b / best / 1e6 / 80.0, W * H / best / 1e3), flush=True)
    enc.close()
