#!/usr/bin/env python3
"""Where a workgroup of the coding kernel spends its life: runs one 32-image launch of the instrumented build
(python tools/build_variant.py stamps -DJPG_PHASE_STAMPS), in which every wave reads the shader clock at its
phase boundaries, and prints the mean length of every phase, the life of a workgroup and, per CU, how many
workgroups are resident on average and how long a slot stands empty between two workgroups.

    JPEG_HIP_LIB=$PWD/jpeg-encoder_amd/variants/stamps.so python tools/phase_stamps.py [--config 3] [--images 32]
"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
import jpeg_encoder_amd as J

NAMES = ["entry -> first pixels arrived", "colour conversion", "barrier A (planes complete)", "transform / MCU before the tile",
         "barrier B (blocks complete)", "mask + code the block", "scan, placement, emit, stores"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=3)
    ap.add_argument("--images", type=int, default=32)
    a = ap.parse_args()
    cfg = bench.CONFIGS[a.config]
    W, H = cfg["width"], cfg["height"]
    ratio = tuple(int(x) for x in cfg["ratio"].split(":"))
    hs, vs = J.get_downsampling_factor(ratio)
    stride = J.host.bmp_stride(W)
    L = J._native.lib()
    if not hasattr(L, "jpg_debug_phase_stamps"):
        raise SystemExit("not the instrumented build: set JPEG_HIP_LIB to variants/stamps.so")
    tw = 1024 if (hs, vs) == (2, 1) else 512
    tiles = ((W + 8 * hs - 1) // (8 * hs) * 8 * hs + tw - 1) // tw * ((H + 8 * vs - 1) // (8 * vs))
    cstride = (tiles + 63) // 64 * 64
    n = a.images
    stamps = torch.zeros((n, cstride, 4, 10), dtype=torch.int64, device="cuda")
    L.jpg_debug_phase_stamps.argtypes = [C.c_void_p]
    L.jpg_debug_phase_stamps.restype = None
    base = [torch.from_numpy(J.host.synth_bgr(W, H, seed=(a.config << 32) | i)).cuda() for i in range(4)]
    imgs = [base[i % 4] if i < 4 else torch.roll(base[i % 4], 48 * (i // 4), dims=1).contiguous() for i in range(n)]
    enc = J.Encoder(ratio, cfg["dct_algorithm"], device=0)
    ptrs = (C.c_void_p * n)(*[t.data_ptr() for t in imgs])
    for rep in range(3):                       # the third launch is the one read
        if rep == 2:
            L.jpg_debug_phase_stamps(stamps.data_ptr())
        enc.encode_batch_device_async(ptrs, W, H, stride, True)
        enc.finish_next()
    torch.cuda.synchronize()
    L.jpg_debug_phase_stamps(None)
    s = stamps.cpu().numpy()[:, :tiles]                     # [img][tile][wave][10]
    t = s[..., :8].astype(np.int64)
    hw, xcc = s[..., 8], s[..., 9]
    d = np.diff(t, axis=-1)                                 # [img][tile][wave][7]
    print("%dx%d %s %s, one launch of %d images, %d workgroups; shader clock ticks (s_memtime), mean over waves"
          % (W, H, cfg["ratio"], cfg["dct_algorithm"], n, n * tiles))
    for k, name in enumerate(NAMES):
        print("  %-34s %8.0f   (wave 0..3: %s)" % (name, d[..., k].mean(), " ".join("%7.0f" % d[:, :, w, k].mean() for w in range(4))))
    life = t[..., 7].max(-1) - t[..., 0].min(-1)            # per workgroup
    print("  life of a workgroup                %8.0f   (min %d, max %d)" % (life.mean(), life.min(), life.max()))
    # per CU: the CU of a workgroup from wave 0's HW_ID (cu_id bits 11:8, sh 12, se 15:13) and XCC_ID
    h0 = hw[:, :, 0]
    cu = ((xcc[:, :, 0] & 0xf) << 16) | (h0 & 0xff00)
    start = t[..., 0].min(-1).ravel()
    end = t[..., 7].max(-1).ravel()
    cu = cu.ravel()
    res, gaps, spans = [], [], []
    for c in np.unique(cu):
        m = cu == c
        st, en = np.sort(start[m]), np.sort(end[m])
        span = en[-1] - st[0]
        res.append((en - st).sum() if False else (end[m] - start[m]).sum() / span)
        spans.append(span)
        # k-th start after the first four pairs with the (k-4)-th end: the slot it took over
        if len(st) > 8:
            gaps.append(np.median(st[4:] - en[:-4]))
    print("  CUs seen %d; workgroups resident per CU %.2f on average; a slot stands empty for %.0f ticks (median) between two workgroups"
          % (len(res), float(np.mean(res)), float(np.median(gaps))))
    print("  busy span of a CU %.0f ticks on average" % float(np.mean(spans)))
    # which SIMD the waves of a workgroup run on (HW_ID bits 5:4)
    simd = (hw >> 4) & 3
    tab = [[int(((simd[:, :, w] == k).sum())) for k in range(4)] for w in range(4)]
    print("  waves 0..3 of a workgroup on SIMD 0..3 (counts): %s" % tab)
    # which die a workgroup runs on, against its number in dispatch order (x fastest, then MCU row, then image)
    x = (xcc[:, :, 0] & 0xf).ravel()
    ident = np.arange(x.size)
    agree = [(int(((ident + k) % 8 == x).sum()), k) for k in range(8)]
    best = max(agree)
    print("  die of workgroup i == (i + %d) mod 8 for %.2f %% of the workgroups" % (best[1], 100.0 * best[0] / x.size))
    enc.close()


if __name__ == "__main__":
    main()
