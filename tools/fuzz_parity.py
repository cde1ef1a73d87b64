"""Randomised parity run on the GPU box: random sizes, ratios, transforms, contents, row orders, pitches and
input kinds (host rows, device rows, batches, stripes with host exchanges, stripes through the chain) through the C ABI, every stuffed scan compared byte for byte with
the CPU oracle.  usage: python tools/fuzz_parity.py [cases] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import jpeg_encoder_amd as J  # noqa: E402
from oracle import refcpu  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
RATIOS = [(4, 4, 4), (4, 2, 2), (4, 2, 0)]
ALGOS = [("bin-dct", refcpu.BIN), ("real-dct-strict", refcpu.REAL)]
encs = {}


def content(w, h, kind, stride):
    rows = np.zeros((h, stride), np.uint8)
    if kind == 0:
        rows[:, :3 * w] = J.host.synth_bgr(w, h, int(rng.integers(1 << 30)))[:, :3 * w]
    elif kind == 1:
        rows[:, :3 * w] = rng.integers(0, 256, (h, 3 * w), dtype=np.uint8)
    elif kind == 2:
        rows[:, :3 * w] = rng.choice(np.array([0, 255], np.uint8), size=(h, 3 * w))
    elif kind == 3:
        m = rng.random((h, 3 * w)) < 0.01
        rows[:, :3 * w][m] = 255
    else:
        rows[:, :3 * w] = int(rng.integers(256))
    return rows


def chain_scan(rows, w, h, stride, ratio, algo, vs, bottom_up, n):
    """MCU-row stripes through the stripe chain: n contexts of this process, a pass queued on all of them in rank order."""
    S = J.stripes
    plan = S.plan_stripes(h, vs, n)
    active = [k for k, (a, b) in enumerate(plan) if b > a]
    mask = sum(1 << k for k in active)
    cap = 2 * max(1 << 16, w * h * 4)                        # two halves: one per pass in flight
    mailbox, gather = L.jpg_chain_mailbox_alloc(), L.jpg_device_alloc(0, cap)
    es, dev = {}, {}
    for k in active:
        es[k] = J.Encoder(ratio, algo)
        es[k].chain_setup(k, n, mailbox, gather, cap)
        dev[k] = torch.from_numpy(np.ascontiguousarray(S.stripe_rows(rows, h, plan[k][0], plan[k][1], bottom_up))).cuda()
    torch.cuda.synchronize()
    for k in active:
        es[k].stripe_device(dev[k].data_ptr(), w, plan[k][1] - plan[k][0], stride, bottom_up)
        es[k].chain_pass_async(mask)
    res = {k: es[k].chain_pass_finish() for k in reversed(active)}
    chain_scan.last = {"stripes": n, "rows": [plan[k] for k in active], "results": {k: res[k] for k in active}}
    total = res[0][4]
    out = np.zeros(total, np.uint8)
    if total:
        J._native.check(None, L.jpg_device_to_host(0, out.ctypes.data, gather, total))
    for e in es.values():
        e.close()
    L.jpg_device_free(0, gather)
    L.jpg_free(mailbox)
    return out


t0 = time.time()
done = fused = static = chained = 0
import ctypes as C  # noqa: E402
L = J._native.lib()
for it in range(cases):
    ratio = RATIOS[int(rng.integers(3))]
    algo, ref_algo = ALGOS[int(rng.integers(2))]
    wide = rng.random() < 0.6
    w = int(rng.integers(512, 2600)) if wide else int(rng.integers(1, 512))
    h = int(rng.integers(1, 120 if wide else 400))
    if wide and rng.random() < 0.1:
        h = int(rng.integers(1024, 1500))                       # tall enough for several bands of MCU rows
    hs, vs = J.get_downsampling_factor(ratio)
    if (hs == 2 and w % 16 == 1) or (vs == 2 and h % 16 == 1):
        w += 1 if hs == 2 and w % 16 == 1 else 0
        h += 1 if vs == 2 and h % 16 == 1 else 0
    stride = J.host.bmp_stride(w) + 4 * int(rng.integers(0, 4)) + (int(rng.integers(0, 4)) if rng.random() < 0.2 else 0)
    bottom_up = bool(rng.integers(2))
    kind = int(rng.integers(5))
    rows = content(w, h, kind, stride)
    want = refcpu.RefImage.from_bgr(rows, w, h, stride, ratio, ref_algo, bottom_up=bottom_up).encode().stuffed_scan()
    enc = encs.setdefault((ratio, algo), J.Encoder(ratio, algo))
    mode = int(rng.integers(6))
    if os.environ.get("FUZZ_CHAIN_BIAS") and rng.random() < 0.5:
        mode = 5                                                # (more of the cases through the stripe chain)
    tile_w = 1024 if (hs, vs) == (2, 1) else 512
    if mode == 5 and w >= tile_w:
        got = chain_scan(rows, w, h, stride, ratio, algo, vs, bottom_up, int(rng.integers(1, 7)))
        chained += 1
    elif mode == 3 or mode == 5:
        # MCU-row stripes on separate contexts, stitched
        n = int(rng.integers(2, 6))
        jobs = [J.stripes.StripeJob(J.Encoder(ratio, algo), k, n, rows, w, h, stride, vs, bottom_up) for k in range(n)]
        dcs = [j.transform() for j in jobs]
        codes = [j.code(dcs) for j in jobs]
        got = np.concatenate([j.place(codes) for j in jobs])
        for j in jobs:
            j.backend.close()
    elif mode == 4 and stride % 4 == 0 and L.jpg_static_scan_capacity(w, h, *ratio) and w * h < 3000000:
        # the encode that allocates nothing
        cap = L.jpg_static_scan_capacity(w, h, *ratio)
        t = torch.from_numpy(rows).cuda()
        out = torch.empty(cap, dtype=torch.uint8, device="cuda")
        res = torch.zeros(32, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        st = L.jpg_encode_device_static(0, J.jpeg_image.DCT_ALGORITHMS[algo], *ratio, t.data_ptr(), w, h, stride, int(bottom_up),
                                        out.data_ptr(), cap, res.data_ptr(), None)
        torch.cuda.synchronize()
        r = J._native.StaticResult.from_buffer_copy(res.cpu().numpy().tobytes())
        if st != 0 or L.jpg_static_status(C.byref(r)) != 0:
            got = np.zeros(0, np.uint8) if st != -12 else want     # (too large for the working array: not a case)
        else:
            got = out[:r.scan_bytes].cpu().numpy()
            static += 1
    elif mode == 0 or mode == 4 or stride % 4:
        if rng.random() < 0.3:
            os.environ["JPG_BANDS"] = str(int(rng.integers(1, 7)))
        got = enc.encode_scan(rows, w, h, stride, bottom_up)
        os.environ.pop("JPG_BANDS", None)
    elif mode == 1:
        t = torch.from_numpy(rows).cuda()
        torch.cuda.synchronize()
        enc.encode_device_async(t.data_ptr(), w, h, stride, bottom_up)
        _, n = enc.finish()
        got = enc.scan(n)
    else:
        n_img = int(rng.integers(2, 6))
        others = [content(w, h, int(rng.integers(5)), stride) for _ in range(n_img - 1)]
        k = int(rng.integers(n_img))
        imgs = others[:k] + [rows] + others[k:]
        dev = [torch.from_numpy(r).cuda() for r in imgs]
        torch.cuda.synchronize()
        enc.encode_batch_device_async([t.data_ptr() for t in dev], w, h, stride, bottom_up)
        enc.finish()
        got = enc.batch_scan(k)
    if not np.array_equal(got, want):
        print("MISMATCH", dict(ratio=ratio, algo=algo, w=w, h=h, stride=stride, bottom_up=bottom_up, kind=kind, mode=mode),
              len(got), len(want), getattr(chain_scan, "last", None) if mode == 5 else "", flush=True)
        # the case, for a closer look: rows, what came out, what was expected
        np.savez_compressed(os.path.join(ROOT, "gpurun_out", "fuzz_mismatch.npz"), rows=rows, got=got, want=want,
                            params=np.array([w, h, stride, int(bottom_up), kind, mode, *ratio]))
        sys.exit(1)
    done += 1
    fused += int(enc.fused_coding())
    if it % 25 == 24:
        print("%d cases ok (%d through the fused coding kernel), %.0f s" % (done, fused, time.time() - t0), flush=True)
print("fuzz ok: %d cases, %d through the fused coding kernel, %d through the entry point that allocates nothing, %d through the stripe chain"
      % (done, fused, static, chained))
