"""The single pixels-to-bits kernel (whole encodes of images at least one tile wide) against the oracle
and against the front + entropy kernel pair it replaces, through the C ABI: edge tiles, tiles whose
bits exceed the assembly window, the dense bit string, batches, and what the getters do afterwards."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RATIOS = [(4, 4, 4), (4, 2, 2), (4, 2, 0)]
REF_ALGO = {"real-dct-strict": "REAL", "bin-dct": "BIN"}   # names of the oracle module's constants
# one full tile exactly; full tiles + a partial one; a partial last MCU column and row; several tile rows
SIZES = [(512, 16), (1000, 37), (1544, 72), (2048 + 24, 40)]


def tile_width(ratio):
    """Pixels per tile; images at least that wide go through the fused coding kernel."""
    return 1024 if tuple(ratio) == (4, 2, 2) else 512


@pytest.fixture()
def two_kernel_encoder(J):
    """Encoder factory whose contexts never use the fused coding kernel."""
    def make(*a, **k):
        os.environ["JPG_TWO_KERNELS"] = "1"
        try:
            return J.Encoder(*a, **k)
        finally:
            del os.environ["JPG_TWO_KERNELS"]
    return make


@pytest.mark.parametrize("ratio", RATIOS)
@pytest.mark.parametrize("size", SIZES)
@pytest.mark.parametrize("noise", [False, True])
def test_fused_coding_matches_oracle_and_two_kernels(J, oracle, two_kernel_encoder, ratio, size, noise):
    w, h = size
    rows = J.host.synth_bgr(w, h, 31 * w + h, noise_only=noise)
    for algo in ("bin-dct", "real-dct-strict"):
        ref = oracle.RefImage.from_bgr(rows, w, h, rows.shape[1], ratio, getattr(oracle, REF_ALGO[algo]))
        ref.encode()
        enc = J.Encoder(ratio, algo)
        scan = enc.encode_scan(rows, w, h)
        assert enc.fused_coding() == (w >= tile_width(ratio))
        np.testing.assert_array_equal(scan, ref.stuffed_scan(), err_msg="%s scan" % algo)
        want_bits, want_nbits = ref.bits()
        nbits = want_nbits
        np.testing.assert_array_equal(enc.bits(nbits), want_bits)     # gathered from the tile slots
        for c in range(3):                                             # computed on request afterwards
            np.testing.assert_array_equal(enc.coeffs(c), ref.coeffs(c))
        two = two_kernel_encoder(ratio, algo)
        np.testing.assert_array_equal(two.encode_scan(rows, w, h), scan)
        assert not two.fused_coding()
    # real-dct (the separable transform) has no oracle bytes: the two device paths must still agree
    enc, two = J.Encoder(ratio, "real-dct"), two_kernel_encoder(ratio, "real-dct")
    np.testing.assert_array_equal(enc.encode_scan(rows, w, h), two.encode_scan(rows, w, h))


def test_narrow_images_keep_the_kernel_pair(J):
    enc = J.Encoder((4, 2, 0), "bin-dct")
    rows = J.host.synth_bgr(496, 64, 5)
    enc.encode_scan(rows, 496, 64)
    assert not enc.fused_coding()
    enc = J.Encoder((4, 4, 4), "bin-dct")
    enc.encode_scan(J.host.synth_bgr(512, 8, 5), 512, 8)
    assert enc.fused_coding()
    enc = J.Encoder((4, 2, 2), "bin-dct")                        # 4:2:2 tiles are 1024 pixels wide
    enc.encode_scan(J.host.synth_bgr(1008, 8, 5), 1008, 8)
    assert not enc.fused_coding()
    enc.encode_scan(J.host.synth_bgr(1024, 8, 5), 1024, 8)
    assert enc.fused_coding()


def test_fused_batch_and_pipeline(J, oracle):
    """Batches and back-to-back encodes on one context; images differ, so a mixed-up slot would show."""
    import torch
    w, h, ratio = 1032, 56, (4, 2, 0)
    rows = [J.host.synth_bgr(w, h, 100 + i, noise_only=(i % 3 == 0)) for i in range(7)]
    want = []
    for r in rows:
        ref = oracle.RefImage.from_bgr(r, w, h, r.shape[1], ratio, oracle.BIN)
        ref.encode()
        want.append(ref.stuffed_scan())
    dev = [torch.from_numpy(r).cuda() for r in rows]
    torch.cuda.synchronize()
    stride = rows[0].shape[1]
    enc = J.Encoder(ratio, "bin-dct")
    enc.encode_batch_device_async([t.data_ptr() for t in dev], w, h, stride, True)
    enc.finish()
    for i in range(len(rows)):
        np.testing.assert_array_equal(enc.batch_scan(i), want[i])
    # three single encodes in flight, taken in order
    for i in range(3):
        enc.encode_device_async(dev[i].data_ptr(), w, h, stride, True)
    for i in range(3):
        _, n = enc.finish_next()
        np.testing.assert_array_equal(enc.scan(n), want[i])


def test_coefficients_after_device_input(J):
    """Rows the caller owns may be gone after the encode: the getters refuse instead of reading them."""
    import torch
    w, h = 1024, 32
    rows = J.host.synth_bgr(w, h, 9)
    t = torch.from_numpy(rows).cuda()
    torch.cuda.synchronize()
    enc = J.Encoder((4, 2, 0), "bin-dct")
    enc.encode_device_async(t.data_ptr(), w, h, rows.shape[1], True)
    enc.finish()
    with pytest.raises(J.JpgError, match="stage called out of order"):
        enc.coeffs(0)
    enc.front_device(t.data_ptr(), w, h, rows.shape[1], True)
    host = J.Encoder((4, 2, 0), "bin-dct")
    host.encode_scan(rows, w, h)
    for c in range(3):
        np.testing.assert_array_equal(enc.coeffs(c), host.coeffs(c))


@pytest.mark.parametrize("ratio", RATIOS)
def test_fused_top_down_rows_and_padded_stride(J, oracle, ratio):
    """Row order and row pitch are the caller's: top-down rows, extra bytes after every row, device-resident
    rows read in place with that pitch."""
    import torch
    w, h = 1096, 83
    base = J.host.synth_bgr(w, h, seed=21)
    stride = base.shape[1] + 20
    rows = np.zeros((h, stride), np.uint8)
    rows[:, :base.shape[1]] = base
    dev = torch.from_numpy(rows).cuda()
    torch.cuda.synchronize()
    for bottom_up in (True, False):
        ref = oracle.RefImage.from_bgr(rows, w, h, stride, ratio, oracle.BIN, bottom_up=bottom_up).encode()
        enc = J.Encoder(ratio, "bin-dct")
        np.testing.assert_array_equal(enc.encode_scan(rows, w, h, stride, bottom_up), ref.stuffed_scan())
        assert enc.fused_coding()
        enc.encode_device_async(dev.data_ptr(), w, h, stride, bottom_up)
        _, n = enc.finish()
        np.testing.assert_array_equal(enc.scan(n), ref.stuffed_scan())


def test_fused_extreme_content(J, oracle):
    """Checkerboards and saturated noise at 4:4:4: long blocks, ZRL runs, many 0xFF bytes, tiles of several
    window passes; the coder's range errors never fire on 8-bit input."""
    w, h = 1024, 48
    rng = np.random.default_rng(3)
    stride = J.host.bmp_stride(w)
    imgs = []
    chk = np.zeros((h, stride), np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    chk[:, :3 * w] = np.repeat((((xx + yy) & 1) * 255).astype(np.uint8), 3, axis=1)
    imgs.append(chk)
    sat = np.zeros((h, stride), np.uint8)
    sat[:, :3 * w] = rng.choice(np.array([0, 255], np.uint8), size=(h, 3 * w))
    imgs.append(sat)
    imp = np.zeros((h, stride), np.uint8)
    imp[::8, 21:3 * w:24] = 255
    imgs.append(imp)
    for rows in imgs:
        for algo, ref_algo in (("bin-dct", oracle.BIN), ("real-dct-strict", oracle.REAL)):
            ref = oracle.RefImage.from_bgr(rows, w, h, stride, (4, 4, 4), ref_algo).encode()
            enc = J.Encoder((4, 4, 4), algo)
            np.testing.assert_array_equal(enc.encode_scan(rows, w, h), ref.stuffed_scan())


@pytest.mark.parametrize("ratio,algo,w,h,n", [
    ((4, 2, 0), "bin-dct", 1032, 200, 3), ((4, 2, 2), "real-dct-strict", 1040, 120, 4), ((4, 4, 4), "bin-dct", 1544, 70, 5),
])
def test_stripes_through_the_fused_kernel(J, oracle, ratio, algo, w, h, n):
    """Stripes of host rows: the rows wait in the context until the stripe is coded, in one kernel, from the
    predictors the stripes above hand over; the last DCs come from the last MCU's pixels."""
    import torch
    S = J.stripes
    rows = J.host.synth_bgr(w, h, seed=17, noise_only=True)
    stride = rows.shape[1]
    _, v = J.get_downsampling_factor(ratio)
    jobs = [S.StripeJob(J.Encoder(ratio, algo), k, n, rows, w, h, stride, v) for k in range(n)]
    dcs = [j.transform() for j in jobs]
    # the same values from the coefficients of a front-kernel run over device-resident rows
    for j, got in zip(jobs, dcs):
        sub = np.ascontiguousarray(S.stripe_rows(rows, h, j.y0, j.y1))
        t = torch.from_numpy(sub).cuda()
        torch.cuda.synchronize()
        enc = J.Encoder(ratio, algo)
        enc.front_device(t.data_ptr(), w, j.y1 - j.y0, stride, True)
        assert tuple(enc.last_dc()) == tuple(got)
        enc.sync()
    codes = [j.code(dcs) for j in jobs]
    pieces = [j.place(codes) for j in jobs]
    want = oracle.RefImage.from_bgr(rows, w, h, stride, ratio, oracle.BIN if algo == "bin-dct" else oracle.REAL).encode()
    assert sum(c[0] for c in codes) == want.bits()[1]
    np.testing.assert_array_equal(np.concatenate(pieces), want.stuffed_scan())


@pytest.mark.parametrize("ratio", RATIOS)
def test_fused_flat_image_many_tiles_per_segment(J, oracle, ratio):
    """A black image codes to the minimum number of bits per block, so a 4 KiB segment of the stream spans
    as many tile slots as it ever can (full tiles alternating with a one-MCU tile)."""
    w, h = 2048 + 16, 1600
    stride = J.host.bmp_stride(w)
    rows = np.zeros((h, stride), np.uint8)
    ref = oracle.RefImage.from_bgr(rows, w, h, stride, ratio, oracle.BIN).encode(threads=3)
    enc = J.Encoder(ratio, "bin-dct")
    scan = enc.encode_scan(rows, w, h)
    assert enc.fused_coding()
    np.testing.assert_array_equal(scan, ref.stuffed_scan())
    bits, nbits = ref.bits()
    np.testing.assert_array_equal(enc.bits(nbits), bits)


@pytest.mark.parametrize("ratio,algo,w,h,n", [
    ((4, 2, 0), "bin-dct", 1032, 200, 3), ((4, 2, 2), "real-dct-strict", 1040, 120, 4), ((4, 4, 4), "bin-dct", 600, 70, 2),
    ((4, 2, 0), "bin-dct", 300, 100, 3),                        # narrower than a tile: the kernel pair
])
def test_stripes_of_device_resident_rows(J, oracle, ratio, algo, w, h, n):
    """Stripes whose rows already sit in device memory (jpg_stripe_device): coded by the single kernel straight
    from the caller's rows, every step queued on all contexts before any is waited for (as one host thread drives
    several GPUs), the pieces gathered device to device at their byte offsets."""
    import ctypes as C
    import torch
    S = J.stripes
    L = J._native.lib()
    rows = J.host.synth_bgr(w, h, seed=23, noise_only=True)
    stride = rows.shape[1]
    _, v = J.get_downsampling_factor(ratio)
    plan = S.plan_stripes(h, v, n)
    dev = torch.from_numpy(rows).cuda()
    torch.cuda.synchronize()
    encs = [J.Encoder(ratio, algo) for _ in range(n)]
    for e, (y0, y1) in zip(encs, plan):
        e.stripe_device(dev.data_ptr() + (h - y1) * stride, w, y1 - y0, stride, True)
        assert e.fused_coding() == (w >= tile_width(ratio))
    for e in encs:
        e.last_dc_async()
    dcs = [e.last_dc() for e in encs]
    for k, e in enumerate(encs):
        e.entropy_stripe_async(dcs[k - 1] if k else (0, 0, 0))
    codes = [e.entropy_stripe_finish() for e in encs]
    for k, e in enumerate(encs):
        assert e.stream_tail() == codes[k][1]
    offsets, heads = S.stitch_plan([c[0] for c in codes], [c[1] for c in codes])
    for k, e in enumerate(encs):
        e.stitch_async(offsets[k], heads[k], k == n - 1)
    lens = [e.stitch_finish()[0] for e in encs]
    total = sum(lens)
    buf = L.jpg_device_alloc(0, total)
    assert buf
    for k, e in enumerate(encs):
        e.scan_copy_to_device(buf + sum(lens[:k]), 0, lens[k])
    for e in encs:
        e.sync()
    got = np.zeros(total, np.uint8)
    assert L.jpg_device_to_host(0, got.ctypes.data, buf, total) == 0
    L.jpg_device_free(0, buf)
    want = oracle.RefImage.from_bgr(rows, w, h, stride, ratio, oracle.BIN if algo == "bin-dct" else oracle.REAL).encode()
    assert sum(c[0] for c in codes) == want.bits()[1]
    np.testing.assert_array_equal(got, want.stuffed_scan())
    for c in range(3):                                            # coefficients on request still work
        first = oracle.RefImage.from_bgr(rows[h - plan[0][1]:], w, plan[0][1], stride, ratio,
                                         oracle.BIN if algo == "bin-dct" else oracle.REAL).encode()
        np.testing.assert_array_equal(encs[0].coeffs(c), first.coeffs(c))
