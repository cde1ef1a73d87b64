"""More GPU parity cases through the C ABI: golden fixtures, extreme content, error behaviour,
buffer reuse, device-pointer input, stripe mode, and size-independent properties at full size."""
import ctypes as C
import hashlib
import io
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "oracle_vectors.json")))
RATIOS = [(4, 4, 4), (4, 2, 2), (4, 2, 0)]


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _ref(oracle, rows, w, h, ratio, algo):
    a = oracle.BIN if algo == "bin-dct" else oracle.REAL
    return oracle.RefImage.from_bgr(rows, w, h, rows.shape[1], ratio, a).encode()


def _unstuff(scan):
    ff = np.nonzero(scan == 0xff)[0]
    assert (ff + 1 < len(scan)).all() and (scan[ff + 1] == 0).all()
    keep = np.ones(len(scan), bool)
    keep[ff + 1] = False
    return scan[keep]


# ---- committed fixtures

@pytest.mark.parametrize("case", [c for c in GOLD["cases"] if c["dct"] == "bin-dct"],
                         ids=lambda c: "%dx%d-%s" % (c["width"], c["height"], "".join(map(str, c["ratio"]))))
def test_golden_fixtures_bin_dct(J, case):
    from tests.golden.make_golden import image, SIZES
    w, h = case["width"], case["height"]
    rows, stride = image(w, h, 1000 + SIZES.index((w, h)))
    enc = J.Encoder(tuple(case["ratio"]), "bin-dct")
    data = enc.encode_file_bytes(rows, w, h)
    assert len(data) == case["file_len"] and hashlib.sha256(data).hexdigest() == case["file_sha256"]
    assert [_sha(enc.coeffs(c)) for c in range(3)] == case["coeffs_sha256"]


@pytest.mark.parametrize("case", [c for c in GOLD["cases"] if c["dct"] == "real-dct"],
                         ids=lambda c: "%dx%d-%s" % (c["width"], c["height"], "".join(map(str, c["ratio"]))))
def test_golden_fixtures_real_dct_strict(J, case):
    from tests.golden.make_golden import image, SIZES
    w, h = case["width"], case["height"]
    rows, stride = image(w, h, 1000 + SIZES.index((w, h)))
    enc = J.Encoder(tuple(case["ratio"]), "real-dct-strict")
    data = enc.encode_file_bytes(rows, w, h)
    assert hashlib.sha256(data).hexdigest() == case["file_sha256"]


def _colour_with_luma(oracle, v):
    """An (r, g, b) whose luma is exactly v in the reference's f32 formula (src/utils/colorspace.rs:5-15):
    grey v gives v or v - 1 (13 of the 256 grey levels lose one to truncation), so look around it."""
    for g in (v, v + 1, v - 1):
        if 0 <= g <= 255 and oracle.rgb_to_ycbcr(g, g, g)[0] == v:
            return (g, g, g)
    for dg in (0, 1, -1, 2, -2):
        for db in range(-8, 9):
            r, g, b = v, min(255, max(0, v + dg)), min(255, max(0, v + db))
            if oracle.rgb_to_ycbcr(r, g, b)[0] == v:
                return (r, g, b)
    raise AssertionError("no colour with luma %d" % v)


def test_wikipedia_block_on_device(J, oracle):
    """The reference's own DCT vector (src/jpeg/dct_quant.rs:242-322) through the device: pixels are chosen so
    that the device's luma plane IS the block, then the reference's bound (mean |error| <= 1, :255,280) and the
    oracle's exact coefficients are asserted for every transform, staged and fused."""
    ref = json.load(open(os.path.join(HERE, "golden", "reference_vectors.json")))["wikipedia_block"]
    blk = np.array(ref["input"], np.uint8).reshape(8, 8)
    rgb = np.array([_colour_with_luma(oracle, int(v)) for v in blk.reshape(-1)], np.uint8).reshape(8, 8, 3)
    rows = np.ascontiguousarray(rgb[::-1, :, ::-1]).reshape(8, 24)           # bottom-up, B,G,R
    first_rows = {"real-dct": [-25, -2, -6, 1, 2, 0, 0, 0, 0, -1, -4, 0, 0, 0, 0, 0],
                  "real-dct-strict": [-25, -2, -6, 1, 2, 0, 0, 0, 0, -1, -4, 0, 0, 0, 0, 0],
                  "bin-dct": [-51, -6, -8, 3, 2, -1, 0, 0, 0, -6, -7, 1, 0, 0, 0, 0]}
    for algo in ("real-dct", "real-dct-strict", "bin-dct"):
        im = J.JpegImage.from_bgr_rows(rows, 8, 8, None, (4, 4, 4), algo)
        np.testing.assert_array_equal(im.y_channel, blk)                    # the transform sees the block itself
        im.chrominance_downsampling()
        im.dct_and_quantization()
        got = im.y_dct_coeffs.reshape(-1).astype(np.int32)
        assert np.abs(got - np.array(ref["expected"])).mean() <= ref["mean_abs_error_threshold"]
        want = oracle.bin_dct_block(blk) if algo == "bin-dct" else oracle.real_dct_block(blk, cos_mode=oracle.COS_CALL)
        if algo == "real-dct":                                              # separable: +-1 at truncation boundaries
            assert np.abs(got - want).max() <= 1
        else:
            assert got[:16].tolist() == first_rows[algo]
            np.testing.assert_array_equal(got, want.astype(np.int32))
        enc = J.Encoder((4, 4, 4), algo)                                    # the fused kernel
        enc.encode_scan(rows, 8, 8)
        np.testing.assert_array_equal(enc.coeffs(0).reshape(-1), got)


# ---- extreme content: long blocks (more than 128 bits, several LDS passes), ZRL runs, 0xFF bytes

def _pattern(kind, w, h, rng):
    px = np.zeros((h, w, 3), np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    if kind == "checker":
        px[...] = (((xx + yy) & 1) * 255)[..., None]
    elif kind == "checker_rgb":
        px[..., 0] = ((xx & 1) * 255)
        px[..., 1] = ((yy & 1) * 255)
        px[..., 2] = (((xx + yy) & 1) * 255)
    elif kind == "impulse":
        px[3::8, 5::8] = 255
    elif kind == "white":
        px[...] = 255
    elif kind == "black":
        pass
    elif kind == "hf_corner":                                     # only the highest frequencies: ZRL then a value
        px[...] = (128 + 100 * np.cos(np.pi * (2 * xx + 1) * 7 / 16) * np.cos(np.pi * (2 * yy + 1) * 7 / 16)
                   ).clip(0, 255).astype(np.uint8)[..., None]
    elif kind == "noise":
        px[...] = rng.integers(0, 256, (h, w, 3))
    elif kind == "saturated_noise":
        px[...] = rng.integers(0, 2, (h, w, 3)) * 255
    stride = (3 * w + 3) // 4 * 4
    rows = np.zeros((h, stride), np.uint8)
    rows[:, :3 * w] = px[::-1, :, ::-1].reshape(h, 3 * w)
    return rows


@pytest.mark.parametrize("kind", ["checker", "checker_rgb", "impulse", "white", "black", "hf_corner", "noise",
                                  "saturated_noise"])
@pytest.mark.parametrize("ratio", RATIOS)
def test_extreme_content_bit_exact(J, oracle, kind, ratio):
    w, h = 200, 120
    rows = _pattern(kind, w, h, np.random.default_rng(3))
    for algo in ("bin-dct", "real-dct-strict"):
        ref = _ref(oracle, rows, w, h, ratio, "bin-dct" if algo == "bin-dct" else "real-dct")
        enc = J.Encoder(ratio, algo)
        scan = enc.encode_scan(rows, w, h)
        np.testing.assert_array_equal(scan, ref.stuffed_scan(), err_msg="%s %s" % (kind, algo))
        bits, nbits = ref.bits()
        np.testing.assert_array_equal(enc.bits(nbits), bits)    # dense stream after a fused encode


def test_medium_image_all_paths(J, oracle):
    """1920x1080 (not a multiple of 16 in height): fused file, dense bits, planes; bin-dct is exact."""
    w, h = 1920, 1080
    rows = J.host.synth_bgr(w, h, seed=21)
    for ratio in RATIOS:
        ref = _ref(oracle, rows, w, h, ratio, "bin-dct")
        enc = J.Encoder(ratio, "bin-dct")
        assert enc.encode_file_bytes(rows, w, h) == ref.file_bytes()


def test_context_reuse_across_sizes_and_epochs(J, oracle):
    """One context, many encodes of changing size: buffers regrow, tagged status words never go stale."""
    enc = J.Encoder((4, 2, 0), "bin-dct")
    rng = np.random.default_rng(0)
    for it in range(40):
        w, h = int(rng.integers(1, 300)), int(rng.integers(1, 200))
        if w % 16 == 1 or h % 16 == 1:
            continue
        rows = J.host.synth_bgr(w, h, seed=it, noise_only=bool(it & 1))
        np.testing.assert_array_equal(enc.encode_scan(rows, w, h), _ref(oracle, rows, w, h, (4, 2, 0), "bin-dct")
                                      .stuffed_scan(), err_msg="iteration %d (%dx%d)" % (it, w, h))


def test_device_pointer_input_and_async_finish(J, oracle):
    import torch
    w, h = 640, 360
    rows = J.host.synth_bgr(w, h, seed=4)
    t = torch.from_numpy(rows).cuda()
    torch.cuda.synchronize()
    enc = J.Encoder((4, 2, 2), "bin-dct")
    for _ in range(3):                                          # back-to-back queued encodes, one wait
        enc.encode_device_async(t.data_ptr(), w, h, rows.shape[1], True)
    nbits, n = enc.finish()
    ref = _ref(oracle, rows, w, h, (4, 2, 2), "bin-dct")
    assert nbits == ref.bits()[1]
    np.testing.assert_array_equal(enc.scan(n), ref.stuffed_scan())
    # the scan is also reachable as a device pointer
    out = torch.empty(n, dtype=torch.uint8, device="cuda")
    C.CDLL("libamdhip64.so.7").hipMemcpy(C.c_void_p(out.data_ptr()), C.c_void_p(enc.scan_device_ptr()),
                                                   C.c_size_t(n), C.c_int(3))
    np.testing.assert_array_equal(out.cpu().numpy(), ref.stuffed_scan())


def test_top_down_rows_and_padded_stride(J, oracle):
    w, h = 77, 45
    base = J.host.synth_bgr(w, h, seed=8)
    stride = base.shape[1] + 12                                  # extra bytes at the end of every row
    rows = np.zeros((h, stride), np.uint8)
    rows[:, :base.shape[1]] = base
    for bottom_up in (True, False):
        ref = oracle.RefImage.from_bgr(rows, w, h, stride, (4, 2, 0), oracle.BIN, bottom_up=bottom_up).encode()
        enc = J.Encoder((4, 2, 0), "bin-dct")
        np.testing.assert_array_equal(enc.encode_scan(rows, w, h, stride, bottom_up), ref.stuffed_scan())


# ---- errors

def test_errors(J):
    with pytest.raises(J.JpgError, match="Invalid chrominance downsampling ratio!"):
        J.Encoder((4, 1, 1), "bin-dct")
    enc = J.Encoder((4, 2, 0), "bin-dct")
    rows = J.host.synth_bgr(33, 20, seed=1)
    with pytest.raises(J.JpgError, match="unsupported dimensions") as e:      # width % 16 == 1 with h = 2
        enc.encode_scan(rows, 33, 20)
    assert e.value.code == -9
    rows = J.host.synth_bgr(32, 17, seed=1)
    with pytest.raises(J.JpgError) as e:                                      # height % 16 == 1 with v = 2
        enc.encode_scan(rows, 32, 17)
    assert e.value.code == -9
    enc444 = J.Encoder((4, 4, 4), "bin-dct")
    enc444.encode_scan(J.host.synth_bgr(33, 17, seed=1), 33, 17)              # fine without subsampling
    with pytest.raises(J.JpgError) as e:                                      # stages out of order
        enc.downsample()
    assert e.value.code == -8
    with pytest.raises(J.JpgError) as e:
        enc.finish()
    assert e.value.code == -8
    L = J._native.lib()
    assert L.jpg_load_bgr(enc._ctx, None, 8, 8, 24, 1) == -6
    rows = J.host.synth_bgr(8, 8, seed=1)
    assert L.jpg_load_bgr(enc._ctx, rows.ctypes.data, 8, 8, 20, 1) == -6      # stride shorter than a row
    assert L.jpg_load_bgr(enc._ctx, rows.ctypes.data, 0, 8, 24, 1) == -6
    enc.encode_scan(rows, 8, 8)
    small = np.zeros(1, np.uint8)
    assert L.jpg_get_scan(enc._ctx, small.ctypes.data, 1) == -11              # caller buffer too small


# ---- stripes on the device

@pytest.mark.parametrize("ratio,algo,w,h,n,noise", [
    ((4, 2, 0), "bin-dct", 640, 480, 2, False), ((4, 2, 0), "bin-dct", 330, 250, 3, True),
    ((4, 2, 2), "bin-dct", 256, 200, 8, True), ((4, 4, 4), "real-dct-strict", 100, 90, 4, False),
    ((4, 2, 0), "bin-dct", 64, 16, 3, True),                    # empty stripes
])
def test_stripes_match_whole_image(J, oracle, ratio, algo, w, h, n, noise):
    S = J.stripes
    rows = J.host.synth_bgr(w, h, seed=5, noise_only=noise)
    stride = rows.shape[1]
    _, v = J.get_downsampling_factor(ratio)
    jobs = [S.StripeJob(J.Encoder(ratio, algo), k, n, rows, w, h, stride, v) for k in range(n)]
    dcs = [j.transform() for j in jobs]
    codes = [j.code(dcs) for j in jobs]
    pieces = [j.place(codes) for j in jobs]
    want = _ref(oracle, rows, w, h, ratio, "bin-dct" if algo == "bin-dct" else "real-dct")
    assert sum(c[0] for c in codes) == want.bits()[1]
    np.testing.assert_array_equal(np.concatenate(pieces), want.stuffed_scan())
    hf, vf = J.get_downsampling_factor(ratio)
    assert J.host.assemble_jpeg(w, h, hf, vf, np.concatenate(pieces)) == want.file_bytes()


# ---- full size: properties that need no oracle run

def test_full_size_properties(J):
    """7680x4320 4:2:0: fused == staged coefficients (bin-dct), un-stuffing the scan gives the dense
    stream, the stripe cut reproduces the scan, and an independent decoder accepts the file."""
    from PIL import Image
    w, h = 7680, 4320
    rows = J.host.synth_bgr(w, h, seed=(3 << 32))
    enc = J.Encoder((4, 2, 0), "bin-dct")
    scan = enc.encode_scan(rows, w, h)
    fused = [enc.coeffs(c).copy() for c in range(3)]
    im = J.JpegImage.from_bgr_rows(rows, w, h, None, (4, 2, 0), "bin-dct")
    im.chrominance_downsampling()
    im.dct_and_quantization()
    for c, name in enumerate(("y", "cb", "cr")):
        np.testing.assert_array_equal(getattr(im, name + "_dct_coeffs"), fused[c])
    im.generate_entropy_encoded_data()
    bits, nb = im.entropy_coded_bits
    np.testing.assert_array_equal(_unstuff(scan), bits)
    # two stripes on two contexts
    S = J.stripes
    jobs = [S.StripeJob(J.Encoder((4, 2, 0), "bin-dct"), k, 2, rows, w, h, rows.shape[1], 2) for k in range(2)]
    dcs = [j.transform() for j in jobs]
    codes = [j.code(dcs) for j in jobs]
    assert sum(c[0] for c in codes) == nb
    np.testing.assert_array_equal(np.concatenate([j.place(codes) for j in jobs]), scan)
    # real-dct file through Pillow
    enc = J.Encoder((4, 2, 0), "real-dct")
    data = enc.encode_file_bytes(rows, w, h)
    dec = Image.open(io.BytesIO(data))
    dec.draft("RGB", (w // 8, h // 8))
    dec.load()
    small = np.asarray(dec, np.float64)
    src = rows[::-1, :3 * w].reshape(h, w, 3)[:, :, ::-1]
    k = h // small.shape[0]
    ref_small = src[:small.shape[0] * k, :small.shape[1] * k].reshape(small.shape[0], k, small.shape[1], k, 3).mean((1, 3))
    psnr = 10 * np.log10(255.0 ** 2 / ((small - ref_small) ** 2).mean())
    assert psnr > 28, psnr


# ---- stripes across processes (two ranks sharing the one GPU of the test box, gloo for the scalars)

def _stripe_rank(rank, world, port, tmp):
    import os
    import sys
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, os.path.dirname(HERE))
    import torch.distributed as dist
    import jpeg_encoder_amd as J
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        w, h, ratio = 1000, 700, (4, 2, 0)
        rows = J.host.synth_bgr(w, h, seed=31, noise_only=True)
        enc = J.Encoder(ratio, "bin-dct", device=0)
        scan = J.stripes.encode_striped(enc, rows, w, h, rows.shape[1], 2, device="cpu")
        if rank == 0:
            whole = J.Encoder(ratio, "bin-dct", device=0).encode_scan(rows, w, h)
            assert np.array_equal(scan, whole)
            open(os.path.join(tmp, "ok"), "w").write("%d" % len(scan))
        else:
            assert scan is None
    finally:
        dist.destroy_process_group()


def test_stripes_two_processes(tmp_path):
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_stripe_rank, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert int(open(tmp_path / "ok").read()) > 0


def test_page_locked_input_and_large_pageable_input(J, oracle):
    """Rows in page-locked memory go by DMA, pageable rows through the bounce buffers (several chunks
    here, with a row stride that is not the packed pitch): same bytes either way."""
    L = J._native.lib()
    w, h = 2500, 1400                                            # 10.5 MB of rows: two bounce chunks
    base = J.host.synth_bgr(w, h, seed=77)
    ref = oracle.RefImage.from_bgr(base, w, h, base.shape[1], (4, 2, 0), oracle.BIN).encode(threads=3).stuffed_scan()
    enc = J.Encoder((4, 2, 0), "bin-dct")
    np.testing.assert_array_equal(enc.encode_scan(base, w, h), ref)
    wide = np.zeros((h, base.shape[1] + 6), np.uint8)            # misaligned stride -> re-pitched while staging
    wide[:, :base.shape[1]] = base
    np.testing.assert_array_equal(enc.encode_scan(wide, w, h, wide.shape[1]), ref)
    p = L.jpg_host_alloc(base.nbytes)
    assert p
    try:
        C.memmove(p, base.ctypes.data, base.nbytes)
        pinned = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), (h, base.shape[1]))
        np.testing.assert_array_equal(enc.encode_scan(pinned, w, h), ref)
    finally:
        enc.close()
        L.jpg_free(p)


def test_kernel_time_stamps(J):
    """jpg_set_timing / jpg_front_time: sampled start/end stamps of the fused kernel."""
    import torch
    w, h = 1920, 1080
    rows = J.host.synth_bgr(w, h, seed=2)
    t = torch.from_numpy(rows).cuda()
    torch.cuda.synchronize()
    enc = J.Encoder((4, 2, 0), "real-dct")
    enc.set_timing(2)
    for _ in range(10):
        enc.encode_device_async(t.data_ptr(), w, h, rows.shape[1], True)
        enc.finish()
    ms, n, images = enc.front_time()
    assert n == 5 and images == 5 and 0.0005 < ms / n < 5.0
    enc.set_timing(0)
    enc.encode_device_async(t.data_ptr(), w, h, rows.shape[1], True)
    enc.finish()
    assert enc.front_time() == (0.0, 0, 0)


@pytest.mark.parametrize("depth", [1, 2, 3])
def test_pipelined_encodes_keep_their_results(J, oracle, depth):
    """Several encodes in flight on one context (stuffing on a second stream, rotating output
    buffers): every image comes out byte-exact, in order, whatever the sizes."""
    import torch
    rng = np.random.default_rng(depth)
    jobs = []
    for i in range(9):
        w, h = int(rng.integers(2, 40)) * 16, int(rng.integers(2, 30)) * 16
        rows = J.host.synth_bgr(w, h, seed=100 + i, noise_only=bool(i % 3 == 0))
        jobs.append((w, h, rows, torch.from_numpy(rows).cuda()))
    torch.cuda.synchronize()
    enc = J.Encoder((4, 2, 0), "bin-dct")
    got, inflight = [], 0
    for w, h, rows, t in jobs:
        if inflight == depth:
            nbits, n = enc.finish_next()
            got.append((nbits, enc.scan(n)))
            inflight -= 1
        enc.encode_device_async(t.data_ptr(), w, h, rows.shape[1], True)
        inflight += 1
    while inflight:
        nbits, n = enc.finish_next()
        got.append((nbits, enc.scan(n)))
        inflight -= 1
    assert len(got) == len(jobs)
    for (w, h, rows, _), (nbits, scan) in zip(jobs, got):
        ref = _ref(oracle, rows, w, h, (4, 2, 0), "bin-dct")
        assert nbits == ref.bits()[1]
        np.testing.assert_array_equal(scan, ref.stuffed_scan())
    with pytest.raises(J.JpgError):
        enc.finish_next()                                       # nothing queued


@pytest.mark.parametrize("ratio,algo,w,h,n", [((4, 2, 2), "bin-dct", 320, 200, 7), ((4, 2, 0), "bin-dct", 1920, 1080, 16),
                                             ((4, 4, 4), "real-dct-strict", 100, 60, 64), ((4, 2, 0), "bin-dct", 48, 40, 1),
                                             ((4, 2, 0), "real-dct-strict", 512, 96, 256)])
def test_batches_match_single_images(J, oracle, ratio, algo, w, h, n):
    """n equally sized images through one launch of each kernel: every image byte-exact; pipelined batches too."""
    import torch
    imgs = [J.host.synth_bgr(w, h, seed=500 + i, noise_only=bool(i % 4 == 1)) for i in range(n)]
    dev = [torch.from_numpy(r).cuda() for r in imgs]
    torch.cuda.synchronize()
    enc = J.Encoder(ratio, algo)
    assert enc.max_batch() >= 64
    ptrs = [t.data_ptr() for t in dev]
    enc.encode_batch_device_async(ptrs, w, h, imgs[0].shape[1], True)
    enc.encode_batch_device_async(ptrs[::-1], w, h, imgs[0].shape[1], True)     # a second batch in flight
    refs = [_ref(oracle, r, w, h, ratio, "bin-dct" if algo == "bin-dct" else "real-dct") for r in imgs]
    for order in (list(range(n)), list(range(n))[::-1]):
        enc.finish_next()
        for i, k in enumerate(order):
            nbits, ln, ptr = enc.batch_result(i)
            assert nbits == refs[k].bits()[1] and ptr
            np.testing.assert_array_equal(enc.batch_scan(i), refs[k].stuffed_scan(), err_msg="image %d" % k)
    # a single image afterwards on the same context
    np.testing.assert_array_equal(enc.encode_scan(imgs[0], w, h), refs[0].stuffed_scan())
    with pytest.raises(J.JpgError):
        enc.encode_batch_device_async(ptrs * (enc.max_batch() // n + 1), w, h, imgs[0].shape[1], True)  # more than jpg_max_batch


@pytest.mark.parametrize("ratio,algo,w,h", [
    ((4, 2, 0), "bin-dct", 1024, 1600),          # 100 MCU rows: four bands
    ((4, 2, 0), "real-dct-strict", 1000, 777),   # ragged height and width, two bands
    ((4, 4, 4), "bin-dct", 600, 1000),           # 125 MCU rows of 8
    ((4, 2, 2), "bin-dct", 1100, 300),
])
def test_host_rows_coded_band_by_band(J, oracle, monkeypatch, ratio, algo, w, h):
    """jpg_encode_scan of rows in host memory copies and codes the image in bands of MCU rows; begin/end leave
    room for the caller's own work in between.  Pageable and page-locked rows, bottom-up and top-down, a pitch
    that is no multiple of four: all the oracle's bytes."""
    import ctypes as C
    L = J._native.lib()
    a = oracle.BIN if algo == "bin-dct" else oracle.REAL
    rows = J.host.synth_bgr(w, h, seed=61, noise_only=True)
    stride = rows.shape[1]
    want = oracle.RefImage.from_bgr(rows, w, h, stride, ratio, a).encode().stuffed_scan()
    enc = J.Encoder(ratio, algo)
    for rep in range(3):                                      # (later calls reuse the context's buffers)
        if rep == 2:
            monkeypatch.setenv("JPG_BANDS", "5")              # pageable rows are one band unless told otherwise
        np.testing.assert_array_equal(enc.encode_scan(rows, w, h), want)
    monkeypatch.delenv("JPG_BANDS")
    # page-locked rows, in two steps
    ptr = L.jpg_host_alloc(rows.nbytes)
    C.memmove(ptr, rows.ctypes.data, rows.nbytes)
    enc._ck(L.jpg_encode_scan_begin(enc._ctx, ptr, w, h, stride, 1))
    header = J.host.jfif_header(w, h, enc.h, enc.v)           # (the host's own work meanwhile)
    p, n = C.c_void_p(), C.c_size_t()
    enc._ck(L.jpg_encode_scan_end(enc._ctx, C.byref(p), C.byref(n)))
    got = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), (n.value,)).copy()
    L.jpg_free(p)
    L.jpg_free(ptr)
    np.testing.assert_array_equal(got, want)
    assert len(header) == 623
    assert L.jpg_encode_scan_end(enc._ctx, C.byref(p), C.byref(n)) < 0      # nothing begun
    for c in range(3):                                        # coefficients on request still work afterwards
        np.testing.assert_array_equal(enc.coeffs(c), oracle.RefImage.from_bgr(rows, w, h, stride, ratio, a).encode().coeffs(c))
    # top-down rows with a pitch of 3*w + 1
    odd = 3 * w + 1
    td = np.zeros((h, odd), np.uint8)
    td[:, :3 * w] = rows[::-1, :3 * w]
    np.testing.assert_array_equal(enc.encode_scan(td, w, h, odd, bottom_up=False), want)
    enc.close()
