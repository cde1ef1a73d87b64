"""The named configurations at their full sizes (GPU): byte-exact bin-dct at 3840x2160 4:4:4, the
coefficient tolerance and PSNR bar at 7680x4320 4:2:0 real-dct, and a 1920x1080 4:2:2 batch
(the 32768x32768 image: tests/test_gpu_oracle_links.py)."""
import io
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _two_kernel_encoder(J, *a, **k):
    """A context whose whole encodes run the front + entropy kernel pair instead of the fused coding kernel."""
    os.environ["JPG_TWO_KERNELS"] = "1"
    try:
        return J.Encoder(*a, **k)
    finally:
        del os.environ["JPG_TWO_KERNELS"]


def _tolerance(enc, ref):
    """(mismatching, total) quantised coefficients; every difference must be +-1."""
    total = bad = 0
    for c in range(3):
        d = np.abs(enc.coeffs(c).astype(np.int32) - ref.coeffs(c).astype(np.int32))
        assert d.max() <= 1
        total += d.size
        bad += int((d != 0).sum())
    return bad, total


def test_config_512x512_420_real_dct(J, oracle):
    """The reference's CPU-runnable configuration: strict bytes == oracle, the separable transform within
    +-1 in at most 0.01 % of coefficients (per-term cosf in the oracle, as the reference evaluates it), and the
    fused coding kernel == the kernel pair."""
    w = h = 512
    rows = J.host.synth_bgr(w, h, seed=(1 << 32))
    ref = oracle.RefImage.from_bgr(rows, w, h, rows.shape[1], (4, 2, 0), oracle.REAL)
    ref.encode(cos_mode=oracle.COS_CALL, threads=3)
    strict = J.Encoder((4, 2, 0), "real-dct-strict")
    assert strict.encode_file_bytes(rows, w, h) == ref.file_bytes()
    assert strict.fused_coding()
    for c in range(3):
        np.testing.assert_array_equal(strict.coeffs(c), ref.coeffs(c))
    enc = J.Encoder((4, 2, 0), "real-dct")
    data = enc.encode_file_bytes(rows, w, h)
    bad, total = _tolerance(enc, ref)
    print("512x512 real-dct mismatches: %d of %d" % (bad, total))
    assert bad <= 1e-4 * total
    assert abs(_psnr(data, rows, w, h) - _psnr(ref.file_bytes(), rows, w, h)) < 0.01
    two = _two_kernel_encoder(J, (4, 2, 0), "real-dct")
    assert two.encode_file_bytes(rows, w, h) == data and not two.fused_coding()


def test_config_3840x2160_444_bin_dct_byte_exact(J, oracle):
    w, h = 3840, 2160
    rows = J.host.synth_bgr(w, h, seed=(2 << 32))
    ref = oracle.RefImage.from_bgr(rows, w, h, rows.shape[1], (4, 4, 4), oracle.BIN).encode(threads=3)
    enc = J.Encoder((4, 4, 4), "bin-dct")
    assert enc.encode_file_bytes(rows, w, h) == ref.file_bytes()


def _psnr(data, rows, w, h):
    from PIL import Image
    im = Image.open(io.BytesIO(data))
    im.load()
    got = np.asarray(im, np.float32)
    src = rows[::-1, :3 * w].reshape(h, w, 3)[:, :, ::-1].astype(np.float32)
    return 10 * np.log10(255.0 ** 2 / float(((got - src) ** 2).mean(dtype=np.float64)))


def test_config_7680x4320_420_real_dct_tolerance_and_psnr(J, oracle):
    """Quantised coefficients within +-1 in at most 0.01 % of coefficients; decoded PSNR within 0.01 dB."""
    w, h = 7680, 4320
    rows = J.host.synth_bgr(w, h, seed=(3 << 32))
    ref = oracle.RefImage.from_bgr(rows, w, h, rows.shape[1], (4, 2, 0), oracle.REAL)
    ref.encode(cos_mode=oracle.COS_TABLE, threads=3)
    enc = J.Encoder((4, 2, 0), "real-dct")
    data = enc.encode_file_bytes(rows, w, h)
    assert enc.fused_coding()
    bad, total = _tolerance(enc, ref)
    print("8K real-dct mismatches: %d of %d (%.5f %%)" % (bad, total, 100.0 * bad / total))
    assert bad <= 1e-4 * total
    p_gpu, p_ref = _psnr(data, rows, w, h), _psnr(ref.file_bytes(), rows, w, h)
    print("PSNR device %.4f dB, oracle %.4f dB" % (p_gpu, p_ref))
    assert abs(p_gpu - p_ref) < 0.01
    # the benchmarked kernel (pixels to bits in one launch) against the kernel pair whose coefficients were
    # just compared with the oracle's: same bytes, from host rows and as a device-resident batch of two
    two = _two_kernel_encoder(J, (4, 2, 0), "real-dct")
    assert two.encode_file_bytes(rows, w, h) == data and not two.fused_coding()
    import torch
    dev = torch.from_numpy(rows).cuda()
    torch.cuda.synchronize()
    enc.encode_batch_device_async([dev.data_ptr(), dev.data_ptr()], w, h, rows.shape[1], True)
    enc.finish()
    scan = np.frombuffer(data, np.uint8)[623:-2]
    for i in range(2):
        np.testing.assert_array_equal(enc.batch_scan(i), scan)
    del dev
    # the strict variant reproduces the oracle's file byte for byte at this size too
    strict = J.Encoder((4, 2, 0), "real-dct-strict")
    assert strict.encode_file_bytes(rows, w, h) == ref.file_bytes()


@pytest.mark.parametrize("algo,n,per_launch", [("bin-dct", 40, 20), ("real-dct-strict", 16, 16)])
def test_config_batch_1920x1080_422(J, oracle, algo, n, per_launch):
    """The image-sharded configuration as one rank runs it: launch sets of equally sized 1920x1080 4:2:2 images
    (1024-pixel tiles, several images per launch of the fused coding kernel), two sets in flight; every image
    byte for byte against the oracle."""
    import torch
    w, h = 1920, 1080
    enc = J.Encoder((4, 2, 2), algo)
    imgs = [J.host.synth_bgr(w, h, seed=(4 << 32) | i, noise_only=(i % 7 == 3)) for i in range(n)]
    dev = [torch.from_numpy(r).cuda() for r in imgs]
    torch.cuda.synchronize()
    stride = imgs[0].shape[1]
    sets = [list(range(k, min(n, k + per_launch))) for k in range(0, n, per_launch)]
    for ids in sets:
        enc.encode_batch_device_async([dev[i].data_ptr() for i in ids], w, h, stride, True)
    assert enc.fused_coding()
    ref_algo = oracle.BIN if algo == "bin-dct" else oracle.REAL
    for ids in sets:
        enc.finish_next()
        for j, i in enumerate(ids):
            ref = oracle.RefImage.from_bgr(imgs[i], w, h, stride, (4, 2, 2), ref_algo).encode(threads=3)
            nbits, ln, _ = enc.batch_result(j)
            assert nbits == ref.bits()[1]
            np.testing.assert_array_equal(enc.batch_scan(j), ref.stuffed_scan(), err_msg="image %d" % i)
