"""Direct links between the device path and the oracle where earlier tests only had a chain: the oracle's entropy
coder run on the DEVICE's separable real-dct coefficients must give the coding kernel's scan; the 1920x1080 4:2:2
batch with the separable transform as it is benchmarked (64 images per launch); and the 32768x32768 image against
an oracle run of the whole image.  All through the C ABI."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _two_kernel_encoder(J, *a, **k):
    os.environ["JPG_TWO_KERNELS"] = "1"
    try:
        return J.Encoder(*a, **k)
    finally:
        del os.environ["JPG_TWO_KERNELS"]


def _oracle_scan_of(oracle, enc, rows, w, h, ratio):
    """The oracle's generate_entropy_encoded_data + stuffing applied to the coefficients the device holds."""
    ref = oracle.RefImage.from_bgr(rows, w, h, rows.shape[1], ratio, oracle.REAL)
    ref.chrominance_downsampling(threads=2)
    for c in range(3):
        ref.set_coeffs(c, enc.coeffs(c))
    ref.generate_entropy_encoded_data()
    return ref.stuffed_scan(), ref.bits()[1]


@pytest.mark.parametrize("w,h,ratio", [(7680, 4320, (4, 2, 0)), (3840, 2160, (4, 4, 4)), (2048, 1080, (4, 2, 2)),
                                       (1544, 72, (4, 2, 0))])
def test_oracle_entropy_coder_on_device_coefficients(J, oracle, w, h, ratio):
    """real-dct (separable): the coding kernel's bytes == the oracle's entropy coder and stuffing run on the
    coefficients the device computed (which the tolerance tests compare with the oracle's own)."""
    rows = J.host.synth_bgr(w, h, seed=(3 << 32) | w)
    enc = J.Encoder(ratio, "real-dct")
    got = enc.encode_scan(rows, w, h)            # host rows: the context keeps its own copy for coeffs()
    assert enc.fused_coding()
    want, want_bits = _oracle_scan_of(oracle, enc, rows, w, h, ratio)
    assert got.size == want.size
    np.testing.assert_array_equal(got, want)


def test_config_batch_1920x1080_422_real_dct_as_benchmarked(J, oracle):
    """64 images in one launch set of the fused coding kernel with the separable transform: every image's scan ==
    the kernel pair's, == the oracle's entropy coder on the device's coefficients; and on a sample of the images
    the coefficients are within +-1 of the oracle's in at most 0.01 % of places."""
    import torch
    w, h, ratio, n = 1920, 1080, (4, 2, 2), 64
    enc = J.Encoder(ratio, "real-dct")
    assert n <= enc.max_batch()
    imgs = [J.host.synth_bgr(w, h, seed=(4 << 32) | (1000 + i), noise_only=(i % 11 == 5)) for i in range(n)]
    dev = [torch.from_numpy(r).cuda() for r in imgs]
    torch.cuda.synchronize()
    stride = imgs[0].shape[1]
    enc.encode_batch_device_async([d.data_ptr() for d in dev], w, h, stride, True)
    assert enc.fused_coding()
    enc.finish()
    scans = [enc.batch_scan(i) for i in range(n)]
    two = _two_kernel_encoder(J, ratio, "real-dct")
    total = bad = 0
    for i in range(n):
        pair = two.encode_scan(imgs[i], w, h)
        assert not two.fused_coding()
        np.testing.assert_array_equal(scans[i], pair, err_msg="image %d: coding kernel vs kernel pair" % i)
        if i % 8 == 0:
            want, _ = _oracle_scan_of(oracle, two, imgs[i], w, h, ratio)
            np.testing.assert_array_equal(scans[i], want, err_msg="image %d: oracle coder on device coefficients" % i)
            ref = oracle.RefImage.from_bgr(imgs[i], w, h, stride, ratio, oracle.REAL).encode(cos_mode=oracle.COS_TABLE, threads=3)
            for c in range(3):
                d = np.abs(two.coeffs(c).astype(np.int32) - ref.coeffs(c).astype(np.int32))
                assert d.max() <= 1
                total += d.size
                bad += int((d != 0).sum())
    print("1920x1080 4:2:2 real-dct mismatches: %d of %d (%.5f %%)" % (bad, total, 100.0 * bad / total))
    assert bad <= 1e-4 * total          # tolerance: +-1 in at most 0.01 % of the coefficients


def test_config_gigapixel_against_the_oracle(J, oracle):
    """32768x32768 4:2:0 bin-dct (131072 tiles, 25 million blocks, more than 2^31 bits of worst-case stream): the
    whole-image encode and the eight-stripe encode, byte for byte against an oracle run of the whole image."""
    S = J.stripes
    tile = J.host.synth_bgr(8192, 4096, seed=(5 << 32) | 1)[:, :3 * 8192]
    rows = np.tile(tile, (8, 4))
    w = h = 32768
    ref = oracle.RefImage.from_bgr(rows, w, h, rows.shape[1], (4, 2, 0), oracle.BIN).encode(threads=3)
    want = ref.stuffed_scan()
    want_bits = ref.bits()[1]
    del ref
    enc = J.Encoder((4, 2, 0), "bin-dct")
    whole = enc.encode_scan(rows, w, h)
    assert enc.fused_coding()
    enc.close()
    assert whole.size == want.size and np.array_equal(whole, want)
    n = 8
    jobs = [S.StripeJob(J.Encoder((4, 2, 0), "bin-dct"), k, n, rows, w, h, rows.shape[1], 2) for k in range(n)]
    dcs = [j.transform() for j in jobs]
    codes = [j.code(dcs) for j in jobs]
    assert sum(c[0] for c in codes) == want_bits
    got = np.concatenate([j.place(codes) for j in jobs])
    assert got.size == want.size and np.array_equal(got, want)
