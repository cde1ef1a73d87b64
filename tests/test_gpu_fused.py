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
        assert enc.fused_coding()
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
