"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Bar: bit-exact for every integer/byte stage, for bin-dct and for real-dct-strict;
for the separable real-dct, quantised coefficients within +-1 in at most 0.01 %
of coefficients (checked on images large enough for the rate to mean something).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RATIOS = [(4, 4, 4), (4, 2, 2), (4, 2, 0)]
ALGOS = ["bin-dct", "real-dct-strict", "real-dct"]
REF_ALGO = {"bin-dct": 1, "real-dct-strict": 0, "real-dct": 0}
# (W, H): aligned, ragged (zero padding in both directions), wider than one 512-pixel tile
SIZES = [(16, 16), (48, 32), (50, 35), (531, 77), (1040, 24)]


def make_rows(J, w, h, seed, noise=False):
    return J.host.synth_bgr(w, h, seed, noise_only=noise)


def ref_encode(oracle, rows, w, h, ratio, algo):
    im = oracle.RefImage.from_bgr(rows, w, h, rows.shape[1], ratio, REF_ALGO[algo])
    planes_full = [im.plane2d(c).copy() for c in range(3)]
    im.chrominance_downsampling()
    planes_down = [im.plane2d(c).copy() for c in range(3)]
    im.dct_and_quantization()
    coeffs = [im.coeffs(c) for c in range(3)]
    im.generate_entropy_encoded_data()
    bits, nbits = im.bits()
    return dict(planes_full=planes_full, planes_down=planes_down, coeffs=coeffs, bits=bits, nbits=nbits,
                scan=im.stuffed_scan(), file=im.file_bytes())


@pytest.mark.parametrize("ratio", RATIOS)
@pytest.mark.parametrize("size", SIZES)
@pytest.mark.parametrize("noise", [False, True])
def test_staged_stages_bit_exact(J, oracle, ratio, size, noise):
    w, h = size
    rows = make_rows(J, w, h, 7 * w + h, noise)
    for algo in ("bin-dct", "real-dct-strict"):
        ref = ref_encode(oracle, rows, w, h, ratio, algo)
        im = J.JpegImage.from_bgr_rows(rows, w, h, None, ratio, algo)
        for c in range(3):
            np.testing.assert_array_equal(im._enc.plane(c), ref["planes_full"][c], err_msg="plane %d" % c)
        im.chrominance_downsampling()
        for c in range(3):
            np.testing.assert_array_equal(im._enc.plane(c), ref["planes_down"][c], err_msg="down plane %d" % c)
        im.dct_and_quantization()
        for c in range(3):
            np.testing.assert_array_equal(im._enc.coeffs(c), ref["coeffs"][c], err_msg="%s coeffs %d" % (algo, c))
        im.generate_entropy_encoded_data()
        bits, nbits = im.entropy_coded_bits
        assert nbits == ref["nbits"]
        np.testing.assert_array_equal(bits, ref["bits"])
        assert im.file_bytes() == ref["file"]


@pytest.mark.parametrize("ratio", RATIOS)
@pytest.mark.parametrize("size", SIZES)
@pytest.mark.parametrize("algo", ["bin-dct", "real-dct-strict"])
def test_fused_file_bit_exact(J, oracle, ratio, size, algo):
    w, h = size
    rows = make_rows(J, w, h, 11 * w + 3 * h)
    ref = ref_encode(oracle, rows, w, h, ratio, algo)
    enc = J.Encoder(ratio, algo)
    scan = enc.encode_scan(rows, w, h)
    np.testing.assert_array_equal(scan, ref["scan"])
    for c in range(3):
        np.testing.assert_array_equal(enc.coeffs(c), ref["coeffs"][c])
    assert enc.encode_file_bytes(rows, w, h) == ref["file"]


@pytest.mark.parametrize("ratio", RATIOS)
def test_real_dct_tolerance(J, oracle, ratio):
    """Separable real-dct vs the reference's summation order: +-1 in <= 0.01 % of coefficients."""
    w, h = 1024, 512
    total = bad = 0
    for noise in (False, True):
        rows = make_rows(J, w, h, 99, noise)
        ref = ref_encode(oracle, rows, w, h, ratio, "real-dct")
        enc = J.Encoder(ratio, "real-dct")
        enc.encode_scan(rows, w, h)
        for c in range(3):
            d = np.abs(enc.coeffs(c).astype(np.int32) - ref["coeffs"][c].astype(np.int32))
            assert d.max() <= 1
            total += d.size
            bad += int((d != 0).sum())
    print("real-dct mismatches: %d of %d (%.5f %%)" % (bad, total, 100.0 * bad / total))
    assert bad <= 1e-4 * total, (bad, total)
