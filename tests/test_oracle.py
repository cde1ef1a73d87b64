"""CPU tests of the oracle (oracle/refcpu.c): the reference's own test vectors, known answers of the
restated arithmetic, frozen outputs, and an independent decoder (Pillow) on its files."""
import hashlib
import io
import json
import os
import re

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = json.load(open(os.path.join(HERE, "golden", "reference_vectors.json")))
GOLD = json.load(open(os.path.join(HERE, "golden", "oracle_vectors.json")))


# ---- vectors the reference's own tests hold

def test_wikipedia_block_real(oracle):
    v = REF["wikipedia_block"]
    for mode in (oracle.COS_CALL, oracle.COS_TABLE):
        got = oracle.real_dct_block(v["input"], cos_mode=mode)
        assert np.abs(got - np.array(v["expected"])).mean() <= v["mean_abs_error_threshold"]
        # the exact answer of the reference's arithmetic (f32, truncating division)
        assert got[:12].tolist() == [-25, -2, -6, 1, 2, 0, 0, 0, 0, -1, -4, 0]


def test_wikipedia_block_bin(oracle):
    v = REF["wikipedia_block"]
    got = oracle.bin_dct_block(v["input"])
    assert np.abs(got - np.array(v["expected"])).mean() <= v["mean_abs_error_threshold"]
    assert got[:16].tolist() == [-51, -6, -8, 3, 2, -1, 0, 0, 0, -6, -7, 1, 0, 0, 0, 0]
    assert oracle.bin_dct_block_raw(v["input"])[:8].tolist() == [-826, -67, -81, 58, 56, -79, 3, -5]


def test_block_walk(oracle):
    v = REF["block_walk"]
    m = np.array(v["matrix"], np.int32)
    out = oracle.block_walk_max(m, v["first"]["block_width"], v["first"]["block_height"], v["first"]["padding"])
    m[v["then_set"]["row"], v["then_set"]["col"]] = v["then_set"]["value"]
    out += oracle.block_walk_max(m, v["second"]["block_width"], v["second"]["block_height"], v["second"]["padding"])
    assert out == v["expected"]


# ---- known answers of the restated arithmetic

def test_dc_scale_truncation(oracle):
    flat = np.full(64, 160, np.uint8)
    assert oracle.real_dct_block(flat)[0] == 15          # 0.25*a0*a0 is 0x3DFFFFFF, then truncation
    assert oracle.bin_dct_block(flat)[0] == 32
    zero = np.zeros(64, np.uint8)                        # a padding block
    assert oracle.real_dct_block(zero)[0] == -63
    assert oracle.bin_dct_block(zero)[0] == -128
    assert not oracle.real_dct_block(flat)[1:].any()


@pytest.mark.parametrize("rgb,ycc", [((255, 255, 255), (255, 128, 128)), ((255, 0, 0), (76, 84, 255)),
                                     ((0, 255, 0), (149, 43, 21)), ((0, 0, 255), (29, 255, 107)),
                                     ((128, 128, 128), (128, 128, 127)), ((0, 0, 0), (0, 128, 128))])
def test_colour_known_answers(oracle, rgb, ycc):
    assert oracle.rgb_to_ycbcr(*rgb) == ycc


COS_HEX = """3f800000 3f800000 3f800000 3f800000 3f800000 3f800000 3f800000 3f800000
3f7b14be 3f54db31 3f0e39d9 3e47c5bc be47c5c2 bf0e39dc bf54db32 bf7b14bf
3f6c835e 3ec3ef15 bec3ef18 bf6c8360 bf6c835e bec3ef0b 3ec3ef1b 3f6c835f
3f54db31 be47c5c2 bf7b14bf bf0e39d6 3f0e39dd 3f7b14be 3e47c5b1 bf54db34
3f3504f3 bf3504f3 bf3504f1 3f3504f7 3f3504f3 bf3504fb bf3504ef 3f3504f4
3f0e39d9 bf7b14bf 3e47c5c8 3f54db2d bf54db34 be47c5bb 3f7b14bf bf0e39e4
3ec3ef15 bf6c835e 3f6c835f bec3ef25 bec3ef06 3f6c835b bf6c8362 3ec3ef25
3e47c5bc bf0e39d6 3f54db2d bf7b14bd 3f7b14c1 bf54db31 3f0e39e9 be47c614"""


def test_cos_table_bits(oracle):
    """cosf of the reference's f32 arguments (src/jpeg/dct_quant.rs:205-216), as bit patterns."""
    want = np.array([int(x, 16) for x in COS_HEX.split()], np.uint32).reshape(8, 8)
    got = oracle.cos_table().view(np.uint32)
    assert (got == want).all()


def test_device_cos_table_matches(oracle):
    """The table compiled into the kernels is the same bit patterns."""
    txt = open(os.path.join(ROOT, "jpeg-encoder_amd", "csrc", "transform.hpp")).read()
    body = txt[txt.index("COS_BITS"):]
    body = body[body.index("{"):body.index("};")]
    vals = [int(x, 16) for x in re.findall(r"0x([0-9a-fA-F]{8})", body)]
    assert len(vals) == 64
    assert (np.array(vals, np.uint32).reshape(8, 8) == oracle.cos_table().view(np.uint32)).all()


def test_tables(oracle):
    zz = oracle.zigzag()
    assert sorted(zz.tolist()) == list(range(64)) and zz[:6].tolist() == [0, 1, 8, 16, 9, 2]
    assert oracle.quant_table(0)[:8].tolist() == [16, 11, 10, 16, 24, 40, 51, 61]
    assert oracle.quant_table(1)[:8].tolist() == [17, 18, 24, 47, 99, 99, 99, 99]
    # canonical Annex-K codes (table ids: 0 YDC, 1 CHDC, 2 YAC, 3 CHAC)
    assert oracle.huffman_code(0, 0) == (0b00, 2)
    assert oracle.huffman_code(1, 0) == (0b00, 2)
    assert oracle.huffman_code(2, 0x00) == (0b1010, 4)
    assert oracle.huffman_code(3, 0x00) == (0b00, 2)
    assert oracle.huffman_code(2, 0xF0) == (0b11111111001, 11)
    assert oracle.huffman_code(3, 0xF0) == (0b1111111010, 10)
    assert oracle.huffman_code(2, 0x0B) is None


def test_ratio_errors(oracle):
    assert oracle.downsampling_factor(4, 4, 4) == (1, 1)
    assert oracle.downsampling_factor(4, 2, 2) == (2, 1)
    assert oracle.downsampling_factor(4, 2, 0) == (2, 2)
    with pytest.raises(oracle.RefError, match="Invalid chrominance downsampling ratio!"):
        oracle.downsampling_factor(4, 1, 1)


# ---- frozen outputs

def _image(case):
    from tests.golden.make_golden import image, SIZES
    k = SIZES.index((case["width"], case["height"]))
    return image(case["width"], case["height"], 1000 + k)


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize("case", GOLD["cases"], ids=lambda c: "%dx%d-%s-%s" % (
    c["width"], c["height"], "".join(map(str, c["ratio"])), c["dct"]))
def test_frozen_outputs(oracle, case):
    rows, stride = _image(case)
    algo = oracle.BIN if case["dct"] == "bin-dct" else oracle.REAL
    for mode in (oracle.COS_CALL, oracle.COS_TABLE):     # per-term cosf and the 64-entry table agree
        im = oracle.RefImage.from_bgr(rows, case["width"], case["height"], stride, tuple(case["ratio"]), algo)
        im.encode(cos_mode=mode)
        bits, nbits = im.bits()
        assert nbits == case["nbits"]
        assert [_sha(im.plane2d(c)) for c in range(3)] == case["planes_sha256"]
        assert [_sha(im.coeffs(c)) for c in range(3)] == case["coeffs_sha256"]
        assert _sha(bits) == case["bits_sha256"]
        assert _sha(im.stuffed_scan()) == case["scan_sha256"]
        data = im.file_bytes()
        assert len(data) == case["file_len"] and hashlib.sha256(data).hexdigest() == case["file_sha256"]
        if "y_coeffs" in case:
            assert im.coeffs(0).reshape(-1).tolist() == case["y_coeffs"]


def test_file_layout(oracle):
    """Marker order and fixed overhead (src/jpeg/file.rs:10-137): 623 header bytes, scan, EOI."""
    from tests.golden.make_golden import image
    rows, stride = image(48, 32, 1001)
    im = oracle.RefImage.from_bgr(rows, 48, 32, stride, (4, 2, 0), oracle.REAL).encode()
    data = im.file_bytes()
    scan = im.stuffed_scan().tobytes()
    assert data[:2] == b"\xff\xd8" and data[-2:] == b"\xff\xd9"
    assert len(data) == 623 + len(scan) + 2 and data[623:-2] == scan
    assert data[2:6] == b"\xff\xe0\x00\x10" and data[6:11] == b"JFIF\0"
    sof = data.index(b"\xff\xc0")
    assert data[sof:sof + 19] == bytes([0xff, 0xc0, 0, 17, 8, 0, 32, 0, 48, 3, 1, 0x22, 0, 2, 0x11, 1, 3, 0x11, 1])
    assert data[623 - 14:623] == bytes([0xff, 0xda, 0, 12, 3, 1, 0x00, 2, 0x11, 3, 0x11, 0, 0x3f, 0])
    # every 0xFF inside the scan is followed by a stuffed zero
    b = np.frombuffer(scan, np.uint8)
    ff = np.nonzero(b == 0xff)[0]
    assert (ff + 1 < len(b)).all() and (b[ff + 1] == 0).all()
    # un-stuffing gives the raw bits back
    bits, nbits = im.bits()
    keep = np.ones(len(b), bool)
    keep[ff + 1] = False
    assert (b[keep] == bits).all() and len(bits) == (nbits + 7) // 8


@pytest.mark.parametrize("ratio", [(4, 4, 4), (4, 2, 2), (4, 2, 0)])
@pytest.mark.parametrize("algo", ["real-dct", "bin-dct"])
def test_pillow_decodes(oracle, ratio, algo):
    """An independent decoder accepts the files; real-dct is a faithful JPEG, bin-dct (no scaling) a dim one."""
    from PIL import Image
    from tests.golden.make_golden import image
    w, h = 130, 67
    rows, stride = image(w, h, 1005)
    a = oracle.BIN if algo == "bin-dct" else oracle.REAL
    if ratio == (4, 2, 0) or ratio == (4, 2, 2):
        assert w % 16 != 1
    im = oracle.RefImage.from_bgr(rows, w, h, stride, ratio, a).encode()
    dec = Image.open(io.BytesIO(im.file_bytes()))
    dec.load()
    assert dec.size == (w, h) and dec.mode == "RGB"
    got = np.asarray(dec, np.float64)
    src = rows[::-1, :3 * w].reshape(h, w, 3)[:, :, ::-1].astype(np.float64)   # bottom-up BGR -> top-down RGB
    mse = ((got - src) ** 2).mean()
    psnr = 10 * np.log10(255.0 ** 2 / mse)
    assert psnr > (15.0 if algo == "real-dct" else 8.0), psnr   # (a busy, noisy test image)


def test_set_coeffs_feeds_the_entropy_coder(oracle):
    """The test aid ref_set_coeffs: the entropy coder run on coefficients put back unchanged gives the same bits;
    with one coefficient changed, different ones; wrong dimensions are refused."""
    import jpeg_encoder_amd as J
    rows = J.host.synth_bgr(48, 32, seed=9)
    a = oracle.RefImage.from_bgr(rows, 48, 32, rows.shape[1], (4, 2, 0), oracle.BIN).encode()
    b = oracle.RefImage.from_bgr(rows, 48, 32, rows.shape[1], (4, 2, 0), oracle.REAL)
    b.chrominance_downsampling()
    for c in range(3):
        b.set_coeffs(c, a.coeffs(c))
    b.generate_entropy_encoded_data()
    assert np.array_equal(a.stuffed_scan(), b.stuffed_scan()) and a.bits()[1] == b.bits()[1]
    y = a.coeffs(0).copy()
    y[0, 0] += 3
    b.set_coeffs(0, y).generate_entropy_encoded_data()
    assert not np.array_equal(a.bits()[0], b.bits()[0])
    with pytest.raises(oracle.RefError):
        b.set_coeffs(1, a.coeffs(0))
