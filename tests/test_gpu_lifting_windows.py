"""GPU parity cases added with the f32 lifting transform, the two bit windows of the coding kernel, and the
fixes to jpg_stream_tail, the stuffing grid and the row-offset range; all through the C ABI, against the oracle."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RATIOS = [(4, 4, 4), (4, 2, 2), (4, 2, 0)]


def _ref(oracle, rows, w, h, ratio, algo):
    a = oracle.BIN if algo == "bin-dct" else oracle.REAL
    return oracle.RefImage.from_bgr(rows, w, h, rows.shape[1], ratio, a).encode()


def _block_image(J, blocks):
    """An image whose luma and chroma blocks take the given 8x8 sample blocks as directly as the colour
    conversion allows: grey pixels (r = g = b = v) give y = v or v - 1, cb, cr = 128 or 127."""
    n = len(blocks)
    w, h = 8 * n, 8
    stride = J.host.bmp_stride(w)
    rows = np.zeros((h, stride), np.uint8)
    for k, b in enumerate(blocks):
        for y in range(8):
            for x in range(8):
                rows[h - 1 - y, 3 * (8 * k + x):3 * (8 * k + x) + 3] = b[y][x]
    return rows, w, h


@pytest.mark.parametrize("ratio", RATIOS)
def test_bin_dct_extreme_blocks_through_every_kernel(J, oracle, ratio):
    """The f32 formulation of the integer lifting transform (floor by fused multiply-add under round-down, reciprocal
    quantiser) against the oracle's i32 code on blocks that drive its intermediate values to their extremes:
    coefficients bit-exact through the staged kernel, the front kernel and the coding kernel."""
    rng = np.random.default_rng(11)
    blocks = []
    for k in range(160):
        kind = k % 8
        if kind == 0:
            b = rng.integers(0, 2, (8, 8)) * 255
        elif kind == 1:
            b = np.indices((8, 8)).sum(0) % 2 * 255
        elif kind == 2:
            b = np.where(np.arange(8)[None, :] < (k // 8) % 8 + 1, 255, 0) * np.ones((8, 1), int)
        elif kind == 3:
            b = np.where(np.arange(8)[:, None] < (k // 8) % 8 + 1, 0, 255) * np.ones((1, 8), int)
        elif kind == 4:
            b = np.zeros((8, 8), int)
            b[(k // 8) % 8, (k // 64) % 8] = 255
        elif kind == 5:
            b = rng.integers(0, 256, (8, 8))
        elif kind == 6:
            b = (np.arange(8)[None, :] * 36 + rng.integers(0, 4, (8, 8))) % 256
        else:
            b = np.full((8, 8), (k * 37) % 256)
        blocks.append(b.astype(np.uint8))
    rows, w, h = _block_image(J, blocks)
    # wide enough for the coding kernel (1280 pixels), a few MCU rows of the same content shifted
    rows = np.concatenate([np.roll(rows, 24 * s, axis=1) for s in range(4)], axis=0)
    h *= 4
    ref = _ref(oracle, rows, w, h, ratio, "bin-dct")
    enc = J.Encoder(ratio, "bin-dct")
    scan = enc.encode_scan(rows, w, h)                     # coding kernel
    assert enc.fused_coding()
    np.testing.assert_array_equal(scan, ref.stuffed_scan())
    for c in range(3):                                     # front kernel (coefficients asked for after the encode)
        np.testing.assert_array_equal(enc.coeffs(c), ref.coeffs(c))
    enc.load_bgr(rows, w, h)                               # staged kernels
    enc.downsample()
    enc.dct_quant()
    for c in range(3):
        np.testing.assert_array_equal(enc.coeffs(c), ref.coeffs(c))
    enc.close()


@pytest.mark.parametrize("window", ["small", "large", None])
@pytest.mark.parametrize("ratio", [(4, 4, 4), (4, 2, 0)])
def test_both_bit_windows_and_the_switch_between_them(J, oracle, monkeypatch, window, ratio):
    """bin-dct tiles of 512 pixels have two coding kernels (512- and 1536-word bit window); JPG_CODED_WINDOW pins
    one, otherwise a context picks per launch from the tile lengths of the encode before.  Photo-like content
    (one pass per tile) and noise (several passes with the small window) alternate, so the unpinned context
    switches back and forth; every scan must equal the oracle's."""
    if window:
        monkeypatch.setenv("JPG_CODED_WINDOW", window)
    else:
        monkeypatch.delenv("JPG_CODED_WINDOW", raising=False)
    w, h = 1100, 80
    enc = J.Encoder(ratio, "bin-dct")
    for it, noise in enumerate([False, True, True, False, False, True]):
        rows = J.host.synth_bgr(w, h, seed=30 + it, noise_only=noise)
        ref = _ref(oracle, rows, w, h, ratio, "bin-dct")
        np.testing.assert_array_equal(enc.encode_scan(rows, w, h), ref.stuffed_scan(), err_msg="encode %d" % it)
    enc.close()


def test_stream_tail_after_a_whole_encode(J, oracle):
    """jpg_stream_tail after jpg_encode_device_async + jpg_finish (no stripe call measured the tail) must report the
    last 32 bits of the stream, not what an earlier stripe left in the slot."""
    import torch
    w, h = 640, 96
    enc = J.Encoder((4, 2, 0), "bin-dct")
    # an earlier stripe-style encode in the same context leaves its tail in a slot's status record
    rows0 = J.host.synth_bgr(w, h, seed=90)
    enc.front_host(rows0, w, h)
    enc.entropy((0, 0, 0))
    stale = enc.stream_tail()
    for seed in (91, 92, 93, 94):
        rows = J.host.synth_bgr(w, h, seed=seed)
        t = torch.from_numpy(rows).cuda()
        torch.cuda.synchronize()
        enc.encode_device_async(t.data_ptr(), w, h, rows.shape[1], True)
        nbits, n = enc.finish()
        bits, nb = _ref(oracle, rows, w, h, (4, 2, 0), "bin-dct").bits()
        assert nb == nbits
        allbits = np.unpackbits(bits)[:nb]
        want = int("".join(map(str, allbits[-32:])), 2)
        got = enc.stream_tail()
        assert got == want, "seed %d: %08x != %08x (stale %08x)" % (seed, got, want, stale)
        got_bits = np.unpackbits(enc.bits(nbits))[:nb]
        np.testing.assert_array_equal(got_bits[-32:], allbits[-32:])
    enc.close()


def test_more_images_than_stuffing_workgroups(J, oracle, monkeypatch):
    """A batch with more images than the stuffing kernel is allowed workgroups (JPG_STUFF_GRID=8, 16 images): every
    image gets one workgroup instead of the launch being refused."""
    import torch
    monkeypatch.setenv("JPG_STUFF_GRID", "8")
    w, h, n = 640, 64, 16
    rows = [J.host.synth_bgr(w, h, seed=200 + k, noise_only=bool(k & 1)) for k in range(n)]
    dev = [torch.from_numpy(r).cuda() for r in rows]
    torch.cuda.synchronize()
    enc = J.Encoder((4, 2, 0), "bin-dct")
    enc.encode_batch_device_async([t.data_ptr() for t in dev], w, h, rows[0].shape[1], True)
    enc.finish_next()
    for k in range(n):
        np.testing.assert_array_equal(enc.batch_scan(k), _ref(oracle, rows[k], w, h, (4, 2, 0), "bin-dct").stuffed_scan(),
                                      err_msg="image %d" % k)
    enc.close()


@pytest.mark.parametrize("grid", [1, 3, 33, 40])
def test_single_image_stuffed_by_few_workgroups(J, oracle, monkeypatch, grid):
    """The stuffing kernel of a single image draws its segments in teams of 32 workgroups, three segments in
    work per workgroup: with 1, 3, 33 or 40 workgroups (one team, a team and a single workgroup, a team and a
    quarter) every workgroup takes many turns, and the scan must not change (noise: several hundred segments, many
    0xFF bytes)."""
    import torch
    monkeypatch.setenv("JPG_STUFF_GRID", str(grid))
    w, h = 2048, 1200
    rows = J.host.synth_bgr(w, h, seed=300 + grid, noise_only=True)
    t = torch.from_numpy(rows).cuda()
    torch.cuda.synchronize()
    enc = J.Encoder((4, 2, 0), "bin-dct")
    want = _ref(oracle, rows, w, h, (4, 2, 0), "bin-dct").stuffed_scan()
    assert want.size > 300 * 4096
    for _ in range(2):                                        # (the second launch finds the counters as the first left them)
        enc.encode_device_async(t.data_ptr(), w, h, rows.shape[1], True)
        _, n = enc.finish()
        np.testing.assert_array_equal(enc.scan(n), want)
    enc.close()


def test_row_stride_beyond_the_offset_range_is_refused(J):
    """The kernels address a tile's rows with 32-bit offsets: a stride at which 16 rows reach 2^31 bytes is refused
    (JPG_ERR_TOO_LARGE) before anything is launched."""
    L = J._native.lib()
    enc = J.Encoder((4, 2, 0), "bin-dct")
    assert L.jpg_encode_device_async(enc._ctx, C.c_void_p(256), 16, 16, 1 << 27, 1) == -12
    assert L.jpg_front_device(enc._ctx, C.c_void_p(256), 16, 16, 1 << 27, 1) == -12
    enc.close()
