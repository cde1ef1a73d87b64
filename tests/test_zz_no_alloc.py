"""jpg_encode_device_static, the encode that allocates nothing: its scan must equal the one of
jpg_encode_device_async + jpg_finish (the entry point it is a variant of) byte for byte, on two calls in a row
through the C ABI, the second with a smaller image (the working array keeps what the first call left)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def encode_static(J, torch, rows, w, h, ratio, algo, stream=None):
    N = J._native
    L = N.lib()
    cap = L.jpg_static_scan_capacity(w, h, *ratio)
    assert cap > 0
    dev = torch.from_numpy(rows).cuda()
    out = torch.full((cap,), 0xA5, dtype=torch.uint8, device="cuda")
    res = torch.zeros(32, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    st = L.jpg_encode_device_static(0, J.jpeg_image.DCT_ALGORITHMS[algo], *ratio, dev.data_ptr(), w, h, rows.shape[1], 1,
                                    out.data_ptr(), cap, res.data_ptr(), stream)
    assert st == 0, st
    torch.cuda.synchronize()
    r = N.StaticResult.from_buffer_copy(res.cpu().numpy().tobytes())
    assert L.jpg_static_status(C.byref(r)) == 0
    return r.nbits, out[:r.scan_bytes].cpu().numpy(), out[r.scan_bytes:r.scan_bytes + 64].cpu().numpy()


def encode_context(J, torch, rows, w, h, ratio, algo):
    dev = torch.from_numpy(rows).cuda()
    torch.cuda.synchronize()
    enc = J.Encoder(ratio, algo)
    enc.encode_device_async(dev.data_ptr(), w, h, rows.shape[1], True)
    nbits, n = enc.finish()
    scan = enc.scan(n)
    enc.close()
    return nbits, scan


@pytest.mark.parametrize("ratio,algo,sizes", [
    ((4, 2, 0), "bin-dct", [(1920, 1080), (1000, 300)]),
    ((4, 2, 2), "real-dct-strict", [(1920, 1080), (1100, 64)]),
    ((4, 4, 4), "bin-dct", [(1544, 600), (600, 72)]),
    ((4, 2, 0), "real-dct", [(2048, 2048), (520, 50)]),
])
def test_static_encode_equals_the_context_encode_twice(J, oracle, ratio, algo, sizes):
    import torch
    assert J._native.lib().jpg_static_init(0) == 0
    for k, (w, h) in enumerate(sizes):                    # the second image is smaller than the first
        rows = J.host.synth_bgr(w, h, seed=300 + k, noise_only=(k == 0))
        nbits, scan, after = encode_static(J, torch, rows, w, h, ratio, algo)
        want_bits, want = encode_context(J, torch, rows, w, h, ratio, algo)
        assert nbits == want_bits
        np.testing.assert_array_equal(scan, want)
        assert (after == 0xA5).all()                      # nothing written past the scan
        if algo != "real-dct" and w * h <= 700000:        # and the oracle's bytes, where it is quick
            ref = oracle.RefImage.from_bgr(rows, w, h, rows.shape[1], ratio,
                                           oracle.BIN if algo == "bin-dct" else oracle.REAL).encode()
            np.testing.assert_array_equal(scan, ref.stuffed_scan())


def test_static_encode_on_a_stream_of_the_caller_and_its_limits(J):
    import torch
    N = J._native
    L = N.lib()
    s = torch.cuda.Stream()
    w, h, ratio = 1024, 512, (4, 2, 0)
    rows = J.host.synth_bgr(w, h, seed=9)
    nbits, scan, _ = encode_static(J, torch, rows, w, h, ratio, "bin-dct", stream=s.cuda_stream)
    want_bits, want = encode_context(J, torch, rows, w, h, ratio, "bin-dct")
    assert nbits == want_bits
    np.testing.assert_array_equal(scan, want)
    buf = torch.zeros(1 << 20, dtype=torch.uint8, device="cuda")
    res = torch.zeros(32, dtype=torch.uint8, device="cuda")
    dev = torch.from_numpy(rows).cuda()
    cap = L.jpg_static_scan_capacity(w, h, *ratio)
    call = lambda ww, hh, st, c, r=(4, 2, 0): L.jpg_encode_device_static(0, N.DCT_BIN, *r, dev.data_ptr(), ww, hh, st, 1,
                                                                          buf.data_ptr(), c, res.data_ptr(), None)
    assert call(w, h, rows.shape[1], 1 << 20) < 0                      # output buffer below the worst case
    assert call(7680, 4320, 7680 * 3, 1 << 40) < 0                     # beyond the working array
    assert L.jpg_static_scan_capacity(7680, 4320, 4, 2, 0) > 0
    assert call(200, 100, 600, 1 << 40) < 0                            # narrower than a tile
    assert call(w, h, rows.shape[1], cap, (4, 1, 1)) < 0               # no such ratio
    torch.cuda.synchronize()
