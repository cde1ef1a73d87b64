"""The stuffing kernel under load (GPU): several contexts and several launch sets competing for the device,
streams of many segments, and results taken out of a pipeline of differently sized images."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_two_contexts_stuff_large_noise_images_concurrently(J, oracle):
    """Two contexts on one device, each coding and stuffing a 7680x4320 uniform-noise image (about 2200 stream
    segments each) at the same time: neither stuffing kernel has the device to itself, yet every wait inside
    it ends (segments are handed out in order) and the bytes are the oracle's."""
    import torch
    w, h = 7680, 4320
    imgs = [J.host.synth_bgr(w, h, seed=900 + k, noise_only=True) for k in range(2)]
    refs = [oracle.RefImage.from_bgr(r, w, h, r.shape[1], (4, 2, 0), oracle.BIN).encode(threads=3).stuffed_scan()
            for r in imgs]
    dev = [torch.from_numpy(r).cuda() for r in imgs]
    torch.cuda.synchronize()
    encs = [J.Encoder((4, 2, 0), "bin-dct") for _ in range(2)]
    stride = imgs[0].shape[1]
    for rnd in range(6):
        for k, e in enumerate(encs):
            e.encode_device_async(dev[(k + rnd) % 2].data_ptr(), w, h, stride, True)
        for k, e in enumerate(encs):
            _, n = e.finish()
            assert n == refs[(k + rnd) % 2].size
            if rnd in (0, 5):
                np.testing.assert_array_equal(e.scan(n), refs[(k + rnd) % 2], err_msg="round %d context %d" % (rnd, k))


def test_noise_batches_three_sets_in_flight(J, oracle):
    """64-image uniform-noise launch sets, three in flight on one context, 20 rounds: the stuffing kernel of one
    set runs beside the coding kernels of the next two, which fill every compute unit's LDS.  No spurious
    error; a sample of every round is compared byte for byte."""
    import torch
    w, h, n = 1024, 512, 64
    imgs = [J.host.synth_bgr(w, h, seed=7000 + i, noise_only=True) for i in range(n)]
    refs = {i: oracle.RefImage.from_bgr(imgs[i], w, h, imgs[i].shape[1], (4, 2, 0), oracle.BIN).encode().stuffed_scan()
            for i in (0, 17, 63)}
    dev = [torch.from_numpy(r).cuda() for r in imgs]
    torch.cuda.synchronize()
    ptrs = [t.data_ptr() for t in dev]
    stride = imgs[0].shape[1]
    enc = J.Encoder((4, 2, 0), "bin-dct")
    inflight = 0
    taken = 0
    for rnd in range(20):
        if inflight == 3:
            enc.finish_next()                                    # raises on any kernel error
            inflight -= 1
            i = (0, 17, 63)[taken % 3]
            np.testing.assert_array_equal(enc.batch_scan(i), refs[i], err_msg="set %d image %d" % (taken, i))
            taken += 1
        enc.encode_batch_device_async(ptrs, w, h, stride, True)
        inflight += 1
    while inflight:
        enc.finish_next()
        inflight -= 1
        for i in refs:
            np.testing.assert_array_equal(enc.batch_scan(i), refs[i])


def test_mixed_sizes_in_flight_and_the_dense_stream(J, oracle):
    """A large and a small image queued on one context: while an encode is pending the dense bit string is
    refused (its buffer belongs to the slot being viewed, the pending kernels still write theirs); afterwards
    it is the oracle's for the image the view shows, whichever size was queued last."""
    import torch
    big = (2048, 1024)
    small = (512, 64)
    data = {}
    for name, (w, h) in (("big", big), ("small", small)):
        rows = J.host.synth_bgr(w, h, seed=w + h, noise_only=True)
        ref = oracle.RefImage.from_bgr(rows, w, h, rows.shape[1], (4, 2, 0), oracle.BIN).encode()
        t = torch.from_numpy(rows).cuda()
        data[name] = (w, h, rows, t, ref)
    torch.cuda.synchronize()
    L = J._native.lib()
    for order in (("big", "small"), ("small", "big")):
        enc = J.Encoder((4, 2, 0), "bin-dct")
        for name in order:
            w, h, rows, t, ref = data[name]
            enc.encode_device_async(t.data_ptr(), w, h, rows.shape[1], True)
        nbits0, n0 = enc.finish_next()                            # the view shows the first image, one is pending
        w, h, rows, t, ref = data[order[0]]
        assert nbits0 == ref.bits()[1]
        np.testing.assert_array_equal(enc.scan(n0), ref.stuffed_scan())
        buf = np.zeros((nbits0 + 7) // 8, np.uint8)
        assert L.jpg_get_bits(enc._ctx, buf.ctypes.data, buf.nbytes) == -8
        nbits1, n1 = enc.finish_next()
        w, h, rows, t, ref = data[order[1]]
        assert nbits1 == ref.bits()[1]
        np.testing.assert_array_equal(enc.scan(n1), ref.stuffed_scan())
        np.testing.assert_array_equal(enc.bits(nbits1), ref.bits()[0])
        enc.close()


def test_entry_points_restore_the_callers_device(J):
    """Calls on a context of another device leave the thread's current device as they found it."""
    import ctypes as C
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    hip = C.CDLL("libamdhip64.so.7")
    cur = C.c_int(-1)
    torch.cuda.set_device(0)
    enc = J.Encoder((4, 2, 0), "bin-dct", device=1)
    assert hip.hipGetDevice(C.byref(cur)) == 0 and cur.value == 0
    rows = J.host.synth_bgr(64, 32, seed=1)
    enc.encode_scan(rows, 64, 32)
    assert hip.hipGetDevice(C.byref(cur)) == 0 and cur.value == 0
    enc.close()
    assert hip.hipGetDevice(C.byref(cur)) == 0 and cur.value == 0
