"""Stripe chain on one GPU: N contexts in one process, a pass queued on all of them before any is waited for; the
scan put together in the gather buffer must equal the whole-image encode (and the oracle's)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _chain_encode_all(J, images, w, h, ratio, algo, n, gather_host=False, cap=None, in_flight=2):
    """images: one array of rows per pass.  Up to `in_flight` passes are queued on every context before the oldest is
    waited for; returns the scan of every pass, read out of its half of the gather buffer."""
    S = J.stripes
    L = J._native.lib()
    _, v = J.get_downsampling_factor(ratio)
    stride = images[0].shape[1]
    plan = S.plan_stripes(h, v, n)
    active = [k for k, (a, b) in enumerate(plan) if b > a]
    mask = sum(1 << k for k in active)
    mailbox = L.jpg_chain_mailbox_alloc()
    assert mailbox
    cap = cap or 2 * (w * h // 2 + (1 << 20))              # two halves, one per pass in flight
    gather = L.jpg_host_alloc(cap) if gather_host else L.jpg_device_alloc(0, cap)
    assert gather
    import torch
    encs = {}
    for k in active:
        encs[k] = J.Encoder(ratio, algo)
        encs[k].chain_setup(k, n, mailbox, gather, cap)
    devs = [{k: torch.from_numpy(np.ascontiguousarray(S.stripe_rows(rows, h, plan[k][0], plan[k][1], True))).cuda()
             for k in active} for rows in images]
    torch.cuda.synchronize()
    outs = []

    def take(p):
        res = {k: encs[k].chain_pass_finish() for k in reversed(active)}
        total = res[0][4]
        assert total == sum(r[1] for r in res.values())
        base = (p & 1) * (cap // 2)
        bit, byte = 0, base
        for k in active:
            assert res[k][2] == bit and res[k][3] == byte, (k, res[k], bit, byte)
            bit += res[k][0]
            byte += res[k][1]
        out = np.zeros(total, np.uint8)
        if gather_host:
            C.memmove(out.ctypes.data, gather + base, total)
        else:
            J._native.check(None, L.jpg_device_to_host(0, out.ctypes.data, gather + base, total))
        outs.append(out)

    try:
        for p in range(len(images)):
            if p >= in_flight:
                take(p - in_flight)
            for k in active:                                  # ascending ranks: a gate only waits for lower ranks
                encs[k].stripe_device(devs[p][k].data_ptr(), w, plan[k][1] - plan[k][0], stride, True)
                encs[k].chain_pass_async(mask)
        for p in range(max(0, len(images) - in_flight), len(images)):
            take(p)
    finally:
        for e in encs.values():
            e.close()
        if gather_host:
            L.jpg_free(gather)
        else:
            L.jpg_device_free(0, gather)
        L.jpg_free(mailbox)
    return outs


def _chain_encode(J, rows, w, h, ratio, algo, n, passes=1, gather_host=False, cap=None):
    outs = _chain_encode_all(J, [rows] * passes, w, h, ratio, algo, n, gather_host, cap, in_flight=1)
    for o in outs[1:]:
        np.testing.assert_array_equal(o, outs[0])
    return outs[0]


@pytest.mark.parametrize("n", [1, 2, 3, 8])
@pytest.mark.parametrize("ratio,algo,w,h", [((4, 2, 0), "bin-dct", 1544, 200), ((4, 4, 4), "bin-dct", 640, 72),
                                            ((4, 2, 2), "real-dct-strict", 1100, 100)])
def test_chain_equals_whole_image(J, oracle, n, ratio, algo, w, h):
    rows = J.host.synth_bgr(w, h, seed=91 + n, noise_only=(n == 3))
    ref = oracle.RefImage.from_bgr(rows, w, h, rows.shape[1], ratio, oracle.BIN if algo == "bin-dct" else oracle.REAL).encode()
    got = _chain_encode(J, rows, w, h, ratio, algo, n, passes=3)
    np.testing.assert_array_equal(got, ref.stuffed_scan())


def test_chain_more_ranks_than_mcu_rows_and_host_gather(J, oracle):
    w, h, ratio = 600, 40, (4, 2, 0)                      # 3 MCU rows, 5 ranks: two without a stripe
    rows = J.host.synth_bgr(w, h, seed=5)
    want = oracle.RefImage.from_bgr(rows, w, h, rows.shape[1], ratio, oracle.BIN).encode().stuffed_scan()
    np.testing.assert_array_equal(_chain_encode(J, rows, w, h, ratio, "bin-dct", 5, passes=2, gather_host=True), want)


@pytest.mark.parametrize("n,gather_host", [(1, False), (4, False), (8, True)])
def test_chain_two_passes_in_flight_of_different_images(J, oracle, n, gather_host):
    """Seven passes over images of different content (so of different lengths and offsets), two of them queued at
    any time: the hand-over records, the halves of the gather buffer and the contexts' two sets of buffers are all
    reused while the pass before is still running."""
    w, h, ratio = 1544, 136, (4, 2, 0)
    images = [J.host.synth_bgr(w, h, seed=700 + i, noise_only=(i % 3 == 1)) for i in range(7)]
    want = [oracle.RefImage.from_bgr(r, w, h, r.shape[1], ratio, oracle.BIN).encode().stuffed_scan() for r in images]
    got = _chain_encode_all(J, images, w, h, ratio, "bin-dct", n, gather_host=gather_host)
    assert len({x.size for x in want}) > 3
    for p, (g, x) in enumerate(zip(got, want)):
        np.testing.assert_array_equal(g, x, err_msg="pass %d" % p)


def test_chain_gather_too_small_is_an_error_not_a_fault(J):
    w, h, ratio = 1024, 256, (4, 2, 0)
    rows = J.host.synth_bgr(w, h, seed=6, noise_only=True)
    with pytest.raises(J._native.JpgError) as e:
        _chain_encode(J, rows, w, h, ratio, "bin-dct", 2, cap=8192)
    assert e.value.code == -13
