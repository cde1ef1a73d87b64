"""Stripe chain on one GPU: N contexts in one process, a pass queued on all of them before any is waited for; the
scan put together in the gather buffer must equal the whole-image encode (and the oracle's)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _chain_encode(J, rows, w, h, ratio, algo, n, passes=1, gather_host=False, cap=None):
    S = J.stripes
    L = J._native.lib()
    _, v = J.get_downsampling_factor(ratio)
    stride = rows.shape[1]
    plan = S.plan_stripes(h, v, n)
    active = [k for k, (a, b) in enumerate(plan) if b > a]
    mask = sum(1 << k for k in active)
    mailbox = L.jpg_chain_mailbox_alloc()
    assert mailbox
    cap = cap or (w * h // 2 + (1 << 20))
    gather = L.jpg_host_alloc(cap) if gather_host else L.jpg_device_alloc(0, cap)
    assert gather
    import torch
    encs, devs = {}, {}
    for k in active:
        encs[k] = J.Encoder(ratio, algo)
        encs[k].chain_setup(k, n, mailbox, gather, cap)
        sub = np.ascontiguousarray(S.stripe_rows(rows, h, plan[k][0], plan[k][1], True))
        devs[k] = torch.from_numpy(sub).cuda()
    torch.cuda.synchronize()
    out = None
    for p in range(passes):
        for k in active:                                  # ascending ranks: a gate only waits for lower ranks
            encs[k].stripe_device(devs[k].data_ptr(), w, plan[k][1] - plan[k][0], stride, True)
            encs[k].chain_pass_async(mask)
        res = {k: encs[k].chain_pass_finish() for k in reversed(active)}
        total = res[active[0]][4] if active[0] == 0 else None
        assert total == sum(r[1] for r in res.values())
        bit = byte = 0
        for k in active:
            assert res[k][2] == bit and res[k][3] == byte, (k, res[k], bit, byte)
            bit += res[k][0]
            byte += res[k][1]
        out = np.zeros(total, np.uint8)
        if gather_host:
            C.memmove(out.ctypes.data, gather, total)
        else:
            J._native.check(None, L.jpg_device_to_host(0, out.ctypes.data, gather, total))
    for e in encs.values():
        e.close()
    if gather_host:
        L.jpg_free(gather)
    else:
        L.jpg_device_free(0, gather)
    L.jpg_free(mailbox)
    return out


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


def test_chain_gather_too_small_is_an_error_not_a_fault(J):
    w, h, ratio = 1024, 256, (4, 2, 0)
    rows = J.host.synth_bgr(w, h, seed=6, noise_only=True)
    with pytest.raises(J._native.JpgError) as e:
        _chain_encode(J, rows, w, h, ratio, "bin-dct", 2, cap=4096)
    assert e.value.code == -13
