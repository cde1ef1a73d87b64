"""CPU tests of the stripe mode's host logic (jpeg-encoder_amd/stripes.py): the plan, the stitch
arithmetic, a lock-step simulation of N ranks, and two real processes over gloo.  The stripe
results come from the oracle (tests/stripe_fakes.py); the GPU backend is tested in
test_gpu_parity.py."""
import os
import time
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def test_plan_covers_the_image(J):
    S = J.stripes
    for H in (1, 7, 16, 17, 100, 4320, 32768):
        for v in (1, 2):
            for n in (1, 2, 3, 8):
                plan = S.plan_stripes(H, v, n)
                assert len(plan) == n and plan[0][0] == 0 and plan[-1][1] == H
                for (a, b), (c, d) in zip(plan, plan[1:]):
                    assert b == c and a <= b
                for a, b in plan[:-1]:
                    assert (a % (8 * v) == 0 or a == H) and (b % (8 * v) == 0 or b == H)
                sizes = [(b - a + 8 * v - 1) // (8 * v) for a, b in plan]
                assert max(sizes) - min(sizes) <= 1
                assert J.host.plan_stripes(H, v, n) == plan               # the C twin used by the CLI
    assert S.plan_stripes(4320, 2, 8)[0] == (0, 544)            # 270 MCU rows: 34,34,34,34,34,34,33,33
    rows = np.arange(10)[:, None] * np.ones((1, 4), int)
    assert S.stripe_rows(rows, 10, 2, 5, bottom_up=True)[:, 0].tolist() == [5, 6, 7]
    assert S.stripe_rows(rows, 10, 2, 5, bottom_up=False)[:, 0].tolist() == [2, 3, 4]


def test_stitch_plan_heads(J):
    S = J.stripes
    rng = np.random.default_rng(5)
    for _ in range(200):
        n = int(rng.integers(1, 7))
        lens = [int(rng.integers(0, 40)) if rng.random() < 0.5 else int(rng.integers(0, 5)) for _ in range(n)]
        streams = [rng.integers(0, 2, L) for L in lens]
        tails = [int("".join(map(str, s[-32:])), 2) if len(s) else 0 for s in streams]
        offsets, heads = S.stitch_plan(lens, tails)
        whole = np.concatenate(streams + [np.zeros(0, int)]).astype(int)
        assert offsets == [int(sum(lens[:k])) for k in range(n)]
        assert J.host.stitch_plan(lens, tails) == (offsets, heads)    # the C twin used by the CLI
        for k in range(n):
            r = offsets[k] % 8
            want = whole[offsets[k] - r:offsets[k]]
            got = np.unpackbits(np.array([heads[k]], np.uint8))
            assert got[:r].tolist() == want.tolist() and not got[r:].any()


def _image(J, w, h, seed, noise=False):
    rows = J.host.synth_bgr(w, h, seed=seed, noise_only=noise)
    return rows, rows.shape[1]


@pytest.mark.parametrize("ratio,algo,w,h,n,noise", [
    ((4, 2, 0), "bin-dct", 64, 80, 2, False), ((4, 2, 0), "bin-dct", 50, 100, 3, True),
    ((4, 2, 2), "real-dct", 48, 40, 4, True), ((4, 4, 4), "bin-dct", 24, 24, 3, False),
    ((4, 2, 0), "real-dct", 40, 16, 3, True),                  # more ranks than MCU rows: empty stripes
    ((4, 4, 4), "bin-dct", 8, 64, 8, False),                   # one tiny stripe per rank
])
def test_lockstep_simulation_matches_whole_image(J, oracle, ratio, algo, w, h, n, noise):
    from tests.stripe_fakes import OracleStripe
    S = J.stripes
    rows, stride = _image(J, w, h, 77, noise)
    a = oracle.BIN if algo == "bin-dct" else oracle.REAL
    _, v = J.get_downsampling_factor(ratio)
    plan = S.plan_stripes(h, v, n)
    jobs = [S.StripeJob(OracleStripe(oracle, rows, w, h, stride, ratio, a, *plan[k]), k, n, rows, w, h, stride, v)
            for k in range(n)]
    dcs = [j.transform() for j in jobs]
    codes = [j.code(dcs) for j in jobs]
    pieces = [j.place(codes) for j in jobs]
    want = oracle.RefImage.from_bgr(rows, w, h, stride, ratio, a).encode()
    assert sum(c[0] for c in codes) == want.bits()[1]
    assert np.array_equal(np.concatenate(pieces), want.stuffed_scan())


def _worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import jpeg_encoder_amd as J
    from oracle import refcpu
    from tests.stripe_fakes import OracleStripe
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        w, h, ratio = 72, 96, (4, 2, 0)
        rows = J.host.synth_bgr(w, h, seed=11, noise_only=True)
        stride = rows.shape[1]
        plan = J.stripes.plan_stripes(h, 2, world)
        be = OracleStripe(refcpu, rows, w, h, stride, ratio, refcpu.BIN, *plan[rank])
        scan = J.stripes.encode_striped(be, rows, w, h, stride, 2, device="cpu")
        # the same with the scalars exchanged through shared host memory instead of gloo
        shm = J.stripes.ShmExchange.from_group()
        for k in range(300):                              # ranks out of step: the records are double-buffered
            got = shm.all_gather([rank * 1000 + k, k])
            assert got == [(r * 1000 + k, k) for r in range(world)]
            if (k + rank) % 7 == 0:
                time.sleep(0.0005)
        scan2 = J.stripes.encode_striped(be, rows, w, h, stride, 2, exchange=shm)
        shm.close()
        if rank == 0:
            want = refcpu.RefImage.from_bgr(rows, w, h, stride, ratio, refcpu.BIN).encode().stuffed_scan()
            assert np.array_equal(scan, want)
            assert np.array_equal(scan2, want)
            assert not os.path.exists(shm.path)
            open(os.path.join(tmp, "ok"), "w").write("%d" % len(scan))
        else:
            assert scan is None and scan2 is None
    finally:
        dist.destroy_process_group()


def test_two_processes_over_gloo(tmp_path):
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert int(open(tmp_path / "ok").read()) > 0


class _ThreadExchange:
    """all_gather / barrier among the threads of one process (the interface of ScalarExchange)."""

    def __init__(self, rank, world, shared):
        self.rank, self.world, self.shared = rank, world, shared

    def all_gather(self, values):
        import threading
        sh = self.shared
        with sh["cv"]:
            gen = sh["gen"]
            sh["slots"][self.rank] = tuple(int(v) for v in values)
            sh["count"] += 1
            if sh["count"] == self.world:
                sh["result"] = list(sh["slots"])
                sh["count"] = 0
                sh["gen"] += 1
                sh["cv"].notify_all()
            else:
                while sh["gen"] == gen:
                    assert sh["cv"].wait(timeout=30)
            return list(sh["result"])

    def barrier(self):
        self.all_gather([])


@pytest.mark.parametrize("failing", [None, 0, 2, "export"])
def test_sharing_rank0_memory_is_decided_by_all_ranks_together(J, failing):
    """The set-up step of the stripe chain that makes an allocation of rank 0 reachable on every rank: when one rank
    cannot open it (or rank 0 cannot allocate or export it) EVERY rank gets None -- so all fall back to the next kind
    of memory together -- those that had opened it have closed it, and rank 0 frees it only after that."""
    import threading
    S = J.stripes
    world = 4
    shared = {"cv": threading.Condition(), "slots": [None] * world, "count": 0, "gen": 0, "result": None}
    log, out = [], [None] * world

    def run(rank):
        ex = _ThreadExchange(rank, world, shared)
        alloc = lambda: (None if failing == 0 else 0x1000)
        export = lambda p: (None if failing == "export" else [7, 0, 0, 0, 0, 0, 0, 1])
        open_ = lambda words: (None if rank == failing else 0x2000 + rank if words == [7, 0, 0, 0, 0, 0, 0, 1] else None)
        out[rank] = S.share_from_rank0(ex, alloc, lambda p: log.append(("free", rank, p)), export, open_,
                                       lambda p: log.append(("close", rank, p)))

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(60)
    if failing is None:
        assert out == [0x1000, 0x2001, 0x2002, 0x2003] and log == []
    else:
        assert out == [None] * world
        closes = [e for e in log if e[0] == "close"]
        frees = [e for e in log if e[0] == "free"]
        if failing == 2:
            assert sorted(closes) == [("close", 1, 0x2001), ("close", 3, 0x2003)]
            assert frees == [("free", 0, 0x1000)] and log.index(frees[0]) > max(log.index(c) for c in closes)
        elif failing == "export":
            assert closes == [] and frees == [("free", 0, 0x1000)]
        else:
            assert log == []
