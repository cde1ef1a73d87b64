"""One large image coded as MCU-row stripes, one stripe per rank.

The reference codes an image serially (src/jpeg/entropy_coding.rs:16-125).  Its scan is
causal in MCU-row order, so it can be cut between MCU rows: a stripe needs from the stripes
above it only (a) the DC value of their last Y, Cb and Cr block (the three predictors of
src/jpeg/entropy_coding.rs:38-40,149-150), (b) the number of bits they produced, which fixes
where its own bits start, and (c) the few bits that share a byte with its first ones.  Each
rank transforms and codes its stripe on its own GPU; the ranks exchange those O(N) scalars,
shift their bits into place, stuff (0xFF -> 0xFF00 must come after the shift) and the pieces
are copied device to device into one buffer at their byte offsets.  Only the last stripe keeps
the zero-padded partial byte.
"""
import numpy as np


def plan_stripes(height, v, n):
    """Rows [y0, y1) of the image (top-down) for each of n ranks: whole MCU rows (8*v pixel rows),
    as even as possible.  A rank may get an empty stripe when there are fewer MCU rows than ranks."""
    mcu_h = 8 * v
    mcu_rows = (height + mcu_h - 1) // mcu_h
    base, extra = divmod(mcu_rows, n)
    out, r = [], 0
    for k in range(n):
        cnt = base + (1 if k < extra else 0)
        out.append((min(height, r * mcu_h), min(height, (r + cnt) * mcu_h)))
        r += cnt
    return out


def stripe_rows(rows, height, y0, y1, bottom_up=True):
    """The file rows holding image rows [y0, y1): BMP files are bottom-up (src/bmp/bmp_image.rs:65-70),
    so a stripe is a contiguous range of the file either way."""
    return rows[height - y1:height - y0] if bottom_up else rows[y0:y1]


def stitch_plan(nbits, tails):
    """nbits[k]: bits of stripe k; tails[k]: its last min(32, nbits[k]) bits, right-aligned.
    Returns (offsets, heads): the bit offset of every stripe in the whole scan and the byte its
    first byte has to be OR-ed with (the offset % 8 bits before it, top-aligned)."""
    offsets, heads, total = [], [], 0
    for k in range(len(nbits)):
        offsets.append(total)
        r = total % 8
        acc, got, j = 0, 0, k - 1
        while got < r and j >= 0:                        # walk back over (possibly tiny) stripes
            take = min(r - got, nbits[j], 32)
            acc |= (tails[j] & ((1 << take) - 1)) << got
            got += take
            j -= 1
        heads.append((acc << (8 - r)) & 0xff if r else 0)
        total += nbits[k]
    return offsets, heads


class StripeJob:
    """The three local steps of one rank; the caller moves the small results between ranks.

    backend: front(rows, width, height, stride, bottom_up), last_dc() -> (y, cb, cr),
    entropy(pred) -> nbits, stream_tail() -> int, stitch(bit_offset, head, is_last) -> (len, tail),
    scan(len) -> uint8 array.  jpeg_encoder_amd.Encoder provides these (front = front_host)."""

    def __init__(self, backend, rank, world, rows, width, height, stride, v, bottom_up=True, stripe_ptr=None):
        """stripe_ptr: device address of this rank's stripe rows when they are already resident on its GPU
        (they must stay valid until code() has returned); rows may then be None."""
        self.backend, self.rank, self.world = backend, rank, world
        self.stripe_ptr = stripe_ptr
        self.plan = plan_stripes(height, v, world)
        self.active = [k for k, (a, b) in enumerate(self.plan) if b > a]
        self.y0, self.y1 = self.plan[rank]
        self.empty = self.y1 <= self.y0
        self.rows, self.width, self.height, self.stride, self.bottom_up = rows, width, height, stride, bottom_up

    def transform(self):
        """Colour conversion, subsampling, DCT and quantisation of the stripe -> its last DCs."""
        if self.empty:
            return (0, 0, 0)
        if self.stripe_ptr is not None:
            self.backend.stripe_device(self.stripe_ptr, self.width, self.y1 - self.y0, self.stride, self.bottom_up)
            return tuple(int(x) for x in self.backend.last_dc())
        sub = stripe_rows(self.rows, self.height, self.y0, self.y1, self.bottom_up)
        front = getattr(self.backend, "front_host", None) or self.backend.front
        front(sub, self.width, self.y1 - self.y0, self.stride, self.bottom_up)
        return tuple(int(x) for x in self.backend.last_dc())

    def code(self, all_dcs):
        """all_dcs[k]: what transform() returned on rank k -> (nbits, tail) of this stripe."""
        if self.empty:
            return (0, 0)
        before = [k for k in self.active if k < self.rank]
        pred = all_dcs[before[-1]] if before else (0, 0, 0)     # src/jpeg/entropy_coding.rs:38-40
        nbits = self.backend.entropy(pred)
        return (int(nbits), int(self.backend.stream_tail()))

    def place(self, all_codes):
        """all_codes[k]: what code() returned on rank k -> this stripe's piece of the stuffed scan."""
        if self.empty:
            return np.zeros(0, np.uint8)
        offsets, heads = stitch_plan([c[0] for c in all_codes], [c[1] for c in all_codes])
        is_last = self.rank == self.active[-1]
        n, _ = self.backend.stitch(offsets[self.rank], heads[self.rank], is_last)
        return self.backend.scan(n)

    def place_on_device(self, all_codes):
        """The same, leaving the piece in the backend's device memory: its length in bytes."""
        if self.empty:
            return 0
        offsets, heads = stitch_plan([c[0] for c in all_codes], [c[1] for c in all_codes])
        n, _ = self.backend.stitch(offsets[self.rank], heads[self.rank], self.rank == self.active[-1])
        return n


class ScalarExchange:
    """The O(N) scalars that link neighbouring stripes travel through the host: an all_gather of a few int64 on
    a gloo (CPU) group.  With an RCCL default group a gloo twin of it is created once."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        if group is None and dist.get_backend() != "gloo":
            group = dist.new_group(backend="gloo")
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)

    def all_gather(self, values):
        import torch
        t = torch.tensor([int(v) for v in values], dtype=torch.int64)
        out = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t, group=self.group)
        return [tuple(int(x) for x in o.tolist()) for o in out]

    def barrier(self):
        self.dist.barrier(group=self.group)


def _unlink_quietly(path):
    import os
    try:
        os.unlink(path)
    except OSError:
        pass


class ShmExchange:
    """The same exchange through a small file in shared host memory (/dev/shm): every rank owns a record of a
    sequence number and up to WORDS int64 and publishes by writing the values, then the number; readers spin on
    the numbers.  One process per GPU on one node, so host memory is shared; an all_gather costs a few
    microseconds instead of a gloo round trip.  Records are double-buffered by the parity of the sequence
    number: a rank can be at most one round ahead of the slowest reader.  `key` must be the same string on
    every rank and new for every job (from_group draws one and spreads it over the process group)."""
    WORDS = 16

    def __init__(self, rank, world, key, timeout=120.0):
        import os
        import time
        self.rank, self.world, self.timeout = rank, world, timeout
        self.path = os.path.join("/dev/shm" if os.path.isdir("/dev/shm") else "/tmp", "jpg_stripes_%s" % key)
        n = world * 2 * (1 + self.WORDS)
        if rank == 0:
            tmp = self.path + ".new"
            np.zeros(n, np.int64).tofile(tmp)
            os.replace(tmp, self.path)                     # appears complete or not at all
        else:
            t0 = time.monotonic()
            while not (os.path.exists(self.path) and os.path.getsize(self.path) == 8 * n):
                if time.monotonic() - t0 > timeout:
                    raise TimeoutError("rank 0 never created %s" % self.path)
                time.sleep(0.001)
        self.mem = np.memmap(self.path, np.int64, "r+", shape=(world, 2, 1 + self.WORDS))
        self.seq = 0
        if rank == 0:                                      # the file goes away with the process even when close() is never reached
            import atexit
            atexit.register(_unlink_quietly, self.path)

    @classmethod
    def from_group(cls, group=None):
        """Draws a key on rank 0 and spreads it with one gloo all_gather."""
        import os
        ex = ScalarExchange(group)
        key = int.from_bytes(os.urandom(7), "little") if ex.rank == 0 else 0
        key = ex.all_gather([key])[0][0]
        out = cls(ex.rank, ex.world, "%x" % key)
        ex.barrier()                                       # every rank has mapped the file
        return out

    def all_gather(self, values):
        import time
        vals = [int(v) for v in values]
        assert len(vals) <= self.WORDS
        self.seq += 1
        mine = self.mem[self.rank, self.seq & 1]
        mine[1:1 + len(vals)] = vals
        mine[0] = self.seq                                 # published (stores are not reordered on x86-64)
        out = []
        t0 = None
        for k in range(self.world):
            rec = self.mem[k, self.seq & 1]
            while int(rec[0]) < self.seq:
                if t0 is None:
                    t0 = time.monotonic()
                elif time.monotonic() - t0 > self.timeout:
                    raise TimeoutError("rank %d did not reach exchange %d" % (k, self.seq))
            out.append(tuple(int(x) for x in rec[1:1 + len(vals)]))
        return out

    def barrier(self):
        self.all_gather([])

    def close(self):
        try:
            self.barrier()
        finally:
            self.mem = None
            if self.rank == 0:
                _unlink_quietly(self.path)


class DeviceGather:
    """Where the stuffed pieces of all ranks meet (the protocol with host exchanges): a buffer owned by rank 0 that
    the other ranks open through an inter-process handle and write their pieces into at their byte offsets with a
    device-to-device copy (over xGMI between GPUs).  Device memory of rank 0 where every rank can open that
    (`kind == "device"`), else page-locked host memory of rank 0 (`kind == "host"`: the pieces travel over PCIe).
    No collective moves scan bytes.  Kept across calls; grown, by all ranks together, when a scan does not fit."""

    def __init__(self, native, device):
        self.N, self.device = native, device
        self.L = native.lib()
        self.cap, self.kind = 0, None
        self.buffer = None                        # the buffer's address in this process
        self._free = None                         # rank 0: how to release it

    def ensure(self, ex, cap):
        """Collective: every rank calls it with the same cap.  (Re)allocates when cap exceeds the present size."""
        import ctypes as C
        if cap <= self.cap:
            return
        self.release(ex)
        cap = max(int(cap), 1 << 20)
        N, L, dev = self.N, self.L, self.device

        def export(p):
            h = (C.c_uint8 * N.IPC_HANDLE_BYTES)()
            return [int(x) for x in np.frombuffer(bytes(h), np.int64)] if L.jpg_ipc_export(p, h) == 0 else None

        def open_(words):
            h = (C.c_uint8 * N.IPC_HANDLE_BYTES).from_buffer_copy(np.array(words, np.int64).tobytes())
            p = C.c_void_p()
            return p.value if L.jpg_ipc_open(dev, h, C.byref(p)) == 0 and p.value else None

        for kind, alloc, free in (("device", lambda: L.jpg_device_alloc(dev, cap), lambda p: L.jpg_device_free(dev, p)),
                                  ("host", lambda: L.jpg_host_alloc(cap), L.jpg_free)):
            addr = share_from_rank0(ex, alloc, free, export, open_, L.jpg_ipc_close)
            if addr is not None:
                self.buffer, self.cap, self.kind = addr, cap, kind
                self._free = free if ex.rank == 0 else None
                return
        raise RuntimeError("no memory of rank 0 can be opened on every rank")

    def release(self, ex):
        """Collective: every rank lets go of the buffer, then rank 0 frees it."""
        if self.buffer is None:
            return
        if ex.rank != 0:
            self.L.jpg_ipc_close(self.buffer)
        ex.barrier()                              # no rank has it mapped any more
        if ex.rank == 0:
            self._free(self.buffer)
        self.buffer, self.cap, self.kind = None, 0, None

    def read(self, n):
        """Rank 0: the first n bytes of the buffer."""
        import ctypes as C
        out = np.zeros(n, np.uint8)
        if n and self.kind == "device":
            self.N.check(None, self.L.jpg_device_to_host(self.device, out.ctypes.data, self.buffer, n))
        elif n:
            C.memmove(out.ctypes.data, self.buffer, n)
        return out

    def close(self, ex=None):
        """With the exchange: collective, as release().  Without: this rank's side only (rank 0 must be the last)."""
        if ex is not None:
            self.release(ex)
        elif self.buffer is not None:
            if self._free:
                self._free(self.buffer)
            else:
                self.L.jpg_ipc_close(self.buffer)
            self.buffer, self.cap, self.kind = None, 0, None


def share_from_rank0(ex, alloc, free, export, open_, close):
    """One allocation of rank 0 made reachable on every rank, or on none: rank 0 allocates (alloc() -> address or None)
    and exports (export(address) -> 8 int64 or None), the others open (open_(words) -> address or None); the ranks
    then tell each other whether they got it.  If any did not, those that did let go again (close(address) on the
    others, free(address) on rank 0 once nobody has it mapped) and every rank returns None: the decision is the same
    on all ranks, so they can fall back together.  `ex`: rank, world, all_gather(values), barrier()."""
    ok, words, mine = 1, [0] * 8, None
    if ex.rank == 0:
        mine = alloc()
        words = export(mine) if mine else None
        ok = int(bool(mine) and words is not None)
        words = words or [0] * 8
    got = ex.all_gather([ok] + list(words))
    ok, addr = int(got[0][0]), mine
    if ok and ex.rank != 0:
        addr = open_(list(got[0][1:]))
        ok = int(addr is not None)
    everyone = min(int(g[0]) for g in ex.all_gather([ok]))
    if everyone:
        return addr
    if ex.rank != 0 and ok and addr:
        close(addr)
    ex.barrier()                          # no rank has it mapped any more
    if ex.rank == 0 and mine:
        free(mine)
    return None


class StripeChain:
    """One process per GPU: the set-up of the stripe chain (jpg_chain_* in include/jpeg_hip.h), in which the ranks'
    kernels hand each other the scalars of a pass through a mailbox in host memory and put the stuffed pieces at
    their byte offsets in one gather buffer, so that a pass is queued at once and waited for once.

    mailbox: page-locked host memory of rank 0, which the other ranks open through an inter-process handle and
    their GPUs read and write directly (host memory is not cached by any GPU).
    gather buffer: device memory of rank 0, opened by the other ranks the same way (`gather="device"`: the pieces
    travel GPU to GPU); where that is not possible -- the open fails on some rank -- every rank falls back to
    page-locked host memory of rank 0 (`gather="host"`: the pieces travel over PCIe), and `self.gather` says which.
    `exchange` (ScalarExchange or ShmExchange) is only used here, to pass the handles around.  Kept across passes;
    close() on every rank."""

    def __init__(self, encoder, exchange, gather_cap, gather="auto"):
        from . import _native as N
        self.N, self.L, self.enc, self.ex = N, N.lib(), encoder, exchange
        self.rank, self.world, self.device = exchange.rank, exchange.world, encoder.device
        # two passes may be in flight: each has its own half of the buffer (gather_cap bytes for one scan)
        self.cap = 2 * ((max(int(gather_cap), 1) + 255) & ~255)
        self.offset = 0                   # rank 0: where the scan of the pass finish() took last begins
        self._mine = []                   # rank 0: (address, free function) of what it allocated
        self._opened = []                 # other ranks: addresses opened through handles
        self.mailbox = self._share(self.L.jpg_chain_mailbox_alloc, self.L.jpg_free, required=True)
        self.gather, self.buffer = None, None
        if gather in ("auto", "device"):
            self.buffer = self._share(lambda: self.L.jpg_device_alloc(self.device, self.cap),
                                      lambda p: self.L.jpg_device_free(self.device, p), required=gather == "device")
            if self.buffer:
                self.gather = "device"
        if not self.buffer:
            self.buffer = self._share(lambda: self.L.jpg_host_alloc(self.cap), self.L.jpg_free, required=True)
            self.gather = "host"
        encoder.chain_setup(self.rank, self.world, self.mailbox, self.buffer, self.cap)
        exchange.barrier()                # every rank is set up before any rank's kernels use the mailbox

    def _share(self, alloc, free, required):
        """Rank 0 allocates and exports, the others open: the address in this process, or None when some rank could
        not get at it (every rank then lets go of it again)."""
        import ctypes as C
        N = self.N

        def export(p):
            h = (C.c_uint8 * N.IPC_HANDLE_BYTES)()
            if self.L.jpg_ipc_export(p, h) != 0:
                return None
            return [int(x) for x in np.frombuffer(bytes(h), np.int64)]

        def open_(words):
            h = (C.c_uint8 * N.IPC_HANDLE_BYTES).from_buffer_copy(np.array(words, np.int64).tobytes())
            p = C.c_void_p()
            return p.value if self.L.jpg_ipc_open(self.device, h, C.byref(p)) == 0 and p.value else None

        addr = share_from_rank0(self.ex, alloc, free, export, open_, self.L.jpg_ipc_close)
        if addr is None:
            if required:
                raise RuntimeError("memory of rank 0 cannot be opened on every rank through an inter-process handle")
            return None
        if self.rank == 0:
            self._mine.append((addr, free))
        else:
            self._opened.append(addr)
        return addr

    def encode_async(self, width, height, stride, v, stripe_ptr=None, rows=None, bottom_up=True):
        """Queues one pass: this rank's stripe (device rows at stripe_ptr, or sliced out of the host array `rows` of the
        whole image) through the chain.  Up to two passes may be queued before finish() takes the first; rank 0 has
        read (or given up) the scan of the pass before the last one by then, whose half of the buffer the new pass
        fills.  True when this rank has a stripe (only then is there something to finish)."""
        plan = plan_stripes(height, v, self.world)
        mask = sum(1 << k for k, (a, b) in enumerate(plan) if b > a)
        y0, y1 = plan[self.rank]
        if y1 <= y0:
            return False
        if stripe_ptr is not None:
            self.enc.stripe_device(stripe_ptr, width, y1 - y0, stride, bottom_up)
        else:
            self.enc.front_host(stripe_rows(rows, height, y0, y1, bottom_up), width, y1 - y0, stride, bottom_up)
        self.enc.chain_pass_async(mask)
        return True

    def finish(self):
        """Waits for the oldest queued pass.  The length of the whole scan on rank 0 -- whose pass ends when every
        piece lies in the gather buffer -- and None elsewhere."""
        res = self.enc.chain_pass_finish()
        if self.rank != 0:
            return None
        self.offset = res[3]              # (rank 0's piece is the first of the scan)
        return res[4]

    def encode(self, width, height, stride, v, stripe_ptr=None, rows=None, bottom_up=True):
        """One pass, queued and waited for."""
        if not self.encode_async(width, height, stride, v, stripe_ptr, rows, bottom_up):
            return None
        return self.finish()

    def read(self, n):
        """Rank 0: the first n bytes of the scan of the pass finish() took last."""
        import ctypes as C
        out = np.zeros(n, np.uint8)
        if n and self.gather == "device":
            self.N.check(None, self.L.jpg_device_to_host(self.device, out.ctypes.data, self.buffer + self.offset, n))
        elif n:
            C.memmove(out.ctypes.data, self.buffer + self.offset, n)
        return out

    def close(self):
        self.enc.sync()
        for p in self._opened:
            self.L.jpg_ipc_close(p)
        self._opened = []
        self.ex.barrier()                 # no rank has rank 0's memory mapped any more
        for p, free in self._mine:
            free(p)
        self._mine = []


def encode_striped(backend, rows, width, height, stride, v, bottom_up=True, group=None, device=None, exchange=None,
                   gather=None, fetch=True, stripe_ptr=None):
    """Runs the stripe of this rank (torch.distributed must be initialised; one process per GPU) and returns the
    whole stuffed scan on rank 0 (None elsewhere; with fetch=False rank 0 gets its length and the scan stays in
    gather's device buffer).  `rows` may hold the whole image or be any object stripe_rows() can slice; only
    this rank's rows are touched.  The scalars go through the host (ScalarExchange, gloo).  The pieces: a GPU
    backend (jpeg_encoder_amd.Encoder) copies its piece device-to-device into rank 0's buffer (DeviceGather);
    any other backend sends it to rank 0 point to point.  `device` is kept for callers of the earlier
    interface and ignored."""
    ex = exchange or ScalarExchange(group)
    rank, world = ex.rank, ex.world
    job = StripeJob(backend, rank, world, rows, width, height, stride, v, bottom_up, stripe_ptr)
    dcs = ex.all_gather(job.transform())
    codes = ex.all_gather(job.code(dcs))
    on_device = hasattr(backend, "scan_copy_to_device")
    if on_device:
        from . import _native
        own_gather = gather is None
        gather = gather or DeviceGather(_native, backend.device)
        n = job.place_on_device(codes)
        bound = 2 * ((sum(c[0] for c in codes) + 7) // 8) + 16 * world       # every byte stuffed, at worst
        gather.ensure(ex, bound)                                 # (every rank computes the same bound)
        lens = [g[0] for g in ex.all_gather([n])]
        if n:
            backend.scan_copy_to_device(gather.buffer + sum(lens[:rank]), -1, n)
        backend.sync()
        ex.barrier()                                             # every piece has landed
        out = None
        if rank == 0:
            out = gather.read(sum(lens)) if fetch or own_gather else sum(lens)
        if own_gather:
            gather.close(ex)                                     # (a buffer made for this call goes with it)
        return out
    import torch
    import torch.distributed as dist
    pgroup = getattr(ex, "group", None)                              # (a ShmExchange has none: the default group)
    if pgroup is None and dist.get_backend() != "gloo":
        raise RuntimeError("host-memory pieces travel over a gloo group")
    piece = job.place(codes)
    lens = [x[0] for x in ex.all_gather([int(piece.size)])]
    if rank != 0:
        if piece.size:
            dist.send(torch.from_numpy(np.ascontiguousarray(piece)), dst=0, group=pgroup)
        return None
    parts = [piece]
    for k in range(1, world):
        buf = torch.empty(lens[k], dtype=torch.uint8)
        if lens[k]:
            dist.recv(buf, src=k, group=pgroup)
        parts.append(buf.numpy())
    return np.concatenate(parts)
