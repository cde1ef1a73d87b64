"""One large image coded as MCU-row stripes, one stripe per rank.

The reference codes an image serially (src/jpeg/entropy_coding.rs:16-125).  Its scan is
causal in MCU-row order, so it can be cut between MCU rows: a stripe needs from the stripes
above it only (a) the DC value of their last Y, Cb and Cr block (the three predictors of
src/jpeg/entropy_coding.rs:38-40,149-150), (b) the number of bits they produced, which fixes
where its own bits start, and (c) the few bits that share a byte with its first ones.  Each
rank transforms and codes its stripe on its own GPU; the ranks exchange those O(N) scalars,
shift their bits into place, stuff (0xFF -> 0xFF00 must come after the shift) and the pieces
are concatenated.  Only the last stripe keeps the zero-padded partial byte.
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

    def __init__(self, backend, rank, world, rows, width, height, stride, v, bottom_up=True):
        self.backend, self.rank, self.world = backend, rank, world
        self.plan = plan_stripes(height, v, world)
        self.active = [k for k, (a, b) in enumerate(self.plan) if b > a]
        self.y0, self.y1 = self.plan[rank]
        self.empty = self.y1 <= self.y0
        self.rows, self.width, self.height, self.stride, self.bottom_up = rows, width, height, stride, bottom_up

    def transform(self):
        """Colour conversion, subsampling, DCT and quantisation of the stripe -> its last DCs."""
        if self.empty:
            return (0, 0, 0)
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


def encode_striped(backend, rows, width, height, stride, v, bottom_up=True, group=None, device=None):
    """Runs the stripe of this rank (torch.distributed must be initialised; one process per GPU) and
    returns the whole stuffed scan on rank 0, None elsewhere.  `rows` may hold the whole image or be
    any object stripe_rows() can slice; only this rank's rows are touched.  `device` is where the
    exchanged tensors live ("cuda:N" for the nccl backend, "cpu" for gloo)."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    device = device or ("cuda" if dist.get_backend(group) == "nccl" else "cpu")

    def all_gather(values):
        t = torch.tensor(values, dtype=torch.int64, device=device)
        out = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(out, t, group=group)
        return [tuple(int(x) for x in o.tolist()) for o in out]

    job = StripeJob(backend, rank, world, rows, width, height, stride, v, bottom_up)
    dcs = all_gather(list(job.transform()))
    codes = all_gather(list(job.code(dcs)))
    piece = job.place(codes)
    lens = [x[0] for x in all_gather([int(piece.size)])]
    # the pieces travel as one padded all_gather (point-to-point sends would do as well: N <= 8)
    cap = max(max(lens), 1)
    buf = torch.zeros(cap, dtype=torch.uint8, device=device)
    if piece.size:
        buf[:piece.size] = torch.from_numpy(piece).to(device)
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf, group=group)
    if rank != 0:
        return None
    return np.concatenate([o[:n].cpu().numpy() for o, n in zip(out, lens)])
