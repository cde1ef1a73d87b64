"""A stripe backend for CPU tests, built on the oracle: what a GPU rank would produce for its
stripe is cut out of the oracle's whole-image results (the scan is causal in MCU-row order, so
the oracle's stream for the rows above a cut is a prefix of the whole stream)."""
import numpy as np


def unpack(data, nbits):
    return np.unpackbits(np.asarray(data, np.uint8))[:nbits]


def stuff(b):
    b = np.asarray(b, np.uint8)
    out = np.zeros(2 * len(b), np.uint8)
    idx = np.arange(len(b)) + np.concatenate([[0], np.cumsum(b == 0xff)[:-1]]) if len(b) else np.zeros(0, int)
    out[idx] = b
    return out[:len(b) + int((b == 0xff).sum())]


class OracleStripe:
    def __init__(self, oracle, rows, width, height, stride, ratio, algo, y0, y1, bottom_up=True):
        self.o, self.rows, self.w, self.h, self.stride = oracle, rows, width, height, stride
        self.ratio, self.algo, self.y0, self.y1, self.bottom_up = ratio, algo, y0, y1, bottom_up
        self.seen_pred = None

    def _top(self, y):
        """Oracle image of rows [0, y) of the picture."""
        sub = self.rows[self.h - y:] if self.bottom_up else self.rows[:y]
        return self.o.RefImage.from_bgr(sub, self.w, y, self.stride, self.ratio, self.algo,
                                        bottom_up=self.bottom_up).encode()

    @staticmethod
    def _last_dcs(im):
        return tuple(int(im.coeffs(c)[-8, -8]) for c in range(3))

    def front(self, sub, width, height, stride, bottom_up):
        want = self.rows[self.h - self.y1:self.h - self.y0] if self.bottom_up else self.rows[self.y0:self.y1]
        assert (width, height, stride) == (self.w, self.y1 - self.y0, self.stride)
        assert np.array_equal(np.asarray(sub), want)
        self.upto = self._top(self.y1)

    def last_dc(self):
        return self._last_dcs(self.upto)

    def entropy(self, pred):
        self.seen_pred = tuple(int(x) for x in pred)
        want = self._last_dcs(self._top(self.y0)) if self.y0 > 0 else (0, 0, 0)
        assert self.seen_pred == want, (self.seen_pred, want)
        start = self._top(self.y0).bits()[1] if self.y0 > 0 else 0
        data, end = self.upto.bits()
        self.bits = unpack(data, end)[start:]
        return len(self.bits)

    def stream_tail(self):
        t = self.bits[-32:]
        return int("".join(map(str, t)), 2) if len(t) else 0

    def stitch(self, bit_offset, head, is_last):
        r = bit_offset % 8
        lead = np.unpackbits(np.array([head], np.uint8))[:r]
        allb = np.concatenate([lead, self.bits])
        whole = len(allb) // 8 * 8
        tail = 0
        if is_last:
            body = np.packbits(allb)                      # zero-padded partial byte is kept
        else:
            body = np.packbits(allb[:whole])
            if len(allb) > whole:
                tail = int(np.packbits(allb[whole:])[0])
        self.piece = stuff(body)
        return len(self.piece), tail

    def scan(self, n):
        return self.piece[:n]
