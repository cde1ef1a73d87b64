"""Host-side mirror of the reference's `JpegImage` (src/jpeg/jpeg_image.rs:8-21)
over the C ABI of libjpeg_hip.so.

The five methods `main` calls in order (src/main.rs:17,35,48,61,66) keep their
names and meaning; `Encoder` is the fused fast path (one kernel from BMP rows
to quantised coefficients).  All compute runs on the GPU; there is no CPU
fallback.
"""
import ctypes as C

import numpy as np

from . import _native as N
from . import host

DCT_ALGORITHMS = {"real-dct": N.DCT_REAL, "bin-dct": N.DCT_BIN, "real-dct-strict": N.DCT_REAL_STRICT,
                  "RealDct": N.DCT_REAL, "BinDct": N.DCT_BIN}


def get_downsampling_factor(ratio):
    """(4,4,4)->(1,1), (4,2,0)->(2,2), (4,2,2)->(2,1) (src/jpeg/sampling.rs:6-22)."""
    table = {(4, 4, 4): (1, 1), (4, 2, 0): (2, 2), (4, 2, 2): (2, 1)}
    try:
        return table[tuple(ratio)]
    except KeyError:
        raise ValueError("Invalid chrominance downsampling ratio!")


def _algo(a):
    return DCT_ALGORITHMS[a] if isinstance(a, str) else int(a)


class Encoder:
    """One jpg_ctx: device buffers + a HIP stream for one (ratio, algorithm)."""

    def __init__(self, ratio=(4, 2, 0), dct_algorithm="real-dct", device=0):
        self._L = N.lib()
        self._ctx = C.c_void_p()
        self.ratio = tuple(ratio)
        self.algo = _algo(dct_algorithm)
        st = self._L.jpg_create(C.byref(self._ctx), device, self.algo, *self.ratio)
        N.check(None, st)
        self.h, self.v = get_downsampling_factor(self.ratio)
        self.device = device
        self.width = self.height = 0

    def close(self):
        if getattr(self, "_ctx", None):
            self._L.jpg_destroy(self._ctx)
            self._ctx = None

    __del__ = close

    def _ck(self, st):
        N.check(self._ctx, st)

    @property
    def stream(self):
        """hipStream_t of the context as an integer."""
        return self._L.jpg_stream(self._ctx)

    def sync(self):
        self._ck(self._L.jpg_sync(self._ctx))

    # ---- staged path -----------------------------------------------------
    def load_bgr(self, rows, width, height, stride=None, bottom_up=True):
        rows = np.ascontiguousarray(rows, np.uint8)
        stride = stride or rows.shape[-1]
        assert rows.size >= stride * (height - 1) + 3 * width
        self.width, self.height = width, height
        self._ck(self._L.jpg_load_bgr(self._ctx, rows.ctypes.data, width, height, stride, int(bottom_up)))

    def load_bgr_device(self, dev_ptr, width, height, stride, bottom_up=True):
        self.width, self.height = width, height
        self._ck(self._L.jpg_load_bgr_device(self._ctx, dev_ptr, width, height, stride, int(bottom_up)))

    def downsample(self):
        self._ck(self._L.jpg_downsample(self._ctx))

    def dct_quant(self):
        self._ck(self._L.jpg_dct_quant(self._ctx))

    def entropy(self, pred=None):
        n = C.c_uint64()
        if pred is None:
            self._ck(self._L.jpg_entropy(self._ctx, C.byref(n)))
        else:
            p = (C.c_int16 * 3)(*[int(x) for x in pred])
            self._ck(self._L.jpg_entropy_stripe(self._ctx, p, C.byref(n)))
        return n.value

    def stuff(self):
        n = C.c_size_t()
        self._ck(self._L.jpg_stuff(self._ctx, C.byref(n)))
        return n.value

    def stitch(self, bit_offset, head_bits, is_last):
        n, tail = C.c_size_t(), C.c_uint8()
        self._ck(self._L.jpg_stitch(self._ctx, bit_offset, head_bits, int(is_last), C.byref(n), C.byref(tail)))
        return n.value, tail.value

    def stream_tail(self):
        """The last min(32, nbits) bits of the stripe's bit string, right-aligned."""
        v = C.c_uint32()
        self._ck(self._L.jpg_stream_tail(self._ctx, C.byref(v)))
        return v.value

    def last_dc(self):
        dc = (C.c_int16 * 3)()
        self._ck(self._L.jpg_last_dc(self._ctx, dc))
        return tuple(dc)

    # the same steps queued without waiting (one host thread driving several contexts)
    def last_dc_async(self):
        self._ck(self._L.jpg_last_dc_async(self._ctx))

    def entropy_stripe_async(self, pred):
        p = (C.c_int16 * 3)(*[int(x) for x in pred])
        self._ck(self._L.jpg_entropy_stripe_async(self._ctx, p))

    def entropy_stripe_finish(self):
        """(nbits, last min(32, nbits) bits right-aligned) of the stripe queued with entropy_stripe_async."""
        n, t = C.c_uint64(), C.c_uint32()
        self._ck(self._L.jpg_entropy_stripe_finish(self._ctx, C.byref(n), C.byref(t)))
        return n.value, t.value

    def stitch_async(self, bit_offset, head_bits, is_last):
        self._ck(self._L.jpg_stitch_async(self._ctx, bit_offset, head_bits, int(is_last)))

    def stitch_finish(self):
        n, tail = C.c_size_t(), C.c_uint8()
        self._ck(self._L.jpg_stitch_finish(self._ctx, C.byref(n), C.byref(tail)))
        return n.value, tail.value

    def stripe_device(self, dev_ptr, width, height, stride, bottom_up=True):
        """A stripe from device-resident rows that stay valid until the stripe is coded."""
        self.width, self.height = width, height
        self._ck(self._L.jpg_stripe_device(self._ctx, dev_ptr, width, height, stride, int(bottom_up)))

    def scan_copy_to_device(self, dst_ptr, dst_device, cap):
        """Queue the copy of the stuffed piece to device memory (a peer copy when dst_device is another GPU)."""
        self._ck(self._L.jpg_scan_copy_to_device(self._ctx, dst_ptr, dst_device, cap))

    # ---- stripe chain: a whole pass of the stripe queued at once, the ranks' kernels hand the scalars on themselves
    def chain_setup(self, rank, world, mailbox_ptr, gather_ptr, gather_cap):
        self._ck(self._L.jpg_chain_setup(self._ctx, rank, world, mailbox_ptr, gather_ptr, gather_cap))

    def chain_pass_async(self, active_mask):
        self._ck(self._L.jpg_chain_pass_async(self._ctx, active_mask))

    def chain_pass_finish(self):
        """(bits of the stripe, bytes of its stuffed piece, bit offset, byte offset, bytes of the whole scan on rank 0)."""
        nbits, off = C.c_uint64(), C.c_uint64()
        n, boff, total = C.c_size_t(), C.c_size_t(), C.c_size_t()
        self._ck(self._L.jpg_chain_pass_finish(self._ctx, C.byref(nbits), C.byref(n), C.byref(off), C.byref(boff), C.byref(total)))
        return nbits.value, n.value, off.value, boff.value, total.value

    # ---- fused path ------------------------------------------------------
    def front_device(self, dev_ptr, width, height, stride, bottom_up=True):
        self.width, self.height = width, height
        self._ck(self._L.jpg_front_device(self._ctx, dev_ptr, width, height, stride, int(bottom_up)))

    def front_host(self, rows, width, height, stride=None, bottom_up=True):
        rows = np.ascontiguousarray(rows, np.uint8)
        stride = stride or rows.shape[-1]
        assert rows.size >= stride * (height - 1) + 3 * width
        self.width, self.height = width, height
        self._ck(self._L.jpg_front_host(self._ctx, rows.ctypes.data, width, height, stride, int(bottom_up)))

    def encode_device_async(self, dev_ptr, width, height, stride, bottom_up=True):
        self.width, self.height = width, height
        self._ck(self._L.jpg_encode_device_async(self._ctx, dev_ptr, width, height, stride, int(bottom_up)))

    def finish(self):
        nbits, n = C.c_uint64(), C.c_size_t()
        self._ck(self._L.jpg_finish(self._ctx, C.byref(nbits), C.byref(n)))
        return nbits.value, n.value

    def encode_batch_device_async(self, dev_ptrs, width, height, stride, bottom_up=True):
        """Queue one launch set for len(dev_ptrs) equally sized device-resident images."""
        n = len(dev_ptrs)
        arr = dev_ptrs if isinstance(dev_ptrs, C.Array) else (C.c_void_p * n)(*dev_ptrs)   # (a prepared array is used as is)
        self.width, self.height = width, height
        self._ck(self._L.jpg_encode_batch_device_async(self._ctx, arr, n, width, height, stride, int(bottom_up)))

    def batch_result(self, i):
        """(nbits, stuffed length, device address of the scan) of image i of the retired batch."""
        nbits, n, p = C.c_uint64(), C.c_size_t(), C.c_void_p()
        self._ck(self._L.jpg_batch_result(self._ctx, i, C.byref(nbits), C.byref(n), C.byref(p)))
        return nbits.value, n.value, p.value

    def batch_scan(self, i):
        _, n, _ = self.batch_result(i)
        out = np.zeros(n, np.uint8)
        if n:
            self._ck(self._L.jpg_get_batch_scan(self._ctx, i, out.ctypes.data, out.nbytes))
        return out

    def max_batch(self):
        return self._L.jpg_max_batch()

    def finish_next(self):
        """Wait for the oldest of (up to two) queued encodes only: (nbits, stuffed length)."""
        nbits, n = C.c_uint64(), C.c_size_t()
        self._ck(self._L.jpg_finish_next(self._ctx, C.byref(nbits), C.byref(n)))
        return nbits.value, n.value

    def encode_scan(self, rows, width, height, stride=None, bottom_up=True):
        """Host rows -> stuffed scan bytes (numpy uint8)."""
        rows = np.ascontiguousarray(rows, np.uint8)
        stride = stride or rows.shape[-1]
        assert rows.size >= stride * (height - 1) + 3 * width
        self.width, self.height = width, height
        p, n = C.c_void_p(), C.c_size_t()
        self._ck(self._L.jpg_encode_scan(self._ctx, rows.ctypes.data, width, height, stride, int(bottom_up),
                                         C.byref(p), C.byref(n)))
        out = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), (n.value,)).copy() if n.value else np.zeros(0, np.uint8)
        self._L.jpg_free(p)
        return out

    def encode_file_bytes(self, rows, width, height, stride=None, bottom_up=True):
        scan = self.encode_scan(rows, width, height, stride, bottom_up)
        return host.assemble_jpeg(width, height, self.h, self.v, scan)

    # ---- results ---------------------------------------------------------
    def plane(self, comp):
        w, h = C.c_int32(), C.c_int32()
        self._ck(self._L.jpg_plane_dims(self._ctx, comp, C.byref(w), C.byref(h)))
        out = np.zeros((h.value, w.value), np.uint8)
        self._ck(self._L.jpg_get_plane(self._ctx, comp, out.ctypes.data, out.nbytes))
        return out

    def coeffs(self, comp):
        w, h = C.c_int32(), C.c_int32()
        self._ck(self._L.jpg_coeff_dims(self._ctx, comp, C.byref(w), C.byref(h)))
        out = np.zeros((h.value, w.value), np.int16)
        self._ck(self._L.jpg_get_coeffs(self._ctx, comp, out.ctypes.data, out.nbytes))
        return out

    def bits(self, nbits):
        out = np.zeros((nbits + 7) // 8, np.uint8)
        if out.size:
            self._ck(self._L.jpg_get_bits(self._ctx, out.ctypes.data, out.nbytes))
        return out

    def scan(self, length):
        out = np.zeros(length, np.uint8)
        if length:
            self._ck(self._L.jpg_get_scan(self._ctx, out.ctypes.data, out.nbytes))
        return out

    def scan_device_ptr(self):
        return self._L.jpg_scan_device(self._ctx)

    def bits_device_ptr(self):
        return self._L.jpg_bits_device(self._ctx)

    def set_timing(self, every=1):
        """Bracket every n-th front launch with events (0 = off)."""
        self._ck(self._L.jpg_set_timing(self._ctx, int(every)))

    def front_time(self):
        """(summed milliseconds, launches, images) of the time-stamped front launches since set_timing."""
        ms, n, im = C.c_double(), C.c_uint64(), C.c_uint64()
        self._ck(self._L.jpg_front_time(self._ctx, C.byref(ms), C.byref(n), C.byref(im)))
        return ms.value, n.value, im.value

    def front_samples(self):
        """Durations (ms) of the individual time-stamped launches since set_timing, in launch order."""
        buf = (C.c_float * 4096)()
        n = C.c_int32()
        self._ck(self._L.jpg_front_samples(self._ctx, buf, 4096, C.byref(n)))
        return [float(buf[k]) for k in range(n.value)]

    def front_bytes(self):
        return self._L.jpg_front_bytes(self._ctx)

    def num_blocks(self):
        return self._L.jpg_num_blocks(self._ctx)

    def fused_coding(self):
        """True when whole encodes of the current image size run the single pixels-to-bits kernel."""
        return bool(self._L.jpg_fused_coding(self._ctx))


class JpegImage:
    """Same five steps, same names, as the reference's JpegImage."""

    def __init__(self, path, width, height, chrominance_downsampling_ratio, dct_algorithm, device=0):
        self.path = path
        self.width, self.height = width, height
        self.chrominance_downsampling_ratio = tuple(chrominance_downsampling_ratio)
        self.dct_algorithm = dct_algorithm
        self._enc = Encoder(self.chrominance_downsampling_ratio, dct_algorithm, device)
        self._nbits = None
        self._scan_len = None

    @classmethod
    def from_bmp(cls, bmp_path, jpeg_path, chrominance_downsampling_ratio=(4, 2, 0), dct_algorithm="real-dct",
                 device=0):
        """src/jpeg/jpeg_image.rs:104-137: read the BMP, convert to YCbCr planes (on the GPU)."""
        get_downsampling_factor(chrominance_downsampling_ratio)
        rows, w, h, stride = host.read_bmp(bmp_path)
        im = cls(jpeg_path, w, h, chrominance_downsampling_ratio, dct_algorithm, device)
        im._enc.load_bgr(rows, w, h, stride, bottom_up=True)
        return im

    @classmethod
    def from_bgr_rows(cls, rows, width, height, jpeg_path=None, chrominance_downsampling_ratio=(4, 2, 0),
                      dct_algorithm="real-dct", stride=None, bottom_up=True, device=0):
        im = cls(jpeg_path, width, height, chrominance_downsampling_ratio, dct_algorithm, device)
        im._enc.load_bgr(rows, width, height, stride, bottom_up)
        return im

    def chrominance_downsampling(self):
        self._enc.downsample()

    def dct_and_quantization(self):
        self._enc.dct_quant()

    def generate_entropy_encoded_data(self):
        self._nbits = self._enc.entropy()

    def generate_file(self, path=None):
        self._scan_len = self._enc.stuff()
        scan = self._enc.scan(self._scan_len)
        h, v = get_downsampling_factor(self.chrominance_downsampling_ratio)
        host.write_jpeg(path or self.path, self.width, self.height, h, v, scan)

    def file_bytes(self):
        self._scan_len = self._enc.stuff()
        scan = self._enc.scan(self._scan_len)
        h, v = get_downsampling_factor(self.chrominance_downsampling_ratio)
        return host.assemble_jpeg(self.width, self.height, h, v, scan)

    # public fields of the reference struct
    @property
    def y_channel(self):
        return self._enc.plane(0)

    @property
    def cb_channel(self):
        return self._enc.plane(1)

    @property
    def cr_channel(self):
        return self._enc.plane(2)

    @property
    def y_dct_coeffs(self):
        return self._enc.coeffs(0)

    @property
    def cb_dct_coeffs(self):
        return self._enc.coeffs(1)

    @property
    def cr_dct_coeffs(self):
        return self._enc.coeffs(2)

    @property
    def entropy_coded_bits(self):
        """(bytes, nbits): BitVec::as_raw_slice() and its length in bits."""
        return self._enc.bits(self._nbits), self._nbits
