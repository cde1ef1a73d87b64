"""ctypes binding of the CPU oracle (oracle/librefcpu.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "librefcpu.so")

REAL, BIN = 0, 1
COS_CALL, COS_TABLE = 0, 1

ERRORS = {
    -1: "Invalid chrominance downsampling ratio!",
    -2: "DC coefficient bit length greater than 11!",
    -3: "AC coefficient bit length greater than 10!",
    -4: "DC Huffman Code Not Found!",
    -5: "AC Huffman Code Not Found!",
    -6: "bad argument",
    -7: "Could not open bitmap image file!\n",
}


class RefError(RuntimeError):
    def __init__(self, code):
        super().__init__(ERRORS.get(code, "error %d" % code))
        self.code = code


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("refcpu.c", "refcpu.h", "Makefile")]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        build()
    L = C.CDLL(_SO)
    u8p, i16p, i32p = C.POINTER(C.c_uint8), C.POINTER(C.c_int16), C.POINTER(C.c_int32)
    L.ref_rgb_to_ycbcr.argtypes = [C.c_uint8, C.c_uint8, C.c_uint8, u8p]
    L.ref_real_dct_block.argtypes = [u8p, u8p, i16p, C.c_int]
    L.ref_bin_dct_block.argtypes = [u8p, u8p, i16p]
    L.ref_bin_dct_block_raw.argtypes = [u8p, i32p]
    L.ref_cos_table.argtypes = [C.POINTER(C.c_float)]
    L.ref_quant_table.argtypes = [C.c_int]
    L.ref_quant_table.restype = u8p
    L.ref_zigzag.restype = C.POINTER(C.c_int)
    L.ref_huffman_code.argtypes = [C.c_int, C.c_uint8, C.POINTER(C.c_uint32), u8p]
    L.ref_downsampling_factor.argtypes = [C.c_int] * 3 + [C.POINTER(C.c_int)] * 2
    L.ref_block_walk_max_i32.argtypes = [i32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, i32p]
    L.ref_block_walk_max_i32.restype = C.c_size_t
    L.ref_from_bgr.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_uint32, C.c_int] + [C.c_int] * 4 + [C.POINTER(C.c_int)]
    L.ref_from_bgr.restype = C.c_void_p
    L.ref_from_bmp_file.argtypes = [C.c_char_p] + [C.c_int] * 4 + [C.POINTER(C.c_int)]
    L.ref_from_bmp_file.restype = C.c_void_p
    L.ref_chrominance_downsampling.argtypes = [C.c_void_p, C.c_int]
    L.ref_dct_and_quantization.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.ref_generate_entropy_encoded_data.argtypes = [C.c_void_p]
    L.ref_generate_file.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.ref_generate_file.restype = C.c_size_t
    L.ref_stuffed_scan.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.ref_stuffed_scan.restype = C.c_size_t
    L.ref_free.argtypes = [C.c_void_p]
    L.ref_plane.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_size_t)]
    L.ref_plane.restype = u8p
    L.ref_coeffs.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.ref_coeffs.restype = i16p
    L.ref_set_coeffs.argtypes = [C.c_void_p, C.c_int, i16p, C.c_int, C.c_int]
    L.ref_bits.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_size_t)]
    L.ref_bits.restype = u8p
    L.ref_bmp_load_rgb.argtypes = [C.c_char_p, i32p, i32p, C.POINTER(C.c_int)]
    L.ref_bmp_load_rgb.restype = C.c_void_p
    L.ref_free_buf.argtypes = [C.c_void_p]
    _lib = L
    return L


def _u8(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def rgb_to_ycbcr(r, g, b):
    out = (C.c_uint8 * 3)()
    lib().ref_rgb_to_ycbcr(r, g, b, out)
    return tuple(out)


def quant_table(chroma):
    return np.ctypeslib.as_array(lib().ref_quant_table(int(chroma)), (64,)).copy()


def zigzag():
    return np.ctypeslib.as_array(lib().ref_zigzag(), (64,)).copy()


def cos_table():
    out = np.zeros(64, np.float32)
    lib().ref_cos_table(out.ctypes.data_as(C.POINTER(C.c_float)))
    return out.reshape(8, 8)


def real_dct_block(block, chroma=False, cos_mode=COS_CALL):
    b = np.ascontiguousarray(block, np.uint8).reshape(64)
    q = quant_table(chroma)
    out = np.zeros(64, np.int16)
    lib().ref_real_dct_block(_u8(b), _u8(q), out.ctypes.data_as(C.POINTER(C.c_int16)), cos_mode)
    return out


def bin_dct_block(block, chroma=False):
    b = np.ascontiguousarray(block, np.uint8).reshape(64)
    q = quant_table(chroma)
    out = np.zeros(64, np.int16)
    lib().ref_bin_dct_block(_u8(b), _u8(q), out.ctypes.data_as(C.POINTER(C.c_int16)))
    return out


def bin_dct_block_raw(block):
    b = np.ascontiguousarray(block, np.uint8).reshape(64)
    out = np.zeros(64, np.int32)
    lib().ref_bin_dct_block_raw(_u8(b), out.ctypes.data_as(C.POINTER(C.c_int32)))
    return out


def huffman_code(table, symbol):
    code, ln = C.c_uint32(), C.c_uint8()
    ok = lib().ref_huffman_code(table, symbol, C.byref(code), C.byref(ln))
    return (code.value, ln.value) if ok else None


def downsampling_factor(a, b, c):
    h, v = C.c_int(), C.c_int()
    e = lib().ref_downsampling_factor(a, b, c, C.byref(h), C.byref(v))
    if e:
        raise RefError(e)
    return h.value, v.value


def block_walk_max(matrix, bw, bh, pad):
    m = np.ascontiguousarray(matrix, np.int32)
    out = np.zeros(m.size + 64, np.int32)
    n = lib().ref_block_walk_max_i32(m.ctypes.data_as(C.POINTER(C.c_int32)), m.shape[1], m.shape[0], bw, bh,
                                     int(pad), out.ctypes.data_as(C.POINTER(C.c_int32)))
    return out[:n].tolist()


def bmp_load_rgb(path):
    w, h, e = C.c_int32(), C.c_int32(), C.c_int()
    p = lib().ref_bmp_load_rgb(os.fsencode(path), C.byref(w), C.byref(h), C.byref(e))
    if not p:
        raise RefError(e.value)
    a = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), (h.value, w.value, 3)).copy()
    lib().ref_free_buf(p)
    return a


class RefImage:
    """One reference JpegImage; methods are named after the reference's (jpeg_image.rs, main.rs:17-66)."""

    def __init__(self, handle):
        self._h = handle

    @classmethod
    def from_bgr(cls, pixels, width, height, stride, ratio=(4, 2, 0), algo=REAL, bottom_up=True):
        px = np.ascontiguousarray(pixels, np.uint8)
        assert px.size >= stride * (height - 1) + 3 * width
        e = C.c_int()
        h = lib().ref_from_bgr(px.ctypes.data, width, height, stride, int(bottom_up), *ratio, algo, C.byref(e))
        if not h:
            raise RefError(e.value)
        return cls(h)

    @classmethod
    def from_bmp(cls, path, ratio=(4, 2, 0), algo=REAL):
        e = C.c_int()
        h = lib().ref_from_bmp_file(os.fsencode(path), *ratio, algo, C.byref(e))
        if not h:
            raise RefError(e.value)
        return cls(h)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().ref_free(self._h)
            self._h = None

    def chrominance_downsampling(self, threads=1):
        lib().ref_chrominance_downsampling(self._h, threads)
        return self

    def dct_and_quantization(self, cos_mode=COS_TABLE, threads=1):
        lib().ref_dct_and_quantization(self._h, cos_mode, threads)
        return self

    def generate_entropy_encoded_data(self):
        e = lib().ref_generate_entropy_encoded_data(self._h)
        if e:
            raise RefError(e)
        return self

    def encode(self, cos_mode=COS_TABLE, threads=1):
        return (self.chrominance_downsampling(threads).dct_and_quantization(cos_mode, threads)
                .generate_entropy_encoded_data())

    def plane(self, comp):
        w, h, n = C.c_int(), C.c_int(), C.c_size_t()
        p = lib().ref_plane(self._h, comp, C.byref(w), C.byref(h), C.byref(n))
        raw = np.ctypeslib.as_array(p, (n.value,)).copy()
        return raw, w.value, h.value

    def plane2d(self, comp):
        raw, w, h = self.plane(comp)
        return raw[: w * h].reshape(h, w)

    def coeffs(self, comp):
        w, h = C.c_int(), C.c_int()
        p = lib().ref_coeffs(self._h, comp, C.byref(w), C.byref(h))
        return np.ctypeslib.as_array(p, (h.value, w.value)).copy()

    def set_coeffs(self, comp, plane):
        """Replace one component's quantised coefficients (same shape as coeffs(comp))."""
        a = np.ascontiguousarray(plane, np.int16)
        e = lib().ref_set_coeffs(self._h, comp, a.ctypes.data_as(C.POINTER(C.c_int16)), a.shape[1], a.shape[0])
        if e:
            raise RefError(e)
        return self

    def bits(self):
        nb, ny = C.c_uint64(), C.c_size_t()
        p = lib().ref_bits(self._h, C.byref(nb), C.byref(ny))
        data = np.ctypeslib.as_array(p, (ny.value,)).copy() if ny.value else np.zeros(0, np.uint8)
        return data, nb.value

    def stuffed_scan(self):
        n = lib().ref_stuffed_scan(self._h, None, 0)
        out = np.zeros(n, np.uint8)
        lib().ref_stuffed_scan(self._h, out.ctypes.data, n)
        return out

    def file_bytes(self):
        n = lib().ref_generate_file(self._h, None, 0)
        out = np.zeros(n, np.uint8)
        lib().ref_generate_file(self._h, out.ctypes.data, n)
        return out.tobytes()
