"""ctypes binding of libjpeg_hip.so (C ABI in include/jpeg_hip.h).

There is no CPU fallback: if the library is missing or a call fails, an
exception is raised.
"""
import ctypes as C
import importlib.util
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# JPEG_HIP_LIB selects another build of the same library (kernel A/B experiments)
LIB_PATH = os.environ.get("JPEG_HIP_LIB") or os.path.join(_HERE, "libjpeg_hip.so")

DCT_REAL, DCT_BIN, DCT_REAL_STRICT = 0, 1, 2

# every symbol include/jpeg_hip.h declares
SYMBOLS = [
    "jpg_strerror", "jpg_last_hip_error", "jpg_create", "jpg_destroy", "jpg_stream", "jpg_sync",
    "jpg_load_bgr", "jpg_load_bgr_device", "jpg_downsample", "jpg_dct_quant", "jpg_entropy", "jpg_stuff",
    "jpg_front_device", "jpg_front_host", "jpg_encode_device_async", "jpg_finish", "jpg_finish_next", "jpg_max_batch", "jpg_encode_batch_device_async", "jpg_batch_result",
    "jpg_get_batch_scan", "jpg_encode_scan", "jpg_set_timing", "jpg_front_time", "jpg_host_alloc", "jpg_free",
    "jpg_entropy_stripe", "jpg_stream_tail", "jpg_last_dc", "jpg_stitch", "jpg_bits_device", "jpg_scan_device",
    "jpg_get_bits", "jpg_get_scan", "jpg_plane_dims", "jpg_coeff_dims", "jpg_get_plane", "jpg_get_coeffs",
    "jpg_front_bytes", "jpg_num_blocks", "jpg_fused_coding",
    "jpg_front_samples", "jpg_stripe_device", "jpg_stripe_host_async", "jpg_last_dc_async", "jpg_entropy_stripe_async", "jpg_entropy_stripe_finish",
    "jpg_stitch_async", "jpg_stitch_finish", "jpg_scan_copy_to_device", "jpg_device_alloc", "jpg_device_free",
    "jpg_device_to_host", "jpg_ipc_export", "jpg_ipc_open", "jpg_ipc_close",
    "jpg_encode_scan_begin", "jpg_encode_scan_end",
    "jpg_static_init", "jpg_static_scan_capacity", "jpg_encode_device_static", "jpg_static_status",
    "jpg_chain_mailbox_bytes", "jpg_chain_mailbox_alloc", "jpg_peer_access",
    "jpg_chain_setup", "jpg_chain_pass_async", "jpg_chain_pass_finish",
]
STATIC_SCRATCH_BYTES = 24 << 20


class StaticResult(C.Structure):
    _fields_ = [("nbits", C.c_uint64), ("scan_bytes", C.c_uint64), ("error", C.c_int32), ("reserved", C.c_uint32 * 3)]

IPC_HANDLE_BYTES = 64


class JpgError(RuntimeError):
    """A negative jpg_status; the message is the reference's panic text where one exists."""

    def __init__(self, code, message, hip_error=0):
        super().__init__("%s (status %d%s)" % (message, code, ", hipError %d" % hip_error if hip_error else ""))
        self.code = code
        self.hip_error = hip_error


_lib = None


def _share_hip_runtime():
    """PyTorch wheels ship their own libamdhip64.so.7.  Two HIP runtimes in one process do not see
    each other's devices, so when PyTorch is installed its copy is loaded first and libjpeg_hip.so
    binds to it by soname, whichever of the two libraries the program touches first.  Without
    PyTorch the system runtime (RUNPATH /opt/rocm) is used."""
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec and spec.submodule_search_locations:
        for d in spec.submodule_search_locations:
            p = os.path.join(d, "lib", "libamdhip64.so")
            if os.path.exists(p):
                C.CDLL(p, mode=C.RTLD_GLOBAL)
                return


def lib():
    """Load libjpeg_hip.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s is missing: build it with `python jpeg-encoder_amd/build.py`" % LIB_PATH)
    _share_hip_runtime()
    L = C.CDLL(LIB_PATH)
    vp, i32, u32, u64p, szp = C.c_void_p, C.c_int32, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_size_t)
    L.jpg_strerror.argtypes = [C.c_int]
    L.jpg_strerror.restype = C.c_char_p
    L.jpg_last_hip_error.argtypes = [vp]
    L.jpg_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    L.jpg_destroy.argtypes = [vp]
    L.jpg_destroy.restype = None
    L.jpg_stream.argtypes = [vp]
    L.jpg_stream.restype = vp
    L.jpg_sync.argtypes = [vp]
    for name in ("jpg_load_bgr", "jpg_load_bgr_device", "jpg_front_device", "jpg_front_host",
                 "jpg_encode_device_async", "jpg_stripe_device", "jpg_stripe_host_async"):
        getattr(L, name).argtypes = [vp, vp, i32, i32, u32, C.c_int]
    L.jpg_downsample.argtypes = [vp]
    L.jpg_dct_quant.argtypes = [vp]
    L.jpg_entropy.argtypes = [vp, u64p]
    L.jpg_stuff.argtypes = [vp, szp]
    L.jpg_finish.argtypes = [vp, u64p, szp]
    L.jpg_finish_next.argtypes = [vp, u64p, szp]
    L.jpg_max_batch.argtypes = []
    L.jpg_encode_batch_device_async.argtypes = [vp, C.POINTER(vp), i32, i32, i32, u32, C.c_int]
    L.jpg_batch_result.argtypes = [vp, i32, u64p, szp, C.POINTER(vp)]
    L.jpg_get_batch_scan.argtypes = [vp, i32, vp, C.c_size_t]
    L.jpg_encode_scan.argtypes = [vp, vp, i32, i32, u32, C.c_int, C.POINTER(vp), szp]
    L.jpg_set_timing.argtypes = [vp, C.c_int]
    L.jpg_front_time.argtypes = [vp, C.POINTER(C.c_double), u64p, u64p]
    L.jpg_host_alloc.argtypes = [C.c_size_t]
    L.jpg_host_alloc.restype = vp
    L.jpg_free.argtypes = [vp]
    L.jpg_free.restype = None
    L.jpg_entropy_stripe.argtypes = [vp, C.POINTER(C.c_int16), u64p]
    L.jpg_last_dc.argtypes = [vp, C.POINTER(C.c_int16)]
    L.jpg_stream_tail.argtypes = [vp, C.POINTER(C.c_uint32)]
    L.jpg_stitch.argtypes = [vp, C.c_uint64, C.c_uint8, C.c_int, szp, C.POINTER(C.c_uint8)]
    L.jpg_bits_device.argtypes = [vp]
    L.jpg_bits_device.restype = vp
    L.jpg_scan_device.argtypes = [vp]
    L.jpg_scan_device.restype = vp
    L.jpg_get_bits.argtypes = [vp, vp, C.c_size_t]
    L.jpg_get_scan.argtypes = [vp, vp, C.c_size_t]
    L.jpg_plane_dims.argtypes = [vp, C.c_int, C.POINTER(i32), C.POINTER(i32)]
    L.jpg_coeff_dims.argtypes = [vp, C.c_int, C.POINTER(i32), C.POINTER(i32)]
    L.jpg_get_plane.argtypes = [vp, C.c_int, vp, C.c_size_t]
    L.jpg_get_coeffs.argtypes = [vp, C.c_int, vp, C.c_size_t]
    L.jpg_front_bytes.argtypes = [vp]
    L.jpg_front_bytes.restype = C.c_uint64
    L.jpg_num_blocks.argtypes = [vp]
    L.jpg_num_blocks.restype = C.c_uint64
    L.jpg_fused_coding.argtypes = [vp]
    L.jpg_fused_coding.restype = C.c_int
    L.jpg_front_samples.argtypes = [vp, C.POINTER(C.c_float), i32, C.POINTER(i32)]
    L.jpg_last_dc_async.argtypes = [vp]
    L.jpg_entropy_stripe_async.argtypes = [vp, C.POINTER(C.c_int16)]
    L.jpg_entropy_stripe_finish.argtypes = [vp, u64p, C.POINTER(C.c_uint32)]
    L.jpg_stitch_async.argtypes = [vp, C.c_uint64, C.c_uint8, C.c_int]
    L.jpg_stitch_finish.argtypes = [vp, szp, C.POINTER(C.c_uint8)]
    L.jpg_scan_copy_to_device.argtypes = [vp, vp, C.c_int, C.c_size_t]
    L.jpg_device_alloc.argtypes = [C.c_int, C.c_size_t]
    L.jpg_device_alloc.restype = vp
    L.jpg_device_free.argtypes = [C.c_int, vp]
    L.jpg_device_free.restype = None
    L.jpg_device_to_host.argtypes = [C.c_int, vp, vp, C.c_size_t]
    L.jpg_ipc_export.argtypes = [vp, C.POINTER(C.c_uint8)]
    L.jpg_ipc_open.argtypes = [C.c_int, C.POINTER(C.c_uint8), C.POINTER(vp)]
    L.jpg_ipc_close.argtypes = [vp]
    L.jpg_encode_scan_begin.argtypes = [vp, vp, i32, i32, u32, C.c_int]
    L.jpg_encode_scan_end.argtypes = [vp, C.POINTER(vp), szp]
    L.jpg_static_init.argtypes = [C.c_int]
    L.jpg_static_scan_capacity.argtypes = [i32, i32, C.c_int, C.c_int, C.c_int]
    L.jpg_static_scan_capacity.restype = C.c_size_t
    L.jpg_encode_device_static.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, i32, i32, u32, C.c_int, vp,
                                           C.c_size_t, vp, vp]
    L.jpg_static_status.argtypes = [C.POINTER(StaticResult)]
    L.jpg_chain_mailbox_bytes.argtypes = []
    L.jpg_chain_mailbox_bytes.restype = C.c_size_t
    L.jpg_chain_mailbox_alloc.argtypes = []
    L.jpg_chain_mailbox_alloc.restype = vp
    L.jpg_peer_access.argtypes = [C.c_int, C.c_int]
    L.jpg_chain_setup.argtypes = [vp, C.c_int, C.c_int, vp, vp, C.c_size_t]
    L.jpg_chain_pass_async.argtypes = [vp, C.c_uint64]
    L.jpg_chain_pass_finish.argtypes = [vp, u64p, szp, u64p, szp, szp]
    _lib = L
    return L


def check(ctx, status):
    if status != 0:
        L = lib()
        hip = L.jpg_last_hip_error(ctx) if ctx else 0
        raise JpgError(status, L.jpg_strerror(status).decode(), hip)
