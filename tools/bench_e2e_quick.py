"""Host rows -> host scan (jpg_encode_scan) for one 8K image, pageable and page-locked rows; prints milliseconds."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jpeg_encoder_amd as J  # noqa: E402

W, H = 7680, 4320
rows = J.host.synth_bgr(W, H, seed=(3 << 32))
L = J._native.lib()
ptr = L.jpg_host_alloc(rows.nbytes)
C.memmove(ptr, rows.ctypes.data, rows.nbytes)
pinned = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), rows.shape)
enc = J.Encoder((4, 2, 0), "real-dct")
for name, src in (("pageable", rows), ("page-locked", pinned)):
    enc.encode_scan(src, W, H)
    ts = []
    for _ in range(8):
        t0 = time.perf_counter()
        enc.encode_scan(src, W, H)
        ts.append((time.perf_counter() - t0) * 1e3)
    print("%s JPG_BANDS=%s: min %.3f median %.3f ms" % (name, os.environ.get("JPG_BANDS", "default"), min(ts), sorted(ts)[len(ts) // 2]), flush=True)
