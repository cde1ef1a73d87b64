"""End-to-end figures that include the PCIe transfers and the file I/O (never the headline value):
host rows -> stuffed scan on the host (jpg_encode_scan), and BMP file -> JPEG file with the CLI."""
import os
import subprocess
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jpeg_encoder_amd as J  # noqa: E402

W, H = 7680, 4320
rows = J.host.synth_bgr(W, H, seed=(3 << 32))
import ctypes as C
import numpy as np
L = J._native.lib()
ptr = L.jpg_host_alloc(rows.nbytes)
C.memmove(ptr, rows.ctypes.data, rows.nbytes)
pinned = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), rows.shape)
for algo in ("real-dct", "bin-dct"):
    for name, src in (("pageable", rows), ("page-locked", pinned)):
        enc = J.Encoder((4, 2, 0), algo)
        enc.encode_scan(src, W, H)
        t0 = time.perf_counter()
        n = 5
        for _ in range(n):
            scan = enc.encode_scan(src, W, H)
        dt = (time.perf_counter() - t0) / n
        print("%s host rows (%s) -> host scan (H2D + encode + D2H): %.2f ms, %.0f MP/s, scan %d B" %
              (algo, name, dt * 1e3, W * H / dt / 1e6, scan.size), flush=True)
        enc.close()
cli = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "jpeg-encoder_amd", "jpeg-encoder")
with tempfile.TemporaryDirectory() as d:
    bmp = os.path.join(d, "in.bmp")
    J.host.write_bmp(bmp, rows, W, H)
    for flags in ([], ["--fused"]):
        t0 = time.perf_counter()
        subprocess.run([cli, "-i", bmp] + flags, check=True, stdout=subprocess.DEVNULL)
        dt = time.perf_counter() - t0
        print("CLI %s BMP file -> JPEG file (process start, file read, device init, encode, file write): %.0f ms, %.1f MP/s"
              % (" ".join(flags) or "(staged)", dt * 1e3, W * H / dt / 1e6), flush=True)
