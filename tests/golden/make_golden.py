#!/usr/bin/env python3
"""Regenerates tests/golden/oracle_vectors.json from the CPU oracle.

The reference (Rust) cannot be built here, so these vectors are outputs of the
oracle's restatement of it, frozen so that a later change of the oracle, of the
synthetic image generator or of the compiler shows up as a diff.  The vectors the
reference's own tests hold are in reference_vectors.json.

    python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import refcpu  # noqa: E402

SIZES = [(16, 16), (48, 32), (50, 35), (8, 8), (1, 1), (130, 67)]
RATIOS = [(4, 4, 4), (4, 2, 2), (4, 2, 0)]
ALGOS = [("real-dct", refcpu.REAL), ("bin-dct", refcpu.BIN)]


def image(w, h, seed):
    """Deterministic test image independent of the library under test (numpy PCG64, fixed seed)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    stride = (3 * w + 3) // 4 * 4
    yy, xx = np.mgrid[0:h, 0:w]
    rows = np.zeros((h, stride), np.uint8)
    for c in range(3):
        base = 128 + 96 * np.sin(2 * np.pi * (xx * (3 + c) + yy * (5 - c)) / 64.0) + rng.integers(-12, 13, (h, w))
        rows[:, c:3 * w:3] = np.clip(base, 0, 255).astype(np.uint8)
    return rows, stride


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def skewed(w, h, ratio):
    hf, vf = refcpu.downsampling_factor(*ratio)
    return (hf == 2 and w % 16 == 1) or (vf == 2 and h % 16 == 1)


def main():
    refcpu.build()
    out = {"generator": "tests/golden/make_golden.py", "images": [], "cases": []}
    for k, (w, h) in enumerate(SIZES):
        rows, stride = image(w, h, 1000 + k)
        out["images"].append({"width": w, "height": h, "stride": stride, "seed": 1000 + k, "sha256": sha(rows)})
        for ratio in RATIOS:
            if skewed(w, h, ratio):
                continue
            for name, algo in ALGOS:
                im = refcpu.RefImage.from_bgr(rows, w, h, stride, ratio, algo).encode(cos_mode=refcpu.COS_CALL)
                bits, nbits = im.bits()
                case = {"width": w, "height": h, "ratio": list(ratio), "dct": name, "nbits": int(nbits),
                        "planes_sha256": [sha(im.plane2d(c)) for c in range(3)],
                        "coeffs_sha256": [sha(im.coeffs(c)) for c in range(3)],
                        "bits_sha256": sha(bits), "scan_sha256": sha(im.stuffed_scan()),
                        "file_sha256": hashlib.sha256(im.file_bytes()).hexdigest(),
                        "file_len": len(im.file_bytes())}
                if w * h <= 64:
                    case["y_coeffs"] = im.coeffs(0).reshape(-1).tolist()
                out["cases"].append(case)
    json.dump(out, open(os.path.join(HERE, "oracle_vectors.json"), "w"), indent=1)
    print("wrote %d cases" % len(out["cases"]))


if __name__ == "__main__":
    main()
