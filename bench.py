#!/usr/bin/env python3
"""Benchmark of the MI355X JPEG encode path.

    python bench.py --gpus N --steps K --warmup W [--config 1|2|3|4|5]

Default workload (--config 3): 7680x4320 24-bit BMP rows, 4:2:0, real-dct, the configuration the headline
metric is quoted on.  One step = every rank encodes `--batch` images (device-resident BMP rows -> stuffed JPEG
scan bytes in HBM) taken in turn from `--resident` distinct synthetic images in its HBM (far more bytes than
the 256 MiB Infinity Cache, so no input is served from cache): colour conversion + subsampling + DCT +
quantisation + Huffman coding (one kernel) and concatenation + byte stuffing (a second kernel), one launch of
each per `--per-launch` images, and every image's result is taken before the step's clock stops.  Images are
independent, so ranks share nothing (weak scaling, no data-path collective).

--config 1: 512x512 4:2:0 real-dct (the reference's CPU-runnable point; the line also carries the oracle's
per-stage CPU times and the single-image host-to-host latency), --config 2: 3840x2160 4:4:4 bin-dct,
--config 4: 1920x1080 4:2:2 batches: the same loop on those images.
--config 5 (--mode stripes): one 32768x32768 4:2:0 image cut into MCU-row stripes, one per rank (strong
scaling): a step codes the image `--passes` times; per pass the ranks exchange three DC predictors, a bit count
and a few bits through shared host memory, shift and stuff their piece and copy it device to device into
rank 0's buffer.

With --gpus N and no WORLD_SIZE in the environment the script starts N processes of itself, one per GPU
(before anything touches a GPU); under torchrun it is one of the ranks.  Rank 0 prints ONE JSON line.
The ranks share no data: the only collectives are the barriers around the timed region and the MAX of the
elapsed times, which travel over a gloo (host) group by default (--control nccl puts them on RCCL).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak

# the named configurations (index = position in the list of configurations the metric comes with, 1-based)
CONFIGS = {
    1: dict(width=512, height=512, ratio="4:2:0", dct_algorithm="real-dct", mode="batch", resident=512, batch=8192,
            per_launch=256),
    2: dict(width=3840, height=2160, ratio="4:4:4", dct_algorithm="bin-dct", mode="batch", resident=64, batch=640,
            per_launch=64),
    3: dict(width=7680, height=4320, ratio="4:2:0", dct_algorithm="real-dct", mode="batch", resident=64, batch=640,
            per_launch=64),
    4: dict(width=1920, height=1080, ratio="4:2:2", dct_algorithm="real-dct", mode="batch", resident=512, batch=4096,
            per_launch=256),
    5: dict(width=32768, height=32768, ratio="4:2:0", dct_algorithm="real-dct", mode="stripes", passes=16),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=3, choices=sorted(CONFIGS),
                    help="named configuration: 1 = 512x512 4:2:0 real-dct (the reference's CPU-runnable point), "
                         "2 = 3840x2160 4:4:4 bin-dct, 3 = 7680x4320 4:2:0 real-dct (headline), "
                         "4 = batches of 1920x1080 4:2:2, 5 = one 32768x32768 4:2:0 image in MCU-row stripes")
    ap.add_argument("--mode", choices=["batch", "stripes"], default=None)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--ratio", default=None)
    ap.add_argument("--dct-algorithm", default=None)
    ap.add_argument("--batch", type=int, default=None, help="images per rank per step")
    ap.add_argument("--resident", type=int, default=None,
                    help="distinct images resident in each rank's HBM, which the images of a step are taken from in turn")
    ap.add_argument("--distinct", type=int, default=4,
                    help="images synthesised from scratch; the other resident ones are shifted copies of them")
    ap.add_argument("--passes", type=int, default=None, help="stripes mode: encodes of the whole image per step")
    ap.add_argument("--isolated", action="store_true",
                    help="also time back-to-back launches of the front kernel alone (single images)")
    ap.add_argument("--contexts", type=int, default=1,
                    help="encoder contexts (each with its own stream and buffers) the launch sets alternate over")
    ap.add_argument("--depth", type=int, default=2, choices=[1, 2, 3], help="launch sets in flight per context")
    ap.add_argument("--per-launch", type=int, default=None,
                    help="images per launch set (jpg_encode_batch_device_async); 1 = one launch set per image.  With "
                         "several sets per step the stuffing kernel of one set runs under the coding kernel of the next")
    ap.add_argument("--time-every", type=int, default=1,
                    help="take the kernel's own start/end time stamps of every n-th coding-kernel launch of the timed "
                         "region (1 = all of them)")
    ap.add_argument("--noise", action="store_true", help="uniform-noise images (entropy stress)")
    ap.add_argument("--control", choices=["gloo", "nccl"], default="gloo",
                    help="what carries the barriers and the MAX of the elapsed times when there are several ranks "
                         "(no image data ever crosses ranks): a gloo group on the host (default) or RCCL")
    ap.add_argument("--stripe-protocol", choices=["chain", "host"], default="chain",
                    help="stripes mode: 'chain' = the ranks' kernels hand each other the DC predictors, bit counts and "
                         "stuffed lengths through a mailbox in host memory and copy their pieces to their byte offsets "
                         "(a pass is queued at once, one wait); 'host' = the host exchanges those values between the "
                         "steps (four waits per pass)")
    ap.add_argument("--gather", choices=["auto", "device", "host"], default="auto",
                    help="stripes mode, chain protocol: where the pieces meet: rank 0's device memory opened by the "
                         "other ranks through an inter-process handle, or shared host memory (the fallback of 'auto')")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--verify", action="store_true",
                    help="stripes mode, small sizes: rank 0 also encodes the whole image in one piece and compares the "
                         "stitched scan with it byte for byte (after the timed region)")
    ap.add_argument("--kernel-iters", type=int, default=40, help="front-kernel launches for --isolated")
    args = ap.parse_args()
    for k, v in CONFIGS[args.config].items():
        if getattr(args, k, None) is None:
            setattr(args, k, v)
    if args.mode == "stripes":
        args.passes = args.passes or 16
    else:
        args.resident = args.resident or min(64, args.batch or 64)
        args.batch = args.batch or args.resident
        args.per_launch = args.per_launch if args.per_launch is not None else 32
    return args


def spawn_ranks(n):
    """--gpus N outside torchrun: start N copies of this script, one rank per GPU, and wait for them.  This
    process never touches a GPU."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    live = list(procs)
    while live:
        for p in list(live):
            r = p.poll()
            if r is None:
                continue
            live.remove(p)
            if r != 0 and rc == 0:
                rc = r if r > 0 else 128 - r
                for q in live:                      # a rank failed: the others would wait for it for ever
                    q.terminate()
        time.sleep(0.05)
    return rc


def cpu_baseline(args):
    """The CPU oracle (restatement of the reference, cosine evaluated per term as the reference does,
    3 DCT threads like src/jpeg/dct_quant.rs:29-60) on a bounded sample of the same workload."""
    from oracle import refcpu
    import jpeg_encoder_amd as J
    refcpu.build()
    ratio = tuple(int(x) for x in args.ratio.split(":"))
    algo = refcpu.BIN if args.dct_algorithm == "bin-dct" else refcpu.REAL
    # about 33 megapixels of the workload (roughly 10 s with the per-term cosine): whole images of it, or a band
    # of full-width rows of an image larger than that
    w, h, n = args.width, args.height, 1
    if w * h > 40e6:
        h = max(16, int(33.2e6 / w) // 16 * 16)
        what = "a band of %d full-width rows of the %dx%d image" % (h, args.width, args.height)
    else:
        n = max(1, int(33.2e6 / (w * h)))
        what = "%d synthetic image%s of %dx%d" % (n, "s" if n > 1 else "", w, h)
    rows = [J.host.synth_bgr(w, h, seed=1000 + k, noise_only=args.noise) for k in range(n)]
    stage = {"from_bmp_colour_conversion_1_thread": 0.0, "chrominance_downsampling_2_threads": 0.0,
             "dct_and_quantization_3_threads": 0.0, "generate_entropy_encoded_data_1_thread": 0.0,
             "write_image_data_stuffing_1_thread": 0.0}
    keys = list(stage)
    t0 = time.perf_counter()
    for r in rows:
        # the thread counts are the reference's: src/jpeg/sampling.rs:83-98 (2), src/jpeg/dct_quant.rs:29-60 (3)
        ts = [time.perf_counter()]
        im = refcpu.RefImage.from_bgr(r, w, h, r.shape[1], ratio, algo)
        ts.append(time.perf_counter())
        im.chrominance_downsampling(threads=2)
        ts.append(time.perf_counter())
        im.dct_and_quantization(cos_mode=refcpu.COS_CALL, threads=3)
        ts.append(time.perf_counter())
        im.generate_entropy_encoded_data()
        ts.append(time.perf_counter())
        im.stuffed_scan()
        ts.append(time.perf_counter())
        for k in range(5):
            stage[keys[k]] += ts[k + 1] - ts[k]
    dt = time.perf_counter() - t0
    return {"value": round(n * w * h / 1e6 / dt, 4), "unit": "MP/s", "cores": 3, "kind": "port",
            "sample": "%s, %s %s, pixels already in memory, no file I/O; %.1f s" % (what, args.ratio, args.dct_algorithm, dt),
            "stage_ms_per_image": {k: round(v / n * 1e3, 3) for k, v in stage.items()},
            "host_cores_available": os.cpu_count()}


def e2e_figures(J, args, ratio, device):
    """PCIe-inclusive figures (never the headline value): rows in host memory -> stuffed scan in host memory
    through jpg_encode_scan (H2D copy, both kernels, D2H copy), from pageable and from page-locked rows."""
    import ctypes as C
    import numpy as np
    W, H = args.width, args.height
    rows = J.host.synth_bgr(W, H, seed=(args.config << 32) | 77, noise_only=args.noise)
    L = J._native.lib()
    ptr = L.jpg_host_alloc(rows.nbytes)
    if not ptr:
        return None
    C.memmove(ptr, rows.ctypes.data, rows.nbytes)
    pinned = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), rows.shape)
    out = {"what": "one %dx%d image, rows in host memory -> stuffed scan in host memory (jpg_encode_scan_begin + _end "
                   "through the C ABI: rows copied and coded band by band, both kernels, D2H copy into a page-locked "
                   "buffer the library allocates meanwhile; the buffer is released outside the timed interval); "
                   "mean and minimum of 8 after one warm-up" % (W, H), "unit": "MP/s"}
    enc = J.Encoder(ratio, args.dct_algorithm, device=device)
    p, n = C.c_void_p(), C.c_size_t()
    for name, src in (("pageable", rows), ("page_locked", pinned)):
        ts = []
        for k in range(9):
            t0 = time.perf_counter()
            enc._ck(L.jpg_encode_scan_begin(enc._ctx, src.ctypes.data, W, H, src.shape[1], 1))
            enc._ck(L.jpg_encode_scan_end(enc._ctx, C.byref(p), C.byref(n)))
            dt = time.perf_counter() - t0
            L.jpg_free(p)
            if k:
                ts.append(dt)
        out[name + "_host_to_host"] = round(W * H / (sum(ts) / len(ts)) / 1e6, 1)
        out[name + "_ms"] = round(sum(ts) / len(ts) * 1e3, 3)
        out[name + "_ms_min"] = round(min(ts) * 1e3, 3)
    enc.close()
    del pinned
    L.jpg_free(ptr)
    # file to file with the command-line front end (the reference's own interface): a process of its own, so the
    # figure includes process start and device initialisation as well as reading the BMP and writing the JPEG
    cli = os.path.join(ROOT, "jpeg-encoder_amd", "jpeg-encoder")
    if os.path.exists(cli):
        import tempfile
        tmp = "/dev/shm" if os.path.isdir("/dev/shm") else None
        with tempfile.TemporaryDirectory(dir=tmp) as d:
            bmp = os.path.join(d, "in.bmp")
            J.host.write_bmp(bmp, rows, W, H)
            ts = []
            for k in range(3):
                t0 = time.perf_counter()
                r = subprocess.run([cli, "--image", bmp, "--output", os.path.join(d, "out.jpeg"), "--subsampling-ratio",
                                    args.ratio, "--dct-algorithm", args.dct_algorithm.replace("-strict", ""), "--fused"],
                                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                if r.returncode != 0:
                    ts = []
                    break
                ts.append(time.perf_counter() - t0)
            if ts:
                out["cli_file_to_file_ms"] = round(min(ts) * 1e3, 1)
                out["cli_file_to_file"] = round(W * H / min(ts) / 1e6, 1)
                out["cli_note"] = ("jpeg-encoder --image in.bmp (files in shared memory): process start, device "
                                   "initialisation, read, encode, write; best of 3")
    return out


def pmc_record(workload, kernel):
    """The committed rocprofv3 --pmc summary (profiles/*_pmc*.json) of this workload and kernel, if any."""
    import glob
    best = None
    for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc*.json"))):
        try:
            d = json.load(open(p))
            if d.get("workload") == workload and d.get("kernel", "k_front") == kernel and d.get("hbm_bytes_per_launch"):
                best = dict(d, _file=os.path.relpath(p, ROOT))
        except Exception:
            pass
    return best


def stats(samples):
    s = sorted(samples)
    n = len(s)
    if not n:
        return None
    return {"mean": sum(s) / n, "min": s[0], "median": s[n // 2] if n % 2 else 0.5 * (s[n // 2 - 1] + s[n // 2]),
            "max": s[-1], "n": n}


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # Standard output carries one JSON line and nothing else: libraries that write there (gloo announces its
    # connections on it) are sent to standard error until that line is printed.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    # rehearsal of the multi-process path on a box with one GPU: every rank on device 0, gloo for the barrier
    rehearsal = os.environ.get("BENCH_REHEARSAL_ONE_GPU") == "1"

    import numpy as np
    import torch
    import torch.distributed as dist
    import jpeg_encoder_amd as J

    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    if rehearsal:
        local = 0
    elif local >= ndev:
        raise SystemExit("rank %d needs GPU %d but the node shows %d (BENCH_REHEARSAL_ONE_GPU=1 puts every rank on "
                         "GPU 0 to rehearse the multi-process path)" % (rank, local, ndev))
    torch.cuda.set_device(local)
    control = "gloo" if rehearsal else args.control
    if world > 1:
        # No image data crosses ranks: the group only carries the barriers and the MAX of the elapsed times, and a
        # host group does that without depending on GPU-to-GPU links.  (--control nccl: the same over RCCL.)
        if control == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    ratio = tuple(int(x) for x in args.ratio.split(":"))
    W, H = args.width, args.height
    stride = J.host.bmp_stride(W)
    _, vs = J.get_downsampling_factor(ratio)
    stripes = args.mode == "stripes"
    encs = [J.Encoder(ratio, args.dct_algorithm, device=local) for _ in range(1 if stripes else max(1, args.contexts))]
    enc = encs[0]
    last = [None]

    if stripes:
        # ---- one image, one MCU-row stripe per rank, resident on the rank's GPU
        S = J.stripes
        plan = S.plan_stripes(H, vs, world)

        def synth_stripe(k):
            """File rows of stripe k: a synthetic tile of its own seed repeated over the stripe."""
            hk = plan[k][1] - plan[k][0]
            bw, bh = min(W, 8192), min(max(hk, 16), 2048)
            base = J.host.synth_bgr(bw, bh, seed=(5 << 32) | k, noise_only=args.noise)[:, :3 * bw]
            out = np.zeros((max(hk, 1), stride), np.uint8)
            line = np.tile(base, (1, (W + bw - 1) // bw))[:, :3 * W]
            for r0 in range(0, hk, bh):
                n = min(bh, hk - r0)
                out[r0:r0 + n, :3 * W] = line[:n]
            return out

        y0, y1 = plan[rank]
        hs = y1 - y0
        rows = synth_stripe(rank)
        dev_rows = torch.from_numpy(rows).cuda()
        del rows
        torch.cuda.synchronize()
        if world > 1:
            ex = S.ShmExchange.from_group()
        else:
            class _Solo:
                rank, world = 0, 1

                def all_gather(self, values):
                    return [tuple(int(v) for v in values)]

                def barrier(self):
                    pass

                def close(self):
                    pass
            ex = _Solo()
        ptr = dev_rows.data_ptr() if hs > 0 else None
        chain = gather = None
        protocol = args.stripe_protocol
        if protocol == "chain":
            # room for the whole stuffed scan: half a byte per pixel (photo-like content needs a tenth of that,
            # uniform noise a bit over a quarter byte)
            try:
                chain = S.StripeChain(enc, ex, W * H // 2 + (1 << 20), gather=args.gather)
            except RuntimeError as e:
                # rank 0's mailbox cannot be opened on every rank (the ranks decide this together, so all of them
                # get here): the earlier protocol with host exchanges between the steps
                if rank == 0:
                    print("stripe chain not available (%s): host exchanges instead" % e, file=sys.stderr, flush=True)
                protocol = "host (the stripe chain could not be set up)"
        if chain:

            queued = [0]

            def step():
                # up to two passes are queued before the first is taken: the host queues a pass while the one before
                # it runs (also across steps; drain() takes what is left before the clock is read)
                for _ in range(args.passes):
                    if queued[0] == 2:
                        last[0] = chain.finish()
                        queued[0] -= 1
                    if chain.encode_async(W, H, stride, vs, stripe_ptr=ptr):
                        queued[0] += 1
        else:
            queued = [0]
            gather = S.DeviceGather(J._native, local)

            def step():
                for _ in range(args.passes):
                    last[0] = S.encode_striped(enc, None, W, H, stride, vs, True, exchange=ex, gather=gather, fetch=False,
                                               stripe_ptr=ptr)

        def drain():
            while chain and queued[0]:
                last[0] = chain.finish()
                queued[0] -= 1
            return last[0]
        launches_per_step = args.passes
        images_per_step = args.passes
    else:
        # ---- independent images: resident inputs, launch sets of per_launch images
        imgs, base = [], []
        for i in range(args.resident):
            if i < max(1, args.distinct):
                rows = J.host.synth_bgr(W, H, seed=(args.config << 32) | (rank * args.resident + i), noise_only=args.noise)
                base.append(rows)
            else:
                # the same picture shifted by a multiple of 16 pixels: different bytes at every address, same statistics
                b = base[i % len(base)]
                rows = np.ascontiguousarray(np.roll(b, 48 * (i // len(base)), axis=1))
            imgs.append(torch.from_numpy(rows).cuda())
        del base
        torch.cuda.synchronize()
        per_launch = max(1, min(args.per_launch if args.per_launch > 0 else args.batch, enc.max_batch(), args.batch))
        import ctypes as C
        groups = []
        for i in range(0, args.batch, per_launch):
            ptrs = [imgs[j % len(imgs)].data_ptr() for j in range(i, min(i + per_launch, args.batch))]
            groups.append((C.c_void_p * len(ptrs))(*ptrs))
        inflight = [0] * len(encs)

        def step():
            # The images of a group go through one launch of each kernel; a context keeps up to `depth` groups in
            # flight, so the stuffing of one group overlaps the coding kernel of the next -- also across steps: a
            # step queues its groups and takes the results (sizes read back) of the groups queued before them as it
            # needs their buffers; drain() takes whatever is still in flight before the clock is read.
            for i, g in enumerate(groups):
                k = i % len(encs)
                if inflight[k] == args.depth:
                    last[0] = encs[k].finish_next()
                    inflight[k] -= 1
                if len(g) == 1:
                    encs[k].encode_device_async(g[0], W, H, stride, True)
                else:
                    encs[k].encode_batch_device_async(g, W, H, stride, True)
                inflight[k] += 1

        def drain():
            for k, e in enumerate(encs):
                while inflight[k]:
                    last[0] = e.finish_next()
                    inflight[k] -= 1
            return last[0]
        launches_per_step = len(groups)
        images_per_step = args.batch

    for _ in range(args.warmup):
        step()
    drain()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        for e in encs:
            e.sync()

    for e in encs:
        e.set_timing(max(1, args.time_every))
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    result = drain()                           # every result of the timed steps has been taken
    barrier()
    dt = time.perf_counter() - t0
    samples, region_ms, region_n, region_images = [], 0.0, 0, 0
    for e in encs:
        samples += e.front_samples()
        ms, n, im = e.front_time()
        region_ms += ms
        region_n += n
        region_images += im
        e.set_timing(0)
    if world > 1:
        tt = torch.tensor([dt], device="cpu" if control == "gloo" else "cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms_per_step = dt / args.steps * 1e3

    workload = "%dx%d %s %s" % (W, H, args.ratio, args.dct_algorithm)

    # ---- roofline of the dominant kernel: the fused coding kernel (colour + subsample + DCT + quant +
    # Huffman coding) where the image size allows it, else the front kernel (the same without coding)
    roof = None
    front_roof = None
    valu_roof = None
    if rank == 0:
        fused = enc.fused_coding()
        kernel = "k_coded" if fused else "k_front"
        if stripes:
            y0, y1 = J.stripes.plan_stripes(H, vs, world)[0]
            # rank 0's stripe: rows read + (at most) its share of the coded bits written
            alg_image = stride * (y1 - y0) + (int(result) // max(world, 1) if fused else 0)
            if not fused:
                alg_image = enc.front_bytes()
        else:
            nbits, scan_len = result
            # k_coded reads the rows and writes only the coded bits; k_front writes 2 bytes per coefficient instead
            alg_image = stride * H + (int(nbits) + 7) // 8 if fused else enc.front_bytes()
        st = stats(samples) or {"mean": region_ms / max(region_n, 1), "min": None, "median": None, "n": region_n}
        rms = region_ms / max(region_n, 1)     # the figure of record: mean duration of the launches inside the timed region
        ipl = region_images / max(region_n, 1)
        alg = int(alg_image * ipl)
        achieved = alg / (rms * 1e-3) / 1e9
        rec = pmc_record(workload, kernel)
        traffic = rec["hbm_bytes_per_launch"] if rec else None
        launches_rank0 = launches_per_step / max(1, len(encs)) if not stripes else launches_per_step
        roof = {"bound": "hbm", "kernel": kernel, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": int(traffic * ipl) if traffic else None,
                # traffic is NOT a measurement of this run: the counters come from a separate rocprofv3 --pmc run
                # of the same kernel on the same workload (hardware counters cannot be read from inside a run)
                "traffic_source": ({"file": rec["_file"], "images_per_launch_in_the_counter_run":
                                    rec.get("images_per_launch_in_the_counter_run"),
                                    "scaled_to_images_per_launch": ipl, "how": rec.get("command")} if rec else None),
                "traffic_ratio": round(traffic / alg_image, 4) if traffic else None,
                "algorithmic_bytes": alg, "kernel_ms": round(rms, 5),
                "kernel_ms_min": round(st["min"], 5) if st["min"] is not None else None,
                "kernel_ms_median": round(st["median"], 5) if st["median"] is not None else None,
                "launches_timed": region_n, "images_per_launch": ipl, "launches_per_step": launches_per_step,
                # the launches of a step run one after another on one stream, so their summed durations fit in the step
                "kernel_ms_x_launches_per_step": round(rms * launches_rank0, 4),
                "fits_in_step": bool(rms * launches_rank0 <= ms_per_step * 1.0005)}
        if args.isolated and not fused and not stripes:
            s = torch.cuda.ExternalStream(enc.stream, device=torch.device("cuda", local))
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            kms = None
            for k in range(4):
                enc.front_device(imgs[k % len(imgs)].data_ptr(), W, H, stride, True)
            enc.sync()
            for rep in range(3):          # average launch duration of the steadiest of three back-to-back series
                ev0.record(s)
                for k in range(args.kernel_iters):
                    enc.front_device(imgs[k % len(imgs)].data_ptr(), W, H, stride, True)
                ev1.record(s)
                enc.sync()
                t = ev0.elapsed_time(ev1) / args.kernel_iters
                kms = t if kms is None else min(kms, t)
            roof["isolated_kernel_ms_per_image"] = round(kms, 5)
        if fused and rec and rec.get("valu_issue_cycles_per_launch"):
            # second roof: vector-ALU issue.  achieved = issue cycles the kernel's instructions need (counted per
            # class by the PMC pass, priced with the per-class issue costs measured by tools/micro/valu_rate.hip)
            # per second; peak = SIMDs x clock.
            clock = rec.get("clock_ghz", 2.4) * 1e9
            simds = rec.get("simds", 1024)
            need = rec["valu_issue_cycles_per_launch"] * ipl
            ach = need / (rms * 1e-3)
            valu_roof = {"bound": "valu-issue", "kernel": kernel, "achieved": round(ach / 1e12, 4),
                         "peak": round(simds * clock / 1e12, 4), "unit": "T issue-cycles/s",
                         "frac": round(ach / (simds * clock), 4), "source": rec.get("source"),
                         "valu_instructions_per_launch": int(rec.get("valu_instructions_per_launch", 0) * ipl)}
            if rec.get("scalar_issue_cycles_per_launch"):
                # scalar ALU instructions also take issue time of the SIMD (4 cycles beside a 2-cycle vector
                # instruction of another wave, 2 beside a 4-cycle one; priced at 3 in the record)
                both = (rec["valu_issue_cycles_per_launch"] + rec["scalar_issue_cycles_per_launch"]) * ipl / (rms * 1e-3)
                valu_roof["frac_with_scalar_instructions"] = round(both / (simds * clock), 4)
                valu_roof["scalar_alu_instructions_per_launch"] = int(rec.get("scalar_alu_instructions_per_launch", 0) * ipl)
        if fused and not stripes:
            # for comparison, outside the timed region: the front kernel alone (coefficients to memory,
            # as the staged calls use it), single-image launches
            for k in range(4):
                enc.front_device(imgs[k % len(imgs)].data_ptr(), W, H, stride, True)
            enc.set_timing(1)
            for k in range(16):
                enc.front_device(imgs[k % len(imgs)].data_ptr(), W, H, stride, True)
            fs = stats(enc.front_samples())
            ms, n, im = enc.front_time()
            enc.set_timing(0)
            fb = enc.front_bytes()
            ach = fb * im / (ms * 1e-3) / 1e9
            front_roof = {"bound": "hbm", "kernel": "k_front", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                          "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "algorithmic_bytes": int(fb),
                          "kernel_ms": round(ms / max(n, 1), 5), "kernel_ms_min": round(fs["min"], 5), "launches_timed": n,
                          "note": "not on the benchmarked path: the kernel k_coded replaces, timed after the region"}

    if world > 1:
        dist.barrier()
    if rank == 0:
        mp = W * H / 1e6
        if stripes:
            total_mp = mp * args.passes * args.steps
            cfg = {"workload": "%dx%d BMP, %s, %s, MCU-row stripes" % (W, H, args.ratio, args.dct_algorithm),
                   "named_config": args.config, "encodes_of_the_image_per_step": args.passes,
                   "stripes": world, "stripe_rows_rank0": J.stripes.plan_stripes(H, vs, world)[0][1],
                   "stripe_protocol": protocol, "gather": chain.gather if chain else gather.kind,
                   "passes_in_flight": 2 if chain else 1, "control_plane": control if world > 1 else None,
                   "parallelism": ("MCU-row stripes x%d, scalars handed on by the kernels through a mailbox in host memory, "
                                   "pieces copied to their offsets by a kernel" if chain else
                                   "MCU-row stripes x%d, scalars through shared host memory, pieces by device-to-device copy") % world,
                   "scan_bytes": int(result)}
            metric = "encode megapixels/sec (%dx%d %s %s in MCU-row stripes, device-resident BMP rows -> stitched stuffed scan on rank 0's GPU)" % (
                W, H, args.ratio, args.dct_algorithm)
        else:
            total_mp = mp * args.batch * world * args.steps
            cfg = {"workload": "%dx%d BMP, %s, %s" % (W, H, args.ratio, args.dct_algorithm), "named_config": args.config,
                   "images_per_rank_per_step": args.batch, "resident_images_per_rank": args.resident,
                   "images_per_launch": per_launch, "contexts_per_rank": len(encs),
                   "launch_sets_in_flight_per_context": args.depth, "global_batch": args.batch * world,
                   "parallelism": "image-sharded x%d" % world, "control_plane": control if world > 1 else None,
                   "scan_bytes_last_image": int(scan_len),
                   "bits_per_pixel": round(nbits / (W * H), 4)}
            metric = "encode megapixels/sec (%s, device-resident BMP rows -> stuffed JPEG scan)" % (
                "8K 4:2:0 real-dct" if (W, H, args.ratio, args.dct_algorithm) == (7680, 4320, "4:2:0", "real-dct") else workload)
        out = {
            "metric": metric,
            "value": round(total_mp / dt, 1),
            "unit": "MP/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "timed_region_s": round(dt, 4),
            "higher_is_better": True,
            "scaling": "strong" if stripes else "weak",
            "vs_baseline": None,
            # bin-dct: the reference's i32 lifting steps are evaluated in f32 under a directed rounding mode, on integer
            # values far below 2^22, so every operation is exact (tools/check_bin_f32.c)
            "dtype": "f32",
            "dtype_note": None if args.dct_algorithm.startswith("real") else "integer-valued f32: exact evaluation of the reference's i32 lifting transform",
            "data": "synthetic" + (" (uniform noise)" if args.noise else ""),
            "config": cfg,
            "roofline": roof,
        }
        if rehearsal and world > 1:
            out["rehearsal"] = "all %d ranks on GPU 0 (BENCH_REHEARSAL_ONE_GPU=1): not a multi-GPU figure" % world
        if valu_roof:
            out["roofline_valu"] = valu_roof
        if front_roof:
            out["roofline_front_kernel"] = front_roof
        if world == 1 and not args.no_e2e and not stripes:
            e2e = e2e_figures(J, args, ratio, local)
            if e2e:
                out["e2e"] = e2e
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args)
        sys.stdout.flush()
        os.dup2(json_fd, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if stripes:
        if args.verify and rank == 0:
            whole = np.concatenate([synth_stripe(k)[:plan[k][1] - plan[k][0]] for k in reversed(range(world))])   # bottom-up file
            ref = J.Encoder(ratio, args.dct_algorithm, device=local)
            want = ref.encode_scan(whole, W, H, stride)
            got = (chain or gather).read(int(result))
            ref.close()
            if not np.array_equal(got, want):
                raise SystemExit("stitched scan differs from the whole-image encode")
            print("verified: %d stitched bytes equal the whole-image scan" % got.size, file=sys.stderr, flush=True)
        if chain:
            chain.close()
        else:
            gather.close(ex)
        ex.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
