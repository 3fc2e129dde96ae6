#!/usr/bin/env python3
"""bench.py -- Mpixels/s decoded on the BASELINE.json workload, one process per GPU.

Workload (BASELINE.json configs[1]): a batch of 1024 synthetic 3840x2160 4:2:0 baseline Q75 JPEGs with DRI = 4 MCUs per
GPU (SURVEY.md 8d recipe, distinct seed per image, > 1 GB of distinct compressed input).  A "step" is one pass of the
hot path over that batch: marker index -> Huffman MCU decode -> dequantise + float32 IDCT + level shift -> interleaved
YCbCr8 output (the reference benchmark's sink, tests/JpegLibrary.Benchmarks/DecoderBenchmark.cs:51-73), with the
compressed bytes already resident in HBM and the pixels left resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--images M] [--workload 4k_dri4|4k_dri0|1080p_q90|512_444|...]

For N > 1 the driver launches it under torch.distributed.run (one rank per GPU); images shard one-per-GPU
(embarrassingly parallel: no data-path collective), scaling is weak (per-GPU batch fixed).  Started WITHOUT torchrun with
--gpus N > 1 it launches those ranks itself (a child `python -m torch.distributed.run ... bench.py`, started before this
process touches a GPU) and relays their line; --multi-inprocess adds the same shard through the library's own in-process
driver (jpgpu_multi_*: one process, a context per device).
Prints ONE JSON line on rank 0.  On the default workload that line also carries `configs`: short passes of the other
BASELINE.json configurations (512_444, 4k_dri0, 1080p_q90, 4k_progressive, het_8192) run AFTER the headline's timed region,
each with its own stage times, K3 roofline fraction and oracle spot check; they never change `value`.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (width, height, subsampling, quality, dri, default images per GPU)
    "4k_dri4": (3840, 2160, "420", 75, 4, 1024),
    "4k_dri0": (3840, 2160, "420", 75, 0, 1024),
    "1080p_q90": (1920, 1080, "420", 90, 4, 1024),
    "512_444": (512, 512, "444", 75, 0, 1),
    # BASELINE.json configs[4]: progressive (SOF2) 4K 4:2:0, libjpeg's default 10-scan script, DRI = 0 (made with Pillow)
    "4k_progressive": (3840, 2160, "420p", 75, 0, 256),
    # the reference's OWN benchmark input (tests/JpegLibrary.Benchmarks/DecoderBenchmark.cs:19-43): an 8192 x 8192 canvas with
    # HETissueSlide.jpg (2048 x 2048, tests/golden) drawn at (0,0), (0,H), (W,0), (W,H), saved as a baseline 4:2:0 Q75 JPEG
    # without restart markers (ImageSharp's defaults; made with Pillow here); timed there: Identify + Decode + RGBA conversion
    "het_8192": (8192, 8192, "420het", 75, 0, 16),
    # config 5 on REAL content (round 5): 3840 x 2160 crops of that canvas's tissue quarter at distinct offsets, re-encoded by Pillow as
    # progressive 4:2:0 Q75 -- photographs have the high-frequency coefficients the sinusoid + noise recipe lacks (its scan 5 is empty)
    "het_progressive": (3840, 2160, "420hetp", 75, 0, 256),
}
DEFAULT_FORMAT = {"het_8192": "rgba_u8"}
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def progressive_image(width, height, quality, seed):
    """SURVEY 8d recipe (sinusoid mix + N(0, 8) noise), encoded as a progressive 4:2:0 JPEG by Pillow / libjpeg-turbo."""
    import io

    from PIL import Image

    rng = np.random.default_rng(seed)
    ph = rng.uniform(0, 2 * np.pi, 4)
    y, x = np.mgrid[0:height, 0:width].astype(np.float32)
    img = np.stack([128 + 70 * np.sin(x / 37 + ph[0]) * np.cos(y / 53 + ph[1]), 128 + 60 * np.cos(x / 91 + y / 29 + ph[2]),
                    128 + 90 * np.sin((x + y) / 67 + ph[3])], axis=-1)
    img += rng.normal(0, 8, img.shape).astype(np.float32)
    out = io.BytesIO()
    Image.fromarray(np.clip(np.rint(img), 0, 255).astype(np.uint8)).save(out, format="JPEG", quality=quality, progressive=True,
                                                                          subsampling="4:2:0")
    return out.getvalue()


def het_canvas(quality):
    """DecoderBenchmark.cs:19-43 with Pillow: `new Image<Rgba32>(4 W, 4 H)` is transparent black, the base image is drawn four
    times into its top-left quarter, SaveAsJpeg drops the alpha: baseline, 4:2:0, quality 75, DRI = 0."""
    import io

    from PIL import Image

    base = Image.open(os.path.join(ROOT, "tests", "golden", "HETissueSlide.jpg")).convert("RGB")
    w, h = base.size
    canvas = Image.new("RGB", (4 * w, 4 * h))
    for pos in ((0, 0), (0, h), (w, 0), (w, h)):
        canvas.paste(base, pos)
    out = io.BytesIO()
    canvas.save(out, format="JPEG", quality=quality, subsampling="4:2:0")
    return out.getvalue()


def het_progressive_batch(n, width, height, quality, first, nthreads):
    """`n` progressive frames: crops of the HETissueSlide canvas's content quarter (4096 x 4096: the slide drawn 2 x 2), crop i at an
    offset of its own."""
    import io

    from PIL import Image

    base = Image.open(os.path.join(ROOT, "tests", "golden", "HETissueSlide.jpg")).convert("RGB")
    w, h = base.size
    quarter = Image.new("RGB", (2 * w, 2 * h))
    for pos in ((0, 0), (0, h), (w, 0), (w, h)):
        quarter.paste(base, pos)
    px = np.asarray(quarter)

    def one(i):
        k = first + i
        x0, y0 = (k * 16) % (2 * w - width + 1), (k * 61) % (2 * h - height + 1)
        out = io.BytesIO()
        Image.fromarray(px[y0:y0 + height, x0:x0 + width]).save(out, format="JPEG", quality=quality, progressive=True, subsampling="4:2:0")
        return out.getvalue()

    with ThreadPoolExecutor(max(1, min(nthreads, 64))) as ex:
        return list(ex.map(one, range(n)))


def progressive_batch(n, width, height, quality, seed0, nthreads):
    # threads, not processes: the GPU is already initialised in this process (numpy and Pillow release the GIL)
    with ThreadPoolExecutor(max(1, min(nthreads, 64))) as ex:
        return list(ex.map(progressive_image, [width] * n, [height] * n, [quality] * n, [seed0 + i for i in range(n)]))


def host_cpu_budget():
    """Hardware threads this process may actually use: the scheduler affinity mask and the cgroup CPU quota both bound it
    (os.cpu_count() reports the machine's threads even inside a container that is granted a few of them)."""
    info = {"cpu_count": os.cpu_count() or 1}
    try:
        info["affinity"] = len(os.sched_getaffinity(0))
    except Exception:
        pass
    quota = None
    try:  # cgroup v2
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        pass
    if quota is None:
        try:  # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    if quota is not None:
        info["cgroup_quota_cpus"] = round(quota, 2)
    return info


def cpu_baseline(files, width, height, max_threads, rgba=False, per_thread=8, thread_cap=0):
    """Reference-equivalent CPU path (oracle/jpegref.c, jref_decode_batch_mt): Identify + Decode into a
    JpegBufferOutputWriter8Bit-style YCbCr8 buffer, mirroring tests/JpegLibrary.Benchmarks/DecoderBenchmark.cs:51-73, one
    independent decoder per native host thread (pthreads; per-thread output buffer allocated and touched before the clock
    starts, one untimed warm decode per thread, 8 images per thread), on a bounded sample of the same images.
    The thread count is swept (8, 16, ... up to the machine's hardware threads) and the BEST rate is the baseline: a
    container is often granted fewer CPUs than os.cpu_count() reports, and oversubscribing those must not be what is timed."""
    from oracle import pyoracle as po

    budget = host_cpu_budget()
    hw = max(1, min(max_threads, budget["cpu_count"]))
    if thread_cap:
        hw = min(hw, thread_cap)
    n = len(files)
    # single thread: ~3-6 s of work
    n1 = max(1, min(n, int(4.0 * 100e6 / (width * height)) or 1))
    s1, px1 = po.decode_batch_mt(files[:n1], 3, 1, warm=True, rgba=rgba)
    single = px1 / 1e6 / s1
    counts = sorted({min(hw, c) for c in (8, 16, 32, 64, 128, 256, hw)})
    sweep, spent = [], 0.0
    best = None
    for threads in counts:
        sample = [files[i % n] for i in range(threads * per_thread)]
        sec, px = po.decode_batch_mt(sample, 3, threads, warm=True, rgba=rgba)
        rate = px / 1e6 / sec
        sweep.append({"threads": threads, "Mpixels/s": round(rate, 1), "wall_s": round(sec, 2), "decodes": len(sample)})
        spent += sec
        if best is None or rate > best[0]:
            best = (rate, threads, len(sample), sec)
        if spent > 40.0:  # bounded: the sweep stops early on a slow host
            break
    value, threads_best, decodes, sall = best
    # cores = the CPUs those threads could actually occupy: never more than the affinity mask / cgroup quota grants
    # (the sweep's best point is often an oversubscribed one by a percent of noise; efficiency is per CPU, not per thread)
    granted = granted_cpus(budget)
    cores = max(1, min(threads_best, granted))
    return {
        "value": round(value, 2),
        "unit": "Mpixels/s",
        "cores": cores,
        "threads": threads_best,
        "kind": "port",
        "sample": f"{decodes} decodes ({per_thread} per thread) of the benchmark's images, Identify+Decode into an interleaved YCbCr8 buffer"
                  f"{' + ConvertYCbCr8ToRgba32' if rgba else ''}, one "
                  f"decoder per native thread, buffers pre-touched, one warm decode per thread ({sall:.2f} s wall); best of the thread "
                  f"sweep; single thread: {single:.1f} Mpixels/s over {n1} images",
        "single_core_value": round(single, 2),
        "scaling_efficiency": round(value / (single * cores), 3),
        "thread_sweep": sweep,
        "host_cpu_budget": budget,
    }


def ingest_inclusive(jl, ctx, batch, files, fmt, rounds, pinned=False, sync_ranks=None):
    """Seconds per batch with the host in the loop: while batch A decodes (decode stream), batch B is parsed (headers only)
    and sent to HBM (upload stream), then the roles swap.  One round = one batch ingested AND one batch decoded.
    pinned=False: the compressed bytes start in pageable host memory and travel through the pinned staging ring (the host
    crew copies them); pinned=True: they lie in page-locked memory and are DMA'd from there (jpgpu_batch_upload_segments
    with JPGPU_UPLOAD_PINNED_ARENA: one read buffer, a few large DMAs, no host thread touches an entropy-coded byte)."""
    other = jl.Batch(ctx)
    up = (lambda b: b.upload_segments(files, fmt, pinned=True, arena=True)) if pinned else (lambda b: b.upload(files, fmt))
    up(other)  # allocates the second set of device buffers (not timed)
    up(batch)
    cur, nxt = batch, other
    if sync_ranks:
        sync_ranks()
    t0 = time.perf_counter()
    for _ in range(rounds):
        cur.decode()
        up(nxt)
        cur.sync()
        cur, nxt = nxt, cur
    elapsed = time.perf_counter() - t0
    stats = nxt.ingest_stats()
    for i in (0, len(files) - 1):
        if cur.result(i).status != 0 or nxt.result(i).status != 0:
            raise RuntimeError("ingest-inclusive loop: an image failed")
    other.close()
    return elapsed / rounds, stats


def make_inputs(jl_sharding, jpegsynth, workload, n_images, rank, gen_threads, distinct=0):
    """Synthetic input of one workload: `n_images` files (views into one buffer), distinct seed per image and per rank."""
    width, height, ss, quality, dri, _ = WORKLOADS[workload]
    if ss in ("420p", "420het", "420hetp"):
        if ss == "420het":
            files_b = [het_canvas(quality)]  # the reference benchmark decodes ONE input over and over: so does every slot of the batch
            distinct = 1
        elif ss == "420hetp":
            files_b = het_progressive_batch(min(n_images, distinct or n_images), width, height, quality, rank * 1009, gen_threads)
        else:
            files_b = progressive_batch(min(n_images, distinct or n_images), width, height, quality, jl_sharding.rank_seed_base(rank), gen_threads)
        ss = "420"
        files_b = [files_b[i % len(files_b)] for i in range(n_images)]
        sizes = np.array([len(f) for f in files_b], dtype=np.int64)
        stride = int(sizes.max())
        buf = np.zeros(stride * n_images + 64, np.uint8)
        for i, f in enumerate(files_b):
            buf[i * stride:i * stride + len(f)] = np.frombuffer(f, np.uint8)
    else:
        buf, sizes, stride = jpegsynth.encode_batch(n_images, width, height, ss, quality, dri, seed0=jl_sharding.rank_seed_base(rank), nthreads=gen_threads)
    files = [buf[i * stride:i * stride + int(sizes[i])] for i in range(n_images)]
    return files, sizes, ss, distinct


FORMATS = {"interleaved_u8": 0, "planar_u8": 1, "rgb_u8": 3, "rgba_u8": 4}


def spot_check(jl, batch, files, fmt, indices):
    """A few images of a decoded batch bit-exact against the oracle (the checker; after the timing)."""
    try:
        from oracle import pyoracle as po
    except ImportError:
        return "oracle unavailable"
    for i in indices:
        if fmt not in (jl.FMT_INTERLEAVED_U8, jl.FMT_RGB_U8, jl.FMT_RGBA_U8):
            continue
        ref, _ = po.decode_8bit(bytes(files[i]))
        if fmt != jl.FMT_INTERLEAVED_U8:
            ref = po.ycbcr8_to_rgb(ref, rgba=(fmt == jl.FMT_RGBA_U8))
        if not np.array_equal(batch.output(i), ref):
            raise RuntimeError(f"parity failure on image {i}")
    return "bit-exact vs oracle"


def measure_config(jl, sharding, jpegsynth, torch, dist, reduce_device, ctx, name, rank, world, gen_threads, steps, warmup, images=0):
    """One of the OTHER BASELINE.json configurations, briefly, behind the headline's timed region: the same step (K1 + K2-family
    + K3 over the batch, inputs resident in HBM, output left in HBM), the same barrier / MAX-reduce around it, stage times by
    HIP events, K3's roofline fraction from its algorithmic bytes, two images against the oracle."""
    width, height, ss, quality, dri, default_images = WORKLOADS[name]
    n_images = images or default_images
    fmt_name = DEFAULT_FORMAT.get(name, "interleaved_u8")
    fmt = {"interleaved_u8": jl.FMT_INTERLEAVED_U8, "planar_u8": jl.FMT_PLANAR_U8, "rgb_u8": jl.FMT_RGB_U8, "rgba_u8": jl.FMT_RGBA_U8}[fmt_name]
    t0 = time.perf_counter()
    files, sizes, ss_eff, _ = make_inputs(sharding, jpegsynth, name, n_images, rank, gen_threads)
    t_gen = time.perf_counter() - t0

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    batch = jl.Batch(ctx).upload(files, fmt)
    try:
        for _ in range(max(1, warmup)):
            batch.decode()
        batch.sync()
        for i in range(n_images):
            r = batch.result(i)
            if r.status != 0:
                raise RuntimeError(f"image {i} failed: status {r.status} detail {r.detail}")
        batch.stage_ms()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            batch.decode()
        batch.sync()
        barrier()
        elapsed = time.perf_counter() - t0
        stage = batch.stage_ms()
        if dist is not None:
            elapsed = sharding.max_over_ranks(dist, elapsed, device=reduce_device)
        totals = batch.totals()
        idct_bytes = totals["blocks"] * 128 + totals["output_bytes"]
        achieved = idct_bytes / (stage["idct"] / 1e3) / 1e9 if stage["idct"] > 0 else 0.0
        res = {
            "workload": f"{n_images} x {width}x{height} {ss_eff} Q{quality} DRI={dri}{' progressive (SOF2)' if ss in ('420p', '420hetp') else ''}"
                        f"{' (HETissueSlide canvas, DecoderBenchmark.cs)' if ss == '420het' else ''} per GPU, output {fmt_name} resident in HBM",
            "value": round(sharding.aggregate_throughput(n_images * width * height, world, steps, elapsed), 1),
            "unit": "Mpixels/s", "n_gpus": world, "steps": steps, "warmup": max(1, warmup),
            "ms_per_step": round(elapsed / steps * 1e3, 3),
            "stage_ms": {k: round(v, 4) for k, v in stage.items()},
            "roofline": {"kernel": "idct_output_kernel", "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "algorithmic_bytes": idct_bytes},
            "gen_s": round(t_gen, 1),
            "marker_fallbacks": batch.marker_fallbacks(),  # groups of K1 that gave up waiting for a predecessor (expected: 0)
        }
        if dri == 0 and ss not in ("420p", "420hetp"):
            res["subseq_rounds"] = batch.subseq_rounds()
            res["subseq_fallbacks"] = batch.subseq_fallbacks()
        if rank == 0:
            if ss == "420het":  # the reference benchmark decodes ONE canvas per call
                one = jl.Batch(ctx).upload(files[:1], fmt)
                one.decode().sync()
                t0 = time.perf_counter()
                for _ in range(5):
                    one.decode().sync()
                res["single_image_decode_ms"] = round((time.perf_counter() - t0) / 5 * 1e3, 3)
                one.close()
            if ss in ("420p", "420hetp") and world == 1:
                ctx2 = jl.Context(ctx.device)
                other = jl.Batch(ctx2).upload(files, fmt)
                for b in (batch, other):
                    b.decode()
                for b in (batch, other):
                    b.sync()
                t0 = time.perf_counter()
                for _ in range(steps):
                    batch.decode()
                    other.decode()
                    batch.sync()
                    other.sync()
                t2 = (time.perf_counter() - t0) / steps
                ok = all(other.result(i).status == 0 for i in (0, n_images - 1))
                res["value_two_in_flight"] = round(2 * n_images * width * height / 1e6 / t2, 1) if ok else None
                other.close()
            res["parity_spot_check"] = spot_check(jl, batch, files, fmt, sorted({0, n_images - 1}))
        return res
    finally:
        batch.close()


LATENCY_CASES = {
    # the reference's own call pattern: ONE image per Decode() (DecoderBenchmark.cs:51-73, apps/JpegDecode/DecodeAction.cs:26-56)
    "512_444": (512, 512, "444", 75, 0),
    "1080p_q90": (1920, 1080, "420", 90, 4),
    "4k_dri4": (3840, 2160, "420", 75, 4),
    "4k_dri0": (3840, 2160, "420", 75, 0),
}


def measure_latency(jl, ctx, jpegsynth, reps=20):
    """One image per call, four shapes, median of `reps` calls each:
      decode_resident_ms   jpgpu_batch_decode + sync of an uploaded one-image batch: file in HBM, pixels left in HBM
      from_host_ms         jpgpu_batch_upload (SetInput + Identify + header parse + H2D from pageable memory) + decode + sync
      with_pixels_back_ms  the reference caller's whole sequence through the decoder mirror (jpgpu_decoder_*): SetInput, Identify,
                           SetOutputWriter(JpegBufferOutputWriter8Bit over a host buffer), Decode() -- pixels in host memory at the end
      cpu_single_core_ms   the same sequence in the CPU restatement (checker), one core -- what one call costs without the GPU"""
    from oracle import pyoracle as po

    def med(f, n):
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            f()
            ts.append(time.perf_counter() - t0)
        ts.sort()
        return ts[len(ts) // 2] * 1e3

    out = {}
    for name, (w, h, ss, q, dri) in LATENCY_CASES.items():
        data = bytes(jpegsynth.encode(w, h, ss, q, dri, seed=4242))
        one = jl.Batch(ctx).upload([data], jl.FMT_INTERLEAVED_U8)
        one.decode().sync()
        one.decode().sync()
        resident = med(lambda: one.decode().sync(), reps)

        def from_host():
            one.upload([data], jl.FMT_INTERLEAVED_U8)
            one.decode().sync()

        from_host()
        t_host = med(from_host, reps)
        ok = one.result(0).status == 0
        one.close()
        buf = np.zeros(w * h * 3, np.uint8)
        dec = jl.JpegDecoder(ctx)

        def mirror():
            dec.SetInput(data)
            dec.Identify()
            dec.SetOutputWriter(jl.JpegBufferOutputWriter8Bit(w, h, 3, buf))
            dec.Decode()

        mirror()
        mirror()
        t_mirror = med(mirror, max(5, reps // 2))
        ref = po.decode_8bit(data)[0]
        ok = ok and bool(np.array_equal(buf.reshape(ref.shape), ref))
        dec.close()
        arr = np.frombuffer(data, np.uint8)
        sec, _ = po.decode_batch_mt([arr] * 3, 3, 1)
        out[name] = {"decode_resident_ms": round(resident, 3), "from_host_ms": round(t_host, 3), "with_pixels_back_ms": round(t_mirror, 3),
                     "cpu_single_core_ms": round(sec / 3 * 1e3, 2), "equal_to_checker": ok, "compressed_KB": round(len(data) / 1e3, 1)}
    out["note"] = ("one image per call (the reference's DecoderBenchmark / DecodeAction pattern), median of %d calls; with_pixels_back = SetInput + "
                   "Identify + SetOutputWriter + Decode through jpgpu_decoder_* into a host buffer, compared with the checker" % reps)
    return out


def measure_next_rows(jl, jpegsynth, gen_threads, images=128):
    """SURVEY 8(f)'s rows either side of the path, briefly, for the driver's one run: the encoder (tools/bench_encode.py's workload:
    4K RGB -> 4:2:0 Q75, standard tables) and the optimizer (tools/bench_optimize.py's: 4K 4:2:0 Q75 DRI = 7, strip), `images` each,
    pixels / files resident in HBM, one output of each against the checker."""
    from concurrent.futures import ThreadPoolExecutor
    from tools.bench_encode import image
    w, h = 3840, 2160
    out = {}
    with ThreadPoolExecutor(max(1, min(gen_threads, 16))) as ex:
        base = list(ex.map(lambda k: image(w, h, k), range(min(images, 8))))
    b = jl.EncodeBatch().upload([base[i % len(base)] for i in range(images)], (2, 2), 75, rgb=True)
    b.encode()
    b.encode()  # (the second encode of an upload sizes the one-pass entropy stage's buffer from the first)
    t0 = time.perf_counter()
    for _ in range(5):
        b.encode()
    dt = (time.perf_counter() - t0) / 5
    check = "not checked"
    try:
        from oracle import pyoracle as po
        check = "byte-exact vs oracle" if b.output(0) == po.encode_8bit(po.rgb_to_ycbcr8(base[0]), 2, 2, 75) else "MISMATCH vs oracle"
    except Exception as e:  # pragma: no cover
        check = f"not checked ({e})"[:120]
    out["encode_4k_420"] = {"value": round(images * w * h / dt / 1e6, 1), "unit": "Mpixels/s", "ms_per_step": round(dt * 1e3, 3), "images": images,
                            "stage_ms": {k: round(v, 3) for k, v in b.stage_ms().items()}, "entropy_stage_one_pass": b.emit_passes()[0] > 0,
                            "parity_spot_check": check}
    b.close()
    buf, sizes, stride = jpegsynth.encode_batch(images, w, h, "420", 75, 7, seed0=1, nthreads=gen_threads)
    files = [bytes(buf[i * stride:i * stride + int(sizes[i])]) for i in range(images)]
    ob = jl.OptimizeBatch().upload(files, True)
    ob.run()
    t0 = time.perf_counter()
    for _ in range(5):
        ob.run()
    dt = (time.perf_counter() - t0) / 5
    check = "not checked"
    try:
        from oracle import pyoracle as po
        check = "byte-exact vs oracle" if ob.output(0) == po.optimize(files[0], True) else "MISMATCH vs oracle"
    except Exception as e:  # pragma: no cover
        check = f"not checked ({e})"[:120]
    out["optimize_4k_420"] = {"value": round(images * w * h / dt / 1e6, 1), "unit": "Mpixels/s", "ms_per_step": round(dt * 1e3, 3), "images": images,
                              "parity_spot_check": check}
    ob.close()
    return out


def free_port():
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(args):
    """`bench.py --gpus N` (N > 1) started without torchrun: the ranks are a fresh child launch, started before this process has
    touched a GPU (a process that has must not be replaced or re-executed); its one line is relayed."""
    import subprocess

    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + [a for a in sys.argv[1:] if a != "--multi-inprocess"]
    log(f"[bench] --gpus {args.gpus} without WORLD_SIZE: launching {' '.join(cmd[1:8])} ...")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env, cwd=ROOT)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    for ln in proc.stdout.splitlines():
        if not ln.startswith("{"):
            log(ln)
    if proc.returncode != 0 or not lines:
        log(f"[bench] the rank launch failed (exit code {proc.returncode})")
        sys.exit(proc.returncode or 1)
    out = json.loads(lines[-1])
    out["launch"] = "self-spawned torch.distributed.run, one rank per GPU"
    if args.multi_inprocess and not args.dry_run:
        try:
            out["multi_inprocess"] = multi_inprocess(args)
            out["value_multi_inprocess"] = out["multi_inprocess"]["value"]
        except Exception as e:  # pragma: no cover
            out["multi_inprocess"] = {"value": None, "error": str(e)[:200]}
    print(json.dumps(out), flush=True)
    sys.exit(0)


def multi_inprocess(args):
    """The same shard through the library's own multi-device driver (include/jpgpu.h jpgpu_multi_*): ONE process, a
    context and two batches per device slot, image i on slot i mod G.  Inputs resident in HBM after one jpgpu_multi_decode; a
    step = every slot's decode issued from this one thread (jpgpu_batch_decode does not wait for the device), then every slot
    waited for.  With fewer devices than --gpus the slots share the devices there are (a plumbing check, not a scaling point)."""
    import torch  # device count only

    import jpeglibrary_amd as jl
    from jpeglibrary_amd import sharding
    from tools import jpegsynth

    width, height, ss, quality, dri, default_images = WORKLOADS[args.workload]
    n_images = args.images or default_images
    fmt_name = args.format or DEFAULT_FORMAT.get(args.workload, "interleaved_u8")
    fmt = {"interleaved_u8": jl.FMT_INTERLEAVED_U8, "planar_u8": jl.FMT_PLANAR_U8, "rgb_u8": jl.FMT_RGB_U8, "rgba_u8": jl.FMT_RGBA_U8}[fmt_name]
    n_dev = max(1, torch.cuda.device_count())
    slots = [g % n_dev for g in range(args.gpus)]
    granted = granted_cpus(host_cpu_budget())
    per_slot = [make_inputs(sharding, jpegsynth, args.workload, n_images, g, args.gen_threads or granted, args.distinct)[0] for g in range(args.gpus)]
    files = [per_slot[i % args.gpus][i // args.gpus] for i in range(n_images * args.gpus)]  # image i -> slot i mod G
    md = jl.MultiDecoder(slots)
    md.decode(files, fmt)  # host parse + H2D + one decode on every slot
    upload_ms, decode_ms = md.upload_ms, md.decode_ms
    shards = [md.shard(g) for g in range(args.gpus)]
    for i in range(len(files)):
        if md.result(i).status != 0:
            raise RuntimeError(f"image {i} failed")
    for _ in range(max(1, args.warmup)):
        for b in shards:
            b.decode()
    for b in shards:
        b.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        for b in shards:
            b.decode()
    for b in shards:
        b.sync()
    elapsed = time.perf_counter() - t0
    res = {"value": round(len(files) * width * height * args.steps / elapsed / 1e6, 1), "unit": "Mpixels/s", "slots": args.gpus, "devices": n_dev,
           "ms_per_step": round(elapsed / args.steps * 1e3, 3), "first_call_upload_ms": round(upload_ms, 2), "first_call_decode_ms": round(decode_ms, 2),
           "parity_spot_check": spot_check(jl, shards[-1], per_slot[-1], fmt, [0, n_images - 1]),
           "note": "jpgpu_multi_*: one process, one context per device slot, image i on slot i mod G, inputs resident, every slot's step issued from one thread"
                   + ("" if n_dev >= args.gpus else f"; {args.gpus} slots share {n_dev} device(s): not a scaling point")}
    md.close()
    return res


def granted_cpus(budget):
    g = min(budget.get("affinity", budget["cpu_count"]), budget["cpu_count"])
    if "cgroup_quota_cpus" in budget:
        g = min(g, max(1, int(round(budget["cgroup_quota_cpus"]))))
    return max(1, g)


def library_sha256():
    import hashlib

    h = hashlib.sha256()
    with open(os.path.join(ROOT, "jpeglibrary_amd", "libjpgpu.so"), "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--images", type=int, default=0, help="images per GPU (default: the workload's batch size)")
    ap.add_argument("--workload", default="4k_dri4", choices=sorted(WORKLOADS))
    ap.add_argument("--format", default=None, choices=["interleaved_u8", "planar_u8", "rgb_u8", "rgba_u8"],
                    help="output layout (default: interleaved_u8; het_8192: rgba_u8, the reference benchmark's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ingest", action="store_true", help="skip the ingest-inclusive (upload beside decode) measurement")
    ap.add_argument("--no-planar-pass", action="store_true", help="skip the short PLANAR_U8 pass behind roofline.read_frac_planar")
    ap.add_argument("--latency", action="store_true", help="also time one image per call (always on for het_8192)")
    ap.add_argument("--latency-only", action="store_true", help="print the one-image-per-call table (configs.latency of the default line) and exit")
    ap.add_argument("--gen-threads", type=int, default=0)
    ap.add_argument("--dist", action="store_true", help="initialise torch.distributed (RCCL) even for a single rank: runs the barrier / "
                                                         "MAX-reduce path of the multi-GPU launch on a one-GPU box")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL): one rank per GPU, the driver's launch.  gloo: REHEARSAL of a world > 1 launch on a box with fewer GPUs "
                         "than ranks -- rank r uses device r mod (devices present), barriers and the MAX-reduce run on the host group; "
                         "everything else (input generation at granted // world threads, crews, the parked CPU baseline, the slowest-rank "
                         "ingest figures, the JSON line) is the code path of the real launch.  Its `value` is NOT a scaling point: the ranks share a device")
    ap.add_argument("--no-configs", action="store_true", help="default workload: skip the short passes of the other BASELINE.json configurations")
    ap.add_argument("--config-scale", type=int, default=1, help="tests: the extra configurations at 1/N of their batch sizes")
    ap.add_argument("--multi-inprocess", action="store_true", help="also run the shard through jpgpu_multi_* (one process, a context per device) "
                                                                     "and report it as value_multi_inprocess")
    ap.add_argument("--dry-run", action="store_true", help="rendezvous, barrier and MAX-reduce of the launch only: no device work, value null "
                                                           "(what a box without GPUs can check of the world > 1 launch)")
    ap.add_argument("--distinct", type=int, default=0, help="experiments only: synthesise this many distinct images and repeat them "
                                                            "to fill the batch (default: every image of the batch is distinct)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)  # (does not return; nothing has touched a GPU yet)
    if args.format is None:
        args.format = DEFAULT_FORMAT.get(args.workload, "interleaved_u8")

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        log(f"warning: WORLD_SIZE={world} but --gpus {args.gpus}; using WORLD_SIZE")
    n_gpus = world

    import torch

    dist = None
    host_group = None
    rehearsal = args.dist_backend == "gloo" or (args.dry_run and torch.cuda.device_count() == 0)
    reduce_device = "cpu" if rehearsal else "cuda"
    if rehearsal:
        local_rank = local_rank % max(1, torch.cuda.device_count())  # ranks share the devices there are
    if world > 1 or args.dist:
        import torch.distributed as dist_

        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        if not (args.dry_run and rehearsal):
            torch.cuda.set_device(local_rank)
        if rehearsal:
            dist.init_process_group("gloo")
            host_group = dist.group.WORLD
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            # host-side rendezvous (gloo): ranks that wait here block in a socket read and leave their CPUs to the rank that is
            # measuring something on the host (the CPU baseline) -- an RCCL barrier would keep a core per waiting rank spinning
            host_group = dist.new_group(backend="gloo")
    elif not args.dry_run:
        torch.cuda.set_device(local_rank)

    if args.dry_run:
        # the launch's own plumbing and nothing else: every rank meets at the barrier, the slowest rank's time is reduced, rank 0 prints
        from jpeglibrary_amd import sharding as sh

        elapsed = 0.5 + 0.25 * rank
        if dist is not None:
            dist.barrier(group=host_group)
            elapsed = sh.max_over_ranks(dist, elapsed, device=reduce_device)
        if rank == 0:
            print(json.dumps({"metric": "dry run (no device work)", "value": None, "unit": "Mpixels/s", "n_gpus": n_gpus, "steps": args.steps,
                              "warmup": args.warmup, "dry_run": True, "slowest_rank_s": elapsed, "scaling": "weak"}), flush=True)
        if dist is not None:
            dist.barrier(group=host_group)
            dist.destroy_process_group()
        return

    import jpeglibrary_amd as jl
    from jpeglibrary_amd import sharding
    from tools import jpegsynth

    if args.latency_only:
        print(json.dumps({"latency": measure_latency(jl, jl.Context(local_rank), jpegsynth)}), flush=True)
        return

    width, height, ss, quality, dri, default_images = WORKLOADS[args.workload]
    kind = "progressive (SOF2, 10 scans)" if ss in ("420p", "420hetp") else ("baseline (HETissueSlide canvas, DecoderBenchmark.cs)" if ss == "420het" else "baseline")
    n_images = args.images or default_images
    fmt = {"interleaved_u8": jl.FMT_INTERLEAVED_U8, "planar_u8": jl.FMT_PLANAR_U8, "rgb_u8": jl.FMT_RGB_U8, "rgba_u8": jl.FMT_RGBA_U8}[args.format]

    # ---- synthetic input: distinct seed per image and per rank.  All ranks generate at once and share the CPUs the job is
    # granted (affinity mask / cgroup quota, not os.cpu_count()): granted // world threads each
    cpu = os.cpu_count() or 1
    budget = host_cpu_budget()
    granted = granted_cpus(budget)
    gen_threads = args.gen_threads or max(1, granted // max(1, world))
    t0 = time.perf_counter()
    files, sizes, ss, args.distinct = make_inputs(sharding, jpegsynth, args.workload, n_images, rank, gen_threads, args.distinct)
    t_gen = time.perf_counter() - t0
    log(f"[rank {rank}] generated {n_images} x {width}x{height} {ss} Q{quality} DRI={dri}: {sizes.sum() / 1e6:.1f} MB in {t_gen:.1f} s ({gen_threads} threads)")

    gen_s_ranks = [round(t_gen, 1)]
    if dist is not None:
        gen_s_ranks = [None] * world
        dist.all_gather_object(gen_s_ranks, round(t_gen, 1), group=host_group)

    ctx = jl.Context(local_rank)
    if world > 1:
        ctx.set_host_threads(max(1, min(16, granted // world)))  # the ranks' ingest crews share the granted CPUs
    batch = jl.Batch(ctx)
    batch.upload(files, fmt)  # first call: device buffers and the pinned staging ring are allocated here
    t0 = time.perf_counter()
    batch.upload(files, fmt)
    t_upload = time.perf_counter() - t0
    ingest = batch.ingest_stats()
    totals = batch.totals()
    log(f"[rank {rank}] host parse + H2D: {t_upload * 1e3:.1f} ms; {ingest}; {totals}")

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        batch.decode()
    batch.sync()
    # every image must have decoded cleanly, and a sample must be bit-exact (checked after timing, below)
    for i in range(n_images):
        r = batch.result(i)
        if r.status != 0 and not os.environ.get("JPGPU_BENCH_EXPERIMENT"):  # (kernel-timing experiments with deliberately broken builds)
            raise RuntimeError(f"image {i} failed: status {r.status} detail {r.detail}")
    if args.warmup:
        batch.stage_ms()  # drop warm-up events

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        batch.decode()
    batch.sync()
    barrier()
    elapsed = time.perf_counter() - t0
    stage = batch.stage_ms()  # HIP-event averages over exactly the timed steps, on the stream the kernels ran on

    if dist is not None:
        elapsed = sharding.max_over_ranks(dist, elapsed, device=reduce_device)
    value = sharding.aggregate_throughput(n_images * width * height, n_gpus, args.steps, elapsed)

    # ---- single-image latency (the reference benchmark decodes ONE image per call): decode alone with the file resident in
    # HBM; SetInput + Identify + Decode from host memory; and the same with the pixels copied back to the host
    latency = None
    if rank == 0 and (args.workload == "het_8192" or args.latency):
        one = jl.Batch(ctx)
        one.upload(files[:1], fmt)
        one.decode().sync()
        reps = 10
        t0 = time.perf_counter()
        for _ in range(reps):
            one.decode().sync()
        t_dec = (time.perf_counter() - t0) / reps
        one.stage_ms()
        for _ in range(reps):
            one.decode()
        one.sync()
        st1 = one.stage_ms()
        t0 = time.perf_counter()
        for _ in range(reps):
            one.upload(files[:1], fmt)
            one.decode().sync()
        t_up = (time.perf_counter() - t0) / reps
        t0 = time.perf_counter()
        for _ in range(3):
            one.upload(files[:1], fmt)
            one.decode().sync()
            one.output(0)
        t_all = (time.perf_counter() - t0) / 3
        latency = {"decode_ms": round(t_dec * 1e3, 3), "upload_decode_ms": round(t_up * 1e3, 3), "upload_decode_download_ms": round(t_all * 1e3, 3),
                   "single_image_Mpixels/s": round(width * height / 1e6 / t_dec, 1), "stage_ms": {k: round(v, 4) for k, v in st1.items()},
                   "subseq_rounds": one.subseq_rounds(),
                   "note": "one image per call, 10 calls each: decode with the file resident in HBM; jpgpu_batch_upload (SetInput + Identify + "
                           "header parse + H2D, pageable memory) + decode; the same + D2H of the pixels into pageable memory"}
        one.close()

    # ---- config 5 is the latency of ONE wave per scan: a single batch of 256 frames leaves three quarters of the SIMDs idle.
    # Two batches in flight (two contexts = two streams, inputs of both resident) is what a caller who has more than one
    # batch does about it; reported beside `value`, never as `value`
    two_in_flight = None
    if rank == 0 and args.workload in ("4k_progressive", "het_progressive") and not os.environ.get("JPGPU_PROG_BY_SCAN"):  # (not under the per-scan profiling switch)
        try:
            ctx2 = jl.Context(local_rank)
            other = jl.Batch(ctx2).upload(files, fmt)
            for _ in range(max(1, args.warmup)):
                batch.decode()
                other.decode()
                batch.sync()
                other.sync()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                batch.decode()
                other.decode()
                batch.sync()
                other.sync()
            t2 = (time.perf_counter() - t0) / args.steps
            ok = all(other.result(i).status == 0 for i in (0, n_images - 1))
            two_in_flight = {"value": round(2 * n_images * width * height / 1e6 / t2, 1) if ok else None, "ms_per_pair": round(t2 * 1e3, 3),
                             "note": "two batches of the same size decoded side by side on two streams of the one device (two contexts), inputs resident"}
            other.close()
            batch.stage_ms()
        except Exception as e:  # pragma: no cover
            two_in_flight = {"value": None, "error": str(e)[:160]}

    # ---- the same workload with the host in the loop (never `value`): upload of batch k+1 beside the decode of batch k, on
    # every rank at once (the ranks' crews share the host), the slowest rank's time per batch; from pageable memory through
    # the staging ring, and from page-locked memory by DMA (zero-copy)
    ingest_res = {}
    if not args.no_ingest:
        rounds = max(4, min(args.steps, 8))
        for key in ("pageable", "pinned"):
            per_batch_s, st, err = float("inf"), None, None
            met = [False]

            def meet():
                met[0] = True
                barrier()
            try:
                src = files
                arena = None
                if key == "pinned":
                    # one page-locked read buffer, the files back to back (64-byte aligned) as an application that reads
                    # its files straight into it would have them
                    offs = np.concatenate([[0], np.cumsum((sizes.astype(np.int64) + 63) // 64 * 64)])
                    arena = ctx.host_alloc(int(offs[-1]) + 64)
                    for i in range(n_images):
                        arena[offs[i]:offs[i] + int(sizes[i])] = files[i]
                    src = [arena[offs[i]:offs[i] + int(sizes[i])] for i in range(n_images)]
                per_batch_s, st = ingest_inclusive(jl, ctx, batch, src, fmt, rounds, pinned=(key == "pinned"), sync_ranks=meet)
                if arena is not None:
                    ctx.host_free(arena)
            except Exception as e:  # pragma: no cover
                err = str(e)[:160]
                per_batch_s = float("inf")
                if not met[0]:
                    barrier()  # the other ranks are waiting at the start line
            if dist is not None:
                per_batch_s = sharding.max_over_ranks(dist, per_batch_s, device=reduce_device)
            ingest_res[key] = (per_batch_s, st, err)
        batch.upload(files, fmt)  # leave the batch as the timed region had it (the spot check below reads it)
        batch.decode().sync()

    # ---- the other BASELINE.json configurations, briefly (default workload only, behind the headline's timed region, never part
    # of `value`).  One GPU: all five on this rank.  N > 1: config 4 (8192 x 1080p Q90 over 8 GPUs = 1024 per rank) on every
    # rank, sharded and reduced like the headline.  Bounded: a configuration that would start beyond the time budget is skipped.
    configs = None
    if args.workload == "4k_dri4" and args.format == "interleaved_u8" and not args.no_configs and not (args.images and args.images < 64):
        configs = {}
        t_cfg = time.perf_counter()
        names = ["512_444", "4k_dri0", "1080p_q90", "4k_progressive", "het_8192"] if world == 1 else ["1080p_q90"]
        for name in names:
            if time.perf_counter() - t_cfg > 150.0 and world == 1:
                configs[name] = {"skipped": "time budget of the extra configurations spent"}
                continue
            try:
                configs[name] = measure_config(jl, sharding, jpegsynth, torch, dist, reduce_device, ctx, name, rank, world, gen_threads,
                                               steps=3 if name in ("4k_progressive",) else 5, warmup=2,
                                               images=max(1, WORKLOADS[name][5] // max(1, args.config_scale)))
            except Exception as e:  # pragma: no cover
                if dist is not None:
                    raise  # (the other ranks are inside the same barriers)
                configs[name] = {"error": str(e)[:200]}
            log(f"[rank {rank}] config {name}: {json.dumps(configs[name])[:300]}")
        if world == 1 and time.perf_counter() - t_cfg <= 150.0:
            try:
                configs.update(measure_next_rows(jl, jpegsynth, gen_threads, images=max(2, 128 // max(1, args.config_scale))))
                log(f"[rank {rank}] next rows: {json.dumps({k: configs[k] for k in ('encode_4k_420', 'optimize_4k_420')})[:400]}")
            except Exception as e:  # pragma: no cover
                configs["encode_4k_420"] = {"error": str(e)[:200]}
        if rank == 0 and world == 1:
            try:
                t_lat = time.perf_counter()
                configs["latency"] = measure_latency(jl, ctx, jpegsynth)
                configs["latency"]["seconds"] = round(time.perf_counter() - t_lat, 1)
                log(f"[rank {rank}] latency: {json.dumps(configs['latency'])[:600]}")
            except Exception as e:  # pragma: no cover
                configs["latency"] = {"error": str(e)[:200]}
        if rank == 0:
            configs["note"] = ("short passes of the other BASELINE.json configurations, run after the headline's timed region with the same step "
                               "definition (inputs resident in HBM, output left in HBM); roofline = K3's algorithmic bytes / its HIP-event time; "
                               f"{time.perf_counter() - t_cfg:.0f} s in all, input generation included")

    if rank == 0:
        # dominant kernel: idct_output_kernel.  Algorithmic bytes per launch: 128 B of int16 coefficients
        # read per 8x8 block + the output bytes written in the chosen layout.
        idct_bytes = totals["blocks"] * 128 + totals["output_bytes"]
        idct_s = stage["idct"] / 1e3
        achieved = idct_bytes / idct_s / 1e9
        # HBM bytes per launch from the PMC counters: only from a committed profile of THIS binary (sha256 of libjpgpu.so
        # recorded by tools/profile_pmc.sh), scaled by the launch's image count; any other binary -> null
        traffic = None
        traffic_source = None
        tpath = os.path.join(ROOT, "profiles", "idct_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath)).get(f"{args.workload}:{args.format}")
                if tj and tj.get("library_sha256") == library_sha256():
                    traffic = int(tj["hbm_bytes_per_image"]) * n_images
                    traffic_source = tj.get("source")
            except Exception:
                traffic = None
        sink = {"interleaved_u8": "interleaved YCbCr8", "planar_u8": "planar YCbCr8", "rgb_u8": "RGB8", "rgba_u8": "RGBA8"}[args.format]
        if args.workload == "4k_dri4":
            metric = "Mpixels/s decoded, 4K 4:2:0 baseline Q75 RST=4, 1 & 8 GPU vs CPU ref"  # BASELINE.json's metric, verbatim
            if args.format != "interleaved_u8":
                metric += f" ({sink} sink)"
        else:
            metric = f"Mpixels/s decoded, {width}x{height} {ss} {kind} Q{quality} DRI={dri}, {sink} sink, vs CPU ref"
        kfmt = {"interleaved_u8": 0, "planar_u8": 1, "rgb_u8": 3, "rgba_u8": 4}[args.format]
        klay = {"420": 3, "422": 2, "444": 1, "gray": 4}.get(ss, 0) if args.format != "planar_u8" else 0
        out = {
            "metric": metric,
            "value": round(value, 1),
            "unit": "Mpixels/s",
            "n_gpus": n_gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            **({"rehearsal": f"{world} ranks over gloo sharing {torch.cuda.device_count()} device(s): plumbing check of the world > 1 launch, "
                             "not a scaling point"} if rehearsal and world > 1 else {}),
            "config": {
                "workload": f"{n_images} x {width}x{height} {ss} {kind} Q{quality} DRI={dri} per GPU, output {args.format} resident in HBM",
                "images_per_gpu": n_images,
                **({"distinct_images": args.distinct} if args.distinct and args.distinct < n_images else {}),
                "compressed_MB_per_gpu": round(totals["compressed_bytes"] / 1e6, 1),
                "sharding": "image-per-GPU, no collective",
            },
            "stage_ms": {k: round(v, 4) for k, v in stage.items()},
            "marker_fallbacks": batch.marker_fallbacks(),  # groups of K1 that gave up waiting for a predecessor (expected: 0)
            **({"subseq_rounds": batch.subseq_rounds()} if dri == 0 and "prog" not in args.workload else {}),
            **({"content": "crops of tests/golden/HETissueSlide.jpg (the reference benchmark's image), progressive re-encode by Pillow"}
               if args.workload == "het_progressive" else {}),
            "roofline": {
                "kernel": f"idct_output_kernel<{kfmt},{klay}>",
                "bound": "hbm",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "read_frac": round(totals["blocks"] * 128 / idct_s / 1e9 / HBM_PEAK_GBS, 4),
                "algorithmic_bytes": idct_bytes,
                "traffic": traffic,
                "traffic_source": traffic_source,
            },
            "host": {"gen_s": round(t_gen, 1), "gen_s_per_rank": gen_s_ranks, "gen_threads_per_rank": gen_threads, "parse_upload_s": round(t_upload, 4),
                     "cpu_count": cpu, "granted_cpus": granted,
                     "ingest": {k: (round(v, 2) if isinstance(v, float) else v) for k, v in ingest.items()}},
        }
        if latency is not None:
            out["latency"] = latency
        if two_in_flight is not None:
            out["value_two_in_flight"] = two_in_flight["value"]
            out["two_in_flight"] = two_in_flight
        for key, name in (("pageable", "value_ingest_inclusive"), ("pinned", "value_ingest_inclusive_pinned")):
            if key not in ingest_res:
                continue
            per_batch_s, st, err = ingest_res[key]
            if err is None and per_batch_s != float("inf"):
                out[name] = round(n_images * n_gpus * width * height / 1e6 / per_batch_s, 1)
                out["host"][f"ingest_inclusive_{key}_ms_per_batch"] = round(per_batch_s * 1e3, 2)
                out["host"][f"ingest_{key}"] = {k: (round(v, 2) if isinstance(v, float) else v) for k, v in st.items()}
            else:
                out[name] = None
                out["host"][f"ingest_inclusive_{key}_error"] = err
        if ingest_res:
            out["host"]["ingest_inclusive_note"] = ("whole job, slowest rank per batch; header-only host parse + H2D of batch k+1 beside the decode of "
                                                    "batch k (two batches per context); pageable: the host crew copies the files into the pinned "
                                                    "staging ring; pinned: the files lie in page-locked memory and are DMA'd from there")
        # north_star's literal bar -- the IDCT stage's share of the HBM READ bandwidth -- is defined on the planar sink
        # (192 B per block): a short pass of the same batch with PLANAR_U8 output, K3 timed by HIP events
        if args.format == "interleaved_u8" and not args.no_planar_pass:
            try:
                pb = jl.Batch(ctx).upload(files, jl.FMT_PLANAR_U8)
                for _ in range(2):
                    pb.decode()
                pb.sync()
                pb.stage_ms()
                for _ in range(5):
                    pb.decode()
                pst = pb.stage_ms()
                ptot = pb.totals()
                pb.close()
                out["roofline"]["read_frac_planar"] = round(ptot["blocks"] * 128 / (pst["idct"] / 1e3) / 1e9 / HBM_PEAK_GBS, 4)
                out["roofline"]["planar_pass"] = {"kernel": "idct_output_kernel<1,0>", "steps": 5, "idct_ms": round(pst["idct"], 4),
                                                  "algorithmic_bytes": ptot["blocks"] * 128 + ptot["output_bytes"],
                                                  "achieved": round((ptot["blocks"] * 128 + ptot["output_bytes"]) / (pst["idct"] / 1e3) / 1e9, 1)}
            except Exception as e:  # pragma: no cover
                out["roofline"]["read_frac_planar"] = None
                out["roofline"]["planar_pass_error"] = str(e)[:120]
        # D2H of the pixels: reported, never part of `value`; a bounded sample of the images, pageable host memory
        try:
            n_d2h = min(n_images, 16)
            t_d = time.perf_counter()
            nbytes = 0
            for i in range(n_d2h):
                o = batch.output(i)
                nbytes += sum(p.nbytes for p in o) if isinstance(o, list) else o.nbytes
            t_d = time.perf_counter() - t_d
            out["host"]["d2h_GBps"] = round(nbytes / t_d / 1e9, 2)
            out["host"]["d2h_sample"] = f"{n_d2h} images, {nbytes / 1e6:.0f} MB"
        except Exception as e:  # pragma: no cover
            out["host"]["d2h_GBps"] = None
            out["host"]["d2h_error"] = str(e)[:80]
        # spot check after timing: a few images bit-exact against the oracle (checker only)
        if os.environ.get("JPGPU_BENCH_EXPERIMENT"):  # kernel-timing experiments with deliberately broken outputs
            out["parity_spot_check"] = "oracle unavailable"
        else:
            out["parity_spot_check"] = spot_check(jl, batch, files, fmt, sorted(set([0, n_images // 2, n_images - 1])))
        if configs is not None:
            out["configs"] = configs
        if args.multi_inprocess and world == 1:
            try:
                batch.close()  # (the in-process driver brings its own contexts and batches)
                out["multi_inprocess"] = multi_inprocess(args)
                out["value_multi_inprocess"] = out["multi_inprocess"]["value"]
            except Exception as e:  # pragma: no cover
                out["multi_inprocess"] = {"value": None, "error": str(e)[:200]}
        # rank 0 alone, the other ranks parked in the host-side barrier below (blocked in a socket read, not spinning)
        if not args.no_cpu_baseline:
            big = width * height > 32e6  # 67-Mpixel frames: two decodes per thread, no more threads than CPUs granted (0.5 GB of buffers each)
            out["cpu_baseline"] = cpu_baseline(files, width, height, cpu, rgba=(args.format == "rgba_u8"), per_thread=2 if big else 8,
                                               thread_cap=granted if big else 0)
            out["cpu_baseline"]["gpu_over_cpu"] = round(value / out["cpu_baseline"]["value"], 1)
        print(json.dumps(out), flush=True)

    if dist is not None:
        dist.barrier(group=host_group)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
