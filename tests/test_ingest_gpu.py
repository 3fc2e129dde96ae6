"""jpgpu_batch_upload: the host reads headers only, the device confirms what lies behind the first SOS header.

SURVEY 8f N1 (ref: JpegDecoder.Identify's full-stream walk, JpegDecoder.cs:75-162 + JpegReader.cs:120-158).  Every case
is checked against the oracle's Identify + Decode, whichever way the file was planned; the ingest statistics say which way.
"""
import threading

import numpy as np
import pytest

import jpeglibrary_amd as jl
from golden_util import read_jpeg
from oracle import pyoracle as po
from tools import jpegsynth

pytestmark = pytest.mark.gpu

NAMES = {0: "OK", 1: "InvalidDataException", 2: "InvalidOperationException", 3: "NotSupportedException", 4: "ArgumentException"}


def _oracle(data):
    try:
        return "OK", po.decode_8bit(data)[0]
    except po.OracleError as e:
        return e.kind, None


def _check_batch(files, expect_header_only=None, ctx=None):
    b = jl.Batch(ctx).upload(files, jl.FMT_INTERLEAVED_U8).decode().sync()
    st = b.ingest_stats()
    problems = []
    for i, f in enumerate(files):
        kind, ref = _oracle(bytes(f))
        res = b.result(i)
        mine = NAMES.get(res.status, str(res.status))
        if mine != kind:
            problems.append((i, kind, mine, res.detail))
        elif ref is not None and not np.array_equal(b.output(i), ref):
            problems.append((i, "samples differ"))
    b.close()
    assert not problems, problems
    if expect_header_only is not None:
        assert st["n_header_only"] == expect_header_only, st
        assert st["n_header_only"] + st["n_full_walk"] <= len(files)
    return st


def test_clean_files_are_planned_from_their_headers():
    files = [jpegsynth.encode(w, h, ss, q, dri, seed=w + h) for (w, h, ss, q, dri) in
             [(512, 512, "444", 75, 0), (640, 368, "420", 75, 4), (331, 177, "422", 80, 3), (100, 75, "gray", 60, 1), (17, 9, "420", 75, 1)]]
    files += [read_jpeg("cramps.jpg"), read_jpeg("lake.jpg")]
    st = _check_batch(files, expect_header_only=len(files))
    assert st["n_full_walk"] == 0


def test_files_that_need_the_full_walk_get_it():
    good = bytes(jpegsynth.encode(160, 96, "420", 75, 2, seed=42))
    body = good[:-2]
    sos = good.index(b"\xff\xda")
    dri = good.index(b"\xff\xdd")
    dri_seg = good[dri:dri + 6]
    no_dri = good[:dri] + good[dri + 6:]
    cases = {
        # header-only plans that the device confirms
        "clean": (good, True),
        "bytes_behind_eoi": (good + bytes(range(1, 200)), True),
        "second_eoi": (good + b"\xff\xd9", True),
        "early_eoi_at_restart": (None, True),  # filled in below
        # the first marker behind the scan is not EOI: Identify / Decode go on walking
        "com_behind_scan": (body + b"\xff\xfe\x00\x04ab\xff\xd9", False),
        "bad_segment_behind_scan": (body + b"\xff\xfe\x00\x09ab\xff\xd9", False),
        # Identify latches the LAST DRI of the file, also one behind the scan: decoded with the wrong interval
        "dri_behind_scan": (body + b"\xff\xdd\x00\x04\x00\x07\xff\xd9", False),
        "dri_only_behind_scan": (no_dri[:-2] + dri_seg + b"\xff\xd9", False),
        "second_sof_behind_scan": (body + good[good.index(b"\xff\xc0"):sos] + b"\xff\xd9", False),
        "no_eoi": (body, False),
        "truncated_in_scan": (good[:sos + 14 + 200], False),
        "zeros_for_eoi": (body + bytes(2), False),
        "multi_scan": (bytes(jpegsynth.encode(96, 64, "444", 75, 0, seed=5, noninterleaved=True)), False),
        "progressive": (read_jpeg("progress.jpg"), False),
        "progressive_restart": (read_jpeg("yellowcat_progressive_restart.jpg"), False),
        "no_scan_at_all": (good[:sos] + b"\xff\xd9", False),
        "sos_before_sof": (good[:2] + good[sos:sos + 14] + good[2:], False),
        "not_a_jpeg": (bytes(range(256)) * 4, None),
        "empty_ish": (b"\xff\xd8", None),
    }
    rsts = [i for i in range(sos + 14, len(good) - 1) if good[i] == 0xFF and 0xD0 <= good[i + 1] <= 0xD7]
    cases["early_eoi_at_restart"] = (good[:rsts[10]] + b"\xff\xd9", True)
    keys = list(cases)
    for k in keys:  # one by one: the statistics say how each file was planned
        data, header_only = cases[k]
        st = _check_batch([data])
        if header_only is not None:
            assert (st["n_header_only"], st["n_full_walk"]) == ((1, 0) if header_only else (0, 1)), (k, st)
    # and all of them in one batch (files of both kinds next to each other in the staging ring)
    st = _check_batch([cases[k][0] for k in keys])
    assert st["n_header_only"] == sum(1 for k in keys if cases[k][1] is True), st


def test_one_unread_byte_in_front_of_the_terminator_on_the_header_only_path():
    """The reference resumes its marker walk one byte INTO the terminating marker when exactly one whole byte was left in
    the bit reader: the header-only path replays that walk from the headers and the bytes behind EOI."""
    good = bytes(jpegsynth.encode(104, 72, "420", 80, 4, seed=91))
    body = good[:-2]
    files = []
    for k in (1, 2, 3):
        files.append(body + bytes([0x5A] * k) + b"\xff\xd9")
        files.append(body + bytes([0x5A] * k) + b"\xff\xd9\xff\xd9")
        files.append(body + bytes([0x5A] * k) + b"\xff\xd9\xff\xfe\x00\x04ab")
        files.append(body + bytes([0x5A] * k) + b"\xff\xd9\xff\xfe\x00\x09ab")
    st = _check_batch(files)
    assert st["n_header_only"] == len(files), st


def test_staging_ring_wraps_and_thread_counts_agree(monkeypatch):
    """More input than the staging ring holds (here 3 slots of 4 MiB: the ring wraps several times), files that straddle
    slots, and the same batch through 1, 3, 7 and the default number of host threads."""
    n = 48
    buf, sizes, stride = jpegsynth.encode_batch(n, 3840, 2160, "420", 75, 4, seed0=7000, nthreads=8)
    files = [buf[i * stride:i * stride + int(sizes[i])] for i in range(n)]
    assert sum(len(f) for f in files) > 10 * (4 << 20)
    monkeypatch.setenv("JPGPU_STAGING_SLOTS", "3")
    monkeypatch.setenv("JPGPU_STAGING_SLOT_MB", "4")
    ctx = jl.Context(0)
    outs = {}
    for threads in (0, 1, 3, 7):
        ctx.set_host_threads(threads)
        b = jl.Batch(ctx).upload(files, jl.FMT_INTERLEAVED_U8).decode().sync()
        st = b.ingest_stats()
        assert st["n_header_only"] == n and st["n_full_walk"] == 0, st
        if threads:
            assert st["threads"] == threads
        assert all(b.result(i).status == 0 for i in range(n))
        outs[threads] = [b.output(i) for i in (0, 3, 4, 11, 12, 31, n - 1)]
        b.close()
    for k, i in enumerate((0, 3, 4, 11, 12, 31, n - 1)):
        ref, _ = po.decode_8bit(bytes(files[i]))
        for threads in outs:
            assert np.array_equal(outs[threads][k], ref), (threads, i)
    ctx.close()


def test_upload_of_one_batch_runs_beside_the_decode_of_another():
    """Two batches of one context: B is uploaded (upload stream, host crew) while A decodes (decode stream)."""
    ctx = jl.Context(0)
    buf, sizes, stride = jpegsynth.encode_batch(24, 1920, 1080, "420", 90, 4, seed0=300, nthreads=8)
    files = [buf[i * stride:i * stride + int(sizes[i])] for i in range(24)]
    a, b = jl.Batch(ctx), jl.Batch(ctx)
    a.upload(files[:12], jl.FMT_INTERLEAVED_U8)
    for _ in range(3):
        a.decode()                                   # asynchronous: returns once the kernels are queued
        b.upload(files[12:], jl.FMT_INTERLEAVED_U8)  # meanwhile
        a.sync()
        b.decode()
        a.upload(files[:12], jl.FMT_INTERLEAVED_U8)
        b.sync()
    a.decode().sync()
    for batch, part in ((a, files[:12]), (b, files[12:])):
        for i in (0, 5, 11):
            assert batch.result(i).status == 0
            assert np.array_equal(batch.output(i), po.decode_8bit(bytes(part[i]))[0])
    a.close()
    b.close()
    ctx.close()


def test_two_contexts_from_two_threads_shard_one_file_list():
    """image i -> GPU i mod G, one context per GPU / thread, no exchange.  With one GPU in the box both
    contexts sit on device 0; what is exercised is that contexts are independent and callable concurrently."""
    n = 24
    files = [jpegsynth.encode(320 + 16 * (i % 5), 200 + 8 * (i % 3), "420" if i % 2 else "444", 75, (i % 4) * 2, seed=900 + i) for i in range(n)]
    world = 2
    got = [None] * n
    errors = []

    def rank_main(rank):
        try:
            ctx = jl.Context(0)
            mine = jl.sharding.shard_indices(n, rank, world)
            b = jl.Batch(ctx).upload([files[i] for i in mine], jl.FMT_INTERLEAVED_U8)
            for _ in range(4):
                b.decode()
            b.sync()
            for k, i in enumerate(mine):
                assert b.result(k).status == 0
                got[i] = b.output(k)
            b.close()
            ctx.close()
        except Exception as e:  # pragma: no cover
            errors.append((rank, repr(e)))

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert sorted(jl.sharding.shard_indices(n, 0, world) + jl.sharding.shard_indices(n, 1, world)) == list(range(n))
    for i in range(n):
        assert np.array_equal(got[i], po.decode_8bit(bytes(files[i]))[0]), i


def test_overlapped_issue_order_gives_the_serial_result(monkeypatch):
    """With JPGPU_OVERLAP=1 jpgpu_batch_decode cuts large batches in two halves and runs the second half's Huffman stage
    beside the first half's output stage (two streams); every 8th call is serial.  Same bytes either way, and against the oracle."""
    n = 26  # 26 x 194 400 blocks: above the 4 Mi block threshold
    monkeypatch.setenv("JPGPU_OVERLAP", "1")
    buf, sizes, stride = jpegsynth.encode_batch(n, 3840, 2160, "420", 75, 4, seed0=4100, nthreads=8)
    files = [buf[i * stride:i * stride + int(sizes[i])] for i in range(n)]
    b = jl.Batch().upload(files, jl.FMT_INTERLEAVED_U8)
    b.decode().sync()                       # first call: serial, with stage events
    serial = [b.output(i).copy() for i in range(n)]
    first = b.stage_ms()
    assert first["huffman"] > 0 and first["idct"] > 0
    p, total = b.output_device_ptr()
    b.decode()                              # serial again (first call after the query) ...
    for _ in range(5):
        b.decode()                          # ... then five overlapped ones
    b.sync()
    st = b.stage_ms()
    assert st["total"] > 0 and st["idct"] > 0
    for i in range(n):
        assert b.result(i).status == 0
        assert np.array_equal(b.output(i), serial[i]), i
    for i in (0, n // 2 - 1, n // 2, n - 1):
        assert np.array_equal(serial[i], po.decode_8bit(bytes(files[i]))[0]), i
    b.close()
    # a corrupted file in each half: error reporting does not depend on the issue order either
    bad = [bytes(f) for f in files]
    for i in (3, n - 2):
        d = bytearray(bad[i])
        d[len(d) // 2: len(d) // 2 + 3] = b"\xff\xc4\x00"
        bad[i] = bytes(d)
    res = []
    for overlap in ("1", "0"):
        monkeypatch.setenv("JPGPU_OVERLAP", overlap)
        bb = jl.Batch().upload(bad, jl.FMT_INTERLEAVED_U8)
        bb.decode().decode().sync()
        res.append([(bb.result(i).status, bb.result(i).detail, bb.result(i).error_interval) for i in range(n)])
        bb.close()
    assert res[0] == res[1]
    assert res[0][3][0] != 0 and res[0][n - 2][0] != 0 and res[0][0][0] == 0


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def test_bench_runs_its_rccl_path_under_torchrun_with_one_rank():
    """The driver launches bench.py under torch.distributed.run with one rank per GPU; with one GPU in the box the same
    launcher, one rank, and --dist make bench.py create the RCCL process group and go through its barrier and its
    MAX-reduce of the elapsed time (device tensor), the code the N > 1 runs depend on."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(root, "bench.py"), "--gpus", "1", "--dist", "--workload", "1080p_q90", "--images", "8", "--steps", "2", "--warmup", "1"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["parity_spot_check"] == "bit-exact vs oracle"
    # what a SCALE run's lines must carry too: the CPU baseline (rank 0, the other ranks parked in a host-side
    # barrier), the ingest-inclusive rates (all ranks at once, pageable and page-locked input) and the planar-sink read fraction
    assert out["cpu_baseline"]["value"] > 0 and out["cpu_baseline"]["kind"] == "port"
    assert out["value_ingest_inclusive"] > 0 and out["value_ingest_inclusive_pinned"] > 0
    assert out["host"]["ingest_pinned"]["n_pinned_dma"] >= 1 and out["host"]["gen_s_per_rank"] and out["roofline"]["read_frac_planar"] > 0


def test_bench_rehearses_a_two_rank_launch_on_one_device():
    """No 8-GPU node has ever run bench.py's world > 1 branch.  The rehearsal mode (--dist-backend gloo) runs exactly that branch
    with two ranks that share the one device there is: input generation and ingest crews at granted // world threads, the
    barrier / MAX-reduce around the timed region, the slowest-rank ingest figures, the CPU baseline on rank 0 with rank 1 parked
    in the host barrier, one JSON line from rank 0 with the whole-job aggregate."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(root, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--workload", "1080p_q90", "--images", "8", "--steps", "2", "--warmup", "1"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "rank 0 alone prints the line"
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["scaling"] == "weak" and "rehearsal" in out
    assert out["config"]["images_per_gpu"] == 8 and out["parity_spot_check"] == "bit-exact vs oracle"
    assert len(out["host"]["gen_s_per_rank"]) == 2 and out["host"]["gen_threads_per_rank"] == max(1, out["host"]["granted_cpus"] // 2)
    assert out["cpu_baseline"]["value"] > 0 and out["value_ingest_inclusive"] > 0 and out["value_ingest_inclusive_pinned"] > 0
    assert out["roofline"]["frac"] > 0 and out["roofline"]["read_frac_planar"] > 0


def test_bench_line_carries_every_baseline_config_and_the_in_process_driver():
    """The driver runs bench.py once; its one line must let a reader check EVERY BASELINE.json configuration (round-4 verdict:
    "only one of five configs is driver-verified").  Scaled down here: the headline at 64 images, the other configurations at
    1/64 of their batch sizes -- each with a rate, stage times, K3's roofline fraction and an oracle spot check -- plus the
    same shard through jpgpu_multi_* (--multi-inprocess)."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--images", "64", "--steps", "2", "--warmup", "1", "--config-scale", "64",
                        "--no-ingest", "--multi-inprocess"], capture_output=True, text=True, timeout=1500, cwd=root, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["parity_spot_check"] == "bit-exact vs oracle"
    cfg = out["configs"]
    for name in ("512_444", "4k_dri0", "1080p_q90", "4k_progressive", "het_8192"):
        c = cfg[name]
        assert c.get("value", 0) > 0 and c["parity_spot_check"] == "bit-exact vs oracle", (name, c)
        assert c["stage_ms"]["idct"] > 0 and 0 < c["roofline"]["frac"] < 1, (name, c)
    assert cfg["4k_dri0"]["subseq_rounds"] >= 2 and cfg["4k_dri0"]["subseq_fallbacks"] == 0
    assert cfg["het_8192"]["single_image_decode_ms"] > 0 and cfg["4k_progressive"]["value_two_in_flight"] > 0
    # ... and the rows either side of the path: encoder and optimizer, each with one output against the checker
    for name in ("encode_4k_420", "optimize_4k_420"):
        assert cfg[name]["value"] > 0 and cfg[name]["parity_spot_check"] == "byte-exact vs oracle", (name, cfg[name])
    assert cfg["encode_4k_420"]["entropy_stage_one_pass"] is True
    # ... and the reference's OWN call pattern (round 6): one image per call, four shapes, each equal to the checker
    lat = cfg["latency"]
    for name in ("512_444", "1080p_q90", "4k_dri4", "4k_dri0"):
        row = lat[name]
        assert row["equal_to_checker"] is True, (name, row)
        assert 0 < row["decode_resident_ms"] <= row["from_host_ms"] * 1.5 and row["with_pixels_back_ms"] > 0 and row["cpu_single_core_ms"] > 0, (name, row)
    assert out["value_multi_inprocess"] > 0 and out["multi_inprocess"]["parity_spot_check"] == "bit-exact vs oracle"


def test_bench_spawns_two_ranks_itself_and_adds_the_in_process_driver():
    """`bench.py --gpus 2` without torchrun: two ranks launched as a child (gloo rehearsal: both on the one device), rank 0's line
    relayed with n_gpus: 2, and the in-process driver's figure for the same two-slot shard beside it."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--workload", "1080p_q90", "--images", "8",
                        "--steps", "2", "--warmup", "1", "--no-ingest", "--no-cpu-baseline", "--multi-inprocess"], capture_output=True, text=True,
                       timeout=900, cwd=root, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["value"] > 0 and "self-spawned" in out["launch"]
    assert out["multi_inprocess"]["slots"] == 2 and out["value_multi_inprocess"] > 0


def test_multi_device_driver_shards_round_robin_and_matches_the_oracle():
    """jpgpu_multi_*: the in-library driver of SURVEY 8e.  One MI355X here, so the device is listed three times (three
    independent contexts, three host threads uploading and decoding at once): image i lands on slot i mod 3 at local index
    i // 3, every image equals the checker -- baseline, DRI = 0, progressive and a corrupted file among them -- and the same
    list through one plain batch gives the same bytes."""
    import io

    from PIL import Image
    files = []
    for i in range(23):
        files.append(bytes(jpegsynth.encode(64 + 16 * (i % 7), 48 + 8 * (i % 5), ["420", "444", "422"][i % 3], 60 + i, [0, 2, 5][i % 3], seed=100 + i)))
    rng = np.random.default_rng(4)
    for k in range(4):
        buf = io.BytesIO()
        Image.fromarray(rng.integers(0, 256, (40 + 9 * k, 56 + 7 * k, 3), dtype=np.uint8)).save(buf, format="JPEG", quality=70 + k, progressive=True)
        files.append(buf.getvalue())
    files.append(files[3][: len(files[3]) // 2])  # truncated: a per-image failure, not a failure of the call
    m = jl.MultiDecoder([0, 0, 0])
    m.decode(files)
    assert len(m) == len(files)
    plain_outs, plain_results = jl.decode_batch(files)
    for i, f in enumerate(files):
        assert m.locate(i) == (i % 3, i // 3)
        res = m.result(i)
        assert res.status == plain_results[i].status and res.detail == plain_results[i].detail
        if res.status == 0:
            assert np.array_equal(m.output(i), po.decode_8bit(f)[0]), i
            assert np.array_equal(m.output(i), plain_outs[i]), i
    assert plain_results[-1].status != 0
    assert m.upload_ms > 0 and m.decode_ms > 0
    # a second call on the same driver, another format, fewer files than slots
    m.decode(files[:2], jl.FMT_RGBA_U8)
    for i in range(2):
        assert np.array_equal(m.output(i), jl.decode_batch([files[i]], jl.FMT_RGBA_U8)[0][0])
    m.close()
    with pytest.raises(jl.JpegError):
        jl.MultiDecoder([])



# ------------------------------------------------------------------------------------------------ round 3: segments, pinned, overlap

def _split(data, cuts):
    """bytes -> list of segments cut at the given offsets (the multi-segment ReadOnlySequence a reference caller may hand over)."""
    cuts = sorted({c for c in cuts if 0 < c < len(data)})
    parts, prev = [], 0
    for c in cuts + [len(data)]:
        parts.append(bytes(data[prev:c]))
        prev = c
    return parts


def _variety():
    """Files of every ingest kind: header-only plans, full walks (progressive, several scans, garbage behind the scan,
    bytes behind EOI that do not close the file), failures, and one whose first scan starts behind the 64 KiB head."""
    good = bytes(jpegsynth.encode(160, 96, "420", 75, 2, seed=42))
    sos = good.index(b"\xff\xda")
    app = b"\xff\xe1" + (65000).to_bytes(2, "big") + bytes(64998)       # pushes the SOS header behind the gathered head
    big_head = good[:2] + app + app + good[2:]
    files = [
        good,
        bytes(jpegsynth.encode(512, 512, "444", 75, 0, seed=3)),
        bytes(jpegsynth.encode(331, 177, "422", 80, 3, seed=9)),
        good + bytes(range(1, 200)),                                     # bytes behind EOI
        good[:-2] + b"\x5a\xff\xd9\xff\xfe\x00\x04ab",                  # one unread byte in front of the terminator + a segment behind EOI
        good[:-2] + b"\xff\xfe\x00\x04ab\xff\xd9",                      # COM behind the scan: full walk
        good[:sos + 14 + 200],                                           # truncated in the scan
        read_jpeg("progress.jpg"),
        read_jpeg("yellowcat_progressive_restart.jpg"),
        read_jpeg("lake.jpg"),
        bytes(jpegsynth.encode(96, 64, "444", 75, 0, seed=5, noninterleaved=True)),
        big_head,
        b"\xff\xd8",
        b"",
    ]
    return files


def test_files_handed_over_as_segment_lists_decode_like_contiguous_ones():
    """jpgpu_batch_upload_segments (ref: JpegDecoder.SetInput(ReadOnlySequence<byte>), JpegDecoder.cs:56-62; a multi-segment
    sequence: apps/JpegDecode/MemoryPoolBufferWriter.cs:166-174 with DecodeAction.cs:81-98's 16 KiB reads): every file cut
    at 16 KiB like the reference app reads it, at awkward places (inside markers, one-byte segments, empty segments), and
    whole -- same per-image status and the same samples as the contiguous upload, which equals the checker."""
    files = _variety()
    plain = jl.Batch().upload(files, jl.FMT_INTERLEAVED_U8).decode().sync()
    for cutter in (lambda d: list(range(16384, len(d), 16384)),
                   lambda d: [1, 2, 3, 4, 20, 21, len(d) // 3, len(d) // 3 + 1, len(d) - 2, len(d) - 1],
                   lambda d: []):
        segs = [_split(f, cutter(f)) for f in files]
        segs[1] = segs[1][:1] + [b""] + segs[1][1:]  # an empty segment in the middle of a file
        b = jl.Batch().upload_segments(segs, jl.FMT_INTERLEAVED_U8).decode().sync()
        st = b.ingest_stats()
        for i, f in enumerate(files):
            r0, r1 = plain.result(i), b.result(i)
            assert (r0.status, r0.detail) == (r1.status, r1.detail), (i, r0.status, r1.status)
            kind, ref = _oracle(f)
            assert NAMES.get(r1.status, str(r1.status)) == kind, (i, kind, r1.status)
            if ref is not None:
                assert np.array_equal(b.output(i), ref), i
        assert st["n_pinned_dma"] == 0
        b.close()
    plain.close()


def test_page_locked_input_goes_to_hbm_without_a_staging_copy():
    """JPGPU_UPLOAD_PINNED: files read into jpgpu_host_alloc'd memory (and a caller-owned array registered with
    jpgpu_host_register) are DMA'd from where they lie -- n_pinned_dma counts the copies -- and decode to the checker's samples;
    multi-segment pinned files too."""
    ctx = jl.Context(0)
    files = _variety()
    total = sum(len(f) for f in files)
    arena = ctx.host_alloc(total + 64 * len(files))
    views, pos = [], 0
    for f in files:
        arena[pos:pos + len(f)] = np.frombuffer(f, np.uint8)
        views.append(arena[pos:pos + len(f)])
        pos += (len(f) + 63) // 64 * 64
    b = jl.Batch(ctx).upload_segments(views, jl.FMT_INTERLEAVED_U8, pinned=True).decode().sync()
    st = b.ingest_stats()
    assert st["n_pinned_dma"] == sum(1 for f in files if len(f)), st
    for i, f in enumerate(files):
        kind, ref = _oracle(f)
        r = b.result(i)
        assert NAMES.get(r.status, str(r.status)) == kind, (i, kind, r.status)
        if ref is not None:
            assert np.array_equal(b.output(i), ref), i
    # JPGPU_UPLOAD_PINNED_ARENA: the device copy mirrors the arena (one DMA for this span) -- with marker look-alikes in the
    # gaps between the files, which nothing may interpret
    junk = np.frombuffer(b"\xff\xd9\xff\xda\xff\x00\xff\xd0\xff\xff\x00" * ((arena.size // 11) + 1), np.uint8)[:arena.size]
    arena[:] = junk
    pos = 0
    for f in files:
        arena[pos:pos + len(f)] = np.frombuffer(f, np.uint8)
        pos += (len(f) + 63) // 64 * 64
    b.upload_segments(views, jl.FMT_INTERLEAVED_U8, arena=True).decode().sync()
    st = b.ingest_stats()
    assert 1 <= st["n_pinned_dma"] <= 2, st
    for i, f in enumerate(files):
        kind, ref = _oracle(f)
        assert NAMES.get(b.result(i).status) == kind, i
        if ref is not None:
            assert np.array_equal(b.output(i), ref), i
    # the same file listed many times (a benchmark repeating a few images): the copies share their bytes in HBM
    rep = [views[1], views[9], views[1], views[1], views[9]]
    b.upload_segments(rep, jl.FMT_INTERLEAVED_U8, arena=True).decode().sync()
    for i, v in enumerate(rep):
        assert np.array_equal(b.output(i), po.decode_8bit(bytes(v))[0]), i
    # the same arena as two segments per file
    segs = [[v[:len(v) // 2], v[len(v) // 2:]] if len(v) > 4 else [v] for v in views]
    b.upload_segments(segs, jl.FMT_INTERLEAVED_U8, pinned=True).decode().sync()
    for i, f in enumerate(files):
        kind, ref = _oracle(f)
        assert NAMES.get(b.result(i).status) == kind, i
        if ref is not None:
            assert np.array_equal(b.output(i), ref), i
    b.close()
    # the caller's own array, page-locked in place
    mine = np.frombuffer(bytearray(files[1]), np.uint8)
    ctx.host_register(mine)
    b2 = jl.Batch(ctx).upload_segments([mine], jl.FMT_RGBA_U8, pinned=True).decode().sync()
    assert np.array_equal(b2.output(0), po.ycbcr8_to_rgb(po.decode_8bit(files[1])[0], rgba=True))
    b2.close()
    ctx.host_unregister(mine)
    ctx.host_free(arena)
    ctx.close()


def test_an_upload_right_after_an_unsynchronised_decode_waits_for_it():
    """uploads run on the upload stream and used to overwrite the inputs of a decode of the SAME batch that nobody
    had waited for.  Now the upload is ordered behind the batch's own device work: decode, re-upload other files at once,
    decode -- many times -- and the samples are the second file set's."""
    buf, sizes, stride = jpegsynth.encode_batch(12, 1920, 1080, "420", 90, 4, seed0=820, nthreads=8)
    files = [buf[i * stride:i * stride + int(sizes[i])] for i in range(12)]
    a, b = files[:6], files[6:]
    batch = jl.Batch().upload(a, jl.FMT_INTERLEAVED_U8)
    for _ in range(6):
        batch.decode()          # not waited for
        batch.upload(b, jl.FMT_INTERLEAVED_U8)
        batch.decode()          # not waited for either
        batch.upload(a, jl.FMT_INTERLEAVED_U8)
    batch.decode()
    batch.upload(b, jl.FMT_INTERLEAVED_U8)
    batch.decode().sync()
    for i in range(6):
        assert batch.result(i).status == 0
        assert np.array_equal(batch.output(i), po.decode_8bit(bytes(b[i]))[0]), i
    batch.close()


def test_multi_device_driver_config4_workload_pinned_and_overlapped():
    """BASELINE config 4's workload (1920x1080 4:2:0 Q90, DRI = 4) through jpgpu_multi_* over three slots (one MI355X here,
    listed three times): 66 images per call from page-locked memory (no staging copy), two calls in flight -- call 1 is
    uploaded while call 0 decodes -- every image of both calls against the checker; then the synchronous form from pageable
    memory gives the same bytes."""
    n = 66
    buf, sizes, stride = jpegsynth.encode_batch(2 * n, 1920, 1080, "420", 90, 4, seed0=5000, nthreads=8)
    m = jl.MultiDecoder([0, 0, 0])
    ctx0 = jl.Batch._borrowed(0, jl._capi.lib.jpgpu_multi_context(m._h, 0), 0).ctx
    arena = C_host_alloc(ctx0, int(stride) * 2 * n)
    arena[:] = buf[:arena.size]
    files = [arena[i * stride:i * stride + int(sizes[i])] for i in range(2 * n)]
    t0 = m.submit(files[:n], jl.FMT_INTERLEAVED_U8, pinned=True)
    t1 = m.submit(files[n:], jl.FMT_INTERLEAVED_U8, pinned=True)
    with pytest.raises(jl.JpegError):
        m.submit(files[:3])          # a third call in flight is refused
    m.wait(t0)
    for i in range(n):
        assert m.result_of(t0, i).status == 0
    st = m.shard_of(t0, 1).ingest_stats()
    assert st["n_pinned_dma"] == len(range(1, n, 3)) and st["n_header_only"] == st["n_pinned_dma"], st
    m.wait(t1)
    for i in range(0, n, 5):
        assert np.array_equal(m.output_of(t0, i), po.decode_8bit(bytes(files[i]))[0]), i
        assert np.array_equal(m.output_of(t1, i), po.decode_8bit(bytes(files[n + i]))[0]), i
    first = [m.output_of(t1, i) for i in (0, 1, 2, n - 1)]
    m.decode([bytes(f) for f in files[n:]])       # synchronous, pageable
    for k, i in enumerate((0, 1, 2, n - 1)):
        assert m.result(i).status == 0
        assert np.array_equal(m.output(i), first[k]), i
    import ctypes as C
    jl._capi.lib.jpgpu_host_free(ctx0._h, C.c_void_p(arena.ctypes.data))
    m.close()


def C_host_alloc(borrowed_ctx, nbytes):
    import ctypes as C

    p = C.c_void_p()
    assert jl._capi.lib.jpgpu_host_alloc(borrowed_ctx._h, nbytes, C.byref(p)) == 0
    return np.frombuffer((C.c_uint8 * nbytes).from_address(p.value), dtype=np.uint8)
