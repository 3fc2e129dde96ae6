#!/usr/bin/env python3
"""tools/trace/dri0_decodes.py WORKLOAD IMAGES DECODES -- upload one batch, decode it DECODES times with ONE wait at the very end.
Run under `rocprofv3 --hip-trace --stats` twice (tools/trace/dri0_hip_trace.sh): the difference of the two runs' HIP call counts,
divided by the difference in DECODES, is what ONE jpgpu_batch_decode issues -- launches, async copies, fills; and the calls that
would make the host wait (hipStreamSynchronize, hipDeviceSynchronize, hipEventSynchronize, synchronous hipMemcpy): none."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    workload, images, decodes = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    import bench
    import jpeglibrary_amd as jl
    from jpeglibrary_amd import sharding
    from tools import jpegsynth

    files, _, _, _ = bench.make_inputs(sharding, jpegsynth, workload, images, 0, 8)
    b = jl.Batch()
    b.upload(files)
    b.decode()
    b.sync()       # first decode of the upload: enqueues the full budget of rounds and learns how many it takes
    b.decode()
    b.sync()       # steady state from here on
    for _ in range(decodes):
        b.decode()
    b.sync()
    assert all(b.result(i).status == 0 for i in range(images))
    print("decodes", decodes, "subseq_rounds", b.subseq_rounds(), "fallbacks", b.subseq_fallbacks())
    b.close()


if __name__ == "__main__":
    main()
