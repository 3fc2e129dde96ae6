"""E1 (fdct_fused_kernel<H, V> or, with JPGPU_ENC_NO_FUSED=1, E1a + E1b) per sampling shape: 64 x 4K RGB, stage times.
    python3 tools/trace/e1_shapes.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import jpeglibrary_amd as jl
from oracle import pyoracle as po
from test_gpu_parity import _enc_image

imgs = [_enc_image(3840, 2160, s) for s in range(4)] * 16
for (H, V) in [(2, 2), (2, 1), (1, 1)]:
    b = jl.EncodeBatch().upload(imgs, (H, V), 75, rgb=True)
    for _ in range(3):
        b.encode()
    t = time.perf_counter()
    for _ in range(5):
        b.encode()
    dt = (time.perf_counter() - t) / 5
    st = b.stage_ms()
    print("luma %d x %d, 64 x 4K: %.2f ms = %.0f Mpx/s; E1 %.3f E2 %.3f E3 %.3f E4 %.3f ms; byte-exact %s" % (
        H, V, dt * 1e3, 64 * 3840 * 2160 / dt / 1e6, st["fdct_quant"], st["block_bits"], st["emit"], st["stuff"],
        b.output(0) == po.encode_8bit(po.rgb_to_ycbcr8(imgs[0]), H, V, 75)))
