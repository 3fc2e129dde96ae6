"""One file through JpegDecoder.Decode() (the caller's buffer as canvas) and through the batch, against the restatement's writer buffer."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import jpeglibrary_amd as jl  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

f = open(sys.argv[1], "rb").read()
px, info, err = po.decode_8bit_partial(f)
print("oracle:", err, px.shape)
d = jl.JpegDecoder()
d.SetInput(f)
d.Identify()
buf = np.zeros(d.Width * d.Height * d.NumberOfComponents, np.uint8)
d.SetOutputWriter(jl.JpegBufferOutputWriter8Bit(d.Width, d.Height, d.NumberOfComponents, buf))
try:
    d.Decode()
    print("mirror: OK")
except jl.JpegError as e:
    print("mirror:", type(e).__name__, e)
got = buf.reshape(px.shape)
diff = np.argwhere((got != px).any(axis=2))
print("mirror diff pixels:", len(diff), diff[:5].tolist(), diff[-5:].tolist())
for c in range(px.shape[2]):
    dc = np.argwhere(got[..., c] != px[..., c])
    print(" comp", c, len(dc), dc[:3].tolist(), dc[-3:].tolist())
outs, res = jl.decode_batch([f])
dd = np.argwhere((np.asarray(outs[0]) != px).any(axis=2))
print("batch status", res[0].status, res[0].detail, "diff pixels:", len(dd), dd[:5].tolist())
for r in (80, 87, 88, 90, 93):
    print("row", r, "oracle", px[r, ::32].tolist(), "mirror", got[r, ::32].tolist())
