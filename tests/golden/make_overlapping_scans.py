#!/usr/bin/env python3
"""Makes tests/golden/stress/baseline_overlapping_scans_*.jpg: baseline (SOF0) frames in which a LATER scan writes a component an
EARLIER scan has written too -- no encoder writes such a frame; a corrupted component selector, or a file spliced from two, does.
The reference decodes scan after scan into the one writer, so the later WriteBlock wins wherever the later scan got to
(JpegHuffmanBaselineScanDecoder.cs:99-134) and the earlier scan's samples stay behind its last block.

  a  interleaved Y Cb Cr scan, then a scan of Y alone with other content          (a later scan covers SOME components)
  b  the same with the Y scan cut short in the middle of its data                  (it fails: the writer keeps scan 1 behind that block)
  c  three single-component scans whose third selects Cb again, cut short          (Cb twice, the second time partly; Cr never)

Run from the repository root:  python3 tests/golden/make_overlapping_scans.py   (uses tools/jpegsynth, the tree's own encoder)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tools import jpegsynth  # noqa: E402


def segments(data):
    """[(marker, start, end)] of the marker segments; an SOS segment runs up to the next marker that is not RSTn."""
    out, i = [], 2
    while i + 4 <= len(data):
        assert data[i] == 0xFF, i
        m = data[i + 1]
        if m == 0xD9:
            out.append((m, i, i + 2))
            break
        ln = (data[i + 2] << 8) | data[i + 3]
        j = i + 2 + ln
        if m == 0xDA:
            while not (data[j] == 0xFF and data[j + 1] not in (0x00, 0xFF) and not 0xD0 <= data[j + 1] <= 0xD7):
                j += 1
        out.append((m, i, j))
        i = j
    return out


def scans(data):
    return [data[a:b] for m, a, b in segments(data) if m == 0xDA]


def main():
    d = os.path.join(ROOT, "tests", "golden", "stress")
    w, h, q, dri = 152, 104, 70, 3
    inter = jpegsynth.encode(w, h, "444", q, dri, seed=41)
    other = jpegsynth.encode(w, h, "444", q, dri, seed=42, noninterleaved=True)
    y2 = scans(other)[0]
    assert inter[-2:] == b"\xff\xd9"
    a = inter[:-2] + y2 + b"\xff\xd9"
    cut = y2[:len(y2) * 3 // 5]
    while cut[-1] == 0xFF:  # (not in the middle of a marker or a stuffed byte)
        cut = cut[:-1]
    b = inter[:-2] + cut + b"\xff\xd9"
    non = bytearray(jpegsynth.encode(w, h, "444", q, 0, seed=43, noninterleaved=True))
    segs = [s for s in segments(bytes(non)) if s[0] == 0xDA]
    third = segs[2]
    assert non[third[1] + 4] == 1 and non[third[1] + 5] == 3  # one component, selector 3 (Cr)
    non[third[1] + 5] = 2  # ... now Cb again
    c = bytes(non[:third[1] + (third[2] - third[1]) * 2 // 3])
    while c[-1] == 0xFF:
        c = c[:-1]
    c += b"\xff\xd9"
    for name, blob in (("a_later_scan_of_one_component", a), ("b_later_scan_cut_short", b), ("c_component_twice_cut_short", c)):
        with open(os.path.join(d, f"baseline_overlapping_scans_{name}.jpg"), "wb") as f:
            f.write(blob)
        print(name, len(blob))


if __name__ == "__main__":
    main()
