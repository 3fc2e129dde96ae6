"""Independent evidence for the rows of SURVEY 8f whose checkers the reference cannot pin (it holds no encoder vectors, no
optimizer bytes, no RGB golden): every restatement below is written HERE, a second or third time, from the
published procedure rather than from the product's or the checker's code, and held against both.

 (a) Huffman code lengths: Figures K.1 / K.3 of T.81 with the reference's selection rule, on an explicit tree; the cost of
     the length-limited ("MostOptimalCoding") code against a textbook package-merge in numpy -- cost and Kraft sum do not depend
     on how ties are broken, so they pin the builders without any reading of .NET's sort;
 (b) .NET's unstable Array.Sort: a third coding in Python, fuzzed against the product's (Span-based layout, .NET 5+) and
     the checker's (index-based layout, .NET Core 3.x) on 10^5 arrays full of ties;
 (c) the YCbCr -> RGB tables: built literally from apps/JpegDecode/JpegYCbCrToRgbConverter.cs:24-132 in numpy float32 /
     float64, all 2^24 (Y, Cb, Cr) triples against the checker's converter (which the GPU tests compare the fused writer with).
CPU only: the product functions used here are host code (jpgpu_build_optimal_huffman_table, jpgpu_net_sort_permutation).
"""
import ctypes as C
import heapq

import numpy as np
import pytest

import jpeglibrary_amd as jl
from jpeglibrary_amd import _capi
from oracle import pyoracle as po


# ------------------------------------------------------------------------------------------------ (a) code lengths

def k1_k3_bits(freq):
    """BITS[16] of JpegHuffmanEncodingTableBuilder.Build(false) (JpegHuffmanEncodingTableBuilder.cs:69-175), independently:
    Figure K.1 as an explicit tree (parent pointers; depth = code size) with the reference's selection rule -- V1 = the FIRST
    entry of least frequency, V2 = the first other entry of least frequency, the merged node stays in V1's place -- one extra
    symbol of frequency 1 reserving the all-ones code, then Figure K.3 as T.81 prints it."""
    f = [int(x) for x in freq if x] + [1]
    n = len(f)
    parent = list(range(n))                # parent pointers of the leaves' chain heads
    alive = list(range(n))                 # live tree roots, in the array order the reference scans
    weight = {i: f[i] for i in range(n)}
    members = {i: [i] for i in range(n)}   # leaves under each root
    depth = [0] * n
    while len(alive) > 1:
        v1 = min(alive, key=lambda i: weight[i])                       # min() returns the first minimum
        v2 = min((i for i in alive if i != v1), key=lambda i: weight[i])
        for leaf in members[v1] + members[v2]:
            depth[leaf] += 1
        weight[v1] += weight[v2]
        members[v1] += members[v2]
        alive.remove(v2)
    bits = [0] * 64
    for d in depth:
        if d:
            bits[d] += 1                                                # bits[i] = number of codes of length i (T.81's BITS(i))
    if max(bits) > 255:
        return None   # the reference counts in bytes (Span<byte>, :118): 256 codes of one length wrap to 0 and Build() throws
    i = max(32, max(depth))
    while True:                                                         # Figure K.3
        while bits[i] > 0:
            j = i - 1
            while True:
                j -= 1
                if bits[j] > 0:
                    break
            bits[i] -= 2
            bits[i - 1] += 1
            bits[j + 1] += 2
            bits[j] -= 1
        i -= 1
        if i == 16:
            break
    while bits[i] == 0:
        i -= 1
    bits[i] -= 1                                                        # the reserved code point leaves the table
    return bits[1:17]


def huffman_cost(weights):
    """Weighted path length of an (unrestricted) Huffman code: unique, however ties are broken."""
    h = list(weights)
    heapq.heapify(h)
    total = 0
    while len(h) > 1:
        a, b = heapq.heappop(h), heapq.heappop(h)
        total += a + b
        heapq.heappush(h, a + b)
    return total


def package_merge_lengths(weights, limit):
    """Textbook package-merge (Larmore / Hirschberg): optimal code lengths with every length <= limit.  Items carry a count
    vector over the symbols; level l = the leaves merged with the packages (pairs, in ascending weight) of level l + 1."""
    w = np.asarray(weights, dtype=np.int64)
    n = len(w)
    if n == 1:
        return np.array([1])
    order = np.argsort(w, kind="stable")
    leaves = [(int(w[i]), np.eye(n, dtype=np.int32)[i]) for i in order]
    level = list(leaves)
    for _ in range(limit - 1):
        packages = [(level[k][0] + level[k + 1][0], level[k][1] + level[k + 1][1]) for k in range(0, len(level) - 1, 2)]
        level = sorted(leaves + packages, key=lambda t: t[0])
    counts = sum(t[1] for t in level[:2 * n - 2])
    return counts


def _random_counts(rng, case):
    freq = np.zeros(256, np.uint32)
    kind = case % 5
    n = int(rng.integers(1, 257)) if kind != 4 else int(rng.integers(30, 60))
    symbols = rng.choice(256, n, replace=False)
    if kind == 0:        # small counts: ties everywhere
        freq[symbols] = rng.integers(1, 4, n)
    elif kind == 1:      # camera-like: heavy tail
        freq[symbols] = np.maximum(1, (rng.pareto(0.7, n) * 10 ** int(rng.integers(1, 8))).clip(1, 2 ** 31)).astype(np.uint32)
    elif kind == 2:      # all equal
        freq[symbols] = int(rng.integers(1, 1000))
    elif kind == 3:      # powers of two with repeats
        freq[symbols] = 2 ** rng.integers(0, 24, n)
    else:                # near-Fibonacci: unrestricted lengths far above 16, Figure K.3 / the limit at work
        fib = [1, 1]
        while len(fib) < n:
            fib.append(fib[-1] + fib[-2])
        freq[symbols] = np.minimum(np.array(fib[:n], dtype=np.int64) + rng.integers(0, 2, n), 2 ** 32 - 1).astype(np.uint32)
    return freq, symbols


def test_code_length_histograms_against_an_independent_figure_k1_k3():
    """1 000 random count vectors: BITS[16] of the product's and of the checker's Build(false) equal the independent
    Figure K.1 / K.3 above -- so only the ORDER of equal-length symbols is left to the reading of .NET's sort (b)."""
    rng = np.random.default_rng(2026)
    limited = overflowed = 0
    for case in range(1000):
        freq, symbols = _random_counts(rng, case)
        if case == 999:        # the one shape that overflows the reference's byte counters, in case the draw above missed it
            freq = np.zeros(256, np.uint32)
            freq[:255] = 502
            symbols = np.arange(255)
        want = k1_k3_bits(freq)
        if want is None:   # 255 symbols in a perfect tree of depth 8: IndexOutOfRangeException in the reference, failure here
            overflowed += 1
            with pytest.raises(jl.JpegError):
                jl.build_optimal_huffman_table(freq)
            with pytest.raises(po.OracleError):
                po.build_optimal_table(freq)
            continue
        bits_p, values_p, _, length_p = jl.build_optimal_huffman_table(freq)
        bits_c, values_c, _, length_c = po.build_optimal_table(freq)
        assert list(bits_p) == want, (case, list(bits_p), want)
        assert list(bits_c) == want, case
        assert sorted(values_p.tolist()) == sorted(symbols.tolist())
        # below the 16-bit limit the code is a Huffman code (with the reserved symbol of weight 1): its cost is THE minimum
        lens = length_p[symbols].astype(np.int64)
        unrestricted = huffman_cost([int(x) for x in freq[symbols]] + [1])
        reserved_len = max(l + 1 for l in range(16) if want[l])  # the reserved code point sits among the longest codes
        cost = int((freq[symbols].astype(np.int64) * lens).sum()) + reserved_len
        if cost != unrestricted:
            limited += 1
            assert cost > unrestricted and lens.max() == 16, case   # only Figure K.3's shortening may cost extra
    assert limited >= 20 and overflowed >= 1  # the near-Fibonacci cases did exercise the limit


def test_most_optimal_coding_is_optimal_among_16_bit_codes():
    """MostOptimalCoding (JpegHuffmanEncodingTableBuilder.cs:289-428): the product's and the checker's lengths cost exactly what
    the textbook package-merge says an optimal code limited to 16 bits costs (one extra symbol of frequency 0 takes the longest
    code, i.e. reserves the all-ones code point), satisfy Kraft with room for that code point, and never beat the bound."""
    rng = np.random.default_rng(77)
    for case in range(300):
        freq, symbols = _random_counts(rng, case)
        if len(symbols) < 2:
            continue
        bits_p, values_p, _, length_p = jl.build_optimal_huffman_table(freq, most_optimal=True)
        bits_c, values_c, _, length_c = po.build_optimal_table(freq, most_optimal=True)
        assert np.array_equal(bits_p, bits_c) and np.array_equal(values_p, values_c), case
        w = [int(x) for x in freq[symbols]] + [0]
        best = package_merge_lengths(w, 16)
        best_cost = int((np.array(w, dtype=np.int64) * best).sum())
        cost = int((freq[symbols].astype(np.int64) * length_p[symbols]).sum())
        assert cost == best_cost, (case, cost, best_cost)
        assert length_p[symbols].max() <= 16 and length_p[symbols].min() >= 1
        kraft = sum(int(bits_p[l]) * 2.0 ** -(l + 1) for l in range(16))
        assert kraft + 2.0 ** -int(length_p[symbols].max()) <= 1.0 + 1e-12, case   # room for the reserved longest code


# ------------------------------------------------------------------------------------------------ (b) Array.Sort

def net_sort_python(keys):
    """Third coding of ArraySortHelper<T>.IntrospectiveSort (comparison on the key alone), as a permutation: iterative, an
    explicit work stack instead of recursion, half-open ranges -- neither the product's nor the checker's shape."""
    n = len(keys)
    a = list(range(n))

    def less(i, j):          # comparer(x, y) < 0
        return keys[i] < keys[j]

    def swap_if_greater(i, j):
        if i != j and keys[a[i]] > keys[a[j]]:
            a[i], a[j] = a[j], a[i]

    def insertion(lo, hi):   # [lo, hi)
        for i in range(lo + 1, hi):
            t = a[i]
            j = i - 1
            while j >= lo and less(t, a[j]):
                a[j + 1] = a[j]
                j -= 1
            a[j + 1] = t

    def sift(lo, i, m):      # 1-based heap of m elements at a[lo:]
        d = a[lo + i - 1]
        while i <= m // 2:
            c = 2 * i
            if c < m and less(a[lo + c - 1], a[lo + c]):
                c += 1
            if not less(d, a[lo + c - 1]):
                break
            a[lo + i - 1] = a[lo + c - 1]
            i = c
        a[lo + i - 1] = d

    def heapsort(lo, hi):
        m = hi - lo
        for i in range(m // 2, 0, -1):
            sift(lo, i, m)
        for i in range(m, 1, -1):
            a[lo], a[lo + i - 1] = a[lo + i - 1], a[lo]
            sift(lo, 1, i - 1)

    if n < 2:
        return a
    depth0 = 2 * n.bit_length()          # 2 * (floor(log2 n) + 1)
    stack = [(0, n, depth0)]
    while stack:
        lo, hi, depth = stack.pop()
        while hi - lo > 1:
            size = hi - lo
            if size <= 16:
                if size == 2:
                    swap_if_greater(lo, lo + 1)
                elif size == 3:
                    swap_if_greater(lo, lo + 1)
                    swap_if_greater(lo, lo + 2)
                    swap_if_greater(lo + 1, lo + 2)
                else:
                    insertion(lo, hi)
                break
            if depth == 0:
                heapsort(lo, hi)
                break
            depth -= 1
            last = hi - 1
            mid = lo + (last - lo) // 2
            swap_if_greater(lo, mid)
            swap_if_greater(lo, last)
            swap_if_greater(mid, last)
            pivot = a[mid]
            a[mid], a[last - 1] = a[last - 1], a[mid]
            left, right = lo, last - 1
            while left < right:
                left += 1
                while less(a[left], pivot):
                    left += 1
                right -= 1
                while less(pivot, a[right]):
                    right -= 1
                if left >= right:
                    break
                a[left], a[right] = a[right], a[left]
            if left != last - 1:
                a[left], a[last - 1] = a[last - 1], a[left]
            # the reference recurses into the right part FIRST and then loops on the left one; the order of the two does
            # not matter for the result (disjoint ranges), the depth budget each gets does
            stack.append((left + 1, hi, depth))
            hi = left
    return a


def _perm(fn, keys):
    k = np.ascontiguousarray(keys, dtype=np.int32)
    out = np.empty(len(k), np.int32)
    fn(k.ctypes.data, len(k), out.ctypes.data)
    return out


def test_three_codings_of_the_unstable_sort_agree():
    lib = po.lib()
    lib.jref_net_sort_permutation.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    lib.jref_net_sort_permutation.restype = None
    prod = _capi.lib.jpgpu_net_sort_permutation
    rng = np.random.default_rng(5)
    unstable = 0
    for case in range(100000):
        n = int(rng.integers(0, 40)) if case % 4 else int(rng.integers(17, 600))
        spread = (2, 3, 5, 17, 1000)[case % 5]
        if case % 97 == 0:      # adversarial shapes for the depth limit / heapsort path: organ pipes, sorted, reversed
            keys = np.concatenate([np.arange(n // 2), np.arange(n - n // 2)[::-1]]).astype(np.int32)
        elif case % 89 == 0:
            keys = np.sort(rng.integers(0, spread, n)).astype(np.int32)[::-1].copy()
        else:
            keys = rng.integers(0, spread, n).astype(np.int32)
        p, c = _perm(prod, keys), _perm(lib.jref_net_sort_permutation, keys)
        assert np.array_equal(p, c), (case, n)
        assert sorted(p.tolist()) == list(range(n)) and np.all(np.diff(keys[p]) >= 0)
        if case % 25 == 0:
            assert net_sort_python(keys.tolist()) == p.tolist(), (case, n)
        if n and not np.array_equal(p, np.argsort(keys, kind="stable")):
            unstable += 1
    assert unstable > 10000  # the order really is the algorithm's, not a stable sort's


def test_median_of_three_killer_reaches_the_heapsort_fallback():
    """A sequence built to defeat median-of-three quicksort (Musser's construction) drives the depth budget to zero: the heapsort
    branch of all three codings is executed and agrees."""
    lib = po.lib()
    lib.jref_net_sort_permutation.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    lib.jref_net_sort_permutation.restype = None
    for n in (1 << 10, 3000, 1 << 12):
        m = n // 2 * 2
        k = m // 2
        keys = [0] * m
        for i in range(1, k + 1):          # Musser 1997: a[2i-1] = i for odd i..., the classic killer layout
            if i % 2 == 1:
                keys[i - 1] = i
                keys[i] = k + i
            keys[k + i - 1] = 2 * i
        keys = np.array(keys, dtype=np.int32) // 3   # and ties on top
        p = _perm(_capi.lib.jpgpu_net_sort_permutation, keys)
        assert np.array_equal(p, _perm(lib.jref_net_sort_permutation, keys))
        assert net_sort_python(keys.tolist()) == p.tolist()


# ------------------------------------------------------------------------------------------------ (c) YCbCr -> RGB

def converter_tables():
    """JpegYCbCrToRgbConverter's constructor + Init (apps/JpegDecode/JpegYCbCrToRgbConverter.cs:24-118), line by line: C#
    `float` arithmetic is numpy float32, Fix (:121-124) widens to double before adding 0.5, Code2V (:126-129) divides in float."""
    f32 = np.float32
    luma_red, luma_green, luma_blue = f32(299) / f32(1000), f32(587) / f32(1000), f32(114) / f32(1000)
    rbw = [f32(0), f32(255), f32(128), f32(255), f32(128), f32(255)]
    shift, one_half = 16, 1 << 15

    def fix(x):
        return int(np.float64(f32(x) * f32(1 << shift)) + 0.5)      # x * (1L << Shift): float * long -> float; + 0.5 -> double

    def code2v(c, rb, rw, cr):
        num = f32(f32(c - int(rb)) * f32(cr))
        den = f32(rw - rb) if int(f32(rw - rb)) != 0 else f32(1.0)
        return int(num / den)

    f1 = f32(f32(2) - f32(2) * luma_red)
    d1 = fix(f1)
    d2 = -fix(f32(f32(luma_red * f1) / luma_green))
    f3 = f32(f32(2) - f32(2) * luma_blue)
    d3 = fix(f3)
    d4 = -fix(f32(f32(luma_blue * f3) / luma_green))
    cr_r, cb_b, cr_g, cb_g, y_t = (np.zeros(256, np.int64) for _ in range(5))
    for i in range(256):
        x = i - 128
        cr = code2v(x, f32(rbw[4] - f32(128.0)), f32(rbw[5] - f32(128.0)), 127)
        cb = code2v(x, f32(rbw[2] - f32(128.0)), f32(rbw[3] - f32(128.0)), 127)
        cr_r[i] = (d1 * cr + one_half) >> shift
        cb_b[i] = (d3 * cb + one_half) >> shift
        cr_g[i] = d2 * cr
        cb_g[i] = d4 * cb + one_half
        y_t[i] = code2v(x + 128, rbw[0], rbw[1], 255)
    clamp = np.zeros(4 * 256, np.uint8)                 # :66-80: [256, 512) identity, [512, 1024) = 255, below 256 stays 0
    clamp[256:512] = np.arange(256)
    clamp[512:] = 255
    return cr_r, cb_b, cr_g, cb_g, y_t, clamp


def test_ycbcr_to_rgb_tables_from_the_reference_source_on_every_triple():
    cr_r, cb_b, cr_g, cb_g, y_t, clamp = converter_tables()
    # the tables in closed form, in float64: Cr * (2 - 2 * 0.299) etc. rounded through 16-bit fixed point
    for i in (0, 1, 64, 127, 128, 129, 200, 255):
        x = i - 128
        assert abs(int(cr_r[i]) - x * 1.402) <= 1 and abs(int(cb_b[i]) - x * 1.772) <= 1
        assert abs(((int(cb_g[i]) + int(cr_g[128])) >> 16) - (-0.344136 * x)) <= 1
    assert np.array_equal(y_t, np.arange(256))
    cb, cr = np.meshgrid(np.arange(256), np.arange(256), indexing="ij")
    cb, cr = cb.reshape(-1), cr.reshape(-1)
    g_term = (cb_g[cb] + cr_g[cr]) >> 16
    for y in range(256):
        ycc = np.stack([np.full(65536, y), cb, cr], axis=1).astype(np.uint8).reshape(256, 256, 3)
        want = np.stack([clamp[256 + y_t[y] + cr_r[cr]], clamp[256 + y_t[y] + g_term], clamp[256 + y_t[y] + cb_b[cb]]], axis=1).reshape(256, 256, 3)
        got = po.ycbcr8_to_rgb(ycc)
        assert np.array_equal(got, want), y
        if y % 64 == 0:
            rgba = po.ycbcr8_to_rgb(ycc, rgba=True)
            assert np.array_equal(rgba[..., :3], want) and np.all(rgba[..., 3] == 255)


@pytest.mark.gpu
def test_fused_rgb_writer_on_every_chroma_pair():
    """The same tables against the GPU's fused conversion: a 4:4:4 image whose Cb / Cr planes sweep all 65 536 pairs at a few
    luma levels (quality 100, so the decoded samples still cover the range); the writer must equal the numpy converter applied
    to the decoder's own YCbCr8 samples."""
    from tools import jpegsynth  # noqa: F401  (the synthetic encoder is RGB-in; build YCbCr directly through Pillow instead)
    import io

    from PIL import Image

    cr_r, cb_b, cr_g, cb_g, y_t, clamp = converter_tables()
    cbg, crg = np.meshgrid(np.arange(256), np.arange(256), indexing="ij")
    files = []
    for y in (0, 37, 128, 201, 255):
        ycc = np.stack([np.full((256, 256), y), cbg, crg], axis=-1).astype(np.uint8)
        buf = io.BytesIO()
        Image.fromarray(ycc, mode="YCbCr").save(buf, format="JPEG", quality=100, subsampling=0)
        files.append(buf.getvalue())
    ycc_out, res = jl.decode_batch(files, jl.FMT_INTERLEAVED_U8)
    rgb_out, res2 = jl.decode_batch(files, jl.FMT_RGB_U8)
    for a, b in zip(ycc_out, rgb_out):
        yy, cb, cr = (a[..., k].astype(np.int64) for k in range(3))
        want = np.stack([clamp[256 + y_t[yy] + cr_r[cr]], clamp[256 + y_t[yy] + ((cb_g[cb] + cr_g[cr]) >> 16)], clamp[256 + y_t[yy] + cb_b[cb]]], axis=-1)
        assert np.array_equal(b, want)
