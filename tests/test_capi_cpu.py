"""CPU-side checks of the product library: it loads, exports every symbol include/jpgpu.h declares, and refuses to
run without a GPU (no CPU fallback).  No compute calls here."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "jpgpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set(re.findall(r"\b(jpgpu_[a-z0-9_]+)\s*\(", text))
    names -= {"jpgpu_write_block_fn"}
    return sorted(names)


def test_library_exports_every_declared_symbol():
    from jpeglibrary_amd import _capi

    declared = _declared_symbols()
    assert len(declared) >= 45
    bound = {name for name, _, _ in _capi.SYMBOLS}
    assert set(declared) == bound, (set(declared) ^ bound)
    lib = C.CDLL(_capi.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name


def test_version_and_strings():
    from jpeglibrary_amd import _capi

    assert _capi.lib.jpgpu_version() == 101
    # the struct every *_result call fills in whole: the binding's mirror must have the library's size
    import ctypes
    assert _capi.lib.jpgpu_sizeof_image_result() == ctypes.sizeof(_capi.ImageResult) == 28
    assert _capi.lib.jpgpu_status_string(1) == b"InvalidDataException"
    assert _capi.lib.jpgpu_detail_string(4) == b"Expect restart marker."


def test_shard_rule_is_image_i_to_gpu_i_mod_g():
    """jpgpu_shard (include/jpgpu.h) and its Python mirror: disjoint, complete, round-robin."""
    from jpeglibrary_amd import _capi, sharding

    for n, world in [(0, 1), (1, 8), (7, 8), (8, 8), (8192, 8), (1000, 3)]:
        seen = []
        for rank in range(world):
            first, stride, count = C.c_int(), C.c_int(), C.c_int()
            _capi.lib.jpgpu_shard(n, rank, world, C.byref(first), C.byref(stride), C.byref(count))
            mine = [first.value + k * stride.value for k in range(count.value)]
            assert mine == list(range(rank, n, world)) == sharding.shard_indices(n, rank, world)
            seen += mine
        assert sorted(seen) == list(range(n))


def test_no_cpu_fallback_without_gpu():
    import jpeglibrary_amd as jl

    if jl.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(jl.NoDeviceError):
        jl.Context(0)
    with pytest.raises(jl.NoDeviceError):
        jl.JpegDecoder()
    with pytest.raises(jl.NoDeviceError):
        jl.MultiDecoder([0, 1])  # the multi-device driver creates its contexts up front: no device, no driver
    with pytest.raises(jl.NoDeviceError):
        jl.EncodeBatch()
    # host-only decoder: parsing works, Decode() refuses (scans run on the GPU only)
    import numpy as np
    from golden_util import read_jpeg

    d = jl.JpegDecoder(host_only=True)
    d.SetInput(read_jpeg("cramps.jpg"))
    d.Identify()
    d.SetOutputWriter(jl.JpegBufferOutputWriter8Bit(d.Width, d.Height, 1, np.zeros(d.Width * d.Height, np.uint8)))
    with pytest.raises(jl.NoDeviceError):
        d.Decode()


def test_product_does_not_import_oracle():
    """The product package must never reach into oracle/ (test infrastructure)."""
    pkg = os.path.join(ROOT, "jpeglibrary_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip")) or f == "Makefile":
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle" not in src.lower().replace("no cpu fallback", ""), os.path.join(dirpath, f)
