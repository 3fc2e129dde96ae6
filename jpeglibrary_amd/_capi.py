"""ctypes binding of libjpgpu.so (the C ABI declared in include/jpgpu.h).

The product has NO CPU fallback: importing this module loads the in-tree HIP library and raises if it is
missing; creating a context raises if no MI355X is visible.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libjpgpu.so")

OK, ERR_INVALID_DATA, ERR_INVALID_OPERATION, ERR_NOT_SUPPORTED, ERR_ARGUMENT, ERR_DEVICE, ERR_NO_DEVICE, ERR_OOM = range(8)
FMT_INTERLEAVED_U8, FMT_PLANAR_U8, FMT_PLANAR_I16, FMT_RGB_U8, FMT_RGBA_U8, FMT_EXTENDED_U16 = 0, 1, 2, 3, 4, 5

DETAIL_NAMES = {0: "NONE", 1: "INVALID_HUFFMAN_CODE", 2: "MARKER_IN_DATA", 3: "STREAM_ENDED", 4: "EXPECT_RESTART",
                5: "MISSING_TABLE", 6: "UNSUPPORTED_FRAME", 7: "BAD_HEADER", 8: "EARLY_EOI", 9: "UNEXPECTED_END"}


class FrameComponent(C.Structure):
    _fields_ = [("identifier", C.c_uint8), ("h", C.c_uint8), ("v", C.c_uint8), ("tq", C.c_uint8)]


class Frame(C.Structure):
    _fields_ = [("width", C.c_uint16), ("height", C.c_uint16), ("precision", C.c_uint8), ("num_components", C.c_uint8),
                ("sof", C.c_uint8), ("reserved", C.c_uint8), ("comp", FrameComponent * 4)]


class ScanComponent(C.Structure):
    _fields_ = [("selector", C.c_uint8), ("td", C.c_uint8), ("ta", C.c_uint8), ("reserved", C.c_uint8)]


class Scan(C.Structure):
    _fields_ = [("num_components", C.c_uint8), ("ss", C.c_uint8), ("se", C.c_uint8), ("ah", C.c_uint8), ("al", C.c_uint8),
                ("reserved", C.c_uint8 * 3), ("comp", ScanComponent * 4)]


class Dht(C.Structure):
    _fields_ = [("present", C.c_uint8), ("bits", C.c_uint8 * 16), ("num_values_minus_0", C.c_uint8),
                ("num_values", C.c_uint16), ("values", C.c_uint8 * 256)]


class PlaneInfo(C.Structure):
    _fields_ = [("offset", C.c_uint64), ("width", C.c_uint32), ("height", C.c_uint32), ("pitch", C.c_uint32)]


class ImageInfo(C.Structure):
    _fields_ = [("status", C.c_int32), ("detail", C.c_int32), ("width", C.c_uint16), ("height", C.c_uint16),
                ("precision", C.c_uint8), ("num_components", C.c_uint8), ("sof", C.c_uint8), ("reserved", C.c_uint8),
                ("restart_interval", C.c_uint32), ("mcus_per_line", C.c_uint32), ("mcus_per_column", C.c_uint32),
                ("blocks_per_mcu", C.c_uint32), ("total_blocks", C.c_uint64), ("out_offset", C.c_uint64),
                ("out_bytes", C.c_uint64), ("coef_offset", C.c_uint64), ("plane", PlaneInfo * 4)]


class ImageResult(C.Structure):
    _fields_ = [("status", C.c_int32), ("detail", C.c_int32), ("error_interval", C.c_uint32), ("decoded_mcus", C.c_uint32),
                ("bytes_consumed", C.c_uint32), ("terminator", C.c_uint32), ("error_block", C.c_uint32)]


class EncodeParams(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("components", C.c_int32), ("luma_h", C.c_int32), ("luma_v", C.c_int32),
                ("quality", C.c_int32), ("input_rgb", C.c_int32), ("optimize_coding", C.c_int32), ("restart_interval", C.c_int32)]


class IngestStats(C.Structure):
    _fields_ = [("threads", C.c_int32), ("n_header_only", C.c_int32), ("n_full_walk", C.c_int32), ("parse_ms", C.c_float),
                ("copy_ms", C.c_float), ("full_walk_ms", C.c_float), ("layout_ms", C.c_float), ("total_ms", C.c_float),
                ("n_pinned_dma", C.c_int32), ("n_linearised", C.c_int32)]


class Segment(C.Structure):
    _fields_ = [("data", C.c_void_p), ("len", C.c_size_t)]


UPLOAD_PINNED = 1
UPLOAD_PINNED_ARENA = 2


WRITE_BLOCK_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_int16), C.c_int, C.c_int, C.c_int)

# every symbol include/jpgpu.h declares: (name, restype, argtypes)
_P = C.c_void_p
SYMBOLS = [
    ("jpgpu_version", C.c_int, []),
    ("jpgpu_sizeof_image_result", C.c_size_t, []),
    ("jpgpu_device_count", C.c_int, []),
    ("jpgpu_create", C.c_int, [C.c_int, C.POINTER(_P)]),
    ("jpgpu_destroy", None, [_P]),
    ("jpgpu_last_error", C.c_char_p, [_P]),
    ("jpgpu_set_host_threads", C.c_int, [_P, C.c_int]),
    ("jpgpu_host_alloc", C.c_int, [_P, C.c_size_t, C.POINTER(C.c_void_p)]),
    ("jpgpu_host_free", C.c_int, [_P, C.c_void_p]),
    ("jpgpu_host_register", C.c_int, [_P, C.c_void_p, C.c_size_t]),
    ("jpgpu_host_unregister", C.c_int, [_P, C.c_void_p]),
    ("jpgpu_shard", None, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("jpgpu_multi_create", C.c_int, [C.POINTER(C.c_int), C.c_int, C.POINTER(_P)]),
    ("jpgpu_multi_destroy", None, [_P]),
    ("jpgpu_multi_devices", C.c_int, [_P]),
    ("jpgpu_multi_last_error", C.c_char_p, [_P]),
    ("jpgpu_multi_decode", C.c_int, [_P, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_int, C.c_int, C.POINTER(C.c_double),
                                     C.POINTER(C.c_double)]),
    ("jpgpu_multi_submit", C.c_int, [_P, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_int, C.c_int, C.c_uint, C.POINTER(C.c_int)]),
    ("jpgpu_multi_wait", C.c_int, [_P, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    ("jpgpu_multi_batch_of", _P, [_P, C.c_int, C.c_int]),
    ("jpgpu_multi_locate", C.c_int, [_P, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("jpgpu_multi_batch", _P, [_P, C.c_int]),
    ("jpgpu_multi_context", _P, [_P, C.c_int]),
    ("jpgpu_batch_ingest_stats", C.c_int, [_P, C.POINTER(IngestStats)]),
    ("jpgpu_batch_progressive_fallbacks", C.c_int, [_P]),
    ("jpgpu_status_string", C.c_char_p, [C.c_int]),
    ("jpgpu_detail_string", C.c_char_p, [C.c_int]),
    ("jpgpu_batch_create", C.c_int, [_P, C.POINTER(_P)]),
    ("jpgpu_batch_destroy", None, [_P]),
    ("jpgpu_batch_upload", C.c_int, [_P, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_int, C.c_int]),
    ("jpgpu_batch_upload_segments", C.c_int, [_P, C.POINTER(Segment), C.POINTER(C.c_int), C.c_int, C.c_int, C.c_uint]),
    ("jpgpu_batch_upload_frames", C.c_int, [_P, C.POINTER(Frame), C.c_void_p, C.c_int, C.c_int]),
    ("jpgpu_batch_decode", C.c_int, [_P]),
    ("jpgpu_batch_run_entropy", C.c_int, [_P]),
    ("jpgpu_batch_run_idct", C.c_int, [_P]),
    ("jpgpu_batch_sync", C.c_int, [_P]),
    ("jpgpu_batch_size", C.c_int, [_P]),
    ("jpgpu_batch_image_info", C.c_int, [_P, C.c_int, C.POINTER(ImageInfo)]),
    ("jpgpu_batch_result", C.c_int, [_P, C.c_int, C.POINTER(ImageResult)]),
    ("jpgpu_batch_output_device", C.c_void_p, [_P, C.POINTER(C.c_uint64)]),
    ("jpgpu_batch_coefficients_device", C.c_void_p, [_P, C.POINTER(C.c_uint64)]),
    ("jpgpu_batch_download_output", C.c_int, [_P, C.c_int, C.c_void_p, C.c_size_t]),
    ("jpgpu_batch_download_coefficients", C.c_int, [_P, C.c_int, C.c_void_p, C.c_size_t]),
    ("jpgpu_batch_upload_coefficients", C.c_int, [_P, C.c_int, C.c_void_p, C.c_size_t]),
    ("jpgpu_batch_stage_ms", C.c_int, [_P, C.POINTER(C.c_float)]),
    ("jpgpu_batch_subseq_rounds", C.c_int, [_P]),
    ("jpgpu_batch_subseq_fallbacks", C.c_int, [_P]),
    ("jpgpu_batch_set_partial_flush", C.c_int, [_P, C.c_int]),
    ("jpgpu_batch_progressive_replays", C.c_int, [_P]),
    ("jpgpu_batch_marker_fallbacks", C.c_int, [_P]),
    ("jpgpu_batch_totals", C.c_int, [_P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    ("jpgpu_decode_scan", C.c_int, [_P, C.POINTER(Frame), C.POINTER(Scan), C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint16,
                                    C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(ImageResult),
                                    C.POINTER(C.c_size_t)]),
    ("jpgpu_progressive_begin", C.c_int, [_P, C.POINTER(Frame), C.POINTER(_P)]),
    ("jpgpu_progressive_scan", C.c_int, [_P, C.POINTER(Scan), C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint16, C.c_void_p, C.c_size_t,
                                         C.POINTER(ImageResult), C.POINTER(C.c_size_t)]),
    ("jpgpu_progressive_output_size", C.c_int, [_P, C.c_int, C.POINTER(C.c_size_t)]),
    ("jpgpu_progressive_dispose", C.c_int, [_P, C.c_int, C.c_void_p, C.c_size_t]),
    ("jpgpu_progressive_dispose_to_writer", C.c_int, [_P, C.c_void_p, C.c_void_p]),
    ("jpgpu_progressive_destroy", None, [_P]),
    ("jpgpu_decoder_set_start_of_frame", C.c_int, [_P, C.c_int]),
    ("jpgpu_decoder_set_frame_header", C.c_int, [_P, C.POINTER(Frame)]),
    ("jpgpu_decoder_set_huffman_table", C.c_int, [_P, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]),
    ("jpgpu_decoder_set_quantization_table", C.c_int, [_P, C.c_int, C.c_int, C.c_void_p]),
    ("jpgpu_decoder_clear_huffman_table", C.c_int, [_P]),
    ("jpgpu_decoder_clear_quantization_table", C.c_int, [_P]),
    ("jpgpu_decoder_process_scan", C.c_int, [_P, C.POINTER(Scan), C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    ("jpgpu_decoder_create", C.c_int, [_P, C.POINTER(_P)]),
    ("jpgpu_decoder_destroy", None, [_P]),
    ("jpgpu_decoder_last_error", C.c_char_p, [_P]),
    ("jpgpu_decoder_set_input", C.c_int, [_P, C.c_void_p, C.c_size_t]),
    ("jpgpu_decoder_identify", C.c_int, [_P, C.c_int, C.POINTER(C.c_int)]),
    ("jpgpu_decoder_try_estimate_quality", C.c_int, [_P, C.POINTER(C.c_float)]),
    ("jpgpu_decoder_width", C.c_int, [_P]),
    ("jpgpu_decoder_height", C.c_int, [_P]),
    ("jpgpu_decoder_precision", C.c_int, [_P]),
    ("jpgpu_decoder_number_of_components", C.c_int, [_P]),
    ("jpgpu_decoder_start_of_frame", C.c_int, [_P]),
    ("jpgpu_decoder_get_maximum_horizontal_sampling", C.c_int, [_P]),
    ("jpgpu_decoder_get_maximum_vertical_sampling", C.c_int, [_P]),
    ("jpgpu_decoder_get_horizontal_sampling", C.c_int, [_P, C.c_int]),
    ("jpgpu_decoder_get_vertical_sampling", C.c_int, [_P, C.c_int]),
    ("jpgpu_decoder_get_restart_interval", C.c_int, [_P]),
    ("jpgpu_decoder_set_restart_interval", C.c_int, [_P, C.c_int]),
    ("jpgpu_decoder_load_tables", C.c_int, [_P, C.c_void_p, C.c_size_t]),
    ("jpgpu_decoder_set_output_writer", C.c_int, [_P, C.c_void_p, C.c_void_p]),
    ("jpgpu_decoder_set_output_buffer8", C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t]),
    ("jpgpu_decoder_decode", C.c_int, [_P]),
    ("jpgpu_decoder_reset", None, [_P]),
    ("jpgpu_decoder_reset_input", None, [_P]),
    ("jpgpu_decoder_reset_header", None, [_P]),
    ("jpgpu_decoder_reset_tables", None, [_P]),
    ("jpgpu_decoder_reset_output_writer", None, [_P]),
    ("jpgpu_encoder_create", C.c_int, [_P, C.POINTER(_P)]),
    ("jpgpu_encoder_destroy", None, [_P]),
    ("jpgpu_encoder_upload", C.c_int, [_P, C.POINTER(C.c_void_p), C.POINTER(EncodeParams), C.c_int]),
    ("jpgpu_encoder_set_quantization_table", C.c_int, [_P, C.c_int, C.c_int, C.c_void_p]),
    ("jpgpu_encoder_encode", C.c_int, [_P]),
    ("jpgpu_encoder_stage_ms", C.c_int, [_P, C.POINTER(C.c_float)]),
    ("jpgpu_encoder_emit_passes", C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("jpgpu_encoder_encoded_size", C.c_int, [_P, C.c_int, C.POINTER(C.c_size_t)]),
    ("jpgpu_encoder_download", C.c_int, [_P, C.c_int, C.c_void_p, C.c_size_t]),
    ("jpgpu_encoder_output_device", C.c_void_p, [_P, C.c_int, C.POINTER(C.c_size_t)]),
    ("jpgpu_encoder_download_coefficients", C.c_int, [_P, C.c_int, C.c_void_p, C.c_size_t]),
    ("jpgpu_optimizer_create", C.c_int, [_P, C.POINTER(_P)]),
    ("jpgpu_optimizer_destroy", None, [_P]),
    ("jpgpu_optimizer_upload", C.c_int, [_P, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_int, C.c_int]),
    ("jpgpu_optimizer_run", C.c_int, [_P]),
    ("jpgpu_optimizer_result", C.c_int, [_P, C.c_int, C.POINTER(ImageResult), C.POINTER(C.c_size_t)]),
    ("jpgpu_optimizer_download", C.c_int, [_P, C.c_int, C.c_void_p, C.c_size_t]),
    ("jpgpu_optimizer_statistics", C.c_int, [_P, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("jpgpu_optimizer_last_ms", C.c_int, [_P, C.POINTER(C.c_float)]),
    ("jpgpu_net_sort_permutation", C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    ("jpgpu_build_optimal_huffman_table", C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.c_void_p, C.c_void_p]),
    ("jpgpu_optimizer_set_most_optimal_coding", C.c_int, [_P, C.c_int]),
]


def _preload_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so (SONAME libamdhip64.so.7) next to libtorch_hip.so, and two
    HIP runtimes in one process cannot both own the GPU.  libjpgpu.so has a plain NEEDED libamdhip64.so.7, so loading
    torch's copy first makes the dynamic linker bind libjpgpu.so to it (SONAME match): one runtime per process, and
    device pointers / streams interoperate with torch.  Without torch installed the system ROCm runtime is used.
    Set JPGPU_HIP_RUNTIME=system to skip the preload."""
    if os.environ.get("JPGPU_HIP_RUNTIME", "") == "system":
        return None
    import importlib.util

    try:
        spec = importlib.util.find_spec("torch")
    except Exception:
        spec = None
    if not spec or not spec.origin:
        return None
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if not os.path.exists(cand):
        return None
    try:
        return C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except OSError:
        return None


def _load():
    _preload_hip_runtime()
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(jpeglibrary_amd has no CPU fallback; the HIP library is the product)")
    lib = C.CDLL(LIB_PATH)
    for name, restype, argtypes in SYMBOLS:
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = restype
        fn.argtypes = argtypes
    # the library fills jpgpu_image_result in whole: a mirror of another size would be overrun
    if lib.jpgpu_version() < 101 or lib.jpgpu_sizeof_image_result() != C.sizeof(ImageResult):
        raise ImportError(f"{LIB_PATH} (version {lib.jpgpu_version()}, jpgpu_image_result of {lib.jpgpu_sizeof_image_result()} bytes) does not "
                          f"match this binding (101, {C.sizeof(ImageResult)} bytes): rebuild it")
    return lib


lib = _load()
