"""jpeglibrary_amd -- MI355X-native baseline-JPEG decode path behind the JpegLibrary decoder API.

Importing the package loads the in-tree HIP library (libjpgpu.so); it raises if the library is missing.
There is no CPU fallback: creating a context without a GPU raises NoDeviceError.
"""
from . import _capi  # noqa: F401  (loads libjpgpu.so, fails loudly when absent)
from . import sharding  # noqa: F401
from .batch import FMT_EXTENDED_U16, FMT_INTERLEAVED_U8, FMT_PLANAR_I16, FMT_PLANAR_U8, FMT_RGB_U8, FMT_RGBA_U8, Batch, decode_batch
from .context import Context, default_context, device_count
from .encoder import EncodeBatch, encode_batch
from .optimizer import JpegOptimizer, OptimizeBatch, build_optimal_huffman_table, optimize_batch
from .decoder import (JpegBlockOutputWriter, JpegBufferOutputWriter8Bit, JpegDecoder, JpegExtendingOutputWriter, JpegFrameComponentSpecificationParameters,
                      JpegFrameHeader, JpegGpuProgressiveScanDecoder, JpegHuffmanDecodingTable, JpegScanComponentSpecificationParameters, JpegScanHeader)
from .jpeg_encoder import (JpegBufferInputReader, JpegEncoder, JpegHuffmanEncodingTable, JpegQuantizationTable, JpegStandardHuffmanEncodingTable,
                           JpegStandardQuantizationTable)
from .multi import MultiDecoder
from .errors import (ArgumentException, DeviceError, InvalidDataException, InvalidOperationException, JpegError,
                     NoDeviceError, NotSupportedException)

__all__ = [
    "Batch", "decode_batch", "MultiDecoder", "JpegEncoder", "JpegQuantizationTable", "JpegStandardQuantizationTable", "JpegHuffmanEncodingTable",
    "JpegStandardHuffmanEncodingTable", "JpegBufferInputReader", "EncodeBatch", "encode_batch", "JpegOptimizer", "OptimizeBatch", "optimize_batch", "build_optimal_huffman_table", "Context", "default_context", "device_count", "JpegDecoder", "JpegBlockOutputWriter",
    "JpegBufferOutputWriter8Bit", "JpegExtendingOutputWriter", "JpegFrameHeader", "JpegFrameComponentSpecificationParameters", "JpegScanHeader",
    "JpegScanComponentSpecificationParameters", "JpegHuffmanDecodingTable", "JpegGpuProgressiveScanDecoder", "FMT_INTERLEAVED_U8", "FMT_PLANAR_U8", "FMT_PLANAR_I16", "FMT_RGB_U8", "FMT_RGBA_U8", "FMT_EXTENDED_U16",
    "JpegError", "InvalidDataException", "InvalidOperationException", "NotSupportedException", "ArgumentException",
    "DeviceError", "NoDeviceError",
]
