"""Mirror of the reference's public encoder surface (src/JpegLibrary/JpegEncoder.cs, JpegQuantizationTable.cs,
JpegStandardQuantizationTable.cs, JpegStandardHuffmanEncodingTable.cs; apps/JpegEncode/JpegBufferInputReader.cs) over the
GPU encoder, so that a test can be written like the reference's own callers (apps/JpegEncode/EncodeAction.cs:38-63):

    encoder = JpegEncoder()
    encoder.SetQuantizationTable(JpegStandardQuantizationTable.ScaleByQuality(JpegStandardQuantizationTable.GetLuminanceTable(0, 0), quality))
    encoder.SetQuantizationTable(JpegStandardQuantizationTable.ScaleByQuality(JpegStandardQuantizationTable.GetChrominanceTable(0, 1), quality))
    encoder.SetHuffmanTable(True, 0, JpegStandardHuffmanEncodingTable.GetLuminanceDCTable())   # or SetHuffmanTable(True, 0): built from the image
    ...
    encoder.AddComponent(1, 0, 0, 0, 2, 2); encoder.AddComponent(2, 1, 1, 1, 1, 1); encoder.AddComponent(3, 1, 1, 1, 1, 1)
    encoder.SetInputReader(JpegBufferInputReader(width, height, 3, ycbcr)); encoder.SetOutput(writer); encoder.Encode()

Same names, argument meaning and exceptions.  What the device path takes is the EncodeAction family: component 1 with sampling
(h, v) on tables 0, optionally components 2 and 3 with sampling 1 x 1 on tables 1; quantisation tables of the caller's choice
(element precision 0); Huffman tables either the four standard ones or all four left to be built from the image; any other
arrangement the reference's encoder would accept raises NotSupportedException here.  Nothing touches the device before Encode().
"""
import numpy as np

from .errors import ArgumentException, InvalidOperationException, NotSupportedException

# ref: JpegStandardQuantizationTable.cs:12-34 (zig-zag order)
_STD_LUMINANCE = (16, 11, 12, 14, 12, 10, 16, 14, 13, 14, 18, 17, 16, 19, 24, 40, 26, 24, 22, 22, 24, 49, 35, 37, 29, 40, 58, 51, 61, 60, 57, 51,
                  56, 55, 64, 72, 92, 78, 64, 68, 87, 69, 55, 56, 80, 109, 81, 87, 95, 98, 103, 104, 103, 62, 77, 113, 121, 112, 100, 120, 92, 101,
                  103, 99)
_STD_CHROMINANCE = (17, 18, 18, 24, 21, 24, 47, 26, 26, 47, 99, 66, 56, 66) + (99,) * 50


class JpegQuantizationTable:
    """ref: JpegQuantizationTable.cs:22-33 -- elements in zig-zag order; element precision 0 = 8 bit, 1 = 12 bit."""

    def __init__(self, elementPrecision=0, identifier=0, elements=None):
        if elements is None:
            self.ElementPrecision, self.Identifier, self.Elements = 0, 0, None  # default(JpegQuantizationTable)
            return
        if len(elements) != 64:
            raise ArgumentException("The length of elements must be 64.")
        self.ElementPrecision = int(elementPrecision)
        self.Identifier = int(identifier)
        self.Elements = tuple(int(e) for e in elements)

    @property
    def IsEmpty(self):
        return self.Elements is None


class JpegStandardQuantizationTable:
    @staticmethod
    def GetLuminanceTable(elementPrecision, identifier):  # :42-45
        return JpegQuantizationTable(elementPrecision, identifier, _STD_LUMINANCE)

    @staticmethod
    def GetChrominanceTable(elementPrecision, identifier):  # :53-56
        return JpegQuantizationTable(elementPrecision, identifier, _STD_CHROMINANCE)

    @staticmethod
    def ScaleByQuality(quantizationTable, quality):  # :64-87
        if quantizationTable.IsEmpty:
            raise ArgumentException("Quantization table is not initialized. (Parameter 'quantizationTable')")
        if quality < 0 or quality > 100:
            raise ArgumentException("Specified argument was out of the range of valid values. (Parameter 'quality')")
        scale = 5000 // quality if quality < 50 else 200 - quality * 2  # quality 0: DivideByZeroException there, ZeroDivisionError here
        return JpegQuantizationTable(quantizationTable.ElementPrecision, quantizationTable.Identifier,
                                     [min(max((x * scale + 50) // 100, 1), 255) for x in quantizationTable.Elements])


class JpegHuffmanEncodingTable:
    """An encoding table object.  The device path knows the four standard ones (JpegStandardHuffmanEncodingTable) by identity."""

    def __init__(self, standard_slot):
        self._standard_slot = standard_slot  # 0 DC luminance, 1 AC luminance, 2 DC chrominance, 3 AC chrominance


class JpegStandardHuffmanEncodingTable:  # ref: JpegStandardHuffmanEncodingTable.cs:142-195
    _tables = [JpegHuffmanEncodingTable(k) for k in range(4)]

    @classmethod
    def GetLuminanceDCTable(cls):
        return cls._tables[0]

    @classmethod
    def GetLuminanceACTable(cls):
        return cls._tables[1]

    @classmethod
    def GetChrominanceDCTable(cls):
        return cls._tables[2]

    @classmethod
    def GetChrominanceACTable(cls):
        return cls._tables[3]


class JpegBufferInputReader:
    """ref: apps/JpegEncode/JpegBufferInputReader.cs:14-20 -- interleaved 8-bit samples, componentCount per pixel."""

    def __init__(self, width, height, componentCount, buffer):
        self.Width, self.Height, self.ComponentCount = int(width), int(height), int(componentCount)
        self.buffer = np.frombuffer(buffer, dtype=np.uint8) if not isinstance(buffer, np.ndarray) else np.ascontiguousarray(buffer, dtype=np.uint8).reshape(-1)
        if self.buffer.size < self.Width * self.Height * self.ComponentCount:
            raise ArgumentException("The buffer is too small for the image.")


class JpegEncoder:
    def __init__(self, ctx=None):
        self._ctx = ctx
        self.MostOptimalCoding = False  # :43
        self._input = None
        self._output = None
        self._quant = []        # SetQuantizationTable order (:102-126)
        self._huffman = {}      # (class, identifier) -> table or None (= to be built), in SetHuffmanTable order
        self._components = []   # AddComponent order
        self.restart_interval = 0  # extension (see jpgpu_encode_params): 0 = what the reference writes

    # ---- setters, with the reference's checks and messages
    def SetInputReader(self, inputReader):  # :84-87
        if inputReader is None:
            raise ArgumentException("Value cannot be null. (Parameter 'inputReader')")
        self._input = inputReader

    def SetOutput(self, output):  # :93-96
        if output is None:
            raise ArgumentException("Value cannot be null. (Parameter 'output')")
        self._output = output

    def SetQuantizationTable(self, table):  # :102-126
        if table.IsEmpty:
            raise ArgumentException("Quantization table is not initialized. (Parameter 'table')")
        if table.ElementPrecision != 0:
            raise InvalidOperationException("Only baseline JPEG is supported.")
        for k, t in enumerate(self._quant):
            if t.Identifier == table.Identifier:
                self._quant[k] = table
                return
        self._quant.append(table)

    def SetHuffmanTable(self, isDcTable, identifier, table=None):  # :137-148; no table = built from the image's statistics
        self._huffman[(0 if isDcTable else 1, int(identifier))] = table

    def AddComponent(self, componentIndex, quantizationTableIdentifier, huffmanDcTableIdentifier, huffmanAcTableIdentifier, horizontalSubsampling,
                     verticalSubsampling):  # :175-239
        if horizontalSubsampling not in (1, 2, 4):
            raise ArgumentException("Subsampling factor can only be 1, 2 or 4. (Parameter 'horizontalSubsampling')")
        if verticalSubsampling not in (1, 2, 4):
            raise ArgumentException("Subsampling factor can only be 1, 2 or 4. (Parameter 'verticalSubsampling')")
        if any(c[0] == componentIndex for c in self._components):
            raise ArgumentException("The component index is already used by another component. (Parameter 'componentIndex')")
        if not any(t.Identifier == quantizationTableIdentifier for t in self._quant):
            raise ArgumentException("Quantization table is not defined. (Parameter 'quantizationTableIdentifier')")
        if (0, huffmanDcTableIdentifier) not in self._huffman:
            raise ArgumentException("Huffman table is not defined. (Parameter 'huffmanDcTableIdentifier')")
        if (1, huffmanAcTableIdentifier) not in self._huffman:
            raise ArgumentException("Huffman table is not defined. (Parameter 'huffmanAcTableIdentifier')")
        # the component captures the quantisation table as it is NOW (:226: a later SetQuantizationTable does not reach it)
        quant = next(t for t in self._quant if t.Identifier == quantizationTableIdentifier)
        self._components.append((int(componentIndex), quant, int(huffmanDcTableIdentifier), int(huffmanAcTableIdentifier), int(horizontalSubsampling),
                                 int(verticalSubsampling)))

    # ---- Encode (:255-291)
    def Encode(self):
        from .encoder import EncodeBatch  # the device is only needed from here on

        if self._output is None:
            raise InvalidOperationException("Output is not specified.")
        if self._input is None:
            raise InvalidOperationException("Input is not specified.")
        if not self._components:
            raise InvalidOperationException("No component is specified.")
        comps = self._components
        reader = self._input
        # what the device path encodes: the EncodeAction arrangement
        if len(comps) not in (1, 3) or reader.ComponentCount != len(comps):
            raise NotSupportedException("1 or 3 components are supported.")
        if [c[0] for c in comps] != [1, 2, 3][:len(comps)]:
            raise NotSupportedException("Component indices 1, 2, 3 in this order are supported.")
        first = comps[0]
        if (first[2], first[3]) != (0, 0) or first[1].Identifier != 0:
            raise NotSupportedException("The first component uses quantization table 0 and Huffman tables 0.")
        for c in comps[1:]:
            if (c[2], c[3], c[4], c[5]) != (1, 1, 1, 1) or c[1].Identifier != 1 or c[1].Elements != comps[1][1].Elements:
                raise NotSupportedException("The other components use quantization table 1, Huffman tables 1 and sampling 1 x 1.")
        needed = [(0, 0), (1, 0)] + ([(0, 1), (1, 1)] if len(comps) == 3 else [])
        tables = [self._huffman[k] for k in needed]
        if all(t is None for t in tables):
            mode = 2 if self.MostOptimalCoding else 1
        elif all(t is not None and t._standard_slot == 2 * k[1] + k[0] for t, k in zip(tables, needed)):
            mode = 0
        else:
            raise NotSupportedException("Huffman tables: either the standard tables or all of them built from the image.")
        # (a single component with tables to be built: the chrominance builders stay empty and BuildTables throws "No symbol is
        # recorded." -- the device path reports exactly that when the stream is asked for)
        for c in comps:  # the stream's DQT holds the CURRENT tables, the components quantise with the ones captured at AddComponent
            if c[1].Elements != next(t for t in self._quant if t.Identifier == c[1].Identifier).Elements:
                raise NotSupportedException("A quantization table was replaced after AddComponent captured it.")
        # every table of the collection is written (WriteQuantizationTables :305-335, WriteHuffmanTables :336-352): the device path
        # writes exactly tables 0 and 1 of each kind, so that is what the collection must hold
        if sorted(t.Identifier for t in self._quant) != [0, 1] or sorted(self._huffman) != [(0, 0), (0, 1), (1, 0), (1, 1)]:
            raise NotSupportedException("Quantization tables 0 and 1 and Huffman tables 0 and 1 (DC and AC) are what the stream carries.")
        if [t.Identifier for t in self._quant] != [0, 1] or list(self._huffman) != [(0, 0), (1, 0), (0, 1), (1, 1)]:
            raise NotSupportedException("Tables are written in the order they were set: 0 before 1, DC before AC.")
        pixels = reader.buffer[:reader.Width * reader.Height * reader.ComponentCount].reshape(reader.Height, reader.Width, reader.ComponentCount)
        batch = EncodeBatch(self._ctx)
        try:
            batch.upload([pixels], (first[4], first[5]), 50, rgb=False, optimize_coding=mode, restart_interval=self.restart_interval)
            batch.set_quantization_table(0, 0, first[1].Elements)
            batch.set_quantization_table(0, 1, (comps[1][1] if len(comps) == 3 else next(t for t in self._quant if t.Identifier == 1)).Elements)
            batch.encode()
            data = batch.output(0)
        finally:
            batch.close()
        out = self._output
        if hasattr(out, "extend"):
            out.extend(data)
        elif hasattr(out, "write"):
            out.write(data)
        else:
            raise ArgumentException("output: a bytearray-like (extend) or file-like (write) object")
