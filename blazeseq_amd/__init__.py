"""blazeseq_amd -- MI355X-native FASTQ batch parsing behind BlazeSeq's API surface.

Host-side mirror (Python over the C ABI of libblazeseq_hip.so) of the reference's
``FastqParser`` / ``ParserConfig`` / ``batches()`` / ``FastqBatch.to_device()``
(blazeseq/fastq/parser.mojo:33-274, blazeseq/fastq/record_batch.mojo:19-244).
All parsing happens in hand-written HIP kernels for gfx950; there is no CPU fallback.
"""
from ._lib import (LIB_PATH, LibraryMissing, OK, ID_NO_AT, SEP_NO_PLUS, SEQ_QUAL_LEN_MISMATCH, ASCII_INVALID,
                   QUALITY_OUT_OF_RANGE, EOF, UNEXPECTED_EOF, BUFFER_EXCEEDED, BUFFER_AT_MAX, OTHER)
from .parser import (ParserConfig, FastqParser, FastqBatch, DeviceFastqBatch, FastqRecord, FastqView, ParseError, Context,
                     Ingest, GzipDecoder, ChunkResult, quality_schema, DEFAULT_BATCH_SIZE, DEFAULT_CAPACITY, MAX_CAPACITY)

__all__ = ["ParserConfig", "FastqParser", "FastqBatch", "DeviceFastqBatch", "FastqRecord", "FastqView", "ParseError", "Context",
           "Ingest", "GzipDecoder", "ChunkResult", "quality_schema", "LibraryMissing", "LIB_PATH"]
from .pyapi import parser, create_parser, PyParser, PyFastqBatch, PyFastqRecord  # noqa: E402

__all__ += ["parser", "create_parser"]
from .fasta import FastaParser, FastaRecord, FastaParserConfig, FastaContext, FastaIngest, Definition  # noqa: E402

__all__ += ["FastaParser", "FastaRecord", "FastaParserConfig", "FastaContext", "FastaIngest", "Definition"]
