"""The reference's Python surface (python/blazeseq/__init__.py:267-321) over the HIP parser: ``parser(path, schema)``
with ``.records``, ``.batches``, ``.batches_with_size(n)``, ``has_more()``, ``next_record()``, ``next_batch(n)``;
records expose ``id / sequence / quality`` as ``str`` plus ``phred_scores`` and ``len()``; batches expose
``num_records() / get_record(i)`` and iterate over records (pinned by the reference's
tests/test_python_bindings.py, replayed in tests/test_gpu_pyapi.py).

Everything below is plumbing over ``FastqParser`` (GPU chunks through the C ABI); .gz files are inflated on the device by the
ingest pipeline (BGZF: one wave per block, csrc/bzq_inflate.hpp; any other gzip file: csrc/bzq_gzip.hpp)."""
from __future__ import annotations

import os
from collections import deque
from typing import Deque, Iterator, List, Optional

from .parser import FastqParser, FastqBatch, FastqRecord, ParseError, EOFError_, quality_schema

_DEFAULT_BATCH_SIZE = 100   # python/blazeseq/__init__.py:30


class PyFastqRecord:
    """FastqRecordProtocol (python/blazeseq/__init__.py:38-66)."""
    __slots__ = ("_r",)

    def __init__(self, rec: FastqRecord):
        self._r = rec

    @property
    def id(self) -> str:
        return self._r.id.decode("latin-1")

    @property
    def sequence(self) -> str:
        return self._r.sequence.decode("latin-1")

    @property
    def quality(self) -> str:
        return self._r.quality.decode("latin-1")

    @property
    def phred_scores(self) -> List[int]:
        """quality byte - phred_offset, wrapping like the reference's UInt8 arithmetic (record.mojo:340-346)."""
        off = self._r.phred_offset
        return [(q - off) & 0xFF for q in self._r.quality]

    def __len__(self) -> int:
        return len(self._r.sequence)

    def __repr__(self) -> str:
        return f"FastqRecord(id={self.id!r}, len={len(self)})"


class PyFastqBatch:
    """FastqBatchProtocol (python/blazeseq/__init__.py:69-82).  Wraps a parser batch, or (when next_batch is mixed
    with record-wise calls that had already buffered records) a plain list of records."""
    __slots__ = ("_b", "_recs")

    def __init__(self, batch: Optional[FastqBatch] = None, records: Optional[List[FastqRecord]] = None):
        self._b = batch
        self._recs = records

    def _records(self) -> List[FastqRecord]:
        if self._recs is None:
            self._recs = self._b.to_records() if self._b.num_records() else []
        return self._recs

    def num_records(self) -> int:
        return len(self._recs) if self._b is None else self._b.num_records()

    def __len__(self) -> int:
        return self.num_records()

    def get_record(self, index: int) -> PyFastqRecord:
        return PyFastqRecord(self._records()[index])

    def __iter__(self) -> Iterator[PyFastqRecord]:
        for r in self._records():
            yield PyFastqRecord(r)

    def to_device(self):
        """FastqBatch.to_device() of the Mojo API (record_batch.mojo:89-90): zero-copy view of the device columns."""
        if self._b is None:
            raise ParseError(10, b"this batch was assembled from buffered records and has no device columns")
        return self._b.to_device()


class _Batches:
    def __init__(self, parser: "PyParser", batch_size: int):
        self._p, self._n = parser, batch_size

    def __iter__(self) -> Iterator[PyFastqBatch]:
        while True:
            b = self._p.next_batch(self._n)
            if b.num_records() == 0:
                return
            yield b


class PyParser:
    """_IterableParser (python/blazeseq/__init__.py:175-224).  Record-wise calls pull records from the GPU parser
    ``_RECORD_PULL`` at a time (one batch view + one copy), not one by one."""
    _RECORD_PULL = 4096

    def __init__(self, parser: FastqParser):
        self._parser = parser
        self._buf: Deque[FastqRecord] = deque()

    def has_more(self) -> bool:
        return bool(self._buf) or self._parser.has_more()

    def _fill(self):
        if not self._buf and self._parser.has_more():
            b = self._parser.next_batch(self._RECORD_PULL, partial_ok=True)   # raises at a failing record
            off = self._parser._ctx.quality_offset_schema
            for r in (b.to_records() if len(b) else []):
                r.phred_offset = off
                self._buf.append(r)

    def next_record(self) -> PyFastqRecord:
        """Raises at the end of the stream with "EOF" in the message, like the extension module."""
        self._fill()
        if not self._buf:
            raise EOFError_(6, b"EOF")
        return PyFastqRecord(self._buf.popleft())

    next_ref_as_record = next_record

    def next_batch(self, max_records: int) -> PyFastqBatch:
        if not self._buf:
            return PyFastqBatch(self._parser.next_batch(max_records))
        recs = [self._buf.popleft() for _ in range(min(max_records, len(self._buf)))]
        if len(recs) < max_records and self._parser.has_more():
            recs += self._parser.next_batch(max_records - len(recs)).to_records()
        return PyFastqBatch(records=recs)

    @property
    def records(self) -> "PyParser":
        return self

    @property
    def batches(self) -> _Batches:
        return _Batches(self, _DEFAULT_BATCH_SIZE)

    def batches_with_size(self, batch_size: int) -> _Batches:
        return _Batches(self, batch_size)

    def __iter__(self) -> "PyParser":
        return self

    def __next__(self) -> PyFastqRecord:
        try:
            return self.next_record()
        except EOFError_:
            raise StopIteration from None


def parser(path: str, quality_schema: str = "generic", parallelism: int = 4) -> PyParser:
    """python/blazeseq/__init__.py:267-290.  ``parallelism`` is the number of ingest reader threads here (the
    reference uses it for gzip decompression threads)."""
    p = os.fspath(path)   # plain, gzip and BGZF files all go through the native ingest pipeline
    return PyParser(FastqParser(p, schema=quality_schema, reader_threads=max(1, int(parallelism))))


create_parser = parser  # python/blazeseq/__init__.py:293
