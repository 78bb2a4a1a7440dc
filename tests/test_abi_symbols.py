"""The C-ABI library loads without a GPU and exports every function include/blazeseq_hip.h declares;
the product package never touches the oracle and has no CPU fallback."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "blazeseq_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bzq_[a-z_0-9]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound():
    from blazeseq_amd import _lib
    names = declared_functions()
    assert len(names) >= 20
    assert sorted(_lib.SYMBOLS) == names          # the Python binding covers the whole header
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), n
    assert _lib.lib().bzq_abi_version() == 2


def test_struct_sizes_match_header(tmp_path):
    """sizeof and field offsets as the C compiler sees include/blazeseq_hip.h == the ctypes mirror in _lib.py."""
    import subprocess
    from blazeseq_amd import _lib
    structs = {"bzq_config": _lib.BzqConfig, "bzq_chunk": _lib.BzqChunk, "bzq_device_batch": _lib.BzqDeviceBatch,
               "bzq_host_batch": _lib.BzqHostBatch, "bzq_shard_summary": _lib.BzqShardSummary,
               "bzq_ingest_stats": _lib.BzqIngestStats, "bzq_shard_plan": _lib.BzqShardPlan,
               "bzq_shard_result": _lib.BzqShardResult, "bzq_nccl_id": _lib.BzqNcclId,
               "bzq_fasta_config": _lib.BzqFastaConfig, "bzq_fasta_chunk": _lib.BzqFastaChunk,
               "bzq_fasta_shard_summary": _lib.BzqFastaShardSummary, "bzq_fasta_shard_plan": _lib.BzqFastaShardPlan,
               "bzq_fasta_shard_result": _lib.BzqFastaShardResult, "bzq_bgzf_block": _lib.BzqBgzfBlock,
               "bzq_device_views": _lib.BzqDeviceViews, "bzq_gzip_stats": _lib.BzqGzipStats}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "blazeseq_hip.h"', "int main(void) {"]
    for cname, ct in structs.items():
        lines.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in [f[:2] for f in ct._fields_]:
            lines.append(f'printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ["return 0; }"]
    src = tmp_path / "sz.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)], check=True)
    out = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for cname, ct in structs.items():
        assert int(out[cname]) == ctypes.sizeof(ct), cname
        for fname, _ in [f[:2] for f in ct._fields_]:
            assert int(out[f"{cname}.{fname}"]) == getattr(ct, fname).offset, f"{cname}.{fname}"
    assert ctypes.sizeof(_lib.BzqConfig) == 72 and ctypes.sizeof(_lib.BzqDeviceBatch) == 88


def test_config_defaults_and_schema_table_without_gpu():
    from blazeseq_amd import _lib
    L = _lib.lib()
    c = _lib.BzqConfig()
    L.bzq_config_default(ctypes.byref(c))
    # ParserConfig defaults, blazeseq/fastq/parser.mojo:60-74 + CONSTS.mojo:26-31
    assert (c.buffer_capacity, c.buffer_max_capacity, c.buffer_growth_enabled, c.check_ascii, c.check_quality,
            c.batch_size, c.q_lower, c.q_upper, c.q_offset) == (256 * 1024, 1 << 30, 0, 0, 0, 4096, 33, 126, 33)
    import blazeseq_amd as B
    assert B.quality_schema("solexa") == (59, 126, 64, True)
    assert B.quality_schema("illumina_1.5") == (66, 126, 64, True)
    assert L.bzq_message_for_code(3) == b"Quality and sequence line do not match in length"


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import blazeseq_amd as B
    with pytest.raises(RuntimeError, match="no HIP device"):
        B.Context()
    with pytest.raises(RuntimeError, match="no HIP device"):
        B.FastqParser(b"@a\nA\n+\n!\n")


def test_product_never_imports_the_oracle():
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "blazeseq_amd")):
        for fn in files:
            if fn.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                text = open(os.path.join(dirpath, fn), errors="ignore").read()
                if re.search(r"(from|import)\s+oracle|bzq_oracle|libbzq_oracle|orc_[a-z_]+\(", text):
                    bad.append(os.path.join(dirpath, fn))
    assert not bad, bad


def test_host_simd_width_is_one_of_the_reference_widths():
    """bzq_host_simd_width = simd_width_of[DType.uint8]() of this host: what the Mojo shim passes as compat_simd_width to be
    bit-exact with the reference binary on the same machine (record.mojo:76-104)."""
    from blazeseq_amd import _lib
    w = _lib.lib().bzq_host_simd_width()
    assert w in (16, 32, 64)
    flags = open("/proc/cpuinfo").read() if os.path.exists("/proc/cpuinfo") else ""
    if " avx512bw" in flags:
        assert w == 64
    elif " avx2" in flags:
        assert w == 32


def test_integration_md_config_struct_matches_the_header():
    """The Mojo `BzqConfig` sketched in INTEGRATION.md lists the fields of `bzq_config` in the header's order (it went stale
    once: `_pad1` where the header had grown `views_only`)."""
    import re
    from blazeseq_amd import _lib
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = md[md.index("struct BzqConfig"):md.index("struct HipFastq")]
    mojo_fields = re.findall(r"^\s+var (\w+):", block, re.M)
    assert mojo_fields == [f[0] for f in _lib.BzqConfig._fields_]


# ---- mojo/blazeseq_hip.mojo: the binding a BlazeSeq maintainer adds (it cannot be compiled here: no Mojo toolchain) -------------

MOJO_SIZES = {"UInt64": 8, "Int64": 8, "Int32": 4, "UInt32": 4, "UInt8": 1, "Int8": 1, "Float32": 4, "Float64": 8, "c_void_ptr": 8}


def _mojo_structs():
    import re
    src = open(os.path.join(ROOT, "mojo", "blazeseq_hip.mojo")).read()
    out = {}
    for m in re.finditer(r"^@fieldwise_init\nstruct (Bzq\w+)\(Copyable, Movable\):[^\n]*\n((?:    var \w+: [^\n]+\n)+)", src, re.M):
        out[m.group(1)] = re.findall(r"^    var (\w+): (.+)$", m.group(2), re.M)
    return src, out


def test_every_struct_of_the_mojo_shim_matches_the_header_field_for_field():
    """field order, field sizes and total size of EVERY struct in mojo/blazeseq_hip.mojo against the ctypes
    mirror (itself pinned against the compiled C header by test_struct_sizes_match_header)."""
    import ctypes as C
    import re
    from blazeseq_amd import _lib
    _, structs = _mojo_structs()
    mirror = {n: getattr(_lib, n) for n in dir(_lib) if isinstance(getattr(_lib, n), type) and issubclass(getattr(_lib, n), C.Structure)
              and n.startswith("Bzq")}
    assert set(structs) == set(mirror), (sorted(set(mirror) - set(structs)), sorted(set(structs) - set(mirror)))

    def size_of(t):
        t = t.strip()
        m = re.fullmatch(r"InlineArray\[(\w+), (\d+)\]", t)
        if m:
            return MOJO_SIZES[m.group(1)] * int(m.group(2))
        if t in MOJO_SIZES:
            return MOJO_SIZES[t]
        return C.sizeof(mirror[t])   # a nested struct
    for name, fields in structs.items():
        st = mirror[name]
        assert [f for f, _ in fields] == [f[0] for f in st._fields_], name
        off = 0
        for (fname, ftype), (_, ctype) in zip(fields, st._fields_):
            assert size_of(ftype) == C.sizeof(ctype), (name, fname, ftype)
            assert getattr(st, fname).offset >= off, (name, fname)   # (natural alignment on both sides: no field overlaps its predecessor)
            off = getattr(st, fname).offset + C.sizeof(ctype)
        m = re.search(r"struct " + name + r"\(Copyable, Movable\):\s+# include/blazeseq_hip.h: (\w+) \((\d+) bytes", open(os.path.join(ROOT, "mojo", "blazeseq_hip.mojo")).read())
        assert m and int(m.group(2)) == C.sizeof(st), name
        assert "} " + m.group(1) + ";" in open(os.path.join(ROOT, "include", "blazeseq_hip.h")).read(), m.group(1)


def test_every_function_type_of_the_mojo_shim_names_an_export_with_the_right_arity():
    import re
    from blazeseq_amd import _lib
    src, _ = _mojo_structs()
    aliases = re.findall(r"^comptime (bzq_\w+)_fn = fn\((.*)\) -> (\w+)$", src, re.M)
    assert len(aliases) >= 30
    for name, args, _ret in aliases:
        assert name in _lib.SYMBOLS, name
        n_args = 0 if not args.strip() else len(re.findall(r"\w+: ", args))
        assert n_args == len(_lib.SYMBOLS[name][1]), (name, n_args, len(_lib.SYMBOLS[name][1]))
    used = set(re.findall(r'get_function\[(bzq_\w+)_fn\]\("(bzq_\w+)"\)', src))
    assert used and all(a == b for a, b in used)   # every call binds the symbol its type alias is named after
    assert {a for a, _ in used} <= {n for n, _, _ in aliases}
