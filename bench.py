#!/usr/bin/env python3
"""bench.py -- FASTQ batch-parse throughput on MI355X (BASELINE.json metric).

Workload (BASELINE.json configs[1]): synthetic 150 bp Illumina FASTQ from the reference's generator
(blazeseq/utils.mojo:831-917, args (num_reads, 150, 150, 33, 73, "generic")), 10 M reads per GPU,
batches(4096), validation off.  The input is generated on the device and is resident in HBM when
the timed region starts.  One "step" = one pass of the whole hot path over that input:
delimiter scan -> record table -> FastqBatch columns (+ per-batch ends) -> every DeviceFastqBatch of the
chunk handed out (bzq_batches: 2442 zero-copy views), i.e. everything
``for batch in parser.batches(4096): batch.to_device()`` does in the reference.

N > 1 (launched by torch.distributed.run, one rank per GPU): BASELINE.json configs[4]'s shard shape --
78.125 M reads = 25 GB per GPU of ONE synthetic stream with 9-digit headers (320 B/record; at N = 8 that is
the 625 M-read, 200 GB file), cut into N byte ranges whose interior cuts sit 144 bytes into a record
(not record aligned).  Each step is the whole protocol of bzq_shard_stitch (C ABI, RCCL bound by the
library itself): shard scan, summary all-gather, the straddling record's remainder sent to its owner,
parse, outcome all-gather.  Weak scaling.  torch.distributed only launches the ranks, hands out the
ncclUniqueId and takes the max of the times.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
# HBM traffic of the dominant kernel per launch: NOT measured in this run (PMC counters need rocprofv3 around the process)
# but READ from the committed rocprofv3 summary of this very command (scripts/gpu_profile.sh -> profiles/<tag>_summary.txt),
# so the number printed is by construction the one in the cited file (tests/test_bench_contract.py pins the parse).
# FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE is doubled per the gfx950 correction of MI355X_MICROARCH.md (HBM section).
TRAFFIC_PROFILE = "profiles/r6_final_summary.txt"
TRAFFIC_KERNEL = "k_fused<false, false, false, false, true>"   # the default emit (per-batch ends folded in); a run without the fold cites the other instantiation
TRAFFIC_KERNEL_UNFOLDED = "k_fused<false, false, false, false>"
TRAFFIC_RECORDS = 10_000_000   # the profiled launch: 10 M x 150 bp


def launch_ranks(nranks, argv, script=None, deadline_s=3600.0, poll_s=0.05):
    """`python3 bench.py --gpus N` typed as ONE command (the reference's harness is one command per run,
    benchmark/throughput/run_throughput_benchmarks.sh:56-62): this process becomes the launcher of N ranks of the same
    script, one per GPU -- RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR (127.0.0.1) / MASTER_PORT (a free one) in each
    child's environment, exactly what torch.distributed.run would set.  Rank 0 owns this process's stdout (its ONE JSON
    line is the launcher's); the other ranks' stdout goes to stderr.  The first rank that exits non-zero is NAMED, the
    others are stopped (by their own PIDs) and the launcher exits with that rank's code; the same when the deadline
    passes.  Returns the exit code (0 = every rank exited 0)."""
    import signal
    import socket
    import subprocess

    def die_with_parent():   # (Linux: a rank must not outlive a launcher that was killed -- it would keep its GPU busy)
        try:
            import ctypes
            ctypes.CDLL(None).prctl(1, signal.SIGTERM)   # PR_SET_PDEATHSIG
        except Exception:   # noqa: BLE001
            pass
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    script = script or os.path.abspath(__file__)
    procs = []
    for r in range(nranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(nranks), LOCAL_WORLD_SIZE=str(nranks),
                   GROUP_RANK="0", ROLE_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), BZQ_BENCH_LAUNCHER_PID=str(os.getpid()))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL between processes needs it on this host driver
        env.setdefault("OMP_NUM_THREADS", "1")
        procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=env, stdout=None if r == 0 else sys.stderr, preexec_fn=die_with_parent))

    def stop_all(signum, frame):   # the launcher is being stopped (a driver's timeout, ^C): take the ranks along, exactly the PIDs started here
        for pr in procs:
            if pr.poll() is None:
                pr.terminate()
        print(f"[bench launcher] stopped by signal {signum}: the ranks were told to stop", file=sys.stderr, flush=True)
        os._exit(128 + signum)
    for sg in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
        try:
            signal.signal(sg, stop_all)
        except (ValueError, OSError):   # (not the main thread: a test harness)
            pass
    t_end = time.monotonic() + deadline_s
    failed = None
    live = set(range(nranks))
    while live and failed is None:
        for r in sorted(live):
            rc = procs[r].poll()
            if rc is None:
                continue
            live.discard(r)
            if rc != 0:
                failed = (r, rc, f"rank {r} of {nranks} (pid {procs[r].pid}) exited with code {rc}")
                break
        if failed is None and live and time.monotonic() > t_end:
            failed = (min(live), 124, f"deadline of {deadline_s:.0f} s passed with rank(s) {sorted(live)} of {nranks} still running")
        if live and failed is None:
            time.sleep(poll_s)
    if failed is None:
        return 0
    for r in sorted(live):   # stop what is left: exactly the PIDs started here
        if procs[r].poll() is None:
            procs[r].terminate()
    t_kill = time.monotonic() + 10.0
    for r in sorted(live):
        try:
            procs[r].wait(timeout=max(0.1, t_kill - time.monotonic()))
        except Exception:   # noqa: BLE001
            procs[r].kill()
            procs[r].wait()
    print(f"[bench launcher] FAILED: {failed[2]}; the other ranks were stopped; no JSON line is valid for this run", file=sys.stderr, flush=True)
    return failed[1] if 0 < failed[1] < 256 else 1


def profile_traffic(path=TRAFFIC_PROFILE, kernel=TRAFFIC_KERNEL):
    """(FETCH_SIZE, WRITE_SIZE) per dispatch of `kernel`, in the counter's unit (KiB), from a scripts/summarize_prof.py text."""
    vals, section, cur = {}, None, None
    try:
        lines = open(os.path.join(ROOT, path)).read().splitlines()
    except OSError:
        return None
    for ln in lines:
        if ln.startswith("== "):
            section = ln.split()[1].rstrip(":")
        elif section in ("pmc_fetch", "pmc_write") and ln.startswith("  ") and not ln.startswith("      "):
            cur = ln.strip().rsplit("  (dispatches", 1)[0]
        elif section in ("pmc_fetch", "pmc_write") and cur == kernel and ln.strip().startswith(("FETCH_SIZE", "WRITE_SIZE")):
            k, v = ln.split()
            vals[k] = float(v)
    return (vals["FETCH_SIZE"], vals["WRITE_SIZE"]) if len(vals) == 2 else None


HBM_ACHIEVABLE_GBS = 6300.0   # /opt/skills/guides/MI355X_MICROARCH.md, HBM section: "8 TB/s peak (spec); ~6.3 TB/s achievable"
PATH_KERNELS_A = ("k_tile_aggregate_h", "k_scan_reduce", "k_scan_down", "k_batch_bases", "k_tail")   # the hot path's other kernels (names of the cited summary)


def path_traffic_bytes(kernel=None):
    """HBM bytes of ONE step of the whole hot path (pass A + scan + batch bases + emit + tail) from the cited summary's counter
    passes: sum over the path's kernels of FETCH_SIZE x 2 + WRITE_SIZE (KiB per dispatch); None when a kernel is missing there."""
    total = 0.0
    for k in PATH_KERNELS_A + (kernel or TRAFFIC_KERNEL,):
        fw = profile_traffic(kernel=k)
        if fw is None:
            return None
        total += (fw[0] * 2 + fw[1]) * 1024.0
    return total


def profile_kernel_avg_ms(path=TRAFFIC_PROFILE, kernel=TRAFFIC_KERNEL):
    """Average launch duration (ms) of `kernel` in the `== kt` (rocprofv3 --kernel-trace --stats) section of the same summary."""
    try:
        lines = open(os.path.join(ROOT, path)).read().splitlines()
    except OSError:
        return None
    section = None
    for ln in lines:
        if ln.startswith("== "):
            section = ln.split()[1].rstrip(":")
        elif section == "kt" and ln.strip().startswith(kernel + " ") and "avg_us=" in ln:
            return float(ln.split("avg_us=")[1].split()[0]) / 1e3
    return None


TRAFFIC_MAX_DRIFT = 0.10   # the cited profile must be of THIS kernel: its average launch time within 10 % of this run's


def measured_traffic_bytes_per_record(run_avg_launch_ms=None, kernel=TRAFFIC_KERNEL):
    """Bytes per record from the cited profile, or (None, why) when the profile is missing or stale: a kernel that changed
    since it was profiled runs at a different speed, and its old traffic figure must not be printed beside the new time."""
    t = profile_traffic(kernel=kernel)
    if t is None:
        return None, f"{TRAFFIC_PROFILE} is missing or holds no FETCH_SIZE / WRITE_SIZE for {kernel}"
    if run_avg_launch_ms is not None:
        prof_ms = profile_kernel_avg_ms(kernel=kernel)
        if prof_ms is None:
            return None, f"{TRAFFIC_PROFILE} holds no kernel-trace duration for {kernel}"
        drift = abs(prof_ms - run_avg_launch_ms) / run_avg_launch_ms
        if drift > TRAFFIC_MAX_DRIFT:
            return None, (f"stale profile: {kernel} averages {prof_ms:.4f} ms in {TRAFFIC_PROFILE} but {run_avg_launch_ms:.4f} ms in this run "
                          f"({drift * 100:.0f} % apart, limit {TRAFFIC_MAX_DRIFT * 100:.0f} %): re-profile (scripts/gpu_profile.sh)")
    return (2 * t[0] + t[1]) * 1024 / TRAFFIC_RECORDS, None


def cpu_baseline(data, reads: int, read_len: int, check: bool, mode: str = "batches", t_budget: float = 12.0):
    """Reference-algorithm CPU baseline: the C restatement of BlazeSeq's streaming parser
    (oracle/bzq_oracle.c), batches(4096) mode, 64 KiB buffer like the reference's own runner
    (benchmark/throughput/run_throughput_blazeseq.mojo:28-40), one core, bounded sample.
    ``data``: the first ``reads`` records of the very input the GPU was timed on (copied back to the host)."""
    from oracle import oracle as O
    cfg = O.make_config(buffer_capacity=64 * 1024, check_ascii=check, check_quality=check, batch_size=4096)
    for _ in range(2):
        O.bench_run(data, cfg, mode)
    times, nrec = [], 0
    t_start = time.perf_counter()
    while time.perf_counter() - t_start < t_budget and len(times) < 15:
        t0 = time.perf_counter()
        nrec, _ = O.bench_run(data, cfg, mode)
        times.append(time.perf_counter() - t0)
    assert nrec == reads
    best = sum(times) / len(times)
    return {"value": round(data.size / best / 1e9, 4), "unit": "GB/s", "cores": 1, "kind": "port",
            "mrecords_per_s": round(reads / best / 1e6, 3),
            "mode": "batches(4096)" if mode == "batches" else "views()",
            "sample": f"the first {reads} reads of the GPU input ({read_len} bp, {data.size} B), {'batches(4096)' if mode == 'batches' else 'views() (next_view per record, run_throughput_blazeseq.mojo:38-45)'}, 64 KiB buffer, "
                      f"validation {'on' if check else 'off'}, mean of {len(times)} runs, in-memory"}


def cpu_baseline_all_cores(data, reads: int, rec_bytes: int, check: bool):
    """The same restated parser, one thread per host core, each on a record-aligned slice of the same bytes
    (baseline (ii)): what the whole host can do when the file is split for it.  ctypes releases the GIL."""
    import threading
    from oracle import oracle as O
    cores = max(1, min(os.cpu_count() or 1, reads))
    per = (reads + cores - 1) // cores
    cfg = O.make_config(buffer_capacity=64 * 1024, check_ascii=check, check_quality=check, batch_size=4096)
    slices = [data[i * per * rec_bytes:min(reads, (i + 1) * per) * rec_bytes] for i in range(cores)]
    slices = [s for s in slices if s.size]
    got = [0] * len(slices)

    REPS = 16   # a slice is ~12 MB (1.5 ms of parsing): repeat it so that thread start-up does not dominate

    def work(i):
        for _ in range(REPS):
            got[i] = O.bench_run(slices[i], cfg, "batches")[0]

    best = None
    for _ in range(3):
        th = [threading.Thread(target=work, args=(i,)) for i in range(len(slices))]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    assert sum(got) == reads
    return {"value": round(REPS * data.size / best / 1e9, 3), "unit": "GB/s", "cores": len(slices), "kind": "port",
            "sample": f"same bytes, {len(slices)} threads on record-aligned slices, each slice parsed {REPS}x, best of 3"}


def inflate_mode(ctx, shard, rec_bytes, dev):
    """Compressed input beside the headline: the first 32 MB of the GPU's own FASTQ compressed by zlib on the host
    (level 6) as (a) a plain gzip file of 16 members and (b) a BGZF file, both decoded on the device; every output byte compared with
    the FASTQ on the device.  Rates are bytes of FASTQ delivered per second; the compressed bytes start in pinned host memory (gzip:
    bzq_gzip_decode, PCIe inclusive) resp. in device memory (BGZF: the kernel alone)."""
    import struct
    import zlib
    import ctypes as C
    import numpy as np
    import torch
    import blazeseq_amd as B
    from blazeseq_amd import _lib as L
    k = (32 << 20) // rec_bytes * rec_bytes
    d_plain = shard[:k]
    plain = d_plain.cpu().numpy().tobytes()
    reps = 48   # 1.6 GB of FASTQ per decode: the decoders work in rounds of ~6 000 waves, and 0.5 GB was one round and a bit
    out = torch.empty(reps * k + (1 << 20), dtype=torch.uint8, device=dev)
    res = {}
    # (a) gzip: one member per copy of the slice
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    member = bytes([0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 3]) + co.compress(plain) + co.flush() + struct.pack("<II", zlib.crc32(plain) & 0xFFFFFFFF, k & 0xFFFFFFFF)
    pin = torch.from_numpy(np.frombuffer(member * reps, dtype=np.uint8).copy()).pin_memory()
    best = None
    for _ in range(3):
        dec = B.GzipDecoder(ctx)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        # the file in pieces of 256 MiB, the next piece on its way to the device while this one is decoded (bzq_gzip_stage)
        host, piece, got, off = pin.numpy(), 256 << 20, 0, 0
        dec.stage(host[:piece])
        dec.stage(host[piece:2 * piece])
        while off < host.size:
            part = host[off:off + piece]
            off += part.size
            if off + piece < host.size:   # two pieces in front of the one being decoded are on their way (the decoder starts piece k + 1 under piece k's last kernels)
                dec.stage(host[off + piece:off + 2 * piece])
            while True:
                nb, more = dec.feed(part, off >= host.size, out.data_ptr() + got, out.numel() - got)
                got += nb
                if not more:
                    break
                part = part[:0]
        dt = time.perf_counter() - t0
        st = dec.stats()
        host_calls = dec.set_option("host_calls", 0)   # calls continued by zlib on the host (a stretch without findable block starts): must be 0 here
        dec.close()
        assert got == reps * k and st.members == reps and host_calls == 0, (got, reps * k, st.members, host_calls)
        best = dt if best is None or dt < best else best
    assert bool((out[:reps * k].view(reps, k) == d_plain.unsqueeze(0)).all()), "gzip: device output differs from the FASTQ"
    res["gzip"] = {"value": round(reps * k / best / 1e9, 3), "unit": "GB/s of FASTQ", "ms": round(best * 1e3, 2), "compressed_mb": round(len(member) * reps / 1e6, 1),
                   "decoder_runs_in_output": int(st.chain_jobs), "restarts": int(st.fallback_jobs), "calls_continued_on_the_host": int(host_calls),
                   "note": f"{reps} members of gzip -6 (zlib) in pinned host memory -> bzq_gzip_decode in 256 MiB pieces (the next two staged meanwhile) -> device, verified; the reference's GZFile way (zlib gzread, one host core): ~0.35 GB/s"}
    del pin
    # (b) BGZF: 65280-byte blocks
    def block(data):
        c2 = zlib.compressobj(6, zlib.DEFLATED, -15)
        payload = c2.compress(data) + c2.flush()
        return (b"\x1f\x8b\x08\x04" + b"\0\0\0\0" + b"\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 18 + len(payload) + 8 - 1) + payload +
                struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))
    one = np.frombuffer(b"".join(block(plain[i:i + 65280]) for i in range(0, k, 65280)), dtype=np.uint8)
    blocks, nblk, consumed, out_bytes = ctx.bgzf_scan(one)
    assert consumed == one.size and out_bytes == k
    comp = torch.from_numpy(np.tile(one, reps)).to(dev)
    tab = (L.BzqBgzfBlock * (nblk * reps))()
    for r in range(reps):
        for i in range(nblk):
            b0 = blocks[i]
            tab[r * nblk + i] = L.BzqBgzfBlock(b0.comp_offset + r * one.size, b0.comp_size, b0.out_size, b0.crc32, 0, b0.out_offset + r * k)
    torch.cuda.synchronize()
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        ctx.bgzf_inflate(comp.data_ptr(), comp.numel(), tab, nblk * reps, out.data_ptr(), out.numel())
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    assert bool((out[:reps * k].view(reps, k) == d_plain.unsqueeze(0)).all()), "BGZF: device output differs from the FASTQ"
    res["bgzf"] = {"value": round(reps * k / best / 1e9, 3), "unit": "GB/s of FASTQ", "ms": round(best * 1e3, 2), "blocks": nblk * reps,
                   "note": "bgzip-style 65280-byte blocks (zlib -6), compressed bytes resident on the device -> bzq_bgzf_inflate (one wave per block, CRC-32 checked), verified"}
    return res


PCIE_PEAK_GBS = 64.0   # PCIe Gen5 x16, one direction (end-to-end figures are quoted against this, never against HBM)


def ingest_mode(shard, rec_bytes, dev, local_rank, target_gb=6.4, threads=8, chunk_mib=256, modes=("plain", "bgzf", "gzip")):
    """File -> records, wall clock, the reference's own method (benchmark/throughput/run_throughput_benchmarks.sh:54-62: the file on a
    RAM-backed filesystem, the whole run timed): a FASTQ file of ~6.4 GB on /dev/shm as plain text, as BGZF and as an ordinary
    multi-member gzip file (zlib level 6; the content is the first 32 MiB of the GPU's own reads, repeated -- compressing 3 GB on one
    host core would take minutes), each through bzq_ingest_open / bzq_ingest_next (io/readers.mojo:86-137 FileReader,
    :283-443 GZFile / RapidgzipReader) until EOF, every record counted.  `value` includes the open; the CPU figures beside it are the
    reference algorithm on the same files on one host core (oracle streaming parser; zlib for the .gz, as GZFile does)."""
    import gzip as _gz
    import struct
    import zlib
    import numpy as np
    import blazeseq_amd as B
    from blazeseq_amd import _lib as L
    from oracle import oracle as O
    k = (32 << 20) // rec_bytes * rec_bytes
    plain = shard[:k].cpu().numpy()
    pbytes = plain.tobytes()
    reps = max(2, int(target_gb * 1e9 / k))
    n_fastq, n_rec = reps * k, reps * (k // rec_bytes)
    d = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else "/tmp"
    tag = f"bzq_bench_{os.getpid()}"
    paths = {m: os.path.join(d, f"{tag}.fastq{ext}") for m, ext in (("plain", ""), ("bgzf", ".bgz"), ("gzip", ".gz"))}
    res = {"file_fastq_gb": round(n_fastq / 1e9, 3), "records": n_rec, "reader_threads": threads, "chunk_mib": chunk_mib, "dir": d,
           "note": "wall clock of open + every chunk until EOF + close.  TWO figures per file kind, side by side: value = a run with the library's defaults and its "
                   "buffer cache empty (what a one-file process gets; whole processes: process_mode) and value_warm_cache = best of 3 runs after it with the cache told to keep a "
                   ".gz stream's buffers too (options pin_cache_bytes = 2 GiB, dev_cache_bytes = 16 GiB: a host that parses file after file; the library's default keeps one "
                   "set of chunk buffers, 1 GiB each); every figure after one untimed pass over the file, like the reference's warm-up runs; "
                   "the file sits on a RAM-backed filesystem like the reference's runs; pcie_frac = bytes that crossed PCIe / s / 64 GB/s"}
    try:
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        member = bytes([0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 3]) + co.compress(pbytes) + co.flush() + struct.pack("<II", zlib.crc32(pbytes) & 0xFFFFFFFF, k & 0xFFFFFFFF)

        def block(data):
            c2 = zlib.compressobj(6, zlib.DEFLATED, -15)
            payload = c2.compress(data) + c2.flush()
            return (b"\x1f\x8b\x08\x04" + b"\0\0\0\0" + b"\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 18 + len(payload) + 8 - 1) + payload +
                    struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))
        bg = b"".join(block(pbytes[i:i + 65280]) for i in range(0, k, 65280))
        for m, piece, trailer in (("plain", pbytes, b""), ("bgzf", bg, block(b"")), ("gzip", member, b"")):
            with open(paths[m], "wb") as f:
                for _ in range(reps):
                    f.write(piece)
                f.write(trailer)
        ctx = B.Context(B.ParserConfig(), "generic", 4096, local_rank, min_record_bytes=256 if rec_bytes >= 256 else 32)
        for m in modes:
            fsize = os.path.getsize(paths[m])
            best = first = None
            for it in range(5):
                if it == 1:   # run 1 opens like the first file of a process (run 0, not reported, is the file's first read: a freshly written tmpfs file reads 3x slower once)
                    # ... and from here on like a host that parses file after file and says so: the library's default cache keeps ONE set of
                    # chunk buffers (1 GiB pinned, 1 GiB device); a .gz stream's pools and FIFOs (~12 GiB) stay cached only if asked for
                    for key, val in (("pin_cache_bytes", 2 << 30), ("dev_cache_bytes", 16 << 30)):
                        ctx.set_option(key, 0)
                        ctx.set_option(key, val)
                t0 = time.perf_counter()
                ing = B.Ingest(ctx, paths[m], chunk_bytes=chunk_mib << 20, n_threads=threads)
                t1 = time.perf_counter()
                taken = total = 0
                laps = []
                while True:
                    ta = time.perf_counter()
                    r = ing.next(taken)
                    laps.append(round((time.perf_counter() - ta) * 1e3, 1))
                    taken = int(r.n_records)
                    total += taken
                    if int(r.status) != L.OK:
                        break
                t2 = time.perf_counter()
                if os.environ.get("BZQ_BENCH_LAPS"):
                    print(f"ingest_mode {m} run {it}: open {round((t1 - t0) * 1e3, 1)} ms, bzq_ingest_next calls {laps}", file=sys.stderr, flush=True)
                ing.close()
                dt = time.perf_counter() - t0
                assert total == n_rec and int(r.status) == L.EOF, (m, total, n_rec, int(r.status), ctx.format_error())
                run = (dt, t1 - t0, time.perf_counter() - t2)
                if it == 1:
                    first = run
                elif it > 1 and (best is None or dt < best[0]):
                    best = run
            # `value` = the FIRST file of the process (library defaults, buffer cache empty: what a one-file run gets); the file-after-file
            # figure (cache told to keep a stream's buffers, best of 3) stands beside it as value_warm_cache (ADVICE r4 / VERDICT r5 next-8)
            res[m] = {"value": round(n_fastq / first[0] / 1e9, 2), "value_warm_cache": round(n_fastq / best[0] / 1e9, 2),
                      "unit": "GB/s of FASTQ", "mrecords_per_s": round(n_rec / first[0] / 1e6, 1),
                      "ms": round(first[0] * 1e3, 1), "open_ms": round(first[1] * 1e3, 1), "close_ms": round(first[2] * 1e3, 1), "file_gb": round(fsize / 1e9, 3),
                      "pcie_frac": round(fsize / first[0] / 1e9 / PCIE_PEAK_GBS, 3),
                      "warm_cache": {"value": round(n_fastq / best[0] / 1e9, 2), "ms": round(best[0] * 1e3, 1), "open_ms": round(best[1] * 1e3, 1),
                                     "close_ms": round(best[2] * 1e3, 1), "pcie_frac": round(fsize / best[0] / 1e9 / PCIE_PEAK_GBS, 3)}}
        for key in ("pin_cache_bytes", "dev_cache_bytes"):   # (give the cached buffers back, and the library's default limits again)
            ctx.set_option(key, 0)
            ctx.set_option(key, 1 << 30)
        ctx.close()
        # the reference algorithm on one host core, same files: plain = read + streaming parse; .gz = zlib inflate (GZFile) + parse on a bounded sample
        cfg = O.make_config(buffer_capacity=64 * 1024, batch_size=4096)
        if "plain" not in res or "gzip" not in res:   # (a sweep of one mode: scripts/sweep_ingest.py)
            return res
        t0 = time.perf_counter()
        host = np.fromfile(paths["plain"], dtype=np.uint8, count=min(n_fastq, 48 * k))
        nrec, _ = O.bench_run(host, cfg, "batches")
        dt = time.perf_counter() - t0
        res["plain"]["cpu_1core"] = {"value": round(host.size / dt / 1e9, 2), "unit": "GB/s", "sample": f"the first {host.size} B of the same file: read + oracle streaming parser, batches(4096), 64 KiB buffer"}
        t0 = time.perf_counter()
        with open(paths["gzip"], "rb") as f:
            comp = f.read(8 * len(member))
        out = _gz.decompress(comp)
        nrec, _ = O.bench_run(np.frombuffer(out, dtype=np.uint8), cfg, "batches")
        dt = time.perf_counter() - t0
        res["gzip"]["cpu_1core"] = {"value": round(len(out) / dt / 1e9, 3), "unit": "GB/s of FASTQ", "sample": f"the first 8 members ({len(out)} B of FASTQ): zlib inflate + oracle streaming parser"}
    finally:
        for q in paths.values():
            try:
                os.remove(q)
            except OSError:
                pass
    return res


REF_40BP = b"ACGTACGTACGTACGTACGTACGTACGTACGTACGTACGT"   # examples/nw_gpu/execution.mojo:36
PIPE_BATCH = 65536                                        # examples/nw_gpu/execution.mojo:34 (BATCH_SIZE)


def pipeline_mode(shard, rec_bytes, dev, local_rank, target_gb=6.4, threads=8, chunk_mib=256, kinds=("plain", "bgzf", "gzip")):
    """File -> records -> consumer with no host round trip: the reference's ONLY GPU use case (examples/nw_gpu/execution.mojo:100-130:
    `next_batch(65536)` -> `batch.to_device(ctx)` -> `nw_kernel`; the v0.1 quality_distribution kernel, CHANGELOG.md:73) and the regime
    this path exists for -- the records never leave the device.  A ~6.4 GB FASTQ file on /dev/shm -> bzq_ingest_next (reader threads ->
    pinned -> H2D -> parse) -> every batch of 65 536 records (zero-copy views of the chunk's columns) -> bzq_batch_nw_scores_dev (global
    alignment score of every read against the example's 40 bp reference) + bzq_batch_quality_by_position_acc (per-cycle quality
    distribution of the whole file in one device table) on the consumer stream, chunk k's consumers under the ingest of chunk k + 1.
    Wall clock open -> last consumer kernel done; GB/s of FASTQ.  Parity: the accumulated table and the sum of all scores must equal
    the oracle's CPU twin (orc_pipeline_run) -- asserted.  Beside it: that twin (parse + the same two reductions) on every host core.
    `value` is the plain file; the same pipeline over the file as BGZF and as an ordinary multi-member gzip file (what reads are
    stored as; both inflated on the device) stands beside it (`bgzf`, `gzip`)."""
    import ctypes as C
    import threading
    import numpy as np
    import torch
    import blazeseq_amd as B
    from blazeseq_amd import _lib as L
    from oracle import oracle as O
    k = (32 << 20) // rec_bytes * rec_bytes
    piece = shard[:k].cpu().numpy()
    reps = max(2, int(target_gb * 1e9 / k))
    n_fastq, n_rec = reps * k, reps * (k // rec_bytes)
    d = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else "/tmp"
    paths = {m: os.path.join(d, f"bzq_bench_pipe_{os.getpid()}.fastq{ext}") for m, ext in (("plain", ""), ("bgzf", ".bgz"), ("gzip", ".gz"))}
    read_len = (rec_bytes - 18) // 2
    res = {"file_fastq_gb": round(n_fastq / 1e9, 3), "records": n_rec, "batch_records": PIPE_BATCH, "reference_bp": len(REF_40BP), "reader_threads": threads,
           "chunk_mib": chunk_mib}
    try:
        import struct
        import zlib
        pb = piece.tobytes()

        def block(data):
            c2 = zlib.compressobj(6, zlib.DEFLATED, -15)
            payload = c2.compress(data) + c2.flush()
            return (b"\x1f\x8b\x08\x04" + b"\0\0\0\0" + b"\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 18 + len(payload) + 8 - 1) + payload +
                    struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))
        pieces = {"plain": (pb, b"")}
        if "bgzf" in kinds:
            pieces["bgzf"] = (b"".join(block(pb[i:i + 65280]) for i in range(0, k, 65280)), block(b""))
        if "gzip" in kinds:
            co = zlib.compressobj(6, zlib.DEFLATED, -15)
            pieces["gzip"] = (bytes([0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 3]) + co.compress(pb) + co.flush() + struct.pack("<II", zlib.crc32(pb) & 0xFFFFFFFF, k & 0xFFFFFFFF), b"")
        for m, (body, trailer) in pieces.items():
            with open(paths[m], "wb") as f:
                for _ in range(reps):
                    f.write(body)
                f.write(trailer)
        del pieces
        # ---- the CPU twin on one piece: the expected table / score sum (the file is the piece repeated, record aligned) and the host figure
        cfg = O.make_config(buffer_capacity=64 * 1024, batch_size=PIPE_BATCH)
        cores = max(1, os.cpu_count() or 1)
        recs_piece = k // rec_bytes
        per = max(64, min(recs_piece // cores, 4096))          # records per thread: a bounded sample (NW is ~6000 cell updates per read)
        slices = [piece[(i * per % (recs_piece - per)) * rec_bytes:][:per * rec_bytes] for i in range(cores)]
        outs = [None] * cores

        def work(i):
            outs[i] = O.pipeline_run(slices[i], cfg, REF_40BP, read_len)
        best_cpu = None
        for _ in range(3):
            th = [threading.Thread(target=work, args=(i,)) for i in range(cores)]
            t0 = time.perf_counter()
            for t in th: t.start()
            for t in th: t.join()
            dt = time.perf_counter() - t0
            best_cpu = dt if best_cpu is None or dt < best_cpu else best_cpu
        assert all(o[0] == per for o in outs)
        t0 = time.perf_counter()
        n1, counts1, ss1 = O.pipeline_run(piece, cfg, REF_40BP, read_len)    # (one core over the whole piece: the expected values)
        one_core_s = time.perf_counter() - t0
        assert n1 == recs_piece
        # ---- the device pipeline
        ctx = B.Context(B.ParserConfig(), "generic", PIPE_BATCH, local_rank, min_record_bytes=256 if rec_bytes >= 256 else 32)
        # The consumer stream is created in the HIGHEST priority class (the parser's).  The runtime maps streams onto a few hardware
        # queues per class and WHICH queue a stream gets depends on how many streams the process created before it: in the full default
        # line a default-class consumer stream gave 37-38 GB/s where the same code alone (--pipeline-only) gave 43 -- same box, A/B
        # (profiles/RESULTS.md, round 6); in the highest class it is 43 in both.  BZQ_PIPE_SIDE_PRIO=0: the default class.
        side = torch.cuda.Stream(device=dev, priority=int(os.environ.get("BZQ_PIPE_SIDE_PRIO", "-1")))
        ctx.set_consumer_stream(side.cuda_stream)
        # the two-chunk lifetime rule (chunk k - 1's consumers through before chunk k + 1 is submitted) is kept by the library on the
        # device (option consumer_guard: the host never blocks for it); BZQ_PIPE_HOST_EVENTS=1: by a host-side event wait instead (A/B)
        host_events = os.environ.get("BZQ_PIPE_HOST_EVENTS", "0") == "1"
        ctx.set_option("consumer_guard", 0 if host_events else 1)
        d_ref = torch.frombuffer(bytearray(REF_40BP), dtype=torch.uint8).to(dev)
        d_counts = torch.zeros(read_len * 128, dtype=torch.int64, device=dev)
        d_scores = torch.empty(n_rec, dtype=torch.int32, device=dev)
        nb_cap = (chunk_mib << 20) // rec_bytes // PIPE_BATCH + 8
        arr = (L.BzqDeviceBatch * nb_cap)()
        nb_out = C.c_uint64()
        def run_kind(path):
            best = None
            for it in range(4):          # run 0: the file's first read + first-use costs, not reported
                d_counts.zero_()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                ing = B.Ingest(ctx, path, chunk_bytes=chunk_mib << 20, n_threads=threads)
                taken = total = chunks = 0
                prev_ev = None
                t_cons = 0.0
                while True:
                    r = ing.next(taken)
                    taken = int(r.n_records)
                    if host_events and prev_ev is not None:
                        prev_ev.synchronize()    # two-chunk lifetime rule by hand: chunk k - 1's consumers are through before chunk k + 1 is submitted
                    assert L.lib().bzq_batches(ctx.h, PIPE_BATCH, arr, nb_cap, C.byref(nb_out)) == 0 and nb_out.value <= nb_cap
                    for b in range(nb_out.value):
                        assert L.lib().bzq_batch_nw_scores_dev(ctx.h, C.byref(arr[b]), C.c_void_p(d_ref.data_ptr()), len(REF_40BP),
                                                               C.c_void_p(d_scores.data_ptr() + 4 * (total + b * PIPE_BATCH))) == 0
                        assert L.lib().bzq_batch_quality_by_position_acc(ctx.h, C.byref(arr[b]), read_len, C.c_void_p(d_counts.data_ptr())) == 0
                    if host_events:
                        prev_ev = torch.cuda.Event(); prev_ev.record(side)
                    total += taken
                    chunks += 1
                    if int(r.status) != L.OK:
                        break
                ing.close()
                side.synchronize()
                dt = time.perf_counter() - t0
                assert total == n_rec and int(r.status) == L.EOF, (total, n_rec, int(r.status))
                if it and (best is None or dt < best):
                    best = dt
            return best, chunks

        def check_outputs(kind):
            # parity on the consumer outputs of the kind's last run: the whole file's table and score sum
            got = d_counts.cpu().numpy().reshape(read_len, 128).astype(np.uint64)
            assert np.array_equal(got, counts1 * np.uint64(reps)), f"pipeline_mode ({kind}): per-position quality table differs from the CPU twin"
            assert int(d_scores.to(torch.int64).sum().item()) == ss1 * reps, f"pipeline_mode ({kind}): NW scores differ from the CPU twin"

        best, chunks = run_kind(paths["plain"])
        check_outputs("plain")
        for m in kinds:
            if m == "plain":
                continue
            if m == "gzip":   # (a .gz stream's pools and FIFO are ~10 GiB: let the library's cache keep them between the runs, as ingest_mode does)
                ctx.set_option("dev_cache_bytes", 16 << 30)
            bm, cm = run_kind(paths[m])
            check_outputs(m)
            res[m] = {"value": round(n_fastq / bm / 1e9, 2), "unit": "GB/s of FASTQ", "ms": round(bm * 1e3, 1), "chunks": cm,
                      "file_gb": round(os.path.getsize(paths[m]) / 1e9, 3), "consumer_outputs_equal_cpu_twin": True}
        for key in ("pin_cache_bytes", "dev_cache_bytes"):   # (give the cached buffers back, and the library's default limits again)
            ctx.set_option(key, 0)
            ctx.set_option(key, 1 << 30)
        ctx.close()
        cpu_bytes = cores * per * rec_bytes
        res.update({"value": round(n_fastq / best / 1e9, 2), "unit": "GB/s of FASTQ", "mrecords_per_s": round(n_rec / best / 1e6, 1), "ms": round(best * 1e3, 1),
                    "chunks": chunks, "malignments_per_s": round(n_rec / best / 1e6, 1),
                    "consumer_outputs_equal_cpu_twin": True,
                    "cpu_all_cores": {"value": round(cpu_bytes / best_cpu / 1e9, 3), "unit": "GB/s of FASTQ", "cores": cores, "kind": "port",
                                      "sample": f"{cores} threads, each orc_pipeline_run (streaming parser batches({PIPE_BATCH}) + NW score of every read vs the 40 bp reference + "
                                                f"per-position quality table) over its own {per} records of the same reads, in memory, best of 3"},
                    "cpu_1core": {"value": round(k / one_core_s / 1e9, 4), "unit": "GB/s of FASTQ", "sample": f"orc_pipeline_run over one {k} B piece of the file"},
                    "speedup_vs_cpu_all_cores": round((n_fastq / best) / (cpu_bytes / best_cpu), 2),
                    "note": "file on /dev/shm -> bzq_ingest_next -> batches(65536) -> bzq_batch_nw_scores_dev + bzq_batch_quality_by_position_acc on the consumer stream "
                            "(no host round trip; chunk k's consumers under chunk k+1's ingest; the lifetime rule kept on the device by option consumer_guard); wall clock open -> last consumer done, best of 3 after one untimed pass; "
                            "PCIe inclusive; the CPU twin works from memory (no file read)"})
    finally:
        for q in paths.values():
            try:
                os.remove(q)
            except OSError:
                pass
    return res


def process_mode(ctx, dev, runs=15, warmup=3, modes=("plain", "bgzf", "gzip")):
    """A WHOLE PROCESS per run, the way the reference times itself (benchmark/throughput/run_throughput_benchmarks.sh:54-62, 141-149:
    hyperfine --warmup 3 --runs 15 around `run_throughput_blazeseq <file> batches` on a 3 GiB file of 14.7 M x 100 bp reads on a
    RAM-backed filesystem; published: batches 4.03, views 5.13 GB/s on the authors' host).  The process is the plain-C driver
    tests/c_driver/bzq_throughput (exec -> bzq_create -> bzq_ingest_open -> every chunk, every batch of 4096 handed out ->
    `records base_pairs` printed -> exit), the clock is around the subprocess: dynamic loading, HIP initialisation, pinning, the
    file, teardown -- everything.  plain = the reference generator's own reads; .bgz / .gz = the first 32 MiB of them, compressed
    once (zlib -6) and repeated to the same size (compressing 3 GiB on one host core would take minutes)."""
    import re
    import statistics
    import struct
    import subprocess
    import zlib
    import numpy as np
    import torch
    exe = os.path.join(ROOT, "tests", "c_driver", "bzq_throughput")
    if not os.path.exists(exe):
        return {"error": "tests/c_driver/bzq_throughput is not built (python __graft_entry__.py)"}
    reads, read_len = 14_700_000, 100
    n = ctx.generate_synthetic_device(reads, read_len, 33, 73, "generic", 0, 0, first=0, count=reads, max_len=read_len)
    buf = torch.empty(n + (1 << 20), dtype=torch.uint8, device=dev)
    ctx.generate_synthetic_device(reads, read_len, 33, 73, "generic", buf.data_ptr(), buf.numel(), first=0, count=reads, max_len=read_len)
    torch.cuda.synchronize()
    host = buf[:n].cpu().numpy()
    del buf
    rec_bytes = n // reads
    d = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else "/tmp"
    tag = f"bzq_proc_{os.getpid()}"
    paths = {m: os.path.join(d, f"{tag}.fastq{ext}") for m, ext in (("plain", ""), ("bgzf", ".bgz"), ("gzip", ".gz"))}
    res = {"file_fastq_gb": round(n / 1e9, 3), "records": reads, "read_len": read_len, "runs": runs, "warmup_runs": warmup, "dir": d,
           "driver": "tests/c_driver/bzq_throughput <file> batches (plain C over the C ABI; 256 MiB chunks, 8 reader threads)",
           "reference_published_gb_s": {"batches": 4.03, "views": 5.13, "records": 2.16, "note": "BASELINE.md: the authors' host, their 3 GiB / 100 bp file; context, not a baseline measured here"},
           "note": "wall clock around a fresh process per run (subprocess: exec, library loading, hipInit, bzq_create, open, every chunk and batch, print, teardown, exit); "
                   "value = file's FASTQ bytes / mean wall; min / max / stdev over the timed runs beside it"}
    try:
        host.tofile(paths["plain"])
        k = (32 << 20) // rec_bytes * rec_bytes
        pbytes = host[:k].tobytes()
        reps = n // k
        expect = {"plain": (reads, reads * read_len), "bgzf": (reps * (k // rec_bytes), reps * (k // rec_bytes) * read_len)}
        expect["gzip"] = expect["bgzf"]

        def block(data):
            c2 = zlib.compressobj(6, zlib.DEFLATED, -15)
            payload = c2.compress(data) + c2.flush()
            return (b"\x1f\x8b\x08\x04" + b"\0\0\0\0" + b"\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 18 + len(payload) + 8 - 1) + payload +
                    struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))
        if "bgzf" in modes:
            bg = b"".join(block(pbytes[i:i + 65280]) for i in range(0, k, 65280))
            with open(paths["bgzf"], "wb") as f:
                for _ in range(reps):
                    f.write(bg)
                f.write(block(b""))
        if "gzip" in modes:
            co = zlib.compressobj(6, zlib.DEFLATED, -15)
            member = bytes([0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 3]) + co.compress(pbytes) + co.flush() + struct.pack("<II", zlib.crc32(pbytes) & 0xFFFFFFFF, k & 0xFFFFFFFF)
            with open(paths["gzip"], "wb") as f:
                for _ in range(reps):
                    f.write(member)
        del host
        for m in modes:
            fastq_bytes = n if m == "plain" else reps * k
            walls, parts = [], []
            env_t = dict(os.environ, BZQ_THROUGHPUT_TIMES="1")   # (one line on stderr at the very end of main: where the process's time went)
            for it in range(warmup + runs):
                t0 = time.perf_counter()
                r = subprocess.run([exe, paths[m], "batches"], capture_output=True, text=True, timeout=120, env=env_t)
                dt = time.perf_counter() - t0
                if r.returncode != 0:
                    raise RuntimeError(f"{m}: bzq_throughput exit {r.returncode}: {r.stderr[-300:]}")
                got = tuple(int(x) for x in r.stdout.split())
                assert got == expect[m], (m, got, expect[m])
                if it >= warmup:
                    walls.append(dt)
                    mt = re.search(r"create ([\d.]+) ms, open ([\d.]+) ms, first chunk ([\d.]+) ms, remaining \d+ chunks ([\d.]+) ms, close\+destroy ([\d.]+) ms, main total ([\d.]+) ms", r.stderr)
                    if mt:
                        v = [float(x) for x in mt.groups()]
                        parts.append(v[:5] + [dt * 1e3 - v[5]])
            mean = statistics.fmean(walls)
            res[m] = {"value": round(fastq_bytes / mean / 1e9, 2), "unit": "GB/s of FASTQ, whole process", "mean_ms": round(mean * 1e3, 1), "stdev_ms": round(statistics.pstdev(walls) * 1e3, 1),
                      "min_ms": round(min(walls) * 1e3, 1), "max_ms": round(max(walls) * 1e3, 1), "best_gb_s": round(fastq_bytes / min(walls) / 1e9, 2),
                      "file_gb": round(os.path.getsize(paths[m]) / 1e9, 3), "stdout": " ".join(str(x) for x in got)}
            if m == "plain":   # the same process leaving at once behind its result line (no close / destroy / runtime teardown): beside `value`, never in its place
                fw = []
                env_f = dict(os.environ, BZQ_THROUGHPUT_FAST_EXIT="1")
                for it in range(warmup + runs):
                    t0 = time.perf_counter()
                    r = subprocess.run([exe, paths[m], "batches"], capture_output=True, text=True, timeout=120, env=env_f)
                    dt = time.perf_counter() - t0
                    assert r.returncode == 0 and tuple(int(x) for x in r.stdout.split()) == expect[m]
                    if it >= warmup:
                        fw.append(dt)
                fm = statistics.fmean(fw)
                res[m]["fast_exit"] = {"value": round(fastq_bytes / fm / 1e9, 2), "mean_ms": round(fm * 1e3, 1), "stdev_ms": round(statistics.pstdev(fw) * 1e3, 1),
                                       "min_ms": round(min(fw) * 1e3, 1), "note": "BZQ_THROUGHPUT_FAST_EXIT=1: _exit(0) behind the result line"}
            if parts:   # mean milliseconds of the timed runs: bzq_create is HIP's start-up; outside_main = exec, dynamic loading, the runtime's teardown
                res[m]["where_ms"] = {kname: round(statistics.fmean(pv[i] for pv in parts), 1)
                                      for i, kname in enumerate(("bzq_create_hip_start_up", "ingest_open", "first_chunk", "remaining_chunks", "close_and_destroy", "outside_main"))}
    finally:
        for q in paths.values():
            try:
                os.remove(q)
            except OSError:
                pass
    return res


def crlf_variant(shard, reads, rec_bytes, id_bytes):
    """The same reads with DOS line ends (the reference corpus' example_dos.fastq at scale): '\\r' before each of a record's four
    newlines.  Record layout of the synthetic input: '@' id '\\n' seq '\\n' '+' '\\n' qual '\\n'."""
    import torch
    L = (rec_bytes - id_bytes - 6) // 2
    m = shard[:reads * rec_bytes].view(reads, rec_bytes)
    cuts = [1 + id_bytes, 1 + id_bytes + 1 + L, 1 + id_bytes + 1 + L + 2, rec_bytes - 1]   # positions of the four newlines
    out = torch.empty((reads, rec_bytes + 4), dtype=torch.uint8, device=shard.device)
    src = dst = 0
    for c in cuts:
        out[:, dst:dst + (c - src)] = m[:, src:c]
        dst += c - src
        out[:, dst] = 13
        out[:, dst + 1] = 10
        dst += 2
        src = c + 1
    return out.view(-1)


def illumina_variant(shard, reads, rec_bytes, id_bytes):
    """The same bases and qualities under headers as a sequencer / the SRA writes them: variable-length, with interior spaces
    ('@SRR001666.<i+1> 071112_SLXA-EAS1_s_7:5:1:<x>:<y> length=150') -- no fixed record stride, ids of 45..58 bytes."""
    import torch
    dev = shard.device
    Lb = (rec_bytes - id_bytes - 6) // 2
    m = shard[:reads * rec_bytes].view(reads, rec_bytes)
    i = torch.arange(reads, device=dev, dtype=torch.int64)
    fields = [(b"@SRR001666.", None), (None, i + 1), (b" 071112_SLXA-EAS1_s_7:5:1:", None), (None, (i * 7919) % 1000), (b":", None),
              (None, (i * 104729) % 1000), (f" length={Lb}\n".encode(), None)]
    W = 80
    hdr = torch.zeros((reads, W), dtype=torch.uint8, device=dev)
    pos = torch.zeros(reads, dtype=torch.int64, device=dev)
    rows = torch.arange(reads, device=dev)
    for lit, num in fields:
        if lit is not None:
            for b in lit:
                hdr[rows, pos] = b
                pos += 1
        else:
            nd = torch.ones_like(num)
            for p in range(1, 9):
                nd += (num >= 10 ** p).to(torch.int64)
            for p in range(8, -1, -1):   # most significant digit first
                has = nd > p
                digit = (num // (10 ** p)) % 10 + 48
                hdr[rows[has], pos[has]] = digit[has].to(torch.uint8)
                pos += has.to(torch.int64)
    body = m[:, 1 + id_bytes + 1:]   # seq '\n' '+' '\n' qual '\n'
    keep = torch.arange(W + body.shape[1], device=dev).unsqueeze(0)
    parts = []
    for r0 in range(0, reads, 1 << 20):   # (a masked select over more than 2^31 elements overflows torch's index arithmetic)
        r1 = min(reads, r0 + (1 << 20))
        full = torch.cat([hdr[r0:r1], body[r0:r1]], dim=1)
        parts.append(full[(keep < pos[r0:r1].unsqueeze(1)) | (keep >= W)])
    return torch.cat(parts)


def fasta_main(args, world, rank, local_rank, dev, distributed, native_comm, dist_dev):
    """FastaParser over benchmark/fasta-parser/generate_synthetic_fasta.mojo's input
    (200-3800 bp, line width 60).  Records are independent, so ranks take equal record ranges of one synthetic file
    (weak scaling, no data-path collective); a step = one parse of the rank's resident shard."""
    import torch
    import torch.distributed as dist
    import blazeseq_amd as B
    per = 1_500_000 if args.reads == 10_000_000 else args.reads
    total = per * world
    ctx = B.FastaContext(B.FastaParserConfig(check_ascii=args.validate), local_rank)
    sharded, comm_ctx, expect = None, None, per
    if distributed:
        # byte-range shards: the rank's records, cut SHIFT bytes into a record on both sides -- its head goes to
        # the rank before it, the head of the rank behind it arrives as halo.  The library's own communicator; torch hands the id around.
        SHIFT = 144
        try:
            comm_ctx = B.Context(B.ParserConfig(), device=local_rank)
            native_comm(comm_ctx)   # (with one checked ring exchange: a transport that does not work shows here, not inside the timed steps)
            sharded = "byte ranges"
        except Exception as e:   # noqa: BLE001 -- said out loud in the JSON line, never silent
            sharded = None
            print(f"[bench] rank {rank}: native communicator failed ({str(e)[:200]}); FASTA falls back to a split by record", file=sys.stderr)
        if world > 1:
            flags = [None] * world
            dist.all_gather_object(flags, sharded)
            sharded = "byte ranges" if all(f == "byte ranges" for f in flags) else None
    if sharded:
        last = rank + 1 == world
        buf = ctx.generate_synthetic_device(total, 200, 3800, 60, first=rank * per, count=per + (0 if last else 1))
        n_own = buf.numel() if last else int(ctx.generate_synthetic_device(total, 200, 3800, 60, first=rank * per, count=per).numel())
        lo, hi = (SHIFT if rank else 0), n_own + (0 if last else SHIFT)
        room = torch.empty(buf.numel() + (1 << 20), dtype=torch.uint8, device=dev)   # the halo lands behind the range
        room[:buf.numel()] = buf
        torch.cuda.synchronize()   # torch's stream wrote it, the library's stream reads it
        shard, n = room[lo:], hi - lo
        cap = shard.numel()
        if world > 1:   # a record belongs to the rank its header line starts in: record rank*per starts SHIFT bytes before the cut
            expect = per + 1 if rank == 0 else (per - 1 if last else per)

        def step():
            r = ctx.shard_stitch(comm_ctx, int(shard.data_ptr()), n, cap)
            assert int(r.stream_status) == 6 and int(r.global_records) == total, (r.stream_status, r.global_records)
            return r.chunk
    else:
        shard = ctx.generate_synthetic_device(total, 200, 3800, 60, first=rank * per, count=per)
        n = shard.numel()

        def step():
            return ctx.parse(int(shard.data_ptr()), n, True)
    for _ in range(args.warmup):
        res = step()
    t_w = time.perf_counter()
    while args.warmup and time.perf_counter() - t_w < args.min_seconds and not distributed:   # clocks settle
        res = step()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kernel_ms = 0.0
    for _ in range(args.steps):
        res = step()
        kernel_ms += res.kernel_ms
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    dt = time.perf_counter() - t0
    assert int(res.status) == 6 and int(res.n_records) == expect, (res.status, res.n_records)
    stats = torch.tensor([dt, float(n), float(res.seq_bytes), float(res.id_bytes)], dtype=torch.float64, device=dist_dev)
    if distributed:
        mx = stats.clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = stats.clone(); dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        dt, tot_n, tot_seq, tot_id = float(mx[0]), float(sm[1]), float(sm[2]), float(sm[3])
    else:
        tot_n, tot_seq, tot_id = float(n), float(res.seq_bytes), float(res.id_bytes)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return
    step_s = dt / args.steps
    A = n + int(res.seq_bytes) + int(res.id_bytes) + 16 * per          # this rank: input once + columns + two ends arrays
    k_s = kernel_ms / args.steps / 1e3
    out = {
        "metric": "FASTA GB/s (200-3800 bp, line width 60, synthetic) vs HBM roofline", "value": round(tot_n / step_s / 1e9, 3), "unit": "GB/s",
        "mrecords_per_s": round(total / step_s / 1e6, 3), "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(step_s * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
        "data": "synthetic (generate_synthetic_fasta_buffer restated on the device), resident in HBM",
        "config": {"workload": f"FASTA, {per} records/GPU of 200-3800 bp wrapped at 60, check_ascii={bool(args.validate)}; "
                               "FastaParser over the whole shard per step",
                   "parallelism": (f"byte-range shards over {world} rank(s), cut 144 bytes into a record (bzq_fasta_shard_stitch over RCCL)"
                                   if sharded else f"records split over {world} rank(s)")},
        "roofline": {"bound": "hbm", "achieved": round(A / k_s / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(A / k_s / 1e9 / HBM_PEAK_GBS, 4), "traffic": None,
                     "note": "all FASTA kernels of one step (tile sums + resolve + scan + emit + finish), hipEvent time on the handle's stream; "
                             "algorithmic bytes = input + sequence and id columns + 16 B/record",
                     "algorithmic_gb_per_step": round(A / 1e9, 3), "kernels_ms": round(k_s * 1e3, 4)},
    }
    if world == 1 and not args.no_cpu_baseline:
        from oracle import fasta as F
        import ctypes as C
        import numpy as np
        host = shard.cpu().numpy()
        k = host.size // 8            # ~0.4 GB of the same bytes, cut at a record start
        k = int(np.flatnonzero(host[k:k + 8192] == ord(">"))[0]) + k
        sample = np.ascontiguousarray(host[:k])
        cnt = (C.c_int64 * 2)()
        times = []
        t_start = time.perf_counter()
        while time.perf_counter() - t_start < 12.0 and len(times) < 10:
            t1 = time.perf_counter()
            st = F.lib().fa_bench_run(sample.ctypes.data, sample.size, cnt)
            times.append(time.perf_counter() - t1)
        assert st == 6
        best = sum(times) / len(times)
        out["cpu_baseline"] = {"value": round(sample.size / best / 1e9, 4), "unit": "GB/s", "cores": 1, "kind": "port",
                               "sample": f"the first {int(cnt[0])} records ({sample.size} B) of the GPU input, oracle/fasta_oracle.c flat parse "
                                         f"(memchr per line + strip + append), mean of {len(times)} runs, in-memory"}
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--reads", type=int, default=0, help="reads per GPU (default: 10 M at --gpus 1 = BASELINE config 2; 78.125 M at --gpus N > 1 = config 5's shard)")
    ap.add_argument("--min-seconds", type=float, default=0.5, help="warm up for at least this long (and at least --warmup steps): clocks settle")
    ap.add_argument("--exchange", choices=["native", "torch"], default="native",
                    help="sharded mode: bzq_shard_stitch over the library's own RCCL binding (default) or the torch.distributed cross-check")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--validate", action="store_true", help="config 3: check_ascii + check_quality, sanger")
    ap.add_argument("--views", action="store_true",
                    help="views() mode: RecordOffsets + id spans into the chunk, no columns (A = B + 52 bytes/record)")
    ap.add_argument("--long-reads", action="store_true",
                    help="config 4: 300 000 reads/GPU of 200..19 800 bases (phred 5..30, sanger); record-aligned shards")
    ap.add_argument("--pass-bytes", type=int, default=0)
    ap.add_argument("--single-pass", action="store_true", help="one fused launch with in-kernel look-back (slower today)")
    ap.add_argument("--service", action="store_true", help="one launch, prefix-service workgroup (bzq_single.hpp)")
    ap.add_argument("--hier", action="store_true", help="one launch, two-level decoupled look-back (bzq_single.hpp)")
    ap.add_argument("--two-pass", action="store_true", help="(default) aggregate + scan + emit kernels")
    ap.add_argument("--stream", action="store_true", help="EXPERIMENTS build: k_stream, one read of the input (profiles/r2_single_read.md)")
    ap.add_argument("--kernels-v1", action="store_true", help="two-pass mode with the first-generation kernels")
    ap.add_argument("--overlap", type=int, default=0, help="overlap pass A of sub-chunk k+1 with the emit of sub-chunk k (needs --pass-bytes)")
    ap.add_argument("--force-sharded", action="store_true", help="run the multi-GPU shard protocol even with one rank")
    ap.add_argument("--ranks-on-one-gpu", type=int, default=0, metavar="N",
                    help="harness check on a one-GPU box: the N ranks torch.distributed.run started all use device 0, the library's "
                         "shared-memory transport (bzq_comm_init_shm) replaces RCCL (which refuses two ranks on one GPU) and gloo "
                         "carries the torch side.  Every world > 1 branch of this file and bzq_shard_stitch with real neighbours run; "
                         "the figure it prints is N ranks sharing ONE GPU, not a scaling point.  Pass --reads (the default shard is 25 GB).")
    ap.add_argument("--no-extra-modes", action="store_true", help="skip the validated / long-reads / views / FASTA figures of the default line")
    ap.add_argument("--ablate", type=int, default=0, help="timing experiments only (results are wrong)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fasta", action="store_true",
                    help="the FASTA path on the reference's FASTA benchmark input (200-3800 bp, line width 60), 1.5 M records/GPU")
    ap.add_argument("--cpu-reads", type=int, default=10_000_000, help="CPU baseline sample: the same 10 M-read workload by default (~15 s of CPU work)")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE", help="bzq_set_option on the headline ctx (A/B runs: --opt fold_rebase=0)")
    ap.add_argument("--from-file", action="store_true",
                    help="sharded mode (--gpus N, or --force-sharded): the synthetic stream is written to ONE file on /dev/shm and every step starts "
                         "at the file -- each rank reads its byte range through bzq_shard_read_range (reader threads -> pinned -> device), then "
                         "bzq_shard_stitch: file-chunk sharding end to end, PCIe inclusive (10 M reads per rank unless --reads says otherwise)")
    ap.add_argument("--reader-threads", type=int, default=8)
    ap.add_argument("--no-ingest-mode", action="store_true", help="skip the file -> records figures (ingest_mode) of the default line")
    ap.add_argument("--no-process-mode", action="store_true", help="skip process_mode (a fresh process per run over the reference's own 3 GiB workload, 3 + 15 runs x 3 file kinds: ~40 s)")
    ap.add_argument("--process-only", action="store_true", help="print only process_mode")
    ap.add_argument("--ingest-chunk-mib", type=int, default=256, help="chunk size of the ingest_mode runs")
    ap.add_argument("--ingest-threads", type=int, default=8, help="reader threads of the ingest_mode runs")
    ap.add_argument("--pipeline-only", action="store_true", help="print only pipeline_mode (file -> records -> device consumers)")
    ap.add_argument("--ingest-only", action="store_true", help="print only ingest_mode (sweeps of the two options above)")
    ap.add_argument("--synchronous", action="store_true", help="single GPU: time only the synchronous steps (submit -> result -> batches), no submit of chunk k + 1 in front of chunk k's batches")
    ap.add_argument("--fail-rank", type=int, default=-1, help="harness check: this rank exits with code 3 before it joins anything (the launcher must name it)")
    ap.add_argument("--launch-deadline", type=float, default=3600.0, help="self-launched ranks (--gpus N without a launcher): stop everything after this many seconds")
    args = ap.parse_args()

    if args.gpus > 1 and int(os.environ.get("WORLD_SIZE", "1")) == 1 and "BZQ_BENCH_LAUNCHER_PID" not in os.environ:
        # the plain command `python3 bench.py --gpus N ...`: start the N ranks here (torch.distributed.run remains welcome: it sets WORLD_SIZE)
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:], deadline_s=args.launch_deadline))

    import torch
    import torch.distributed as dist
    import blazeseq_amd as B
    from blazeseq_amd import _lib as L
    from blazeseq_amd import sharded

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if args.fail_rank == rank and world > 1:
        print(f"[bench rank {rank}] --fail-rank: exiting with code 3", file=sys.stderr, flush=True)
        sys.exit(3)
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE is {world}: the launcher (python -m torch.distributed.run, or this file's own) must start exactly --gpus ranks")
    one_gpu = args.ranks_on_one_gpu > 0
    if one_gpu:
        if args.ranks_on_one_gpu != world:
            raise SystemExit(f"--ranks-on-one-gpu {args.ranks_on_one_gpu} needs exactly that many ranks (WORLD_SIZE is {world})")
        if args.reads == 0:
            raise SystemExit("--ranks-on-one-gpu: pass --reads (the default 25 GB shard per rank does not fit N times on one GPU)")
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist_dev = torch.device("cpu") if one_gpu else dev   # where the tensors of torch's own collectives live (gloo: host)
    sharded_mode = world > 1 or args.force_sharded
    if sharded_mode:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    def native_comm(c):
        """The library's own communicator on ctx `c`: RCCL (the id travels through torch), or shared memory with --ranks-on-one-gpu."""
        if one_gpu:
            box = [f"bench{os.getpid()}_{int(time.time() * 1e3) % 1000000}" if rank == 0 else None]
            if world > 1:
                dist.broadcast_object_list(box, src=0)
            c.comm_init_shm(rank, world, box[0], 4 << 20)
        else:
            box = [B.Context.comm_unique_id() if rank == 0 else None]
            if world > 1:
                dist.broadcast_object_list(box, src=0)
            c.comm_init(rank, world, box[0])
        c.comm_selftest()   # one checked ring exchange (send/recv + all-gather) before anything is timed

    config5 = world > 1 and not args.long_reads and not args.fasta
    default_reads = args.reads == 0
    if args.reads == 0:
        args.reads = 78_125_000 if (config5 and not args.from_file) else 10_000_000
    if args.fasta:
        return fasta_main(args, world, rank, local_rank, dev, sharded_mode, native_comm, dist_dev)

    cfg = B.ParserConfig(check_ascii=args.validate, check_quality=args.validate,
                         quality_schema="sanger" if args.validate else None, views_only=args.views)
    # min_record_bytes only sizes the per-record arrays (8 B per possible record; library default 32): 256 for reads >= 100 bp
    # keeps them at n / 256 entries instead of n / 32.  Named in config below.
    min_record_bytes = 256 if not args.long_reads and args.read_len >= 100 else 32
    ctx = B.Context(cfg, "generic", 4096, local_rank, pass_bytes=args.pass_bytes, min_record_bytes=min_record_bytes)
    ctx.set_option("timing_detail", 1)
    if args.process_only:   # (sweeps: only the whole-process figures)
        print(json.dumps({"process_mode": process_mode(ctx, dev)}))
        return
    if args.hier or args.service or args.single_pass or args.kernels_v1:   # EXPERIMENTS build only (BLAZESEQ_HIP_LIB=.../libblazeseq_hip_exp.so)
        ctx.set_option("single_pass", 3 if args.hier else 2 if args.service else (1 if (args.single_pass and not args.kernels_v1) else 0))
        ctx.set_option("kernels_v2", 0 if args.kernels_v1 else 1)
    if args.stream:
        ctx.set_option("stream", 1)
    if args.ablate:
        ctx.set_option("ablate", args.ablate)
    ctx.set_option("overlap", args.overlap)
    ctx.set_option("timing_detail", 0 if args.overlap else 1)
    for kv in args.opt:
        key, val = kv.split("=", 1)
        ctx.set_option(key, int(val))

    # ---- synthetic input, generated on the device (record i depends only on i) ------------------
    if args.long_reads and args.reads == 10_000_000:
        args.reads = 300_000
    total_reads = args.reads * world
    gen = (dict(read_len=200, max_len=19_800, min_phred=5, max_phred=30, schema="sanger") if args.long_reads else
           dict(read_len=args.read_len, max_len=None, min_phred=33, max_phred=73, schema="generic"))

    def generate(d_out=0, cap=0, first=0, count=None):
        return ctx.generate_synthetic_device(total_reads, gen["read_len"], gen["min_phred"], gen["max_phred"], gen["schema"],
                                             d_out, cap, first=first, count=count, max_len=gen["max_len"])

    slack = 1 << 20  # room for the halo (one record) behind the shard
    SHIFT = 144      # interior cuts of the byte-range shards sit 144 bytes into a record: never record aligned
    if args.long_reads:
        # variable record sizes: every rank takes an equal number of records (record-aligned shards)
        total_bytes = generate(count=total_reads)
        lo = generate(count=rank * args.reads) if rank else 0
        n = generate(first=rank * args.reads, count=args.reads)
        rec_bytes = n // args.reads   # mean, for the workload description only
        buf = torch.empty(n + slack, dtype=torch.uint8, device=dev)
        generate(buf.data_ptr(), buf.numel(), first=rank * args.reads, count=args.reads)
        shard = buf
    else:
        rec_bytes = generate(count=1)
        total_bytes = rec_bytes * total_reads
        shift = SHIFT if (world > 1 and rec_bytes > 2 * SHIFT) else 0
        lo = total_bytes * rank // world + (shift if rank else 0)
        hi = total_bytes * (rank + 1) // world + (shift if rank + 1 < world else 0)
        n = hi - lo
        # records that overlap [lo, hi + one record): generated in place, the shard is a 16-byte aligned view of them
        i0, i1 = lo // rec_bytes, min(total_reads, (hi + rec_bytes - 1) // rec_bytes + 1)
        off = lo - i0 * rec_bytes
        pad = (-off) % 16
        buf = torch.empty(pad + (i1 - i0) * rec_bytes + slack, dtype=torch.uint8, device=dev)
        ctx.generate_synthetic_device(total_reads, args.read_len, 33, 73, "generic", buf.data_ptr() + pad, buf.numel() - pad,
                                      first=i0, count=i1 - i0)
        shard = buf[pad + off:]
        assert shard.data_ptr() % 16 == 0 and shard.numel() >= n + slack
    torch.cuda.synchronize()

    if args.pipeline_only:
        print(json.dumps({"pipeline_mode": pipeline_mode(shard, rec_bytes, dev, local_rank, threads=args.ingest_threads, chunk_mib=args.ingest_chunk_mib)}))
        return
    if args.ingest_only:   # (sweeps of --ingest-chunk-mib / --ingest-threads: only the file -> records figures)
        print(json.dumps({"ingest_mode": ingest_mode(shard, rec_bytes, dev, local_rank, threads=args.ingest_threads, chunk_mib=args.ingest_chunk_mib)}))
        return
    exchange = None
    if sharded_mode:
        exchange = args.exchange
        if exchange == "native":
            try:   # the library binds librccl itself; torch only hands the id around
                native_comm(ctx)
            except Exception as e:   # noqa: BLE001 -- said out loud in the JSON line, never silent
                exchange = f"torch (native communicator failed: {str(e)[:200]})"
                print(f"[bench] rank {rank}: {exchange}", file=sys.stderr)
        if world > 1:   # all ranks must take the same path
            flags = [None] * world
            dist.all_gather_object(flags, exchange)
            exchange = "native" if all(f == "native" for f in flags) else next(f for f in flags if f != "native")

    import ctypes as C

    file_path = None
    if args.from_file:
        if not sharded_mode or args.long_reads or exchange != "native":
            raise SystemExit("--from-file needs the sharded mode over the library's own communicator (--gpus N or --force-sharded), fixed-length reads")
        # ONE file for all ranks: rank 0 creates it, every rank writes its own byte range at its offset, then reads it back per step
        box = [os.path.join("/dev/shm" if os.path.isdir("/dev/shm") else "/tmp", f"bzq_bench_stream_{os.getpid()}.fastq") if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(box, src=0)
        file_path = box[0]
        if rank == 0:
            with open(file_path, "wb") as f:
                f.truncate(total_bytes)
        if world > 1:
            dist.barrier()
        host_bytes = shard[:n].cpu().numpy()
        fd = os.open(file_path, os.O_WRONLY)
        try:
            off = 0
            while off < n:
                off += os.pwrite(fd, host_bytes[off:off + (256 << 20)], lo + off)
        finally:
            os.close(fd)
        del host_bytes
        if world > 1:
            dist.barrier()

    def side_mode(d_buf, nbytes, nrec, validate, what, min_rec=None):
        """One more BASELINE configuration beside the headline: its own ctx, batch mode, every batch handed out; host-timed
        steps bracketed by synchronisations, plus the dominant kernel's roofline fraction from the ctx's HIP events."""
        c2 = B.Context(B.ParserConfig(check_ascii=validate, check_quality=validate, quality_schema="sanger" if validate else None),
                       "generic", 4096, local_rank, min_record_bytes=min_rec or min_record_bytes)
        c2.set_option("timing_detail", 1)
        arr = (L.BzqDeviceBatch * (nrec // 4096 + 2))()
        nout = C.c_uint64()

        def one():
            c2.submit_device(d_buf.data_ptr(), nbytes, 0, True)
            r = c2.result()
            assert L.lib().bzq_batches(c2.h, 4096, arr, len(arr), C.byref(nout)) == 0
            return r
        for _ in range(max(2, args.warmup)):
            r = one()
        torch.cuda.synchronize()
        t_w = time.perf_counter()
        while time.perf_counter() - t_w < min(0.25, args.min_seconds):
            r = one()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        me = mt = 0.0
        for _ in range(args.steps):
            r = one()
            me += r.ms_emit; mt += r.ms_total
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t1) / args.steps
        assert int(r.n_records) == nrec and r.status == L.EOF and nout.value == (nrec + 4095) // 4096, (r.n_records, r.status, c2.format_error())
        A2 = nbytes + int(r.seq_bytes) + int(r.qual_bytes) + int(r.id_bytes) + 16 * nrec
        o = {"value": round(nbytes / dt / 1e9, 3), "unit": "GB/s", "mrecords_per_s": round(nrec / dt / 1e6, 3), "ms_per_step": round(dt * 1e3, 4),
             "input_gb": round(nbytes / 1e9, 3), "kernels_ms": round(mt / args.steps, 4),
             "roofline_frac": round(A2 / (me / args.steps / 1e3) / 1e9 / HBM_PEAK_GBS, 4), "roofline_kernel": "k_fused<LB=false> (emit)",
             "roofline_path_frac": round(A2 / (mt / args.steps / 1e3) / 1e9 / HBM_PEAK_GBS, 4),
             # chunks parsed twice because pass A's "no id is stripped" hypothesis failed (all submits of this ctx, warm-up included)
             "hypothesis_retries": int(L.lib().bzq_set_option(c2.h, b"stream_fallbacks", 0)),
             "submits": int(L.lib().bzq_set_option(c2.h, b"n_submits", 0)), "note": what}
        c2.close()
        return o

    nb_cap = args.reads // 4096 + 2
    batch_arr = (L.BzqDeviceBatch * nb_cap)()     # the step hands out every DeviceFastqBatch of the chunk (batches(4096))
    nb_out = C.c_uint64()

    def step():
        if not sharded_mode:
            ctx.submit_device(shard.data_ptr(), n, 0, True)
            res = ctx.result()
            if not args.views:
                rc = L.lib().bzq_batches(ctx.h, 4096, batch_arr, nb_cap, C.byref(nb_out))
                assert rc == 0
            return res, None, None
        if file_path is not None:   # the rank's byte range of the file -> device -> stitch
            p_, n_, cap_ = ctx.shard_read_range(file_path, lo, hi, slack, args.reader_threads)
            sr = ctx.shard_stitch(p_, n_, cap_)
            return sr.chunk, [sr.global_records, sr.global_bases, sr.global_bytes], (sr.first_error_record if sr.first_error_record >= 0 else sharded.NO_ERROR)
        if exchange == "native":
            sr = ctx.shard_stitch(shard.data_ptr(), n, shard.numel())
            return sr.chunk, [sr.global_records, sr.global_bases, sr.global_bytes], (sr.first_error_record if sr.first_error_record >= 0 else sharded.NO_ERROR)
        res, plan, totals, first_err = sharded.parse_sharded(ctx, shard, n, lo)
        return res, totals, first_err

    # warm-up: at least --warmup steps AND at least --min-seconds of them (a 10 ms warm-up leaves the clocks unsettled);
    # in sharded mode every rank must run the same number of steps (the protocol is collective)
    warm_done = 0
    for _ in range(args.warmup):
        step(); warm_done += 1
    torch.cuda.synchronize()
    if warm_done:
        t_w = time.perf_counter(); step(); torch.cuda.synchronize(); one = max(1e-5, time.perf_counter() - t_w); warm_done += 1
        extra = int(min(5000, max(0.0, args.min_seconds) / one))
        if sharded_mode and world > 1:
            tx = torch.tensor([extra], dtype=torch.int64, device=dist_dev)
            dist.all_reduce(tx, op=dist.ReduceOp.MAX)
            extra = int(tx.item())
        for _ in range(extra):
            step(); warm_done += 1
    torch.cuda.synchronize()
    if sharded_mode:
        dist.barrier()
    # Single GPU: the K timed steps run the way a host walks a file under the double-buffer contract (include/blazeseq_hip.h: a chunk's
    # results stay valid until the SECOND following submit): take chunk k's result, submit chunk k + 1 AT ONCE, hand out chunk k's
    # batches under the parse of chunk k + 1.  K submits, K results, K x 2442 batch views inside the timed region, nothing skipped; the
    # same K steps with every step synchronous (submit -> result -> batches, the GPU idle while the host hands out the batches and
    # prepares the next submit) are timed right before and reported beside it (`synchronous`).
    sync_elapsed = None
    pipelined = not sharded_mode and not args.ablate and not args.overlap and not args.synchronous
    if pipelined:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        sync_elapsed = time.perf_counter() - t0
    ms_emit = ms_agg = ms_scan = ms_rebase = ms_kernels = 0.0
    totals = first_err = None
    t0 = time.perf_counter()
    if pipelined:
        ctx.submit_device(shard.data_ptr(), n, 0, True)
        for i in range(args.steps):
            res = ctx.result()
            if i + 1 < args.steps:
                ctx.submit_device(shard.data_ptr(), n, 0, True)
            if not args.views:
                rc = L.lib().bzq_batches(ctx.h, 4096, batch_arr, nb_cap, C.byref(nb_out))
                assert rc == 0
            ms_emit += res.ms_emit; ms_agg += res.ms_aggregate; ms_scan += res.ms_scan
            ms_rebase += res.ms_rebase; ms_kernels += res.ms_total
    else:
        for _ in range(args.steps):
            res, totals, first_err = step()
            ms_emit += res.ms_emit; ms_agg += res.ms_aggregate; ms_scan += res.ms_scan
            ms_rebase += res.ms_rebase; ms_kernels += res.ms_total
    torch.cuda.synchronize()
    if sharded_mode:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if sharded_mode:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dist_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # What the roofline's own event marks cost the headline: the same K pipelined steps once more with option timing_detail = 0 (two
    # events per submit instead of five; a mark between two kernels is ~6 us of an idle queue and more to query on the host:
    # scripts/step_gaps.py on a kernel trace).  Reported beside `value`, never in its place -- the roofline's kernel time needs the marks.
    plain_elapsed = None
    if pipelined:
        ctx.set_option("timing_detail", 0)
        for _ in range(2):
            ctx.submit_device(shard.data_ptr(), n, 0, True); ctx.result()
        torch.cuda.synchronize()
        tp = time.perf_counter()
        ctx.submit_device(shard.data_ptr(), n, 0, True)
        for i in range(args.steps):
            rp = ctx.result()
            if i + 1 < args.steps:
                ctx.submit_device(shard.data_ptr(), n, 0, True)
            if not args.views:
                assert L.lib().bzq_batches(ctx.h, 4096, batch_arr, nb_cap, C.byref(nb_out)) == 0
        torch.cuda.synchronize()
        plain_elapsed = time.perf_counter() - tp
        assert int(rp.n_records) == int(res.n_records) and rp.status == res.status
        ctx.set_option("timing_detail", 1)

    # ---- correctness guard on the timed configuration (size-independent properties) --------------
    recs = int(res.n_records)
    if args.ablate:
        global_records, global_bytes = args.reads, n
    elif not sharded_mode:
        assert recs == args.reads and res.status == L.EOF, (recs, res.status, ctx.format_error())
        assert args.views or (int(res.seq_bytes) == int(res.qual_bytes) and (args.long_reads or int(res.seq_bytes) == args.read_len * recs))
        if not args.views:   # every batch handed out: 4096 records each, the last one the remainder, seq_len = the batch's quality bytes
            assert nb_out.value == (recs + 4095) // 4096 and batch_arr[0].num_records == min(4096, recs)
            assert batch_arr[nb_out.value - 1].num_records == recs - 4096 * (nb_out.value - 1)
            assert args.long_reads or batch_arr[0].seq_len == args.read_len * batch_arr[0].num_records
        global_records, global_bytes = recs, n
    else:
        assert totals[0] == total_reads and first_err == sharded.NO_ERROR, (totals, first_err)
        global_records, global_bytes = totals[0], total_bytes

    rank_info = None
    if sharded_mode:
        # what the first 8-GPU run executes for the first time, said out loud: every rank's device, the GPU's NUMA node, how many CPUs its
        # reader threads are bound to -- and how many ranks' rows the library's own summary all-gather delivered (bzq_shard_stitch
        # stamps every row with its rank: `ranks_seen` must equal N)
        q = lambda key: int(L.lib().bzq_set_option(ctx.h, key, 0))
        mine = {"rank": rank, "device": q(b"device"), "numa_node": (lambda v: None if v == 255 else v)(q(b"numa_node")), "reader_cpus_bound": q(b"numa_cpus"),
                "reader_threads": args.reader_threads if file_path is not None else None, "ranks_seen": q(b"ranks_seen") if exchange == "native" else None}
        print(f"[bench rank {rank}] device {mine['device']}, NUMA node {mine['numa_node']}, reader threads bound to {mine['reader_cpus_bound']} CPUs, "
              f"ranks seen by the summary all-gather: {mine['ranks_seen']}", file=sys.stderr, flush=True)
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        rank_info = gathered
        dist.barrier()
        dist.destroy_process_group()
    if file_path is not None and rank == 0:
        try:
            os.remove(file_path)
        except OSError:
            pass
    if rank == 0:
        steps = args.steps
        folded = L.lib().bzq_set_option(ctx.h, b"last_folded", 0) == 1
        sec_per_step = elapsed / steps
        # algorithmic bytes of this rank's launch: input once + the three columns + ends/id_ends
        A_total = n + int(res.seq_bytes) + int(res.qual_bytes) + int(res.id_bytes) + 16 * recs
        if args.views:   # input once + five i64 offsets + id span (8 + 4) per record
            A_total = n + 52 * recs
        per_rank_records = recs
        A = A_total / max(1, recs)
        emit_s = ms_emit / steps / 1e3
        path_s = ms_kernels / steps / 1e3
        dom_ms, dom_bytes = ms_emit, A_total
        if args.views:
            # views mode: the dominant kernel is the line pass (k_tile_lines, the `aggregate` slot of the timing), which reads the
            # input once and writes 4 bytes per newline; the join (the `emit` slot) moves 20 + 52 bytes per record
            dom_ms, dom_bytes = ms_agg, n + 16 * recs
            emit_s = dom_ms / steps / 1e3
        out = {
            "metric": "FASTQ GB/s + Mrecords/s (150 bp synthetic) at 1/2/4/8 MI355X vs HBM roofline",
            "value": round(global_bytes / sec_per_step / 1e9, 3),
            "unit": "GB/s",
            "mrecords_per_s": round(global_records / sec_per_step / 1e6, 3),
            "n_gpus": world, "steps": steps, "warmup": args.warmup, "warmup_steps_run": warm_done,
            "ms_per_step": round(sec_per_step * 1e3, 4),
            "step": (("result(k) -> submit(k+1): chunk k+1 is submitted as soon as chunk k's result is taken (double-buffer contract; views mode hands out no batches); "
                      "K submits + K results inside the timed region" if args.views else
                      "result(k) -> submit(k+1) -> batches(k): every batch view of chunk k is handed out under the parse of chunk k+1 (double-buffer contract); "
                      "K submits + K results + K x all batches inside the timed region") if pipelined else "submit -> result -> batches, synchronous"),
            "synchronous": ({"value": round(global_bytes / (sync_elapsed / steps) / 1e9, 3), "unit": "GB/s", "ms_per_step": round(sync_elapsed / steps * 1e3, 4),
                             "note": "the same K steps, each submit -> result -> batches with nothing in flight in between (rounds 1-5's timed loop)"}
                            if sync_elapsed is not None else None),
            "without_detail_events": ({"value": round(global_bytes / (plain_elapsed / steps) / 1e9, 3), "unit": "GB/s", "ms_per_step": round(plain_elapsed / steps * 1e3, 4),
                                       "note": "the same K pipelined steps with option timing_detail = 0: two events per submit instead of the five the roofline's kernel times need"}
                                      if plain_elapsed is not None else None),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": (f"synthetic long reads 200..19800 bases (BASELINE config 4), {args.reads} reads/GPU "
                                    f"(mean {rec_bytes} B/record), batches(4096), validation " if args.long_reads else
                                    (f"BASELINE config 5 shard shape{'' if default_reads else ' (scaled by --reads)'}: {total_reads} reads of one synthetic stream "
                                     f"({total_bytes / 1e9:.1f} GB), byte-range shards cut {SHIFT} B into a record; " if config5 else "") +
                                    f"synthetic {args.read_len} bp Illumina FASTQ, {args.reads} reads/GPU "
                                    f"({rec_bytes} B/record), {'views() mode (offsets + id spans, no columns)' if args.views else 'batches(4096)'}, validation ")
                                   + f"{'ascii+quality (sanger)' if args.validate else 'off'}, input resident in HBM",
                       "records_per_gpu": args.reads, "record_bytes": rec_bytes, "batch_size": 4096,
                       "min_record_bytes": min_record_bytes,
                       "ranks_on_one_gpu": (f"{world} ranks share device 0 (harness check: shared-memory transport + gloo, not a scaling point)" if one_gpu else None),
                       "parallelism": (f"{'record-aligned' if args.long_reads else 'byte-range'} shards x{world}"
                                       if world > 1 else "single GPU"),
                       "pass_bytes": args.pass_bytes, "exchange": exchange},
            "fraction_of_hbm_peak_input_rate": round(global_bytes / world / sec_per_step / 1e9 / HBM_PEAK_GBS, 4),
            "ranks_seen": (min(r["ranks_seen"] for r in rank_info) if rank_info and exchange == "native" else None),
            "ranks": rank_info,
            "from_file": ({"path": file_path, "file_gb": round(total_bytes / 1e9, 3), "reader_threads": args.reader_threads,
                           "pcie_frac_per_gpu": round(global_bytes / world / sec_per_step / 1e9 / PCIE_PEAK_GBS, 3),
                           "note": "every step starts at the file: bzq_shard_read_range (this rank's byte range: pread -> pinned -> H2D) + bzq_shard_stitch; "
                                   "PCIe inclusive, so `value` here is NOT the HBM-resident headline and the roofline objects below describe the kernels only"}
                          if file_path is not None else None),
            "roofline": {
                "bound": "hbm", "kernel": "k_stream" if args.stream else "k_tile_lines (views mode: line entries)" if args.views else "k_tile_emit" if args.kernels_v1 else ("k_single<look-back>" if args.hier else "k_single<service>" if args.service else ("k_fused<LB=true>" if args.single_pass else "k_fused<LB=false>")),
                "achieved": round(dom_bytes / emit_s / 1e9, 2) if emit_s > 0 else None,
                "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(dom_bytes / emit_s / 1e9 / HBM_PEAK_GBS, 4) if emit_s > 0 else None,
                # HBM bytes per launch from the PMC counters of the same command (rocprofv3, separate --pmc passes;
                # FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md), committed under profiles/.
                # Only quoted for the profiled configuration (150 bp, validation off, two-pass default).
                "traffic": None,   # filled in below, for the profiled configuration only
                "traffic_unit": "GB per launch",
                "traffic_source": f"{TRAFFIC_PROFILE}: {TRAFFIC_KERNEL if folded else TRAFFIC_KERNEL_UNFOLDED} FETCH_SIZE*2 + WRITE_SIZE (KiB), scaled per record; read from the file at run time, not measured in this run; "
                                  f"withheld when the file's kernel-trace duration of that kernel is more than {int(TRAFFIC_MAX_DRIFT * 100)} % from this run's avg_launch_ms",
                "algorithmic_gb_per_launch": round(dom_bytes / 1e9, 3),
                "algorithmic_bytes_per_record": round(dom_bytes / max(1, recs), 1),
                "avg_launch_ms": round(dom_ms / steps / max(1, int(res.n_passes)), 4),
                "launches_per_step": int(res.n_passes),
            },
            "roofline_path": {
                "note": ("whole hot path (pass A k_tile_aggregate_h + scan + k_batch_bases + emit k_fused; per-batch ends and the record-length check inside the emit, "
                         "k_finish = the chunk totals), hipEvent time on the ctx stream" if folded else
                         "whole hot path (pass A k_tile_aggregate_h + scan + emit k_fused + k_rebase), hipEvent time on the ctx stream"),
                "achieved": round(A_total / path_s / 1e9, 2) if path_s > 0 else None,
                "frac": round(A_total / path_s / 1e9 / HBM_PEAK_GBS, 4) if path_s > 0 else None,
                "ms": {"aggregate": round(ms_agg / steps, 4), "scan": round(ms_scan / steps, 4),
                       "emit": round(ms_emit / steps, 4), ("finish" if folded else "rebase"): round(ms_rebase / steps, 4),
                       "kernels_total": round(ms_kernels / steps, 4)},
            },
        }
        profiled_config = (args.read_len == 150 and not args.long_reads and not args.views and not args.validate and not args.stream
                           and not args.single_pass and not args.service and not args.hier and not args.kernels_v1)
        if profiled_config:
            bpr, why = measured_traffic_bytes_per_record(out["roofline"]["avg_launch_ms"], TRAFFIC_KERNEL if folded else TRAFFIC_KERNEL_UNFOLDED)
            if bpr is not None:
                out["roofline"]["traffic"] = round(bpr * per_rank_records / 1e9, 3)
                # the WHOLE path's HBM bytes per step (the input is read twice: pass A + emit), so that the re-read is visible in the line
                # itself: path_traffic / algorithmic = what two passes cost over one; achievable_frac = the rate the path's kernels move
                # those bytes at, against the ~6.3 TB/s the guide calls achievable on this part (the roofline `frac` stays against 8 TB/s)
                pt = path_traffic_bytes(TRAFFIC_KERNEL if folded else TRAFFIC_KERNEL_UNFOLDED)
                if pt is not None and path_s > 0:
                    pt *= per_rank_records / TRAFFIC_RECORDS
                    out["roofline"].update({
                        "path_traffic": round(pt / 1e9, 3), "path_traffic_unit": "GB per step (all kernels of the hot path: FETCH_SIZE*2 + WRITE_SIZE of the cited summary)",
                        "path_traffic_over_algorithmic": round(pt / A_total, 3),
                        "path_rate": round(pt / path_s / 1e9, 1), "achievable_peak": HBM_ACHIEVABLE_GBS,
                        "achievable_frac": round(pt / path_s / 1e9 / HBM_ACHIEVABLE_GBS, 4),
                        "ceiling_note": ("two reads of the input are this design's floor: at the achievable rate the path's traffic takes "
                                         f"{pt / HBM_ACHIEVABLE_GBS / 1e6:.3f} ms, i.e. at most {n * HBM_ACHIEVABLE_GBS / pt / HBM_PEAK_GBS:.3f} of HBM peak as input rate; "
                                         "the north-star's 0.40 is not reachable with a separate pass A (DESIGN 12: the single-read designs measured slower)")})
            else:
                out["roofline"]["traffic_withheld"] = why
        extras = world == 1 and not args.ablate and not sharded_mode and not args.no_extra_modes
        if extras and not args.views:
            # the same input through views mode (parser.views(): offsets + id spans into the chunk, no columns), as an
            # extra figure next to the headline batch-mode `value`
            vcfg = B.ParserConfig(check_ascii=args.validate, check_quality=args.validate,
                                  quality_schema="sanger" if args.validate else None, views_only=True)
            vctx = B.Context(vcfg, "generic", 4096, local_rank)
            for _ in range(max(1, args.warmup)):
                vctx.submit_device(shard.data_ptr(), n, 0, True); vres = vctx.result()
            torch.cuda.synchronize()
            tv = time.perf_counter()
            for _ in range(args.steps):
                vctx.submit_device(shard.data_ptr(), n, 0, True); vres = vctx.result()
            torch.cuda.synchronize()
            tv = (time.perf_counter() - tv) / args.steps
            assert int(vres.n_records) == recs and vres.status == L.EOF
            out["views_mode"] = {"value": round(n / tv / 1e9, 3), "unit": "GB/s", "mrecords_per_s": round(recs / tv / 1e6, 3),
                                 "ms_per_step": round(tv * 1e3, 4), "algorithmic_bytes_per_record": round((n + 52 * recs) / recs, 1),
                                 "note": "config.views_only: RecordOffsets + id spans into the chunk, no columns; one read of the input"}
            vctx.close()
        if extras and not args.views and not args.validate and not args.long_reads and args.read_len == 150:
            # BASELINE configs[2]: the SAME bytes with ParserConfig(check_ascii=True, check_quality=True), 'sanger' -- batch mode
            out["validated_mode"] = side_mode(shard, n, recs, True, "BASELINE config 3: same 150 bp input, check_ascii + check_quality, 'sanger'; batches(4096)")
        if extras and not args.views and not args.long_reads:
            # BASELINE configs[3]: long reads, 300 000 x 200..19 800 bases (phred 5..30, sanger), irregular record boundaries
            lr_n = ctx.generate_synthetic_device(300_000, 200, 5, 30, "sanger", 0, 0, first=0, count=300_000, max_len=19_800)
            lr = torch.empty(lr_n + (1 << 20), dtype=torch.uint8, device=dev)
            ctx.generate_synthetic_device(300_000, 200, 5, 30, "sanger", lr.data_ptr(), lr.numel(), first=0, count=300_000, max_len=19_800)
            torch.cuda.synchronize()
            out["long_reads_mode"] = side_mode(lr, lr_n, 300_000, False, "BASELINE config 4: 300 000 reads of 200..19 800 bases (mean ~10 kb), batches(4096), validation off",
                                               min_rec=32)
            del lr
        if extras and not args.views and not args.long_reads:
            # the FASTA path (`bench.py --fasta` is its own full line) on the reference's FASTA benchmark
            # input, a third of the size, as one more figure next to the headline
            fctx = B.FastaContext(B.FastaParserConfig(check_ascii=args.validate), local_rank)
            ft = fctx.generate_synthetic_device(500_000, 200, 3800, 60)
            fms = []
            for _ in range(args.warmup + min(args.steps, 5)):
                fres = fctx.parse(int(ft.data_ptr()), ft.numel(), True)
                fms.append(fres.kernel_ms)
            assert int(fres.status) == 6 and int(fres.n_records) == 500_000
            fk = sum(fms[args.warmup:]) / len(fms[args.warmup:]) / 1e3
            fA = ft.numel() + int(fres.seq_bytes) + int(fres.id_bytes) + 16 * 500_000
            out["fasta_mode"] = {"value": round(ft.numel() / fk / 1e9, 3), "unit": "GB/s", "mrecords_per_s": round(0.5 / fk, 3),
                                 "kernels_ms": round(fk * 1e3, 4), "roofline_frac": round(fA / fk / 1e9 / HBM_PEAK_GBS, 4),
                                 "note": "FastaParser path, 500 k records of 200-3800 bp wrapped at 60 (1.02 GB) resident in HBM; kernel time"}
            del ft
            fctx.close()
        if extras and not args.views and not args.long_reads and args.read_len == 150:
            try:
                out["inflate_mode"] = inflate_mode(ctx, shard, rec_bytes, dev)
            except Exception as e:   # noqa: BLE001 -- a side figure must not take the headline line down; said out loud
                out["inflate_mode"] = {"error": str(e)[:300]}
        if extras and not args.views and not args.long_reads and not args.validate and args.read_len == 150:
            # other shapes of ordinary input beside the fixed-shape headline: the reference's published workload
            # (generate_synthetic_fastq.mojo:36-42: 100 bp, 219 B/record, 3 GiB), DOS line ends, sequencer-style variable-length ids
            try:
                n100 = 14_700_000
                b100 = ctx.generate_synthetic_device(n100, 100, 33, 73, "generic", 0, 0, first=0, count=n100)
                d100 = torch.empty(b100 + (1 << 20), dtype=torch.uint8, device=dev)
                ctx.generate_synthetic_device(n100, 100, 33, 73, "generic", d100.data_ptr(), d100.numel(), first=0, count=n100)
                torch.cuda.synchronize()
                out["read100_mode"] = side_mode(d100, b100, n100, False, "the reference's published benchmark workload: 14.7 M x 100 bp (219 B/record, 3.2 GB), batches(4096), validation off; "
                                                "BlazeSeq publishes 4.03 GB/s (batches) / 5.13 GB/s (views) for it on its own host", min_rec=128)
                del d100
                idb = rec_bytes - 2 * args.read_len - 6
                dcr = crlf_variant(shard, recs, rec_bytes, idb)
                torch.cuda.synchronize()
                out["crlf_mode"] = side_mode(dcr, dcr.numel(), recs, False, "the headline reads with DOS line ends (\\r\\n: every id loses its \\r to _strip_spaces, so pass A's "
                                             "no-strip hypothesis fails: the exact pass A is used, sticky after the first contradiction)")
                del dcr
                dil = illumina_variant(shard, recs, rec_bytes, idb)
                torch.cuda.synchronize()
                out["illumina_mode"] = side_mode(dil, dil.numel(), recs, False, "the headline reads under sequencer-style headers of variable length with interior spaces "
                                                 "('@SRR001666.<n> 071112_SLXA-EAS1_s_7:5:1:<x>:<y> length=150'): no fixed record stride")
                del dil
            except Exception as e:   # noqa: BLE001 -- side figures must not take the headline line down; said out loud
                out["side_workloads_error"] = str(e)[:300]
        if extras and not args.views and not args.long_reads and args.read_len == 150 and not args.no_ingest_mode:
            try:
                out["ingest_mode"] = ingest_mode(shard, rec_bytes, dev, local_rank, threads=args.ingest_threads, chunk_mib=args.ingest_chunk_mib)
            except Exception as e:   # noqa: BLE001
                out["ingest_mode"] = {"error": str(e)[:300]}
        if extras and not args.views and not args.long_reads and args.read_len == 150 and not args.no_ingest_mode:
            try:
                out["pipeline_mode"] = pipeline_mode(shard, rec_bytes, dev, local_rank, threads=args.ingest_threads, chunk_mib=args.ingest_chunk_mib)
            except Exception as e:   # noqa: BLE001 -- a side figure must not take the headline line down; said out loud
                out["pipeline_mode"] = {"error": str(e)[:300]}
        if extras and not args.views and not args.long_reads and args.read_len == 150 and not args.no_process_mode:
            try:
                out["process_mode"] = process_mode(ctx, dev)
            except Exception as e:   # noqa: BLE001
                out["process_mode"] = {"error": str(e)[:300]}
        if world == 1 and not args.no_cpu_baseline:
            k = min(args.cpu_reads, recs)
            host = shard[:k * rec_bytes].cpu().numpy() if not args.long_reads else shard[:n].cpu().numpy()
            out["cpu_baseline"] = cpu_baseline(host, k if not args.long_reads else recs, args.read_len, args.validate)
            # BlazeSeq's other CPU mode (run_throughput_blazeseq.mojo:38-45: `for view in parser.views()`), beside views_mode
            out["cpu_baseline_views"] = cpu_baseline(host, k if not args.long_reads else recs, args.read_len, args.validate, mode="views", t_budget=8.0)
            if not args.long_reads and not args.views:
                out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(host, k, rec_bytes, args.validate)
        # RCCL writes a version banner to C stdio; flush it first so that the JSON is the last line
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
