// Host ingest pipeline: file -> pinned double buffers -> device, feeding the chunk parser.
// Replaces FileReader.read_to_buffer + BufferedReader._fill_buffer/_compact_from (blazeseq/io/readers.mojo:86-137,
// blazeseq/io/buffered.mojo:239-290) for plain files: instead of one read() per 64 KiB window and a memmove of the
// partial record, a producer thread reads chunk k+1 with several pread() threads into pinned memory while chunk k
// is copied to the device on a separate HIP stream and chunk k-1 is being consumed; the bytes after the last
// record handed out (the "carry") are moved device-to-device in front of the next chunk.
//
// Included by bzq_api.hip (same translation unit: it needs the ctx's stream).  Host code only.
#pragma once

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <future>

#include <errno.h>
#include <fcntl.h>
#include <sched.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

namespace bzq {

// chunks in flight between the reader threads and the parser: k is read while k-1 travels / is inflated and is parsed.
// Three since round 3: with the asm symbol loop the device inflate of a 256 MiB chunk (4 000 blocks) fills a third of the device,
// and a third chunk's blocks in flight took BGZF end to end from 17.9 to 27-33 GB/s (16 GB file); a .gz decoded on the device reads
// its compressed pieces ahead into the third slot.  The price is a third pinned buffer of reserve + chunk bytes at every open
// (~13 ms when the three are pinned side by side, ingest_open_common; bench.py's ingest_mode reports open_ms / close_ms).
constexpr int INGEST_SLOTS = 3;

struct IngestSlot {
    uint8_t* pinned = nullptr;     // reserve + chunk_bytes
    uint8_t* dev = nullptr;        // reserve + chunk_bytes + 64
    hipEvent_t h2d_done = nullptr; // recorded on the copy stream behind the chunk's H2D
    uint64_t file_off = 0, len = 0;
    bool eof = false;
    int state = 0;                 // 0 free, 1 H2D enqueued (guarded by the mutex)
};

} // namespace bzq

struct bzq_ingest {
    bzq_ctx* ctx = nullptr;
    int device = 0;                // copied at open: close must not look at a ctx that may already be gone
    int fd = -1;
    int fd_direct = -1;            // the same file opened O_DIRECT (option "ingest_direct"): reads bypass the page cache
    int numa_node = -1;            // NUMA node of the GPU (reader threads are bound to its CPUs), -1 = unknown
    std::vector<int> numa_cpus;
    uint64_t file_size = 0, chunk_bytes = 0, reserve = 0;
    int n_threads = 4;
    // compressed input (the reference's GZFile / RapidgzipReader, io/readers.mojo:283-443): 0 plain, 1 gzip stream
    // (zlib gzread, serial), 2 BGZF (blocked gzip: blocks inflate independently on the reader threads)
    int compression = 0;
    gzFile gz = nullptr;
    uint64_t bgzf_off = 0;         // compressed offset of the next BGZF block
    // BGZF inflated on the device (bzq_inflate.hpp): the slot's pinned buffer carries the COMPRESSED blocks to comp_dev, the
    // kernel writes the chunk into the slot's device buffer; first_bad travels back behind it
    int gpu_inflate = 0;
    int inflate_ms = 0;                                    // EXPERIMENTS build: eight blocks per wave (experiments/csrc/bzq_inflate_ms.hpp) instead of one
    uint8_t* comp_dev[bzq::INGEST_SLOTS] = {};
    bzq::inf::DevBlock* tab_dev[bzq::INGEST_SLOTS] = {};
    bzq::inf::DevBlock* tab_pinned[bzq::INGEST_SLOTS] = {};
    int64_t tab_cap = 0;
    unsigned long long* bad_dev = nullptr;      // [INGEST_SLOTS]
    unsigned long long* bad_pinned = nullptr;   // [INGEST_SLOTS]
    double ratio_est = 0.30;       // compressed / inflated bytes of the last chunk: how much to read for the next one
    // any other gzip file inflated on the device (bzq_gzip.hpp): the compressed pieces travel through the slot's pinned buffer,
    // the decoder's output collects in a device FIFO (its size per piece is not known beforehand) and leaves chunk by chunk
    bzq_gzip* gz_dev = nullptr;
    // ONE buffer (rounds 3-5 kept a second one of the same size that every chunk's remainder moved to: 1.5 GiB more per open .gz
    // at the default chunk size, and a copy of up to a piece's output per chunk): chunks leave at gz_head, pieces arrive behind the
    // bytes that wait; when less than half the buffer is free behind them, what waits -- less than a chunk -- moves to the front
    uint8_t* gz_fifo = nullptr;
    uint64_t gz_head = 0;
    uint64_t gz_cap = 0, gz_have = 0, gz_off = 0;   // capacity, bytes waiting (from gz_head on), file offset of the next compressed byte
    bool gz_more = false, gz_done = false;
    // read-ahead: while piece k is decoded, helper threads read piece k + 1 into the other slot's pinned buffer
    std::atomic<uint64_t> gz_piece{0}; // compressed bytes per piece: starts at half a chunk, then follows the file's compression ratio (gz_fill_fifo)
    // read-ahead: one thread reads piece after piece into the three slots' pinned buffers, up to two pieces in front of the one being
    // decoded, and sends each on to the device (gz_stage) -- the decoder starts piece k + 1's finder and decoders while piece k's last
    // kernels run (bzq_gzip.hpp, round 4), so piece k + 2 must be under way by then
    std::thread gz_reader;
    std::mutex gz_mu;
    std::condition_variable gz_cv;
    struct GzPiece { uint64_t len = 0; bool last = false, ok = true; };
    GzPiece gz_ring[bzq::INGEST_SLOTS];
    uint64_t gz_rd_ready = 0, gz_rd_taken = 0;   // pieces read and published / pieces whose decode call has returned (their buffer is free)
    bool gz_stop = false, gz_all_handed = false;
    std::string gz_read_err;
    bzq::IngestSlot slot[bzq::INGEST_SLOTS];
    hipStream_t copy_stream = nullptr;   // H2D of the chunks
    // device inflate: consecutive chunks' copies and kernels alternate between TWO streams, so that they overlap -- [0] is copy_stream,
    // [1] is the ingest's own.  Not one per slot: a process gets 4 hardware queues per device (GPU_MAX_HW_QUEUES) and further streams
    // SHARE them -- with three inflate streams beside the parser's and the null stream the parser's kernels sat in a queue behind the
    // NEXT chunk's 10 ms inflate kernel and the pipeline ran in lockstep (rocprofv3 timeline, round 4: 34 -> 4x GB/s)
    hipStream_t inflate_stream[bzq::INGEST_SLOTS] = {};
    int n_inflate_streams = 2;
    hipEvent_t dev_free[bzq::INGEST_SLOTS] = {}; // recorded on the ctx stream once slot i's device buffer may be overwritten
    bool dev_free_valid[bzq::INGEST_SLOTS] = {};
    std::thread producer;
    std::mutex mu;
    std::condition_variable cv;
    bool stop = false;
    int64_t produced = 0;          // chunks whose H2D has been enqueued
    int64_t released = 0;          // chunks whose device buffer the consumer has released (k-th release frees slot k%2)
    int64_t next_k = 0;            // next chunk the consumer will parse
    std::string io_error;
    // consumer state
    bool have_prev = false, finished = false;
    int32_t final_status = 0;
    uint64_t prev_n = 0;           // bytes of the previous chunk as submitted (carry + body)
    const uint8_t* prev_ptr = nullptr;   // its first byte on the device (inside a slot, or in big[])
    // a carry larger than the reserve in front of a slot's body (a record or a batch longer than chunk/8): the chunk is
    // assembled in a buffer of its own, grown on demand -- the reference has no record-size limit either
    uint8_t* big[bzq::INGEST_SLOTS] = {};
    bool quiesce_device = true;   // bzq_ingest_close names the streams instead
    hipStream_t quiesce_stream = nullptr, quiesce_stream2 = nullptr;   // whose work may still touch the buffers at the close (ingest_free)
    uint64_t big_cap[bzq::INGEST_SLOTS] = {};
    uint64_t prev_stream_pos = 0;
    bzq_chunk prev_res{};
    bzq_ingest_stats stats{};
    std::chrono::steady_clock::time_point t_open;
};

namespace bzq {

inline double seconds_since(std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

// bind the calling thread to the CPUs of the GPU's NUMA node (the pinned buffers live there; best effort)
inline void bind_to_cpus(const std::vector<int>& cpus) {
    if (cpus.empty()) return;
    cpu_set_t set;
    CPU_ZERO(&set);
    for (int c : cpus) if (c >= 0 && c < CPU_SETSIZE) CPU_SET(c, &set);
    (void)sched_setaffinity(0, sizeof(set), &set);
}

// Fill dst[0, len) from the file at off with n_threads pread() workers (slices of >= 8 MiB, 4 KiB aligned).  fd_direct >= 0:
// whole 4 KiB blocks are read O_DIRECT (no page-cache copy: storage DMA lands in the pinned buffer, which is page aligned and
// has room up to the next block boundary); a filesystem that refuses O_DIRECT reads (EINVAL) gets buffered reads instead.
inline bool parallel_pread(int fd, uint8_t* dst, uint64_t off, uint64_t len, int n_threads, std::string& err, int fd_direct = -1,
                           const std::vector<int>* cpus = nullptr) {
    if (len == 0) return true;
    const uint64_t min_slice = 8ull << 20;
    int nt = (int)std::min<uint64_t>((uint64_t)std::max(1, n_threads), (len + min_slice - 1) / min_slice);
    std::atomic<bool> ok{true};
    const bool direct_ok = fd_direct >= 0 && (off & 4095u) == 0 && ((uintptr_t)dst & 4095u) == 0;
    auto work = [&](uint64_t a, uint64_t b) {
        if (cpus) bind_to_cpus(*cpus);
        bool direct = direct_ok;
        while (a < b) {
            ssize_t r;
            if (direct) {   // length rounded up to whole blocks: a read that reaches EOF just returns fewer bytes
                const uint64_t want = (std::min<uint64_t>(b - a, 1ull << 30) + 4095u) & ~4095ull;
                r = pread(fd_direct, dst + a, (size_t)want, (off_t)(off + a));
                if (r < 0 && errno == EINVAL) { direct = false; continue; }
                if (r > (ssize_t)(b - a)) r = (ssize_t)(b - a);
            } else {
                r = pread(fd, dst + a, (size_t)std::min<uint64_t>(b - a, 1ull << 30), (off_t)(off + a));
            }
            if (r <= 0) { ok = false; return; }
            a += (uint64_t)r;
            if (direct && (a & 4095u)) direct = false;   // short read: finish the tail buffered
        }
    };
    std::vector<std::thread> th;
    const uint64_t per = (((len + nt - 1) / nt) + 4095u) & ~4095ull;
    for (int i = 1; i < nt; ++i) th.emplace_back(work, std::min(len, per * i), std::min(len, per * (i + 1)));
    work(0, std::min(len, per));
    for (auto& t : th) t.join();
    if (!ok) err = "pread failed or file truncated while reading";
    return ok;
}

// ---- compressed sources --------------------------------------------------------------------------------------------

// BGZF block header (SAM spec 4.1): gzip member with FEXTRA and a 'B','C' subfield holding BSIZE = block size - 1.
// Returns the block size, 0 if the 18 bytes at p are not such a header.
inline uint32_t bgzf_block_size(const uint8_t* p) {
    if (p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || !(p[3] & 4)) return 0;
    if (p[10] != 6 || p[11] != 0 || p[12] != 'B' || p[13] != 'C' || p[14] != 2 || p[15] != 0) return 0;
    return (uint32_t)(p[16] | (p[17] << 8)) + 1u;
}

struct BgzfBlock { uint64_t coff; uint32_t csize, usize; uint64_t uoff; uint32_t crc; };

// Inflate a run of BGZF blocks (already read into `comp`) into dst with n_threads workers.
inline bool bgzf_inflate_blocks(const std::vector<BgzfBlock>& blocks, const uint8_t* comp, uint64_t comp_base, uint8_t* dst,
                                int n_threads, std::string& err) {
    std::atomic<size_t> next{0};
    std::atomic<bool> ok{true};
    auto work = [&]() {
        z_stream zs;
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= blocks.size() || !ok) return;
            const BgzfBlock& b = blocks[i];
            if (b.usize == 0) continue;
            memset(&zs, 0, sizeof zs);
            if (inflateInit2(&zs, -15) != Z_OK) { ok = false; return; }
            zs.next_in = const_cast<Bytef*>(comp + (b.coff - comp_base) + 18);
            zs.avail_in = b.csize - 18 - 8;
            zs.next_out = dst + b.uoff;
            zs.avail_out = b.usize;
            const int rc = inflate(&zs, Z_FINISH);
            inflateEnd(&zs);
            if (rc != Z_STREAM_END || zs.avail_out != 0) { ok = false; return; }
            if ((uint32_t)crc32(0L, dst + b.uoff, b.usize) != b.crc) { ok = false; return; }   // RFC 1952 8.: CRC-32 of the uncompressed data
        }
    };
    std::vector<std::thread> th;
    const int nt = (int)std::min<size_t>((size_t)std::max(1, n_threads), blocks.size());
    for (int i = 1; i < nt; ++i) th.emplace_back(work);
    work();
    for (auto& t : th) t.join();
    if (!ok) err = "BGZF block failed to inflate (corrupt or truncated file)";
    return ok;
}

// Next chunk of DECOMPRESSED bytes into dst (capacity cap).  Returns false on error; *eof when the input is exhausted.
inline bool read_compressed_chunk(bzq_ingest* g, uint8_t* dst, uint64_t cap, uint64_t* out_len, bool* eof, std::string& err) {
    *out_len = 0; *eof = false;
    if (g->compression == 1) {
        uint64_t got = 0;
        while (got < cap) {
            const int r = gzread(g->gz, dst + got, (unsigned)std::min<uint64_t>(cap - got, 1u << 30));
            if (r < 0) { int en = 0; err = std::string("gzread: ") + gzerror(g->gz, &en); return false; }
            if (r == 0) {
                // a truncated or corrupt stream ends "cleanly" for gzread; gzerror tells (Z_BUF_ERROR: unexpected EOF)
                int en = Z_OK;
                const char* msg = gzerror(g->gz, &en);
                if (en != Z_OK && en != Z_STREAM_END) { err = std::string("gzread: ") + (msg ? msg : "stream error"); return false; }
                *eof = true;
                break;
            }
            got += (uint64_t)r;
        }
        *out_len = got;
        return true;
    }
    // BGZF: collect whole blocks until the chunk is full
    if (g->compression != 2) { err = "internal: read_compressed_chunk on a file that is neither gzip nor BGZF"; return false; }
    std::vector<BgzfBlock> blocks;
    uint64_t usum = 0, coff = g->bgzf_off;
    uint8_t hdr[18], tail[8];
    while (coff < g->file_size) {
        if (coff + 28 > g->file_size || pread(g->fd, hdr, 18, (off_t)coff) != 18) { err = "BGZF: truncated block header"; return false; }
        const uint32_t bs = bgzf_block_size(hdr);
        if (!bs || bs < 26 || coff + bs > g->file_size) { err = "BGZF: bad block header"; return false; }
        if (pread(g->fd, tail, 8, (off_t)(coff + bs - 8)) != 8) { err = "BGZF: truncated block"; return false; }
        const uint32_t crc = (uint32_t)tail[0] | ((uint32_t)tail[1] << 8) | ((uint32_t)tail[2] << 16) | ((uint32_t)tail[3] << 24);
        const uint32_t us = (uint32_t)tail[4] | ((uint32_t)tail[5] << 8) | ((uint32_t)tail[6] << 16) | ((uint32_t)tail[7] << 24);
        if (us > 65536) { err = "BGZF: block claims more than 64 KiB"; return false; }
        if (usum + us > cap) break;
        blocks.push_back({coff, bs, us, usum, crc});
        usum += us;
        coff += bs;
    }
    if (!blocks.empty()) {
        const uint64_t c0 = blocks.front().coff, c1 = blocks.back().coff + blocks.back().csize;
        std::vector<uint8_t> comp((size_t)(c1 - c0));
        if (!parallel_pread(g->fd, comp.data(), c0, c1 - c0, g->n_threads, err)) return false;
        if (!bgzf_inflate_blocks(blocks, comp.data(), c0, dst, g->n_threads, err)) return false;
    }
    g->bgzf_off = coff;
    *out_len = usum;
    *eof = coff >= g->file_size;
    return true;
}

// buffers of slot i (allocated at open: doing it in the reader, when it first gets to the slot, puts the pinning of the second
// slot on the path of chunk 1 -- 43 instead of 50 GB/s on a 3 GB file)
inline bool ingest_alloc_inflate(bzq_ingest* g, int i) {
    if (i == 0) g->inflate_stream[0] = g->copy_stream;
    else if (i < g->n_inflate_streams && cache::stream_pool().get(g->device, &g->inflate_stream[i]) != hipSuccess) return false;
    return cache::get_device(g->device, g->chunk_bytes + 64, &g->comp_dev[i]) &&
           cache::get_device(g->device, (size_t)g->tab_cap * sizeof(bzq::inf::DevBlock) + 16, &g->tab_dev[i]) &&
           (g->tab_pinned[i] = (bzq::inf::DevBlock*)cache::host_small().get((size_t)g->tab_cap * sizeof(bzq::inf::DevBlock) + 16)) != nullptr;
}
// pinned_bytes: what the slot's host buffer must take behind the reserve (0: the slot has none).  Pinning is the expensive part
// of an open (~0.1 s per GiB): a plain file stages whole chunks there, a .gz decoded on the device only pieces of compressed
// bytes, in two of the three slots.
inline bool ingest_alloc_slot(bzq_ingest* g, int i, uint64_t pinned_bytes) {
    return (!pinned_bytes || cache::get_pinned(g->device, g->reserve + pinned_bytes, &g->slot[i].pinned)) &&   // (bzq_bufcache.hpp: the next open of the process takes them from there)
           cache::get_device(g->device, g->reserve + g->chunk_bytes + 64, &g->slot[i].dev) &&
           hipEventCreateWithFlags(&g->slot[i].h2d_done, hipEventDisableTiming) == hipSuccess &&
           hipEventCreateWithFlags(&g->dev_free[i], hipEventDisableTiming) == hipSuccess;
}

// Device inflate: the next run of whole BGZF blocks, still COMPRESSED, into `pinned` (capacity cap); their table (payload
// offsets relative to `pinned`) into tab.  The read is sized from the last chunk's compression ratio; a short read gives a
// smaller chunk, never a wrong one.
inline bool read_bgzf_window(bzq_ingest* g, uint8_t* pinned, uint64_t cap, bzq::inf::DevBlock* tab, int64_t* n_blocks, uint64_t* comp_len,
                             uint64_t* out_len, bool* eof, std::string& err, uint64_t out_cap) {
    *n_blocks = 0; *comp_len = 0; *out_len = 0; *eof = false;
    const uint64_t remaining = g->file_size - g->bgzf_off;
    if (remaining == 0) { *eof = true; return true; }
    uint64_t want = (uint64_t)((double)out_cap * g->ratio_est * 1.05) + (2ull << 20);
    want = std::min<uint64_t>({want, cap, remaining});
    if (!parallel_pread(g->fd, pinned, g->bgzf_off, want, g->n_threads, err, g->fd_direct, &g->numa_cpus)) return false;
    uint64_t off = 0, usum = 0;
    int64_t k = 0;
    while (off + 28 <= want && k < g->tab_cap) {
        const uint32_t bs = bgzf_block_size(pinned + off);
        if (!bs || bs < 26) { if (k == 0) { err = "BGZF: bad block header"; return false; } break; }   // the next call meets it first
        if (off + bs > want) break;
        const uint8_t* t = pinned + off + bs - 4;
        const uint32_t us = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
        if (us > 65536) { if (k == 0) { err = "BGZF: block claims more than 64 KiB"; return false; } break; }
        if (usum + us > out_cap) break;
        const uint32_t crc = (uint32_t)t[-4] | ((uint32_t)t[-3] << 8) | ((uint32_t)t[-2] << 16) | ((uint32_t)t[-1] << 24);
        tab[k++] = bzq::inf::DevBlock{off + 18, usum, bs - 26, us, crc, 0u};
        usum += us; off += bs;
    }
    if (k == 0) { err = want == remaining ? "BGZF: truncated block" : "BGZF: a block does not fit the chunk"; return false; }
    if (usum) g->ratio_est = std::max(0.02, (double)off / (double)usum);
    g->bgzf_off += off;
    *n_blocks = k; *comp_len = off; *out_len = usum; *eof = g->bgzf_off >= g->file_size;
    return true;
}

// Device gzip: decode pieces of the file into the FIFO until it holds a chunk (or the stream ends).  Pieces are as large as half the
// FIFO takes decoded, at most a chunk (enough block starts to fill the device), and are read AHEAD by one thread (gz_read_ahead):
// up to two pieces in front of the one being decoded, into the three slots' pinned buffers in turn, each sent on to the device at
// once (gz_stage) -- the decoder starts piece k + 1's finder and decoders while piece k's last kernels run, so piece k + 2 must be
// on its way by then.  The FIFO is sized so that a piece's output always fits behind a chunk that is still waiting in it (6 chunks;
// beyond a ratio of 10 the decoder keeps the rest and is asked again).
inline void gz_read_ahead(bzq_ingest* g) {
    for (uint64_t seq = 0;; ++seq) {
        {
            std::unique_lock<std::mutex> lk(g->gz_mu);
            g->gz_cv.wait(lk, [&] { return g->gz_stop || seq - g->gz_rd_taken < (uint64_t)INGEST_SLOTS; });
            if (g->gz_stop) return;
        }
        const uint64_t off = g->gz_off;
        if (off >= g->file_size) return;   // (the piece in front was the last)
        const uint64_t len = std::min<uint64_t>(g->gz_piece.load(), g->file_size - off);
        g->gz_off = off + len;
        const int buf = (int)(seq % INGEST_SLOTS);
        uint8_t* dst = g->slot[buf].pinned + g->reserve;
        std::string err;
        const bool ok = parallel_pread(g->fd, dst, off, len, g->n_threads, err, g->fd_direct, &g->numa_cpus);
        // on to the device at once: the copy runs behind the decoding of the pieces in front (a failure here only means that the
        // piece is copied by its gz_decode)
        if (ok && g->gz_dev) (void)bzq::gz::gz_stage(g->gz_dev, dst, len);
        const bool last = off + len >= g->file_size;
        {
            std::unique_lock<std::mutex> lk(g->gz_mu);
            g->gz_ring[buf].len = len; g->gz_ring[buf].last = last; g->gz_ring[buf].ok = ok;
            if (!ok) g->gz_read_err = err;
            g->gz_rd_ready = seq + 1;
            g->gz_cv.notify_all();
        }
        if (!ok || last) return;
    }
}
inline bool gz_fill_fifo(bzq_ingest* g, std::string& err) {
    if (!g->gz_reader.joinable() && !g->gz_all_handed && g->gz_rd_ready == 0) g->gz_reader = std::thread(gz_read_ahead, g);
    while (g->gz_have < g->chunk_bytes && !g->gz_done) {
        if (g->gz_have == 0) g->gz_head = 0;
        if (g->gz_head && g->gz_cap - g->gz_head - g->gz_have < g->gz_cap / 2) {
            // what waits moves to the front, on the decoder's stream (behind the chunk copies that read the bytes in front of it, in
            // front of the kernels that write behind it).  It is less than a chunk and gz_head at least one, so source and
            // destination do not overlap -- pieces of at most gz_head bytes, front to back, hold for any sizes.
            for (uint64_t done = 0; done < g->gz_have;) {
                const uint64_t m = std::min<uint64_t>(g->gz_head, g->gz_have - done);
                const hipError_t he = hipMemcpyAsync(g->gz_fifo + done, g->gz_fifo + g->gz_head + done, m, hipMemcpyDeviceToDevice, g->gz_dev->stream);
                if (he != hipSuccess) { err = std::string("gzip: moving the FIFO's bytes to its front: ") + hipGetErrorString(he); return false; }
                done += m;
            }
            g->gz_head = 0;
        }
        const uint64_t free_bytes = g->gz_cap - g->gz_head - g->gz_have;
        const uint8_t* src = nullptr;
        uint64_t want = 0;
        bool file_done = g->gz_all_handed, took = false;
        if (!g->gz_more && !g->gz_all_handed) {
            std::unique_lock<std::mutex> lk(g->gz_mu);
            g->gz_cv.wait(lk, [&] { return g->gz_rd_ready > g->gz_rd_taken; });
            const int buf = (int)(g->gz_rd_taken % INGEST_SLOTS);
            if (!g->gz_ring[buf].ok) { err = g->gz_read_err; return false; }
            src = g->slot[buf].pinned + g->reserve;
            want = g->gz_ring[buf].len;
            file_done = g->gz_ring[buf].last;
            if (file_done) g->gz_all_handed = true;
            took = true;
        }
        uint64_t got = 0;
        int32_t more = 0;
        const int drc = bzq::gz::gz_decode(g->gz_dev, src, want, file_done, g->gz_fifo + g->gz_head + g->gz_have, free_bytes, &got, &more);
        if (took) {   // the piece's pinned buffer is free (what the decoder keeps of it, it keeps in a copy)
            std::unique_lock<std::mutex> lk(g->gz_mu);
            g->gz_rd_taken += 1;
            g->gz_cv.notify_all();
        }
        if (drc < 0) { err = g->gz_dev->err; return false; }
        g->gz_have += got;
        // the pieces to come: as large as half the FIFO takes decoded (the decode kernel works in rounds of ~6 000 decoder
        // waves: 256 MiB of 2 x compressible FASTQ are 13 000, 128 MiB one round and a bit, which costs a sixth of the rate), at
        // most a chunk (the pinned buffers' size) -- a file that compresses 6 x keeps its pieces of 128 MiB and does not overflow the FIFO
        static const bool fixed_piece = getenv("BZQ_GZ_FIXED_PIECE") != nullptr;   // A/B switch: pieces stay half a chunk
        if (fixed_piece) {}
        else if (want && got && !more) {
            const double ratio = (double)got / (double)want;
            const uint64_t fit = (uint64_t)((double)(g->gz_cap / 2) / (ratio > 1.0 ? ratio : 1.0)) & ~4095ull;
            g->gz_piece = std::min<uint64_t>(g->chunk_bytes, std::max<uint64_t>(fit, std::min<uint64_t>(g->chunk_bytes, 1ull << 20)));
            static const uint64_t piece_cap = getenv("BZQ_GZ_PIECE_MIB") ? (uint64_t)atoll(getenv("BZQ_GZ_PIECE_MIB")) << 20 : 0;   // sweeps (profiles/r5_gzip_piece_sweep.txt): 0 / unset = no cap
            if (piece_cap) g->gz_piece = std::min<uint64_t>(g->gz_piece.load(), piece_cap);
        } else if (more && g->gz_piece.load() > (2ull << 20)) g->gz_piece = (g->gz_piece.load() / 2) & ~4095ull;
        g->gz_more = more != 0;
        g->gz_done = (file_done && !more) || g->gz_dev->finished;
        // (a call without new input that delivers nothing is fine while the file has more to give -- the host continuation of a
        // stretch without findable block starts may need the next piece to finish its block -- and a bug behind the file's end)
        if (!got && !want && !more && !g->gz_done && file_done) { err = "gzip: the decoder made no progress"; return false; }
    }
    return true;
}

inline void ingest_producer(bzq_ingest* g) {
    // a failed HIP call stops the pipeline with an error the consumer reports: a chunk is never published unless its copy
    // was enqueued successfully
    auto fail = [&](const char* what, hipError_t e) {
        std::unique_lock<std::mutex> lk(g->mu);
        g->io_error = std::string(what) + ": " + hipGetErrorString(e);
        g->stop = true;
        g->cv.notify_all();
    };
    hipError_t he;
    bind_to_cpus(g->numa_cpus);
    if ((he = hipSetDevice(g->device)) != hipSuccess) return fail("reader: hipSetDevice", he);
    uint64_t off = 0;   // offset in the (decompressed) stream
    for (int64_t k = 0;; ++k) {
        const int b = (int)(k % INGEST_SLOTS);
        IngestSlot& s = g->slot[b];
        // the pinned buffer of this slot was last used by chunk k - INGEST_SLOTS: its H2D must have finished
        if (k >= INGEST_SLOTS && (he = hipEventSynchronize(s.h2d_done)) != hipSuccess) return fail("reader: waiting for the previous copy", he);
        {
            std::unique_lock<std::mutex> lk(g->mu);
            if (g->stop) return;
        }
        uint64_t len = 0, comp_len = 0;
        int64_t n_blocks = 0;
        // one block is decoded by one wave at its own pace (~20 ms for 64 KiB): a chunk is too few blocks to fill the device,
        // so the inflate kernels of consecutive chunks run side by side on two streams
        const hipStream_t cs = g->gz_dev ? g->gz_dev->stream : (g->gpu_inflate ? g->inflate_stream[k % g->n_inflate_streams] : g->copy_stream);   // (buffers of slot b were last used by chunk k - 3, possibly on the other stream: the wait for h2d_done above covers them)
        bool eof = false, ok;
        const auto t0 = std::chrono::steady_clock::now();
        std::string err;
        if (g->gz_dev) {
            ok = gz_fill_fifo(g, err);
            len = std::min<uint64_t>(g->gz_have, g->chunk_bytes);
            eof = g->gz_done && g->gz_have <= g->chunk_bytes;
        } else if (g->gpu_inflate) {
            ok = read_bgzf_window(g, s.pinned + g->reserve, g->chunk_bytes, g->tab_pinned[b], &n_blocks, &comp_len, &len, &eof, err, g->chunk_bytes);
        } else if (g->compression == 0) {
            // A file of several chunks starts with SHORT ones (an eighth, a quarter, half a chunk): read -> copy -> parse of chunk 0
            // are in series, nothing overlaps them, and with a whole first chunk that was 20-35 ms of a fresh process's ~100 ms for
            // the file (bench.py process_mode).  From chunk 3 on the chunk size is the caller's.  BZQ_INGEST_RAMP=0: off (A/B).
            static const bool ramp = !(getenv("BZQ_INGEST_RAMP") && getenv("BZQ_INGEST_RAMP")[0] == '0');
            uint64_t want_len = g->chunk_bytes;
            if (ramp && k < 3 && g->file_size > 2 * g->chunk_bytes && g->chunk_bytes >= (8ull << 20)) want_len = (g->chunk_bytes >> (3 - k)) & ~4095ull;
            len = std::min<uint64_t>(want_len, g->file_size - off);
            ok = parallel_pread(g->fd, s.pinned + g->reserve, off, len, g->n_threads, err, g->fd_direct, &g->numa_cpus);
            eof = off + len >= g->file_size;
        } else {
            ok = read_compressed_chunk(g, s.pinned + g->reserve, g->chunk_bytes, &len, &eof, err);
        }
        if (!ok) {
            std::unique_lock<std::mutex> lk(g->mu);
            g->io_error = err; g->stop = true; g->cv.notify_all();
            return;
        }
        const double rs = seconds_since(t0);
        // the device buffer of this slot was last used by chunk k - INGEST_SLOTS: wait until the consumer released it
        {
            std::unique_lock<std::mutex> lk(g->mu);
            g->stats.read_s += rs;
            g->stats.bytes_read += len;
            g->cv.wait(lk, [&] { return g->stop || g->released >= k - (INGEST_SLOTS - 1); });
            if (g->stop) return;
            he = g->dev_free_valid[b] ? hipStreamWaitEvent(cs, g->dev_free[b], 0) : hipSuccess;
        }
        if (he != hipSuccess) return fail("reader: hipStreamWaitEvent", he);
        if (g->gz_dev) {   // the chunk leaves the FIFO; what is left stays where it is (gz_fill_fifo moves it when room is short)
            if (len && (he = hipMemcpyAsync(s.dev + g->reserve, g->gz_fifo + g->gz_head, len, hipMemcpyDeviceToDevice, cs)) != hipSuccess) return fail("reader: FIFO to chunk", he);
            g->gz_head += len; g->gz_have -= len;
        } else if (g->gpu_inflate) {
            g->bad_pinned[b] = ~0ull;   // (the consumer read the previous verdict of this slot two chunks ago)
            if (n_blocks) {
                if ((he = cache::pinned_pool().h2d(g->comp_dev[b], s.pinned + g->reserve, comp_len, cs)) != hipSuccess ||   // (lazy pinning, block by block: bzq_bufcache.hpp)
                    (he = hipMemcpyAsync(g->tab_dev[b], g->tab_pinned[b], (size_t)n_blocks * sizeof(bzq::inf::DevBlock), hipMemcpyHostToDevice, cs)) != hipSuccess ||
                    (he = hipMemsetAsync(g->bad_dev + b, 0xFF, sizeof(unsigned long long), cs)) != hipSuccess)
                    return fail("reader: host to device copy (compressed blocks)", he);
#if BZQ_EXPERIMENTS
                if (g->inflate_ms) {
                    bzq::inf::launch_bgzf_inflate_ms(bzq::inf::ArgsMs{g->comp_dev[b], comp_len, g->tab_dev[b], n_blocks, s.dev + g->reserve, g->bad_dev + b, nullptr, nullptr}, cs);
                } else
#endif
                {
                    bzq::inf::Args ia{g->comp_dev[b], comp_len, g->tab_dev[b], n_blocks, s.dev + g->reserve, g->bad_dev + b};
                    hipLaunchKernelGGL(bzq::inf::k_bgzf_inflate, dim3((unsigned)((n_blocks + bzq::inf::WAVES - 1) / bzq::inf::WAVES)), dim3(BLOCK), 0, cs, ia);
                }
                if ((he = hipGetLastError()) != hipSuccess ||
                    (he = hipMemcpyAsync(g->bad_pinned + b, g->bad_dev + b, sizeof(unsigned long long), hipMemcpyDeviceToHost, cs)) != hipSuccess)
                    return fail("reader: device inflate", he);
            }
        } else if (len && (he = cache::pinned_pool().h2d(s.dev + g->reserve, s.pinned + g->reserve, len, cs)) != hipSuccess)
            return fail("reader: host to device copy", he);
        if ((he = hipEventRecord(s.h2d_done, cs)) != hipSuccess) return fail("reader: hipEventRecord", he);
        s.file_off = off; s.len = len; s.eof = eof;
        off += len;
        {
            std::unique_lock<std::mutex> lk(g->mu);
            g->produced = k + 1;
            g->cv.notify_all();
        }
        if (s.eof) return;
    }
}

inline void ingest_free(bzq_ingest* g) {
    if (!g) return;
    // BZQ_CLOSE_TRACE=1: where a close spends its time, to stderr (a close that waits for somebody else's kernel shows itself here)
    static const bool trace = getenv("BZQ_CLOSE_TRACE") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!trace) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[bzq close] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    {
        std::unique_lock<std::mutex> lk(g->mu);
        g->stop = true;
        g->cv.notify_all();
    }
    if (g->producer.joinable()) g->producer.join();
    { std::unique_lock<std::mutex> lk(g->gz_mu); g->gz_stop = true; g->gz_cv.notify_all(); }
    if (g->gz_reader.joinable()) g->gz_reader.join();   // (a read-ahead still writing into a slot's pinned buffer)
    lap("threads joined");
    (void)hipSetDevice(g->device);
    for (int i = 1; i < INGEST_SLOTS; ++i)
        if (g->inflate_stream[i]) { (void)hipStreamSynchronize(g->inflate_stream[i]); lap("inflate stream synchronised"); cache::stream_pool().put(g->device, g->inflate_stream[i]); }
    if (g->copy_stream) { (void)hipStreamSynchronize(g->copy_stream); lap("copy stream synchronised"); cache::stream_pool().put(g->device, g->copy_stream); }   // (kept for the process, not destroyed: bzq_bufcache.hpp)
    // Buffers that go back to the cache skip hipFree's implicit wait, so whoever may still touch them is waited for HERE -- and only
    // they: the parser's stream (it may still read the last chunk) and, in views mode, the ctx's consumer stream (bzq_device_views point
    // INTO the chunk).  Not the whole device (rounds 4-5 did that): a caller's kernels on other streams -- a consumer of the batches'
    // columns, which live in the ctx, not here -- are none of this close's business.
    if (g->quiesce_device) (void)hipDeviceSynchronize();   // (the FASTA ingest, and an open that failed half way: as before)
    if (g->quiesce_stream) (void)hipStreamSynchronize(g->quiesce_stream);
    if (g->quiesce_stream2) (void)hipStreamSynchronize(g->quiesce_stream2);
    lap("parser stream(s) synchronised");
    for (int i = 0; i < INGEST_SLOTS; ++i) {
        cache::pinned_pool().put(g->slot[i].pinned);
        cache::device_pool().put(g->slot[i].dev);
        if (g->big[i]) (void)hipFree(g->big[i]);   // (only a stream whose batch outgrew the reserve has one: this free does wait for the device)
        cache::device_pool().put(g->comp_dev[i]);
        cache::device_pool().put(g->tab_dev[i]);
        cache::host_small().put(g->tab_pinned[i], (size_t)g->tab_cap * sizeof(bzq::inf::DevBlock) + 16);
        if (g->slot[i].h2d_done) (void)hipEventDestroy(g->slot[i].h2d_done);
        if (g->dev_free[i]) (void)hipEventDestroy(g->dev_free[i]);
    }
    cache::device_pool().put(g->bad_dev);
    cache::host_small().put(g->bad_pinned, INGEST_SLOTS * sizeof(unsigned long long));
    lap("buffers and events returned");
    if (g->gz) gzclose(g->gz);
    if (g->gz_dev) bzq::gz::gz_free(g->gz_dev);
    cache::device_pool().put(g->gz_fifo);
    if (g->fd >= 0) close(g->fd);
    if (g->fd_direct >= 0) close(g->fd_direct);
    delete g;
}

} // namespace bzq
