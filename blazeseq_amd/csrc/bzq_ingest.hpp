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

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

namespace bzq {

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
    uint64_t file_size = 0, chunk_bytes = 0, reserve = 0;
    int n_threads = 4;
    bzq::IngestSlot slot[2];
    hipStream_t copy_stream = nullptr;
    hipEvent_t dev_free[2] = {nullptr, nullptr}; // recorded on the ctx stream once slot i's device buffer may be overwritten
    bool dev_free_valid[2] = {false, false};
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
    uint64_t prev_off = 0;         // offset of its first byte inside the slot's device buffer
    uint64_t prev_stream_pos = 0;
    bzq_chunk prev_res{};
    bzq_ingest_stats stats{};
    std::chrono::steady_clock::time_point t_open;
};

namespace bzq {

inline double seconds_since(std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

// Fill dst[0, len) from the file at off with n_threads pread() workers (slices of >= 8 MiB).
inline bool parallel_pread(int fd, uint8_t* dst, uint64_t off, uint64_t len, int n_threads, std::string& err) {
    if (len == 0) return true;
    const uint64_t min_slice = 8ull << 20;
    int nt = (int)std::min<uint64_t>((uint64_t)std::max(1, n_threads), (len + min_slice - 1) / min_slice);
    std::atomic<bool> ok{true};
    auto work = [&](uint64_t a, uint64_t b) {
        while (a < b) {
            const ssize_t r = pread(fd, dst + a, (size_t)std::min<uint64_t>(b - a, 1ull << 30), (off_t)(off + a));
            if (r <= 0) { ok = false; return; }
            a += (uint64_t)r;
        }
    };
    std::vector<std::thread> th;
    const uint64_t per = (len + nt - 1) / nt;
    for (int i = 1; i < nt; ++i) th.emplace_back(work, std::min(len, per * i), std::min(len, per * (i + 1)));
    work(0, std::min(len, per));
    for (auto& t : th) t.join();
    if (!ok) err = "pread failed or file truncated while reading";
    return ok;
}

inline void ingest_producer(bzq_ingest* g) {
    (void)hipSetDevice(g->device);
    uint64_t off = 0;
    for (int64_t k = 0;; ++k) {
        IngestSlot& s = g->slot[k & 1];
        // the pinned buffer of this slot was last used by chunk k-2: its H2D must have finished
        if (k >= 2) (void)hipEventSynchronize(s.h2d_done);
        {
            std::unique_lock<std::mutex> lk(g->mu);
            if (g->stop) return;
        }
        const uint64_t len = std::min<uint64_t>(g->chunk_bytes, g->file_size - off);
        const auto t0 = std::chrono::steady_clock::now();
        std::string err;
        if (!parallel_pread(g->fd, s.pinned + g->reserve, off, len, g->n_threads, err)) {
            std::unique_lock<std::mutex> lk(g->mu);
            g->io_error = err; g->stop = true; g->cv.notify_all();
            return;
        }
        const double rs = seconds_since(t0);
        // the device buffer of this slot was last used by chunk k-2: wait until the consumer released it
        {
            std::unique_lock<std::mutex> lk(g->mu);
            g->stats.read_s += rs;
            g->stats.bytes_read += len;
            g->cv.wait(lk, [&] { return g->stop || g->released >= k - 1; });
            if (g->stop) return;
            if (g->dev_free_valid[k & 1]) (void)hipStreamWaitEvent(g->copy_stream, g->dev_free[k & 1], 0);
        }
        if (len) (void)hipMemcpyAsync(s.dev + g->reserve, s.pinned + g->reserve, len, hipMemcpyHostToDevice, g->copy_stream);
        (void)hipEventRecord(s.h2d_done, g->copy_stream);
        s.file_off = off; s.len = len; s.eof = (off + len >= g->file_size);
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
    {
        std::unique_lock<std::mutex> lk(g->mu);
        g->stop = true;
        g->cv.notify_all();
    }
    if (g->producer.joinable()) g->producer.join();
    (void)hipSetDevice(g->device);
    if (g->copy_stream) { (void)hipStreamSynchronize(g->copy_stream); (void)hipStreamDestroy(g->copy_stream); }
    for (int i = 0; i < 2; ++i) {
        if (g->slot[i].pinned) (void)hipHostFree(g->slot[i].pinned);
        if (g->slot[i].dev) (void)hipFree(g->slot[i].dev);
        if (g->slot[i].h2d_done) (void)hipEventDestroy(g->slot[i].h2d_done);
        if (g->dev_free[i]) (void)hipEventDestroy(g->dev_free[i]);
    }
    if (g->fd >= 0) close(g->fd);
    delete g;
}

} // namespace bzq
