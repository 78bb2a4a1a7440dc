// Process-wide cache of the LARGE buffers a file-backed stream needs (round 4).
//
// Opening a file for the ingest pipeline pins three chunk-sized host buffers and allocates their device twins; closing it unpins
// and frees them.  Measured on MI355X (bench.py ingest_mode, 256 MiB chunks): ~40 ms for the open and ~40 ms for the close, i.e.
// 80 of the 195 ms a 6.4 GB file takes from open to close -- hipHostMalloc pins pages at ~20 GB/s and hipHostFree unpins them at
// the same rate.  A host that parses file after file (the reference's runner opens one parser per file,
// benchmark/throughput/run_throughput_blazeseq.mojo:28-55) pays that per file.  The buffers are therefore handed back to this cache
// instead of the driver and the next open of the process takes them from here: same capacity class, same device.
//
// Rules: a buffer is only put back when no work that touches it is in flight (the callers synchronise first -- the cache skips the
// implicit device synchronisation of hipFree); a request is served by the smallest cached buffer of the device with
// want <= capacity <= 1.5 x want; the cache holds at most `limit` bytes per kind (options pin_cache_bytes / dev_cache_bytes, default
// 1 GiB each = one set of chunk buffers -- rounds 3-4 kept 2 GiB pinned and 8 GiB device by default; 0 drops everything held and turns the cache off; environment BZQ_BUF_CACHE=0 does the same for a
// whole process); what does not fit goes back to the driver.  Buffers still cached when the process ends are the operating system's
// to reclaim (the HIP runtime may already be gone in a static destructor).
#pragma once
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <mutex>
#include <unordered_map>
#include <vector>
#include <sys/mman.h>

namespace bzq { namespace cache {

// Host buffers are PINNED LAZILY (round 5).  What hipHostMalloc costs is not the pinning but the pages: 49 ms for 288 MiB, of which
// touching (zeroing) the fresh pages is 50 and registering them with the driver 4 (scripts/probes/mmap_register_probe.hip) -- and
// three such buffers were 175-225 ms of a fresh process's bzq_ingest_open, more than its whole file took to parse.  A host buffer
// is therefore anonymous memory (transparent huge pages asked for) that nobody touches at the open; the reader threads' own pread
// is what faults its pages in, eight threads side by side, and `pin` registers it in blocks of 32 MiB just before the first copy
// that reads from a block (hipHostRegister: ~0.5 ms per block once its pages exist).  A block stays registered while the buffer
// lives, in the cache too.
constexpr uint64_t PIN_BLOCK = 32ull << 20;
struct Entry { void* p; uint64_t cap; int device; uint64_t pinned_blocks = 0; /* bit k: block k is registered (caps up to 2 GiB) */ };

struct Pool {
    const bool pinned;
    std::mutex mu;
    std::vector<Entry> idle;
    std::unordered_map<void*, Entry> out;   // handed out: capacity and device by pointer
    uint64_t held = 0, limit;
    uint64_t hits = 0, misses = 0;
    // Buffers below 1 MiB (block tables, verdict words) are kept as well, up to SMALL_MAX of them whatever `limit` says (unless the cache
    // is off): handing them to hipFree would make every close a device-wide wait -- hipFree waits for ALL streams of the device.
    static constexpr int SMALL_MAX = 64;
    int n_small = 0;
    Pool(bool pin, uint64_t lim) : pinned(pin), limit(lim) {
        const char* e = getenv("BZQ_BUF_CACHE");
        if (e && e[0] == '0') limit = 0;
    }
    static bool tracing() { static const bool t = getenv("BZQ_BUF_CACHE_TRACE") != nullptr; return t; }   // (debug: every driver call that takes > 2 ms, to stderr)
    hipError_t raw_alloc(void** p, uint64_t n) {
        const auto t0 = std::chrono::steady_clock::now();
        hipError_t e = hipSuccess;
        if (pinned) {
            void* m = mmap(nullptr, (size_t)n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            if (m == MAP_FAILED) { *p = nullptr; e = hipErrorOutOfMemory; }
            else { (void)madvise(m, (size_t)n, MADV_HUGEPAGE); *p = m; }
        } else e = hipMalloc(p, n);
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (tracing() && ms > 2.0) fprintf(stderr, "[bzq cache] %s(%.1f MiB): %.1f ms\n", pinned ? "mmap" : "hipMalloc", n / 1048576.0, ms);
        return e;
    }
    void raw_free(const Entry& en) {
        const auto t0 = std::chrono::steady_clock::now();
        if (pinned) {
            for (uint64_t k = 0; k * PIN_BLOCK < en.cap; ++k)
                if (en.pinned_blocks >> k & 1ull) (void)hipHostUnregister((uint8_t*)en.p + k * PIN_BLOCK);
            (void)munmap(en.p, (size_t)en.cap);
        } else (void)hipFree(en.p);
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (tracing() && ms > 2.0) fprintf(stderr, "[bzq cache] %s: %.1f ms\n", pinned ? "unregister + munmap" : "hipFree", ms);
    }
    // Host buffers: bytes [from, to) of buffer p are about to be read by a copy engine -- register the blocks they touch that are not
    // registered yet.  Their pages should exist by now (the caller has just written them); a block that cannot be registered is
    // left as it is (the copy then goes through the runtime's own staging: slower, not wrong).
    void pin(void* p, uint64_t from, uint64_t to) {
        if (!pinned || !p || to <= from) return;
        std::lock_guard<std::mutex> lk(mu);
        auto it = out.find(p);
        if (it == out.end()) return;
        Entry& en = it->second;
        if (to > en.cap) to = en.cap;
        for (uint64_t k = from / PIN_BLOCK; k * PIN_BLOCK < to && k < 64; ++k) {
            if (en.pinned_blocks >> k & 1ull) continue;
            const uint64_t b0 = k * PIN_BLOCK, b1 = std::min<uint64_t>(en.cap, b0 + PIN_BLOCK);
            const auto t0 = std::chrono::steady_clock::now();
            const hipError_t e = hipHostRegister((uint8_t*)en.p + b0, (size_t)(b1 - b0), hipHostRegisterDefault);
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (tracing() && ms > 2.0) fprintf(stderr, "[bzq cache] hipHostRegister(block %llu, %.1f MiB): %.1f ms\n", (unsigned long long)k, (b1 - b0) / 1048576.0, ms);
            if (e == hipSuccess) en.pinned_blocks |= 1ull << k; else (void)hipGetLastError();
        }
    }

    // hipMemcpyAsync(dst, src, n, host to device, st) for a source inside a buffer of this pool: the blocks it touches are registered
    // first, and the copy is issued block by block -- every block is a registration of its own, and one copy must not span two.
    // A source that is not in a buffer of this pool is copied as it is.
    hipError_t h2d(void* dst, const void* src, uint64_t n, hipStream_t st) {
        if (!n) return hipSuccess;
        const uint8_t* base = nullptr;
        {
            std::lock_guard<std::mutex> lk(mu);
            for (const auto& kv : out)
                if ((const uint8_t*)src >= (const uint8_t*)kv.second.p && (const uint8_t*)src + n <= (const uint8_t*)kv.second.p + kv.second.cap) { base = (const uint8_t*)kv.second.p; break; }
        }
        if (!pinned || !base) return hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, st);
        const uint64_t off = (uint64_t)((const uint8_t*)src - base);
        pin((void*)base, off, off + n);
        for (uint64_t a = off; a < off + n;) {
            const uint64_t b = std::min<uint64_t>(off + n, (a / PIN_BLOCK + 1) * PIN_BLOCK);
            const hipError_t e = hipMemcpyAsync((uint8_t*)dst + (a - off), base + a, (size_t)(b - a), hipMemcpyHostToDevice, st);
            if (e != hipSuccess) return e;
            a = b;
        }
        return hipSuccess;
    }

    // debug: BZQ_BUF_CACHE_POISON=<byte> fills every buffer handed out with that byte (a user of the buffers that relies on what a
    // fresh allocation happens to contain shows itself)
    void poison(void* p, uint64_t n) {
        static const char* e = getenv("BZQ_BUF_CACHE_POISON");
        if (!e) return;
        const int v = (int)strtol(e, nullptr, 0) & 0xFF;
        if (pinned) memset(p, v, n);   // (touches every page: a debugging aid, not for timing)
        else { (void)hipMemset(p, v, n); (void)hipDeviceSynchronize(); }
    }
    hipError_t get(int device, uint64_t want, void** outp) {
        const hipError_t e = get_(device, want, outp);
        if (e == hipSuccess) poison(*outp, want);
        return e;
    }
    // *out gets a buffer of at least `want` bytes on / for `device` (the caller has set the device)
    hipError_t get_(int device, uint64_t want, void** outp) {
        *outp = nullptr;
        {
            std::lock_guard<std::mutex> lk(mu);
            int best = -1;
            for (int i = 0; i < (int)idle.size(); ++i)
                if (idle[i].device == device && idle[i].cap >= want && idle[i].cap <= want + want / 2 && (best < 0 || idle[i].cap < idle[best].cap)) best = i;
            if (best >= 0) {
                const Entry e = idle[best];
                idle.erase(idle.begin() + best);
                held -= e.cap;
                if (e.cap < (1ull << 20)) --n_small;
                out[e.p] = e;
                *outp = e.p;
                ++hits;
                return hipSuccess;
            }
            ++misses;
        }
        void* p = nullptr;
        hipError_t err = raw_alloc(&p, want);
        if (err != hipSuccess) {   // the cache itself may be what is in the way
            trim(0, true);
            err = raw_alloc(&p, want);
            if (err != hipSuccess) return err;
        }
        std::lock_guard<std::mutex> lk(mu);
        out[p] = Entry{p, want, device, 0};
        *outp = p;
        return hipSuccess;
    }
    // a pointer from get(); nothing in flight may touch it any more
    void put(void* p) {
        if (!p) return;
        Entry e{p, 0, 0, 0};
        bool keep = false, known = false;
        {
            std::lock_guard<std::mutex> lk(mu);
            auto it = out.find(p);
            if (it != out.end()) {
                e = it->second;
                known = true;
                out.erase(it);
                const bool small = e.cap < (1ull << 20);
                if (limit && ((small && n_small < SMALL_MAX) || (!small && held + e.cap <= limit))) { idle.push_back(e); held += e.cap; n_small += small ? 1 : 0; keep = true; }
            }
        }
        if (!keep && known) raw_free(e);
    }
    // the counters, read under the lock (the option queries of bzq_set_option: another thread may be inside get / put)
    uint64_t hits_now() { std::lock_guard<std::mutex> lk(mu); return hits; }
    uint64_t held_now() { std::lock_guard<std::mutex> lk(mu); return held; }
    // give buffers back to the driver until at most `keep_bytes` are held; set_limit: that is also the new limit
    void trim(uint64_t keep_bytes, bool keep_limit) {
        std::vector<Entry> drop;
        {
            std::lock_guard<std::mutex> lk(mu);
            if (!keep_limit) limit = keep_bytes;
            while (held > keep_bytes && !idle.empty()) {
                drop.push_back(idle.back());
                held -= idle.back().cap;
                if (idle.back().cap < (1ull << 20)) --n_small;
                idle.pop_back();
            }
        }
        for (const Entry& e : drop) raw_free(e);
    }
};

// defaults: ONE set of chunk buffers each (three 288 MiB slots) -- what a process that reads plain or BGZF files one after the other
// needs.  A host that decodes .gz after .gz raises dev_cache_bytes (its pools and FIFO are ~8 GiB per stream).
inline Pool& pinned_pool() { static Pool* p = new Pool(true, 1ull << 30); return *p; }
inline Pool& device_pool() { static Pool* p = new Pool(false, 1ull << 30); return *p; }

// Small host buffers the DEVICE writes into (verdict words, block tables): really pinned (hipHostMalloc), and never given back to the
// driver while the process lives -- hipHostFree, like hipFree, waits for every stream of the device.  A free list by size; at most
// HOST_SMALL_MAX buffers are kept, the rest is freed (with that wait).
struct HostSmall {
    std::mutex mu;
    std::vector<std::pair<void*, uint64_t>> idle;
    static constexpr size_t HOST_SMALL_MAX = 32;
    void* get(uint64_t n) {
        {
            std::lock_guard<std::mutex> lk(mu);
            for (size_t i = 0; i < idle.size(); ++i)
                if (idle[i].second == n) { void* p = idle[i].first; idle.erase(idle.begin() + (long)i); return p; }
        }
        void* p = nullptr;
        if (hipHostMalloc(&p, (size_t)n, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        return p;
    }
    void put(void* p, uint64_t n) {
        if (!p) return;
        {
            std::lock_guard<std::mutex> lk(mu);
            if (idle.size() < HOST_SMALL_MAX) { idle.emplace_back(p, n); return; }
        }
        (void)hipHostFree(p);
    }
};
inline HostSmall& host_small() { static HostSmall* h = new HostSmall(); return *h; }

// Helper streams (non-blocking, default priority class: copy / inflate / finder streams of the ingest and the gzip decoder) are kept
// for the process instead of being destroyed at a close: hipStreamDestroy finishes the stream's hardware queue with a marker
// (0.3-0.6 ms even when idle) -- and the runtime maps streams onto FOUR hardware queues per priority class, so that marker can land
// behind a long kernel of a caller's stream that happens to share the queue (seen once in round 6: a close waited 1.5 s for another
// stream's spin kernel).  A stream from here continues in order behind whatever its previous user left; callers synchronise it
// before they put it back if work of theirs may still be in flight.
struct StreamPool {
    std::mutex mu;
    std::vector<std::pair<int, hipStream_t>> idle;   // (device, stream)
    static constexpr size_t MAX_IDLE = 24;
    hipError_t get(int device, hipStream_t* out) {
        {
            std::lock_guard<std::mutex> lk(mu);
            for (size_t i = 0; i < idle.size(); ++i)
                if (idle[i].first == device) { *out = idle[i].second; idle.erase(idle.begin() + (long)i); return hipSuccess; }
        }
        // Helper streams live in the LOWEST priority class: a class has hardware queues of its own, so a helper stream never shares
        // one with a caller's default-class stream (a consumer's) nor with the parser's (highest class).  Measured in the default
        // bench line (round 6): with pooled helper streams of the default class the BGZF pipeline figure fell from 35 to 22 GB/s of
        // FASTQ -- the consumer stream had landed in a hardware queue with an inflate stream, and 5 ms inflate kernels sat in front of
        // its kernels -- in the lowest class it is 33-35 again and nothing else moved.  BZQ_HELPER_PRIO=default: the default class (A/B).
        static const bool low = [] { const char* e = getenv("BZQ_HELPER_PRIO"); return !(e && e[0] == 'd'); }();
        int lo = 0, hi = 0;
        if (low && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && lo > hi) return hipStreamCreateWithPriority(out, hipStreamNonBlocking, lo);
        return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
    }
    void put(int device, hipStream_t s) {
        if (!s) return;
        {
            std::lock_guard<std::mutex> lk(mu);
            if (idle.size() < MAX_IDLE) { idle.emplace_back(device, s); return; }
        }
        (void)hipStreamDestroy(s);
    }
};
inline StreamPool& stream_pool() { static StreamPool* p = new StreamPool(); return *p; }

template <class T> inline bool get_pinned(int device, uint64_t want, T** out) { return pinned_pool().get(device, want, (void**)out) == hipSuccess; }
template <class T> inline bool get_device(int device, uint64_t want, T** out) { return device_pool().get(device, want, (void**)out) == hipSuccess; }

}} // namespace bzq::cache
