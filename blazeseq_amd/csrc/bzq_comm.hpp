// bzq_comm.hpp -- the multi-GPU shard protocol behind the C ABI (SURVEY.md 8b last row, 8e): bzq_comm_init /
// bzq_shard_stitch / bzq_global_counts.  New design; the reference is a single process.  A host in any language reaches
// it the way the reference reaches libz (blazeseq/io/readers.mojo:226-280: OwnedDLHandle + get_function, raw pointers and
// sizes, negative int = error) -- and this file binds RCCL the same way: librccl.so.1 is dlopen'ed when a communicator is
// created, so a process that never shards does not need it.
//
// One process per GPU, the stream cut into contiguous BYTE ranges (not record aligned), rank r holds range r:
//   1. every rank scans its shard once (pass A + tile scan + first four newlines)        -> all-gather of 8 x int64 per rank
//   2. every rank plans (bzq_plan_shards, a pure function of the gathered summaries): the line index of its first byte, the
//      leading bytes that belong to a record an earlier rank owns (its "head"), the bytes it receives behind its own (its
//      "halo" = the heads of the following ranks up to the end of its last record -- more than one rank when a record is
//      longer than a whole shard)                                                       -> one grouped send/recv per rank
//   3. every rank parses [own bytes + halo] (pass A of step 1 is reused)                 -> all-gather of 8 x int64 per rank
//   4. only when the stream ends in bytes that are not a record: the reference's BufferedReader window is walked through
//      the ranks' record ends in rank order, so that BUFFER_EXCEEDED / UNEXPECTED_EOF / accepted-last-record come out
//      exactly where the sequential parser gives them (parser.mojo:464-510, buffered.mojo:276-279)
// No bulk data crosses xGMI: messages are bytes to kilobytes, latency bound.
//
// Transports: RCCL (ncclAllGather / ncclSend / ncclRecv on the ctx stream, one GPU per rank) and "shm" -- the same three
// steps through a POSIX shared-memory segment on the host (SURVEY.md 8e "fallback via host"): same-node ranks under any
// GPU assignment, including several ranks on ONE GPU, which RCCL refuses ("Duplicate GPU detected") and which is how the
// two-process tests run on a one-GPU box.
//
// Included by bzq_api.hip inside its extern "C" block.
#pragma once

#include <dlfcn.h>
#include <sched.h>
#include <sys/mman.h>

struct bzq_comm {
    int rank = 0, nranks = 1;
    int kind = 0;   // 1 RCCL, 2 shm
    // ---- RCCL, bound at run time
    void* dl = nullptr;
    void* nccl = nullptr;   // ncclComm_t
    int (*p_CommInitRank)(void**, int, bzq_nccl_id, int) = nullptr;
    int (*p_CommDestroy)(void*) = nullptr;
    int (*p_CommAbort)(void*) = nullptr;
    int (*p_AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*p_Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*p_Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*p_GroupStart)() = nullptr;
    int (*p_GroupEnd)() = nullptr;
    const char* (*p_GetErrorString)(int) = nullptr;
    int64_t* d_row = nullptr;   // [ROW]
    int64_t* d_all = nullptr;   // [nranks * ROW]
    int64_t* h_row = nullptr;   // pinned
    int64_t* h_all = nullptr;   // pinned
    // ---- shm
    int shm_fd = -1;
    uint8_t* seg = nullptr;
    size_t seg_bytes = 0;
    uint64_t halo_cap = 0;
    std::string shm_name;
    // ---- deadlines: every exchange has a host-side limit (option "comm_timeout_ms"); when it passes, the call fails with the
    // ranks that never arrived BY NAME.  Over RCCL nothing tells a rank who is missing, so the ranks of one node also keep a
    // presence board in shared memory ("/bzq_board_<hash of the ncclUniqueId>": one counter per rank, bumped on entering an
    // exchange); the shm transport keeps the same counters inside its own segment.
    int board_fd = -1;
    std::atomic<uint64_t>* board = nullptr;   // [nranks] exchanges entered so far
    std::string board_name;
    uint64_t exchanges = 0;                   // exchanges this rank has entered
};

namespace {

constexpr int COMM_ROW = 8;   // int64 words per rank and exchange
constexpr int NCCL_U8 = 1, NCCL_I64 = 4;   // ncclUint8, ncclInt64 (rccl.h)

struct ShmHeader {
    std::atomic<uint32_t> magic, count, gen;
    uint32_t nranks;
    uint64_t halo_cap;
    std::atomic<uint32_t> attached, go;   // the attach handshake of bzq_comm_init_shm
    uint8_t pad[32];
};
static_assert(sizeof(ShmHeader) == 64, "ShmHeader is one cache line");
constexpr uint32_t SHM_MAGIC = 0x425A5131u;   // "BZQ1"

// segment: header | one arrival counter per rank | the rows of an all-gather | one halo slot per rank
size_t shm_board_bytes(int nranks) { return ((size_t)nranks * 8 + 63) & ~(size_t)63; }
std::atomic<uint64_t>* shm_board(bzq_comm* m) { return (std::atomic<uint64_t>*)(m->seg + sizeof(ShmHeader)); }
int64_t* shm_rows(bzq_comm* m) { return (int64_t*)(m->seg + sizeof(ShmHeader) + shm_board_bytes(m->nranks)); }
uint8_t* shm_halo(bzq_comm* m, int r) { return m->seg + sizeof(ShmHeader) + shm_board_bytes(m->nranks) + (size_t)m->nranks * COMM_ROW * 8 + (size_t)r * m->halo_cap; }

double comm_timeout_s(const bzq_ctx* c);   // option "comm_timeout_ms" (bzq_api.hip)

extern "C++" {   // (this file is included inside an extern "C" block)
// "rank 2, rank 5": the ranks whose arrival counter is behind `want` (nobody: an empty string)
std::string comm_missing(bzq_comm* m, uint64_t want) {
    std::string who;
    if (!m->board) return who;
    for (int r = 0; r < m->nranks; ++r)
        if (m->board[r].load(std::memory_order_acquire) < want) who += (who.empty() ? "rank " : ", rank ") + std::to_string(r);
    return who;
}
// this rank enters its next exchange
uint64_t comm_arrive(bzq_comm* m) {
    m->exchanges += 1;
    if (m->board) m->board[m->rank].store(m->exchanges, std::memory_order_release);
    return m->exchanges;
}
std::string comm_timeout_text(bzq_ctx* c, bzq_comm* m, uint64_t want, const char* what) {
    const std::string who = comm_missing(m, want);
    char t[32];
    snprintf(t, sizeof t, "%.1f", comm_timeout_s(c));
    return std::string("bzq_comm: ") + what + " did not complete within " + t + " s on rank " + std::to_string(m->rank) +
           (who.empty() ? std::string(" (every rank entered it: the transport itself hangs)") : ": " + who + " never entered it");
}
}   // extern "C++"

int shm_barrier(bzq_ctx* c, bzq_comm* m, const char* what = "a barrier") {
    ShmHeader* h = (ShmHeader*)m->seg;
    const uint64_t want = comm_arrive(m);
    const uint32_t g = h->gen.load(std::memory_order_acquire);
    if (h->count.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)m->nranks) {
        h->count.store(0, std::memory_order_relaxed);
        h->gen.store(g + 1, std::memory_order_release);
        return 0;
    }
    const auto t0 = std::chrono::steady_clock::now();
    const double limit = comm_timeout_s(c);
    for (uint32_t spins = 0; h->gen.load(std::memory_order_acquire) == g; ++spins) {
        if ((spins & 63) == 63) sched_yield();
        if ((spins & 0xFFF) == 0xFFF && bzq::seconds_since(t0) > limit) { c->err = comm_timeout_text(c, m, want, what); return BZQ_ERR_IO; }
    }
    return 0;
}

// RCCL: wait for the ctx stream with a deadline instead of hipStreamSynchronize (a collective a peer never joins does not
// return, and neither does a synchronisation behind it)
int comm_stream_wait(bzq_ctx* c, bzq_comm* m, uint64_t want, const char* what) {
    const auto t0 = std::chrono::steady_clock::now();
    const double limit = comm_timeout_s(c);
    for (uint32_t spins = 0;; ++spins) {
        const hipError_t q = hipStreamQuery(c->stream);
        if (q == hipSuccess) return 0;
        if (q != hipErrorNotReady) { c->err = std::string(what) + ": " + hipGetErrorString(q); return BZQ_ERR_HIP; }
        if (spins < 2000) sched_yield(); else usleep(50);
        if ((spins & 0xFF) == 0xFF && bzq::seconds_since(t0) > limit) {
            c->err = comm_timeout_text(c, m, want, what);
            // the collective will never finish: take the communicator down so that the stream (and the process) can go on
            if (m->nccl && m->p_CommAbort) { (void)m->p_CommAbort(m->nccl); m->nccl = nullptr; }
            return BZQ_ERR_IO;
        }
    }
}

#define NCCLCHK(c, m, call)                                                                                         \
    do {                                                                                                            \
        const int r_ = (call);                                                                                      \
        if (r_ != 0) {                                                                                              \
            (c)->err = std::string(#call) + ": " + ((m)->p_GetErrorString ? (m)->p_GetErrorString(r_) : "RCCL error"); \
            return BZQ_ERR_HIP;                                                                                     \
        }                                                                                                           \
    } while (0)

// all-gather of COMM_ROW int64 per rank, host to host.  `row` may be nullptr when d_row already holds the row on the
// stream (RCCL only: the scan packed it there).
int comm_gather(bzq_ctx* c, const int64_t* row, int64_t* all, const char* what = "an all-gather") {
    bzq_comm* m = c->comm;
    if (!m) { if (row) memcpy(all, row, COMM_ROW * 8); return 0; }
    if (m->kind == 1 && !m->nccl) { c->err = std::string(what) + ": the communicator was taken down after an exchange timed out (bzq_comm_destroy + bzq_comm_init to start over)"; return BZQ_ERR_IO; }
    if (m->kind == 1) {
        if (row) {
            memcpy(m->h_row, row, COMM_ROW * 8);
            HIPCHK(c, hipMemcpyAsync(m->d_row, m->h_row, COMM_ROW * 8, hipMemcpyHostToDevice, c->stream));
        }
        const uint64_t want = comm_arrive(m);
        NCCLCHK(c, m, m->p_AllGather(m->d_row, m->d_all, COMM_ROW, NCCL_I64, m->nccl, c->stream));
        HIPCHK(c, hipMemcpyAsync(m->h_all, m->d_all, (size_t)m->nranks * COMM_ROW * 8, hipMemcpyDeviceToHost, c->stream));
        int wrc;
        if ((wrc = comm_stream_wait(c, m, want, what))) return wrc;
        memcpy(all, m->h_all, (size_t)m->nranks * COMM_ROW * 8);
        return 0;
    }
    int rc;
    memcpy(shm_rows(m) + (size_t)m->rank * COMM_ROW, row, COMM_ROW * 8);
    if ((rc = shm_barrier(c, m, what))) return rc;
    memcpy(all, shm_rows(m), (size_t)m->nranks * COMM_ROW * 8);
    return shm_barrier(c, m, what);   // nobody overwrites its row before everybody has read
}

// row = {bytes, newlines, first four newlines, first byte | last byte << 8, room behind the shard's bytes}
static __global__ void k_pack_summary(const ChunkState* st, int64_t n, int64_t room, int64_t* row, int64_t stamp) {
    if (threadIdx.x || blockIdx.x) return;
    row[0] = n; row[1] = st->P;
    for (int i = 0; i < 4; ++i) row[2 + i] = st->first_nl[i];
    row[6] = (int64_t)st->edge_first | ((int64_t)st->edge_last << 8) | (stamp << 16); row[7] = room;
}

void comm_free(bzq_comm* m) {
    if (!m) return;
    if (m->kind == 1) {
        if (m->nccl && m->p_CommDestroy) (void)m->p_CommDestroy(m->nccl);
        if (m->d_row) (void)hipFree(m->d_row);
        if (m->d_all) (void)hipFree(m->d_all);
        if (m->h_row) (void)hipHostFree(m->h_row);
        if (m->h_all) (void)hipHostFree(m->h_all);
        // the library handle stays open: RCCL keeps process-wide state and is shared with other users (torch)
        if (m->board) munmap((void*)m->board, (size_t)m->nranks * 8);
        if (m->board_fd >= 0) close(m->board_fd);
        if (m->rank == 0 && !m->board_name.empty()) shm_unlink(m->board_name.c_str());
    } else if (m->kind == 2) {
        if (m->seg) munmap(m->seg, m->seg_bytes);
        if (m->shm_fd >= 0) close(m->shm_fd);
        if (m->rank == 0 && !m->shm_name.empty()) shm_unlink(m->shm_name.c_str());
    }
    delete m;
}

} // namespace

// ---- planning: a pure function of the gathered summaries (CPU-testable; tests/test_shard_plan.py) -----------------------

int32_t bzq_plan_shards(const bzq_shard_summary* all, int32_t nranks, bzq_shard_plan* out) {
    if (!all || !out || nranks <= 0) return BZQ_ERR_ARG;
    uint64_t lines = 0;
    uint8_t prev_byte = 10;
    int owner = -1, last_owner = -1;
    for (int r = 0; r < nranks; ++r) {
        const bzq_shard_summary& s = all[r];
        bzq_shard_plan& p = out[r];
        memset(&p, 0, sizeof(p));
        p.lines_before = lines; p.prev_last_byte = prev_byte; p.head_dst = -1; p.halo_first_src = -1;
        if (s.n_bytes == 0) continue;
        uint64_t head = 0;
        const int p0 = (int)(lines & 3);
        if (!(prev_byte == 10 && p0 == 0)) {
            const int k = 4 - p0;   // newlines left of the record that started in an earlier shard
            head = s.first_nl[k - 1] >= 0 ? (uint64_t)s.first_nl[k - 1] + 1 : s.n_bytes;   // the whole shard: the record goes on
        }
        p.head_bytes = head;
        if (head > 0) {
            if (owner < 0) return BZQ_ERR_ARG;   // cannot happen: the stream's first byte starts a record
            bzq_shard_plan& o = out[owner];
            p.head_dst = owner;
            p.halo_offset = o.halo_bytes;
            o.halo_bytes += head;
            if (o.halo_first_src < 0) o.halo_first_src = r;
            o.halo_n_src = r - o.halo_first_src + 1;
        }
        if (head < s.n_bytes) { owner = r; last_owner = r; }
        lines += s.n_newlines;
        prev_byte = s.last_byte;
    }
    out[last_owner >= 0 ? last_owner : 0].is_last = 1;
    return 0;
}

// ---- communicators --------------------------------------------------------------------------------------------------------

int32_t bzq_comm_get_unique_id(bzq_nccl_id* id_out) {
    if (!id_out) return BZQ_ERR_ARG;
    void* dl = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!dl) dl = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!dl) return BZQ_ERR_IO;
    auto f = (int (*)(bzq_nccl_id*))dlsym(dl, "ncclGetUniqueId");
    return f && f(id_out) == 0 ? 0 : BZQ_ERR_HIP;
}

int32_t bzq_comm_destroy(bzq_ctx* c) {
    if (!c) return BZQ_ERR_ARG;
    if (c->comm) { (void)hipSetDevice(c->device); (void)hipStreamSynchronize(c->stream); comm_free(c->comm); c->comm = nullptr; }
    return 0;
}

int32_t bzq_comm_init(bzq_ctx* c, int32_t rank, int32_t nranks, const void* nccl_id) {
    if (!c || nranks <= 0 || rank < 0 || rank >= nranks || !nccl_id) return BZQ_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    (void)bzq_comm_destroy(c);
    bzq_comm* m = new bzq_comm();
    m->rank = rank; m->nranks = nranks; m->kind = 1;
    {
        m->dl = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!m->dl) m->dl = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!m->dl) { c->err = std::string("bzq_comm_init: cannot load librccl.so.1: ") + dlerror(); delete m; return BZQ_ERR_IO; }
#define BZQ_SYM(field, name) *(void**)(&m->field) = dlsym(m->dl, name)
        BZQ_SYM(p_CommInitRank, "ncclCommInitRank"); BZQ_SYM(p_CommDestroy, "ncclCommDestroy"); BZQ_SYM(p_CommAbort, "ncclCommAbort"); BZQ_SYM(p_AllGather, "ncclAllGather");
        BZQ_SYM(p_Send, "ncclSend"); BZQ_SYM(p_Recv, "ncclRecv"); BZQ_SYM(p_GroupStart, "ncclGroupStart"); BZQ_SYM(p_GroupEnd, "ncclGroupEnd");
        BZQ_SYM(p_GetErrorString, "ncclGetErrorString");
#undef BZQ_SYM
        if (!m->p_CommInitRank || !m->p_CommDestroy || !m->p_AllGather || !m->p_Send || !m->p_Recv || !m->p_GroupStart || !m->p_GroupEnd) {
            c->err = "bzq_comm_init: librccl.so.1 lacks a required symbol"; delete m; return BZQ_ERR_IO;
        }
        bzq_nccl_id id;
        memcpy(&id, nccl_id, sizeof(id));
        const int r = m->p_CommInitRank(&m->nccl, nranks, id, rank);
        if (r != 0) {
            c->err = std::string("ncclCommInitRank: ") + (m->p_GetErrorString ? m->p_GetErrorString(r) : "failed");
            m->nccl = nullptr; comm_free(m);
            return BZQ_ERR_HIP;
        }
        if (hipMalloc((void**)&m->d_row, COMM_ROW * 8) != hipSuccess || hipMalloc((void**)&m->d_all, (size_t)nranks * COMM_ROW * 8) != hipSuccess ||
            hipHostMalloc((void**)&m->h_row, COMM_ROW * 8, hipHostMallocDefault) != hipSuccess ||
            hipHostMalloc((void**)&m->h_all, (size_t)nranks * COMM_ROW * 8, hipHostMallocDefault) != hipSuccess) {
            c->err = "bzq_comm_init: staging buffers"; comm_free(m); return BZQ_ERR_NOMEM;
        }
        // presence board of the node's ranks (best effort: without it a timeout still fails the call, it just cannot say who).
        // Every rank opens-or-creates the same name (a fresh object is zero-filled; counters only grow), derived from the id.
        {
            uint64_t hsh = 1469598103934665603ull;
            for (size_t i = 0; i < sizeof(id.internal); ++i) hsh = (hsh ^ (uint8_t)id.internal[i]) * 1099511628211ull;
            char nm[64];
            snprintf(nm, sizeof nm, "/bzq_board_%016llx", (unsigned long long)hsh);
            m->board_name = nm;
            m->board_fd = shm_open(nm, O_CREAT | O_RDWR, 0600);
            if (m->board_fd >= 0 && ftruncate(m->board_fd, (off_t)((size_t)nranks * 8)) == 0) {
                void* p = mmap(nullptr, (size_t)nranks * 8, PROT_READ | PROT_WRITE, MAP_SHARED, m->board_fd, 0);
                if (p != MAP_FAILED) m->board = (std::atomic<uint64_t>*)p;
            }
            if (!m->board) { if (m->board_fd >= 0) close(m->board_fd); m->board_fd = -1; m->board_name.clear(); }
        }
    }
    c->comm = m;
    return 0;
}

// A segment of a crashed run may still carry the name (fully initialised, magic set, the same nranks).  Rank 0 always unlinks
// the name and creates a fresh segment; a rank > 0 that got there first must not settle on the stale one.  The handshake: a
// rank > 0 attaches (counter), then waits for `go` to CHANGE -- rank 0 sets it on ITS segment once nranks - 1 ranks have
// attached there -- and while it waits it keeps checking that the name still leads to the segment it holds (same inode).  On a
// stale segment `go` never changes and the name moves on as soon as rank 0 arrives: the rank drops it and starts over.
int32_t bzq_comm_init_shm(bzq_ctx* c, int32_t rank, int32_t nranks, const char* name, uint64_t halo_capacity) {
    if (!c || nranks <= 0 || rank < 0 || rank >= nranks || !name || !*name) return BZQ_ERR_ARG;
    (void)bzq_comm_destroy(c);
    bzq_comm* m = new bzq_comm();
    m->rank = rank; m->nranks = nranks; m->kind = 2;
    m->halo_cap = halo_capacity ? halo_capacity : (4ull << 20);
    m->shm_name = std::string("/bzq_") + name;
    m->seg_bytes = sizeof(ShmHeader) + shm_board_bytes(nranks) + (size_t)nranks * COMM_ROW * 8 + (size_t)nranks * m->halo_cap;
    const auto t0 = std::chrono::steady_clock::now();
    auto drop = [&]() {
        if (m->seg) { munmap(m->seg, m->seg_bytes); m->seg = nullptr; }
        if (m->shm_fd >= 0) { close(m->shm_fd); m->shm_fd = -1; }
    };
    auto fail = [&](const std::string& msg, int32_t code) { c->err = msg; const int r0 = m->rank; if (r0 != 0) m->shm_name.clear(); comm_free(m); return code; };
    if (rank == 0) {
        shm_unlink(m->shm_name.c_str());   // a stale segment of a crashed run
        m->shm_fd = shm_open(m->shm_name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
        if (m->shm_fd < 0 || ftruncate(m->shm_fd, (off_t)m->seg_bytes) != 0) return fail("bzq_comm_init_shm: cannot create " + m->shm_name, BZQ_ERR_IO);
        void* p = mmap(nullptr, m->seg_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, m->shm_fd, 0);
        if (p == MAP_FAILED) return fail("bzq_comm_init_shm: mmap failed", BZQ_ERR_IO);
        m->seg = (uint8_t*)p;
        ShmHeader* h = (ShmHeader*)m->seg;   // (a fresh segment is zero-filled)
        h->count.store(0); h->gen.store(0); h->attached.store(0); h->go.store(0); h->nranks = (uint32_t)nranks; h->halo_cap = m->halo_cap;
        h->magic.store(SHM_MAGIC, std::memory_order_release);
        while (h->attached.load(std::memory_order_acquire) != (uint32_t)(nranks - 1)) {
            if (bzq::seconds_since(t0) > 120.0) return fail("bzq_comm_init_shm: only " + std::to_string(h->attached.load()) + " of " + std::to_string(nranks - 1) + " peers attached to " + m->shm_name + " within 120 s", BZQ_ERR_IO);
            usleep(200);
        }
        h->go.store(1, std::memory_order_release);
    } else {
        for (;;) {
            if (bzq::seconds_since(t0) > 120.0) return fail("bzq_comm_init_shm: rank 0 never offered " + m->shm_name, BZQ_ERR_IO);
            drop();
            m->shm_fd = shm_open(m->shm_name.c_str(), O_RDWR, 0600);
            struct stat st;
            if (m->shm_fd < 0 || fstat(m->shm_fd, &st) != 0 || (size_t)st.st_size < m->seg_bytes) { usleep(1000); continue; }   // not created / not sized yet
            void* p = mmap(nullptr, m->seg_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, m->shm_fd, 0);
            if (p == MAP_FAILED) return fail("bzq_comm_init_shm: mmap failed", BZQ_ERR_IO);
            m->seg = (uint8_t*)p;
            ShmHeader* h = (ShmHeader*)m->seg;
            auto name_moved_on = [&]() {   // the name no longer leads to the segment we hold
                const int fd2 = shm_open(m->shm_name.c_str(), O_RDWR, 0600);
                if (fd2 < 0) return true;
                struct stat s2;
                const bool moved = fstat(fd2, &s2) != 0 || s2.st_ino != st.st_ino || s2.st_dev != st.st_dev;
                close(fd2);
                return moved;
            };
            bool again = false;
            while (h->magic.load(std::memory_order_acquire) != SHM_MAGIC) {
                if (bzq::seconds_since(t0) > 120.0) return fail("bzq_comm_init_shm: segment never initialised", BZQ_ERR_IO);
                usleep(1000);
                if (name_moved_on()) { again = true; break; }
            }
            if (again) continue;
            const bool agree = h->nranks == (uint32_t)nranks && h->halo_cap == m->halo_cap;
            const uint32_t g0 = h->go.load(std::memory_order_acquire);
            if (agree) h->attached.fetch_add(1, std::memory_order_acq_rel);
            for (uint32_t spins = 0;; ++spins) {
                if (agree && h->go.load(std::memory_order_acquire) != g0) break;
                if ((spins & 15) == 15 && name_moved_on()) { again = true; break; }
                if (bzq::seconds_since(t0) > 120.0)
                    return fail(agree ? "bzq_comm_init_shm: rank 0 never completed the handshake on " + m->shm_name
                                      : std::string("bzq_comm_init_shm: ranks disagree on nranks / halo capacity"), agree ? BZQ_ERR_IO : BZQ_ERR_ARG);
                usleep(200);
            }
            if (!again) break;
        }
    }
    c->comm = m;
    m->board = shm_board(m);   // (inside the segment: unmapped with it)
    return shm_barrier(c, m, "the attach barrier");   // everybody is attached (rank 0 may unlink the name only after all have opened it)
}

// Heads travel to their owners (step 3 of the protocol; also what bzq_comm_selftest drives with synthetic plans): rank q's first
// head_bytes bytes go behind the n bytes of rank head_dst, at halo_offset.  One grouped ncclSend / ncclRecv per rank -- a rank with
// nothing to send and nothing to receive still opens and closes its (empty) group -- or two barriers around the host segment.  A
// group that was opened is closed whatever happens inside it; a failure only this rank sees is noted (lrc / lerr) and the barriers
// are still met; the return value is a failure of the transport itself (a deadline that passed included).
extern "C++" {
template <typename Plan>
int comm_exchange_heads(bzq_ctx* c, bzq_comm* m, uint8_t* d_shard, uint64_t n, const std::vector<Plan>& plans, int& lrc, std::string& lerr) {
    const int P = m ? m->nranks : 1, me = m ? m->rank : 0;
    if (!m || P <= 1) return 0;
    const Plan& pl = plans[(size_t)me];
    auto note_hip = [&](hipError_t e, const char* what) { if (e != hipSuccess && !lrc) { lrc = BZQ_ERR_HIP; lerr = std::string(what) + ": " + hipGetErrorString(e); } };
    int rc;
    if (m->kind == 1 && !m->nccl) { c->err = "the exchange of the shard heads: the communicator was taken down after an exchange timed out"; return BZQ_ERR_IO; }
    if (m->kind == 1) {
        auto nccl_note = [&](int r, const char* what) { if (r != 0 && !lrc) { lrc = BZQ_ERR_HIP; lerr = std::string(what) + ": " + (m->p_GetErrorString ? m->p_GetErrorString(r) : "RCCL error"); } return r; };
        const uint64_t want = comm_arrive(m);
        if (nccl_note(m->p_GroupStart(), "ncclGroupStart") == 0) {
            if (pl.head_bytes > 0) nccl_note(m->p_Send(d_shard, (size_t)pl.head_bytes, NCCL_U8, pl.head_dst, m->nccl, c->stream), "ncclSend");
            for (int q = pl.halo_first_src; q >= 0 && q < pl.halo_first_src + pl.halo_n_src; ++q)
                if (plans[(size_t)q].head_bytes > 0 && plans[(size_t)q].head_dst == me)
                    nccl_note(m->p_Recv(d_shard + n + plans[(size_t)q].halo_offset, (size_t)plans[(size_t)q].head_bytes, NCCL_U8, q, m->nccl, c->stream), "ncclRecv");
            nccl_note(m->p_GroupEnd(), "ncclGroupEnd");
        }
        // (with a deadline: a peer that never posts its side leaves this rank's stream in the group for ever)
        if ((rc = comm_stream_wait(c, m, want, "the exchange of the heads (grouped ncclSend / ncclRecv)"))) return rc;
        return 0;
    }
    if (pl.head_bytes > 0) note_hip(hipMemcpy(shm_halo(m, me), d_shard, (size_t)pl.head_bytes, hipMemcpyDeviceToHost), "hipMemcpy(head to segment)");
    if ((rc = shm_barrier(c, m, "the exchange of the heads"))) return rc;
    for (int q = pl.halo_first_src; q >= 0 && q < pl.halo_first_src + pl.halo_n_src; ++q)
        if (plans[(size_t)q].head_bytes > 0 && plans[(size_t)q].head_dst == me)
            note_hip(hipMemcpyAsync(d_shard + n + plans[(size_t)q].halo_offset, shm_halo(m, q), (size_t)plans[(size_t)q].head_bytes, hipMemcpyHostToDevice, c->stream), "hipMemcpyAsync(halo)");
    note_hip(hipStreamSynchronize(c->stream), "hipStreamSynchronize(halo)");   // the bytes have left the segment before a peer may overwrite it, and are on the device before anything reads them
    return shm_barrier(c, m, "the exchange of the heads");
}
}   // extern "C++"

// Every exchange the protocol makes, with the shapes it makes them in and the data checked on arrival, before the first real
// step (a host calls it once after bzq_comm_init*): a transport that does not work, or a peer that is not there, is reported
// here -- by name, within the deadline -- instead of inside a step.  It drives comm_gather and comm_exchange_heads themselves:
//   round 0  the summary all-gather from a DEVICE-resident row (the path bzq_shard_stitch takes over RCCL) and from a host row;
//   round 1  every rank but the first sends a head of 40 000 bytes (a long read's remainder) to the rank before it;
//   round 2  only the odd ranks send (313 bytes): the even ranks' groups hold a receive only, the last even rank's group may be
//            EMPTY (nothing to send, nothing to receive -- the group is still opened and closed, as in a real step);
//   round 3  a record longer than a shard: ranks 1 .. min(3, P - 1) all send to rank 0, which receives them at three offsets;
//   round 4  the ring of the earlier versions (rank r -> r + 1; one rank: to itself) -- over RCCL only (the shm slots are per
//            sender and the rounds above cover them).
static __global__ void k_selftest_fill(uint32_t* p, uint32_t seed, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = seed * 2654435761u + (uint32_t)i * 40503u;
}
static __global__ void k_selftest_check(const uint32_t* p, uint32_t seed, int n, int* bad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && p[i] != seed * 2654435761u + (uint32_t)i * 40503u) atomicAdd(bad, 1);
}
static __global__ void k_selftest_row(int64_t* row, int64_t rank) {
    if (threadIdx.x < COMM_ROW) row[threadIdx.x] = rank * 1000 + threadIdx.x;
}

int32_t bzq_comm_selftest(bzq_ctx* c) {
    if (!c) return BZQ_ERR_ARG;
    bzq_comm* m = c->comm;
    if (!m) return 0;
    HIPCHK(c, hipSetDevice(c->device));
    const int P = m->nranks, me = m->rank;
    constexpr uint64_t OWN = 65536, ROOM = 3 * 40000 + 64;   // "shard" bytes in front, room for the halo behind
    uint8_t* d = nullptr;
    HIPCHK(c, hipMalloc((void**)&d, OWN + ROOM + 64));
    int* d_bad = (int*)(d + OWN + ROOM);
    int rc = 0, lrc = 0;
    std::string lerr;
    int64_t bad_total = 0;
    std::vector<int64_t> all((size_t)P * COMM_ROW);
    auto done = [&](int r) { (void)hipFree(d); return r; };

    // round 0: both forms of the all-gather
    if (m->kind == 1) {
        hipLaunchKernelGGL(k_selftest_row, dim3(1), dim3(64), 0, c->stream, m->d_row, (int64_t)me);
        if ((rc = comm_gather(c, nullptr, all.data(), "the selftest's all-gather (device row)"))) return done(rc);
        for (int r = 0; r < P; ++r)
            for (int w = 0; w < COMM_ROW; ++w)
                if (all[(size_t)r * COMM_ROW + w] != (int64_t)r * 1000 + w) bad_total += 1;
    }
    {
        int64_t row[COMM_ROW];
        for (int w = 0; w < COMM_ROW; ++w) row[w] = (int64_t)me * 77 + w;
        if ((rc = comm_gather(c, row, all.data(), "the selftest's all-gather (host row)"))) return done(rc);
        for (int r = 0; r < P; ++r)
            for (int w = 0; w < COMM_ROW; ++w)
                if (all[(size_t)r * COMM_ROW + w] != (int64_t)r * 77 + w) bad_total += 1;
    }

    // rounds 1-3: comm_exchange_heads with synthetic plans (what bzq_plan_shards would produce for such streams)
    for (int round = 1; round <= 3 && P > 1; ++round) {
        std::vector<bzq_shard_plan> plans((size_t)P);
        for (int r = 0; r < P; ++r) { memset(&plans[(size_t)r], 0, sizeof(bzq_shard_plan)); plans[(size_t)r].head_dst = -1; plans[(size_t)r].halo_first_src = -1; }
        auto send = [&](int from, int to, uint64_t bytes) {
            bzq_shard_plan &p = plans[(size_t)from], &o = plans[(size_t)to];
            p.head_bytes = bytes; p.head_dst = to; p.halo_offset = o.halo_bytes;
            o.halo_bytes += bytes;
            if (o.halo_first_src < 0) o.halo_first_src = from;
            o.halo_n_src = from - o.halo_first_src + 1;
        };
        if (round == 1) { for (int r = 1; r < P; ++r) send(r, r - 1, 40000); }
        else if (round == 2) { for (int r = 1; r < P; r += 2) send(r, r - 1, 313); }
        else { for (int r = 1; r < P && r <= 3; ++r) send(r, 0, 40000 - 1000 * (uint64_t)r); }
        if (m->kind == 2)
            for (int r = 0; r < P; ++r)
                if (plans[(size_t)r].head_bytes > m->halo_cap) { c->err = "bzq_comm_selftest: the communicator's halo capacity is below 40 000 bytes"; return done(BZQ_ERR_ARG); }
        // own bytes: a pattern that names (rank, round); the halo room: cleared
        const int words = (int)(OWN / 4);
        hipLaunchKernelGGL(k_selftest_fill, dim3((words + 255) / 256), dim3(256), 0, c->stream, (uint32_t*)d, (uint32_t)(me * 16 + round), words);
        (void)hipMemsetAsync(d + OWN, 0, ROOM + 64, c->stream);
        // (the exchange takes a shard that IS there, like bzq_shard_stitch's: the shm transport copies it out on the null stream)
        if (hipStreamSynchronize(c->stream) != hipSuccess && !lrc) { lrc = BZQ_ERR_HIP; lerr = "bzq_comm_selftest: filling the pattern"; }
        if ((rc = comm_exchange_heads(c, m, d, OWN, plans, lrc, lerr))) return done(rc);
        const bzq_shard_plan& pl = plans[(size_t)me];
        for (int q = pl.halo_first_src; !lrc && q >= 0 && q < pl.halo_first_src + pl.halo_n_src; ++q) {
            const bzq_shard_plan& src = plans[(size_t)q];
            if (src.head_bytes == 0 || src.head_dst != me) continue;
            // (heads are whole words in rounds 1 and 3; round 2's 313 bytes: the 78 whole words)
            const int nw = (int)(src.head_bytes / 4);
            if (src.halo_offset % 4 == 0)
                hipLaunchKernelGGL(k_selftest_check, dim3((nw + 255) / 256), dim3(256), 0, c->stream, (const uint32_t*)(d + OWN + src.halo_offset), (uint32_t)(q * 16 + round), nw, d_bad);
        }
        int bad = 0;
        if (!lrc && (hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess)) { lrc = BZQ_ERR_HIP; lerr = "bzq_comm_selftest: reading the verdict back"; }
        bad_total += bad;
    }

    // round 4 (RCCL): the ring, P == 1 included (send and receive to itself inside one group)
    if (m->kind == 1 && !lrc) {
        constexpr int WORDS = 1024;
        const int to = (me + 1) % P, from = (me + P - 1) % P;
        uint32_t* w = (uint32_t*)d;
        hipLaunchKernelGGL(k_selftest_fill, dim3(WORDS / 256), dim3(256), 0, c->stream, w, (uint32_t)(me + 1), WORDS);
        (void)hipMemsetAsync(w + WORDS, 0, WORDS * 4, c->stream);
        (void)hipMemsetAsync(d_bad, 0, 4, c->stream);
        const uint64_t want = comm_arrive(m);
        NCCLCHK(c, m, m->p_GroupStart());
        NCCLCHK(c, m, m->p_Send(w, WORDS * 4, NCCL_U8, to, m->nccl, c->stream));
        NCCLCHK(c, m, m->p_Recv(w + WORDS, WORDS * 4, NCCL_U8, from, m->nccl, c->stream));
        NCCLCHK(c, m, m->p_GroupEnd());
        if ((rc = comm_stream_wait(c, m, want, "the selftest's ring exchange"))) return done(rc);
        int bad = 0;
        hipLaunchKernelGGL(k_selftest_check, dim3(WORDS / 256), dim3(256), 0, c->stream, w + WORDS, (uint32_t)(from + 1), WORDS, d_bad);
        if (hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) { lrc = BZQ_ERR_HIP; lerr = "bzq_comm_selftest: reading the verdict back"; }
        bad_total += bad;
    }
    (void)hipFree(d);
    // the verdicts: everybody learns of everybody's (collective failure, like the protocol's)
    int64_t row[COMM_ROW] = {bad_total, lrc, 0, 0, 0, 0, 0, 0};
    if ((rc = comm_gather(c, row, all.data(), "the selftest's all-gather of the verdicts"))) return rc;
    for (int r = 0; r < P; ++r) {
        const int64_t* w = &all[(size_t)r * COMM_ROW];
        if (w[1] != 0) { c->err = r == me ? lerr : "bzq_comm_selftest: rank " + std::to_string(r) + " failed (" + std::to_string(w[1]) + ")"; return r == me ? lrc : BZQ_ERR_IO; }
        if (w[0] != 0) { c->err = "bzq_comm_selftest: rank " + std::to_string(r) + " received " + std::to_string(w[0]) + " wrong words"; return BZQ_ERR_IO; }
    }
    return 0;
}

// ---- the protocol ---------------------------------------------------------------------------------------------------------

// Failures are collective.  Between the first and the last exchange of a call a rank never returns on its own: a failure that
// only this rank sees (a HIP error, a refused allocation, a parse that cannot run) is kept in `lrc`, the rank goes on through
// every exchange of the protocol with empty contributions, and the failure travels in the next gathered row (summary row word 6
// bits 32.., outcome row word 7); every rank then returns together right after that gather.  Over RCCL there is no timeout, so a
// rank that bailed out alone would leave its peers in ncclAllGather / ncclRecv for ever.  What cannot be saved is a failing
// transport itself (comm_gather's own errors).
int32_t bzq_shard_stitch(bzq_ctx* c, uint8_t* d_shard, uint64_t n, uint64_t capacity, bzq_shard_result* out) {
    if (!c || !out) return BZQ_ERR_ARG;
    bzq_comm* m = c->comm;
    const int P = m ? m->nranks : 1, me = m ? m->rank : 0;
    memset(out, 0, sizeof(*out));
    out->first_error_record = -1; out->error_rank = -1;
    int rc;
    int lrc = 0;            // this rank's own failure so far
    std::string lerr;       // ... and its text
    auto note = [&](int r) { if (r < 0 && !lrc) { lrc = r; lerr = c->err; } return r; };
    auto note_hip = [&](hipError_t e, const char* what) { if (e != hipSuccess && !lrc) { lrc = BZQ_ERR_HIP; lerr = std::string(what) + ": " + hipGetErrorString(e); } };
    auto everybody_fails = [&](const std::vector<int64_t>& rows, int word, int shift, const char* where) -> int {   // 0 = nobody
        for (int r = 0; r < P; ++r) {
            const int64_t code = rows[(size_t)r * COMM_ROW + word] >> shift;
            if (code == 0) continue;
            c->tail_mode = 0;
            if (r == me && lrc) { c->err = lerr; return lrc; }
            c->err = std::string("bzq_shard_stitch: rank ") + std::to_string(r) + " failed (" + std::to_string(shift ? -code : code) + ") " + where;
            return BZQ_ERR_IO;
        }
        return 0;
    };
    if ((!d_shard && n) || capacity < n) { c->err = "bzq_shard_stitch: bad shard buffer"; note(BZQ_ERR_ARG); }
    else if (((uintptr_t)d_shard & 15u) != 0) { c->err = "device shard must be 16-byte aligned"; note(BZQ_ERR_ARG); }
    else if (c->cfg.views_only) { c->err = "bzq_shard_stitch: views mode is a single-chunk mode (shards deliver batch columns)"; note(BZQ_ERR_ARG); }
    if (!lrc) note_hip(hipSetDevice(c->device), "hipSetDevice");

    // 1. scan + summary all-gather (RCCL: the row is packed on the device, one synchronisation for both)
    std::vector<bzq_shard_summary> sums((size_t)P);
    std::vector<int64_t> all((size_t)P * COMM_ROW);
    if (!lrc) note(shard_scan_enqueue(c, d_shard, n));
    if (!lrc && m && m->kind == 1 && n > 0) {
        hipLaunchKernelGGL(k_pack_summary, dim3(1), dim3(1), 0, c->stream, (const ChunkState*)c->d_state, (int64_t)n, (int64_t)(capacity - n), m->d_row, (int64_t)(me + 1));
        if ((rc = comm_gather(c, nullptr, all.data(), "the all-gather of the shard summaries"))) return rc;
        shard_scan_finish(c, d_shard, n, &sums[(size_t)me]);
    } else {
        int64_t row[COMM_ROW] = {0, 0, -1, -1, -1, -1, 0, 0};
        if (!lrc) note_hip(hipStreamSynchronize(c->stream), "hipStreamSynchronize(scan)");
        if (!lrc) {
            shard_scan_finish(c, d_shard, n, &sums[(size_t)me]);
            const bzq_shard_summary& s = sums[(size_t)me];
            const int64_t r2[COMM_ROW] = {(int64_t)s.n_bytes, (int64_t)s.n_newlines, s.first_nl[0], s.first_nl[1], s.first_nl[2], s.first_nl[3],
                                          (int64_t)s.first_byte | ((int64_t)s.last_byte << 8) | ((int64_t)(me + 1) << 16), (int64_t)(capacity - n)};
            memcpy(row, r2, sizeof(row));
        } else {
            row[6] = ((int64_t)(uint32_t)(-lrc) << 32) | ((int64_t)(me + 1) << 16);   // an empty shard that says why
        }
        if ((rc = comm_gather(c, row, all.data(), "the all-gather of the shard summaries"))) return rc;
    }
    if ((rc = everybody_fails(all, 6, 32, "before the shards were exchanged"))) return rc;
    uint64_t stream_pos = 0, total_bytes = 0;
    c->ranks_seen = 0;   // rows of the all-gather that carry their rank's stamp (bits 16..31 of word 6): what the transport really delivered
    for (int r = 0; r < P; ++r) {
        const int64_t* w = &all[(size_t)r * COMM_ROW];
        if (((w[6] >> 16) & 0xFFFF) == r + 1) c->ranks_seen += 1;
        bzq_shard_summary& s = sums[(size_t)r];
        s.n_bytes = (uint64_t)w[0]; s.n_newlines = (uint64_t)w[1];
        for (int i = 0; i < 4; ++i) s.first_nl[i] = w[2 + i];
        s.first_byte = (uint8_t)(w[6] & 0xFF); s.last_byte = (uint8_t)((w[6] >> 8) & 0xFF);
        if (r < me) stream_pos += s.n_bytes;
        total_bytes += s.n_bytes;
    }

    // 2. plan: a pure function of the gathered rows -- whatever it refuses, it refuses on every rank
    std::vector<bzq_shard_plan> plans((size_t)P);
    if ((rc = bzq_plan_shards(sums.data(), P, plans.data()))) { c->err = "bzq_shard_stitch: inconsistent shard summaries"; return rc; }
    const bzq_shard_plan pl = plans[(size_t)me];
    out->plan = pl;
    // every rank knows every rank's room (row[7]): a halo that does not fit fails the call on ALL ranks, before anything is
    // exchanged
    for (int r = 0; r < P; ++r)
        if ((int64_t)plans[(size_t)r].halo_bytes > all[(size_t)r * COMM_ROW + 7]) {
            c->err = "bzq_shard_stitch: rank " + std::to_string(r) + "'s shard buffer has no room for its halo (" + std::to_string(plans[(size_t)r].halo_bytes) +
                     " bytes behind " + std::to_string(sums[(size_t)r].n_bytes) + ", room for " + std::to_string(all[(size_t)r * COMM_ROW + 7]) + ")";
            return BZQ_ERR_ARG;
        }
    if (m && m->kind == 2 && P > 1)
        for (int r = 0; r < P; ++r)   // (the capacity is the same on every rank: all of them fail together)
            if (plans[(size_t)r].head_bytes > m->halo_cap) { c->err = "bzq_shard_stitch(shm): rank " + std::to_string(r) + "'s head of " + std::to_string(plans[(size_t)r].head_bytes) + " bytes exceeds the halo capacity the communicator was created with"; return BZQ_ERR_ARG; }

    // 3. heads travel to their owners
    if ((rc = comm_exchange_heads(c, m, d_shard, n, plans, lrc, lerr))) return rc;

    // 4. parse own bytes + halo (a rank whose whole shard is the middle of somebody else's record delivers nothing)
    const bool owner = n > 0 && pl.head_bytes < n;
    bzq_chunk res{};
    res.error_record = -1;
    c->tail_mode = 0;
    if (owner && !lrc) {
        c->tail_mode = pl.is_last ? 1 : 0;
        int prc = bzq_submit_shard(c, d_shard, n, pl.halo_bytes, pl.lines_before, pl.prev_last_byte, stream_pos, pl.is_last);
        if (prc >= 0) prc = bzq_chunk_result(c, &res);
        if (prc < 0) { note(prc); c->tail_mode = 0; res = bzq_chunk{}; res.error_record = -1; }
    } else if (pl.is_last && !lrc) {
        res.status = BZQ_EOF;   // an empty stream
    }
    auto install_empty = [&]() {   // a rank that delivers nothing must not hand out the batches of an earlier step
        res = bzq_chunk{}; res.error_record = -1; res.status = BZQ_EOF;
        c->res = res; c->have_result = true; c->pending = false;
    };
    if (!owner && !lrc) { const int32_t st = res.status; install_empty(); res.status = st; c->res.status = st; }

    // 5. outcomes
    auto gather_outcomes = [&](std::vector<int64_t>& rows) {
        const bool failed = res.status > 0 && res.status != BZQ_EOF;
        int64_t row[COMM_ROW] = {(int64_t)res.n_records, (int64_t)res.seq_bytes, (int64_t)n, failed ? res.error_record : -1, res.status,
                                 owner && c->tail_pending && !lrc ? 1 : 0, owner ? 1 : 0, lrc};
        rows.assign((size_t)P * COMM_ROW, 0);
        return comm_gather(c, row, rows.data(), "the all-gather of the outcomes");
    };
    std::vector<int64_t> oc;
    if ((rc = gather_outcomes(oc))) { c->tail_mode = 0; return rc; }
    if ((rc = everybody_fails(oc, 7, 0, "while parsing its shard"))) return rc;
    bool walk = false;
    for (int r = 0; r < P; ++r) {
        const int64_t* w = &oc[(size_t)r * COMM_ROW];
        if (w[3] >= 0) { walk = false; break; }   // an earlier failing record ends the stream before its tail is looked at
        if (w[5]) walk = true;
    }

    // 6. cold path: the reference's window, walked through every rank's record ends in rank order
    if (walk) {
        int64_t st_row[COMM_ROW] = {0, 0, 0, 0, 0, 0, 0, 0};   // w, end, cap, eof, head, valid
        std::vector<int64_t> st_all((size_t)P * COMM_ROW);
        Window s;
        int64_t head = 0;
        bool have = false;
        for (int round = 0; round < P; ++round) {
            if (round == me && owner && !lrc) {
                if (!have) { window_start(s, c->cfg, (int64_t)total_bytes); head = (int64_t)stream_pos + c->cur_first_header; }
                const int64_t nrec = c->h_state->P > 0 ? (c->h_state->P >> 2) : 0;   // complete records of this rank
                std::vector<int64_t> re((size_t)std::min<int64_t>(nrec, (int64_t)res.n_records));   // the records this rank delivered
                if (!re.empty()) note_hip(hipMemcpy(re.data(), c->o().rec_end.p, re.size() * 8, hipMemcpyDeviceToHost), "hipMemcpy(record ends)");
                if (!lrc) {
                    window_walk(s, head, re.data(), re.size(), (int64_t)stream_pos, c->cfg);
                    if (c->tail_pending) {
                        bool acc = false; int ph = 0; int64_t cap = c->cfg.buffer_capacity;
                        const int code = window_classify(s, head, c->cfg, res.tail_phase, c->h_state->tail_nonblank != 0, &acc, &ph, &cap);
                        c->tail_mode = 2; c->tail_code = code; c->tail_accept = acc; c->tail_phase_dec = ph; c->tail_cap_dec = cap;
                    }
                    st_row[0] = s.w; st_row[1] = s.end; st_row[2] = s.cap; st_row[3] = s.eof ? 1 : 0; st_row[4] = head; st_row[5] = 1;
                }
            }
            if ((rc = comm_gather(c, st_row, st_all.data(), "the walk of the reader's window through the ranks"))) { c->tail_mode = 0; return rc; }
            const int64_t* w = &st_all[(size_t)round * COMM_ROW];
            if (w[5] && round != me) { s = Window(); s.N = (int64_t)total_bytes; s.w = w[0]; s.end = w[1]; s.cap = w[2]; s.eof = w[3] != 0; head = w[4]; have = true; }
        }
        if (owner && c->tail_mode == 2 && !lrc) {   // the last owner: the same chunk again, now with the decision
            c->pending = true; c->have_result = false;
            if (note(bzq_chunk_result(c, &res)) < 0) { res = bzq_chunk{}; res.error_record = -1; }
        }
        if ((rc = gather_outcomes(oc))) { c->tail_mode = 0; return rc; }
        if ((rc = everybody_fails(oc, 7, 0, "while walking the reader's window over its records"))) return rc;
    }
    c->tail_mode = 0;

    // 7. everything global, derived identically on every rank.  The sequential parser stops at the first failing record: the
    // sums end there (global_records == first_error_record on a failing stream) and the ranks behind it deliver nothing.
    uint64_t acc_rec = 0;
    int last_owner = -1;
    for (int r = 0; r < P; ++r) {
        const int64_t* w = &oc[(size_t)r * COMM_ROW];
        out->global_bytes += (uint64_t)w[2];
        if (out->first_error_record >= 0) continue;
        if (r == me) out->records_before = acc_rec;
        if (w[3] >= 0) { out->first_error_record = (int64_t)acc_rec + w[3]; out->error_rank = r; out->stream_status = (int32_t)w[4]; }
        acc_rec += (uint64_t)w[0];
        out->global_records += (uint64_t)w[0]; out->global_bases += (uint64_t)w[1];
        if (w[6]) last_owner = r;
    }
    if (out->first_error_record < 0) out->stream_status = last_owner >= 0 ? (int32_t)oc[(size_t)last_owner * COMM_ROW + 4] : BZQ_EOF;
    if (out->error_rank >= 0 && me > out->error_rank) {   // behind the failing record: the sequential parser never got here
        out->records_before = out->global_records;
        install_empty();
    }
    out->chunk = res;
    out->stream_pos = stream_pos;
    c->shard_totals[0] = out->global_records; c->shard_totals[1] = out->global_bases; c->shard_totals[2] = out->global_bytes;
    c->have_shard_totals = true;
    return 0;
}

int32_t bzq_global_counts(bzq_ctx* c, uint64_t out[3]) {
    if (!c || !out) return BZQ_ERR_ARG;
    if (!c->have_shard_totals) { c->err = "bzq_global_counts: no bzq_shard_stitch has completed on this ctx"; return BZQ_ERR_ARG; }
    for (int i = 0; i < 3; ++i) out[i] = c->shard_totals[i];
    return 0;
}
