/* The reference's throughput runner (benchmark/throughput/run_throughput_blazeseq.mojo:28-55, mode "batches") over the drop-in
 * boundary, as a process of its own: exec -> bzq_create -> bzq_ingest_open -> every chunk, every batch of 4096 handed out ->
 * "<records> <base_pairs>" on stdout -> exit.  What `hyperfine` times in benchmark/throughput/run_throughput_benchmarks.sh:54-62 is
 * a whole process like this one; bench.py's process_mode does the same with it (fresh process per run, 3 warm-up + 15 runs).
 *
 *   bzq_throughput FILE [batches] [chunk_mib] [threads]
 *
 * BZQ_THROUGHPUT_FAST_EXIT=1: _exit(0) right behind the result line (see below).
 * BZQ_THROUGHPUT_TIMES=1: the phases' wall clock on stderr (library load is before main: the caller's clock has it).
 * Plain C (gcc -std=c11): no Python, no torch. */
#define _POSIX_C_SOURCE 200809L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include "blazeseq_hip.h"

static double now_ms(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec * 1e3 + (double)ts.tv_nsec * 1e-6;
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: bzq_throughput FILE [batches] [chunk_mib] [threads]\n"); return 2; }
    if (argc > 2 && strcmp(argv[2], "batches")) { fprintf(stderr, "only the runner's `batches` mode is a process of its own here\n"); return 2; }
    const uint64_t chunk_bytes = argc > 3 ? (uint64_t)atoll(argv[3]) << 20 : 0;
    const int threads = argc > 4 ? atoi(argv[4]) : 8;
    const uint32_t batch = 4096;
    const double t0 = now_ms();

    bzq_config cfg;
    bzq_config_default(&cfg);
    cfg.batch_size = (int32_t)batch;
    cfg.buffer_capacity = 64 * 1024;      /* the runner's ParserConfig: 64 KiB buffer, growth off, validation off */
    cfg.buffer_growth_enabled = 0;
    bzq_ctx* ctx = NULL;
    int rc = bzq_create(0, &cfg, &ctx);
    if (rc) { fprintf(stderr, "bzq_create failed (%d): %s\n", rc, bzq_last_error(NULL)); return 3; }
    const double t1 = now_ms();

    bzq_ingest* in = NULL;
    if ((rc = bzq_ingest_open(ctx, argv[1], chunk_bytes, threads, &in)) < 0) { fprintf(stderr, "bzq_ingest_open failed (%d): %s\n", rc, bzq_last_error(ctx)); return 3; }
    const double t2 = now_ms();

    unsigned long long total_reads = 0, total_base_pairs = 0;
    uint64_t taken = 0, cap = 0;
    bzq_device_batch* arr = NULL;
    double t_first = 0;
    int status = BZQ_OK, chunks = 0;
    for (;;) {
        bzq_chunk ch;
        rc = bzq_ingest_next(in, taken, &ch, NULL);
        if (rc < 0) { fprintf(stderr, "bzq_ingest_next failed (%d): %s\n", rc, bzq_last_error(ctx)); return 3; }
        if (!chunks++) t_first = now_ms();
        status = ch.status;
        /* whole batches only while more input follows (the remainder is carried into the next chunk) */
        uint64_t usable = ch.n_records;
        if (status == BZQ_OK) usable -= usable % batch;
        {
            const uint64_t nb = (usable + batch - 1) / batch;
            if (nb > cap) { cap = nb * 2 + 16; arr = realloc(arr, cap * sizeof *arr); if (!arr) return 4; }
            uint64_t n_out = 0;
            if (nb && (rc = bzq_batches(ctx, batch, arr, nb, &n_out)) < 0) { fprintf(stderr, "bzq_batches: %s\n", bzq_last_error(ctx)); return 3; }
            for (uint64_t b = 0; b < nb; ++b) {   /* `for batch in parser.batches(4096)` */
                total_reads += (unsigned long long)arr[b].num_records;
                total_base_pairs += (unsigned long long)arr[b].seq_len;
            }
        }
        taken = usable;
        if (status != BZQ_OK) break;
    }
    const double t3 = now_ms();
    if (status != BZQ_EOF) {
        char msg[4096];
        const int64_t n = bzq_format_error(ctx, 0, msg, sizeof msg);
        fprintf(stderr, "stream ended with status %d: %.*s\n", status, (int)(n > 0 ? (n < 4095 ? n : 4095) : 0), msg);
    }
    printf("%llu %llu\n", total_reads, total_base_pairs);
    fflush(stdout);
    /* BZQ_THROUGHPUT_FAST_EXIT=1: the answer is out, leave at once -- no close, no destroy, no teardown of the HIP runtime (the kernel
     * reclaims everything); what many command line tools do, and ~50-70 ms of a 350 ms process here.  Off by default: the figure
     * bench.py reports as `value` is the process that cleans up after itself, the fast exit stands beside it. */
    if (status == BZQ_EOF && getenv("BZQ_THROUGHPUT_FAST_EXIT")) _exit(0);
    bzq_ingest_close(in);
    bzq_destroy(ctx);
    free(arr);
    const double t4 = now_ms();
    if (getenv("BZQ_THROUGHPUT_TIMES"))
        fprintf(stderr, "bzq_throughput: create %.1f ms, open %.1f ms, first chunk %.1f ms, remaining %d chunks %.1f ms, close+destroy %.1f ms, main total %.1f ms\n",
                t1 - t0, t2 - t1, t_first - t2, chunks - 1, t3 - t_first, t4 - t3, t4 - t0);
    return status == BZQ_EOF ? 0 : 1;
}
