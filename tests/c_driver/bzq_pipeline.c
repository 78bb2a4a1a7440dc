/* The reference's GPU example (examples/nw_gpu/execution.mojo:100-130: next_batch(65536) -> batch.to_device(ctx) -> nw_kernel ->
 * ctx.synchronize()) over the drop-in boundary from a plain-C host with no HIP binding of its own: file -> bzq_ingest_next -> every
 * batch of 65 536 records (zero-copy views of the chunk's device columns) -> bzq_batch_nw_scores_dev against the example's 40 bp
 * reference + bzq_batch_quality_by_position_acc into one device table; nothing leaves the device until the end.  The consumers run on
 * the ctx stream here (no consumer stream set), behind each chunk's parse and in front of the next one's: stream order is the
 * two-chunk lifetime rule.
 *
 *   bzq_pipeline FILE [max_positions]
 *
 * stdout: "<records> <sum of all scores> <sum over the table> <FNV-1a of the table>".  Plain C (gcc -std=c11). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "blazeseq_hip.h"

#define PIPE_BATCH 65536u
static const char REF_40BP[] = "ACGTACGTACGTACGTACGTACGTACGTACGTACGTACGT";   /* examples/nw_gpu/execution.mojo:36 */

#define CHECK(call) do { int32_t rc_ = (call); if (rc_ < 0) { fprintf(stderr, "%s failed (%d): %s\n", #call, (int)rc_, bzq_last_error(ctx)); return 3; } } while (0)

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: bzq_pipeline FILE [max_positions]\n"); return 2; }
    const int32_t max_pos = argc > 2 ? atoi(argv[2]) : 150;
    const int32_t ref_len = (int32_t)strlen(REF_40BP);
    bzq_config cfg;
    bzq_config_default(&cfg);
    cfg.batch_size = (int32_t)PIPE_BATCH;
    bzq_ctx* ctx = NULL;
    int rc = bzq_create(0, &cfg, &ctx);
    if (rc) { fprintf(stderr, "bzq_create failed (%d): %s\n", rc, bzq_last_error(NULL)); return 3; }
    bzq_ingest* in = NULL;
    CHECK(bzq_ingest_open(ctx, argv[1], 8u << 20, 4, &in));

    void *d_ref = NULL, *d_counts = NULL, *d_scores = NULL;
    const size_t table_bytes = (size_t)max_pos * 128 * 8;
    uint64_t* table = calloc(1, table_bytes);
    uint64_t scores_cap = 0;
    if (!table) return 4;
    CHECK(bzq_device_alloc(ctx, 256, &d_ref));
    CHECK(bzq_copy_to_device(ctx, d_ref, REF_40BP, (size_t)ref_len));
    CHECK(bzq_device_alloc(ctx, table_bytes, &d_counts));
    CHECK(bzq_copy_to_device(ctx, d_counts, table, table_bytes));   /* zeroes */

    unsigned long long records = 0;
    long long score_sum = 0;
    uint64_t taken = 0, cap = 0;
    bzq_device_batch* arr = NULL;
    int32_t* h_scores = NULL;
    int status = BZQ_OK;
    for (;;) {
        bzq_chunk ch;
        CHECK(bzq_ingest_next(in, taken, &ch, NULL));
        status = ch.status;
        const uint64_t n = ch.n_records, nb = (n + PIPE_BATCH - 1) / PIPE_BATCH;
        if (nb > cap) { cap = nb * 2 + 4; arr = realloc(arr, cap * sizeof *arr); if (!arr) return 4; }
        if (n > scores_cap) {   /* (the scores of one chunk; grown when a chunk has more records than any before it) */
            if (d_scores) { CHECK(bzq_consumer_synchronize(ctx)); CHECK(bzq_device_free(ctx, d_scores)); }
            scores_cap = n + n / 4 + 1024;
            CHECK(bzq_device_alloc(ctx, scores_cap * 4, &d_scores));
            h_scores = realloc(h_scores, scores_cap * 4);
            if (!h_scores) return 4;
        }
        uint64_t n_out = 0;
        if (nb) CHECK(bzq_batches(ctx, PIPE_BATCH, arr, nb, &n_out));
        for (uint64_t b = 0; b < nb; ++b) {   /* `for batch in batches: d = batch.to_device(ctx); nw_kernel(d); quality_distribution(d)` */
            CHECK(bzq_batch_nw_scores_dev(ctx, &arr[b], (const uint8_t*)d_ref, ref_len, (int32_t*)d_scores + b * PIPE_BATCH));
            CHECK(bzq_batch_quality_by_position_acc(ctx, &arr[b], max_pos, (uint64_t*)d_counts));
        }
        if (n) {   /* this driver prints the SUM of the scores: they come back per chunk (a real consumer would keep them on the device) */
            CHECK(bzq_consumer_synchronize(ctx));
            CHECK(bzq_copy_to_host(ctx, h_scores, d_scores, (size_t)n * 4));
            for (uint64_t r = 0; r < n; ++r) score_sum += h_scores[r];
        }
        records += n;
        taken = n;
        if (status != BZQ_OK) break;
    }
    CHECK(bzq_consumer_synchronize(ctx));
    CHECK(bzq_copy_to_host(ctx, table, d_counts, table_bytes));
    unsigned long long total = 0, fnv = 1469598103934665603ull;
    for (size_t i = 0; i < (size_t)max_pos * 128; ++i) {
        total += table[i];
        for (int k = 0; k < 8; ++k) { fnv ^= (table[i] >> (8 * k)) & 0xFFu; fnv *= 1099511628211ull; }
    }
    if (status != BZQ_EOF) {
        char msg[4096];
        const int64_t m = bzq_format_error(ctx, 0, msg, sizeof msg);
        fprintf(stderr, "stream ended with status %d: %.*s\n", status, (int)(m > 0 ? (m < 4095 ? m : 4095) : 0), msg);
    }
    printf("%llu %lld %llu %llu\n", records, score_sum, total, fnv);
    bzq_ingest_close(in);
    (void)bzq_device_free(ctx, d_ref); (void)bzq_device_free(ctx, d_counts); (void)bzq_device_free(ctx, d_scores);
    bzq_destroy(ctx);
    free(arr); free(table); free(h_scores);
    return status == BZQ_EOF ? 0 : 1;
}
