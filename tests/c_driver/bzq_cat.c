/* A plain-C consumer of the drop-in boundary (include/blazeseq_hip.h): what a foreign host (the Mojo shim of
 * INTEGRATION.md, or any FFI) does with it -- no Python, no torch.
 *
 *   bzq_cat FILE [batch_size] [chunk_bytes] [check]   ->  one line per record on stdout:  id \t sequence \t quality
 *                                                         then "# records=N batches=B status=S" and, on a failing
 *                                                         stream, the reference's error text after "# error: "
 *
 * Compiled as C (gcc -std=c11) on purpose: the header must be usable from C. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "blazeseq_hip.h"

static void die(bzq_ctx* ctx, const char* what, int rc) {
    fprintf(stderr, "%s failed (%d): %s\n", what, rc, bzq_last_error(ctx));
    exit(2);
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: bzq_cat FILE [batch_size] [chunk_bytes] [check]\n"); return 2; }
    const uint32_t batch = argc > 2 ? (uint32_t)atoi(argv[2]) : 4096;
    const uint64_t chunk_bytes = argc > 3 ? (uint64_t)atoll(argv[3]) : 0;
    const int check = argc > 4 ? atoi(argv[4]) : 0;

    bzq_config cfg;
    bzq_config_default(&cfg);
    cfg.batch_size = (int32_t)batch;
    cfg.check_ascii = check;
    cfg.check_quality = check;
    bzq_ctx* ctx = NULL;
    int rc = bzq_create(0, &cfg, &ctx);
    if (rc) { fprintf(stderr, "bzq_create failed (%d): %s\n", rc, bzq_last_error(NULL)); return 3; }

    bzq_ingest* in = NULL;
    if ((rc = bzq_ingest_open(ctx, argv[1], chunk_bytes, 4, &in)) < 0) die(ctx, "bzq_ingest_open", rc);

    uint64_t total = 0, batches = 0, taken = 0, before = 0;
    int status = BZQ_OK;
    size_t cap_b = 0, cap_r = 0;
    uint8_t *q = NULL, *s = NULL, *id = NULL;
    int64_t *ends = NULL, *id_ends = NULL;
    for (;;) {
        bzq_chunk ch;
        rc = bzq_ingest_next(in, taken, &ch, NULL);
        if (rc < 0) die(ctx, "bzq_ingest_next", rc);
        status = ch.status;
        /* whole batches only while more input follows (the remainder is carried into the next chunk) */
        uint64_t usable = ch.n_records;
        if (status == BZQ_OK) usable -= usable % batch;
        for (uint64_t first = 0; first < usable; first += batch) {
            bzq_device_batch db;
            const uint32_t want = (uint32_t)((usable - first) < batch ? (usable - first) : batch);
            if ((rc = bzq_batch_view(ctx, first, want, &db)) < 0) die(ctx, "bzq_batch_view", rc);
            const int64_t need = db.sequence_bytes > db.seq_len ? db.sequence_bytes : db.seq_len;
            if ((size_t)need + 1 > cap_b || (size_t)db.total_id_bytes + 1 > cap_b) {
                cap_b = (size_t)(need > db.total_id_bytes ? need : db.total_id_bytes) * 2 + 64;
                q = realloc(q, cap_b); s = realloc(s, cap_b); id = realloc(id, cap_b);
            }
            if ((size_t)db.num_records > cap_r) {
                cap_r = (size_t)db.num_records * 2;
                ends = realloc(ends, cap_r * 8); id_ends = realloc(id_ends, cap_r * 8);
            }
            bzq_host_batch hb;
            memset(&hb, 0, sizeof hb);
            hb.quality_bytes = q; hb.sequence_bytes = s; hb.id_bytes = id; hb.ends = ends; hb.id_ends = id_ends;
            if ((rc = bzq_batch_to_host(ctx, &db, &hb)) < 0) die(ctx, "bzq_batch_to_host", rc);
            for (int64_t r = 0; r < db.num_records; ++r) {   /* FastqBatch.get_record, record_batch.mojo:116-150 */
                const int64_t e0 = r ? ends[r - 1] : 0, i0 = r ? id_ends[r - 1] : 0;
                fwrite(id + i0, 1, (size_t)(id_ends[r] - i0), stdout); fputc('\t', stdout);
                fwrite(s + e0, 1, (size_t)(ends[r] - e0), stdout); fputc('\t', stdout);
                fwrite(q + e0, 1, (size_t)(ends[r] - e0), stdout); fputc('\n', stdout);
            }
            total += (uint64_t)db.num_records;
            ++batches;
        }
        taken = usable;
        if (status != BZQ_OK) break;
        before += usable;
    }
    printf("# records=%llu batches=%llu status=%d\n", (unsigned long long)total, (unsigned long long)batches, status);
    if (status != BZQ_EOF) {
        char msg[4096];
        const int64_t n = bzq_format_error(ctx, before, msg, sizeof msg);
        if (n > 0) { printf("# error: "); fwrite(msg, 1, (size_t)(n < 4095 ? n : 4095), stdout); printf("\n"); }
    }
    bzq_ingest_close(in);
    bzq_destroy(ctx);
    free(q); free(s); free(id); free(ends); free(id_ends);
    return 0;
}
