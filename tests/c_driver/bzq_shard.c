/* One rank of the multi-GPU shard protocol from plain C (include/blazeseq_hip.h: bzq_comm_init / bzq_comm_init_shm /
 * bzq_shard_stitch / bzq_global_counts) -- no Python and no torch in the process; start one process per rank.
 *
 *   bzq_shard shm  RANK NRANKS NAME   FILE [device] [check] [buffer_capacity]
 *   bzq_shard rccl RANK NRANKS IDFILE FILE [device] [check] [buffer_capacity]     (rank 0 writes the ncclUniqueId to IDFILE)
 *
 * The rank takes bytes [size*RANK/NRANKS, size*(RANK+1)/NRANKS) of FILE (not record aligned), runs the protocol and prints
 * its records "id \t sequence \t quality", then
 *   # rank=R records=N before=B global_records=G global_bases=S global_bytes=Y status=T stream_status=U first_error=E error_rank=K
 * and, on the rank that holds the stream's terminal error, "# error: " + the reference's text. */
#define _POSIX_C_SOURCE 200809L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "blazeseq_hip.h"

static void die(bzq_ctx* ctx, const char* what, int rc) {
    fprintf(stderr, "%s failed (%d): %s\n", what, rc, bzq_last_error(ctx));
    exit(2);
}

int main(int argc, char** argv) {
    if (argc < 6) { fprintf(stderr, "usage: bzq_shard shm|rccl RANK NRANKS NAME|IDFILE FILE [device] [check] [buffer_capacity]\n"); return 2; }
    const int rccl = strcmp(argv[1], "rccl") == 0;
    const int rank = atoi(argv[2]), nranks = atoi(argv[3]);
    const int device = argc > 6 ? atoi(argv[6]) : 0;
    const int check = argc > 7 ? atoi(argv[7]) : 0;
    const long long bufcap = argc > 8 ? atoll(argv[8]) : 0;

    FILE* f = fopen(argv[5], "rb");
    if (!f) { perror(argv[5]); return 2; }
    fseek(f, 0, SEEK_END);
    const uint64_t size = (uint64_t)ftell(f);
    const uint64_t lo = size * (uint64_t)rank / (uint64_t)nranks, hi = size * (uint64_t)(rank + 1) / (uint64_t)nranks, n = hi - lo;
    uint8_t* host = malloc(n ? n : 1);
    fseek(f, (long)lo, SEEK_SET);
    if (n && fread(host, 1, n, f) != n) { fprintf(stderr, "short read\n"); return 2; }
    fclose(f);

    bzq_config cfg;
    bzq_config_default(&cfg);
    cfg.check_ascii = check; cfg.check_quality = check;
    if (bufcap > 0) cfg.buffer_capacity = bufcap;
    bzq_ctx* ctx = NULL;
    int rc = bzq_create(device, &cfg, &ctx);
    if (rc) { fprintf(stderr, "bzq_create failed (%d): %s\n", rc, bzq_last_error(NULL)); return 3; }

    if (rccl) {
        bzq_nccl_id id;
        if (rank == 0) {
            if ((rc = bzq_comm_get_unique_id(&id)) != 0) die(ctx, "bzq_comm_get_unique_id", rc);
            char tmp[4096];
            snprintf(tmp, sizeof tmp, "%s.tmp", argv[4]);
            FILE* o = fopen(tmp, "wb");
            if (!o || fwrite(&id, 1, sizeof id, o) != sizeof id) { perror("id file"); return 2; }
            fclose(o);
            rename(tmp, argv[4]);
        } else {
            for (int tries = 0;; ++tries) {
                FILE* i = fopen(argv[4], "rb");
                if (i && fread(&id, 1, sizeof id, i) == sizeof id) { fclose(i); break; }
                if (i) fclose(i);
                if (tries > 60000) { fprintf(stderr, "no id file\n"); return 2; }
                struct timespec ts = {0, 1000000};
                nanosleep(&ts, NULL);
            }
        }
        if ((rc = bzq_comm_init(ctx, rank, nranks, &id)) != 0) die(ctx, "bzq_comm_init", rc);
    } else {
        if ((rc = bzq_comm_init_shm(ctx, rank, nranks, argv[4], 0)) != 0) die(ctx, "bzq_comm_init_shm", rc);
    }

    /* BZQ_SHARD_TIMEOUT_MS: the deadline of every exchange; BZQ_SHARD_INJECT_STALL="R:SECONDS[:early]": rank R sleeps before the
     * stitch (or, with :early, before the selftest) -- its peers must fail within the deadline, naming it */
    if (getenv("BZQ_SHARD_TIMEOUT_MS") && (rc = bzq_set_option(ctx, "comm_timeout_ms", atoll(getenv("BZQ_SHARD_TIMEOUT_MS")))) != 0) die(ctx, "bzq_set_option", rc);
    const char* stall = getenv("BZQ_SHARD_INJECT_STALL");
    int stall_rank = -1, stall_early = 0;
    double stall_s = 0;
    if (stall) { stall_rank = atoi(stall); const char* q = strchr(stall, ':'); if (q) { stall_s = atof(q + 1); stall_early = strstr(q + 1, ":early") != NULL; } }
    if (stall_rank == rank && stall_early) { struct timespec ts = {(time_t)stall_s, (long)((stall_s - (time_t)stall_s) * 1e9)}; nanosleep(&ts, NULL); }
    if ((rc = bzq_comm_selftest(ctx)) != 0) die(ctx, "bzq_comm_selftest", rc);
    if (stall_rank == rank && !stall_early) { struct timespec ts = {(time_t)stall_s, (long)((stall_s - (time_t)stall_s) * 1e9)}; nanosleep(&ts, NULL); }

    uint64_t capacity = n + (4u << 20);   /* room for the halo */
    void* d_shard = NULL;
    const char* lib_read = getenv("BZQ_SHARD_LIB_READ");   /* N > 0: the LIBRARY reads the rank's byte range of the file (N reader threads) */
    if (lib_read && atoi(lib_read) > 0) {
        uint8_t* p = NULL;
        uint64_t got = 0;
        if ((rc = bzq_shard_read_range(ctx, argv[5], lo, hi, 4u << 20, atoi(lib_read), &p, &got, &capacity)) != 0) die(ctx, "bzq_shard_read_range", rc);
        if (got != n) { fprintf(stderr, "bzq_shard_read_range: %llu bytes, expected %llu\n", (unsigned long long)got, (unsigned long long)n); return 2; }
        d_shard = p;
    } else {
        if ((rc = bzq_device_alloc(ctx, capacity, &d_shard)) != 0) die(ctx, "bzq_device_alloc", rc);
        if ((rc = bzq_copy_to_device(ctx, d_shard, host, n)) != 0) die(ctx, "bzq_copy_to_device", rc);
    }

    bzq_shard_result res;
    /* BZQ_SHARD_INJECT_MISALIGN=R: rank R hands in a pointer the library refuses -- a failure only that rank sees; the call
     * must fail on EVERY rank (nobody is left waiting in an exchange) */
    const char* inj = getenv("BZQ_SHARD_INJECT_MISALIGN");
    uint8_t* shard_ptr = (uint8_t*)d_shard + ((inj && atoi(inj) == rank) ? 1 : 0);
    if ((rc = bzq_shard_stitch(ctx, shard_ptr, n, capacity, &res)) != 0) die(ctx, "bzq_shard_stitch", rc);

    /* this rank's records, the reference's FastqBatch.get_record walk over the chunk columns (record_batch.mojo:116-150) */
    bzq_chunk* ch = &res.chunk;
    uint8_t *q = malloc(ch->qual_bytes + 1), *s = malloc(ch->seq_bytes + 1), *id = malloc(ch->id_bytes + 1);
    int64_t *ends = malloc(ch->n_records * 8 + 8), *id_ends = malloc(ch->n_records * 8 + 8);
    if ((rc = bzq_chunk_cumulative_ends(ctx, ch)) < 0) { fprintf(stderr, "bzq_chunk_cumulative_ends: %d\n", rc); return 1; }   /* (ABI 2: produced on demand) */
    if (ch->n_records) {
        if ((rc = bzq_copy_to_host(ctx, q, ch->d_qual, ch->qual_bytes)) || (rc = bzq_copy_to_host(ctx, s, ch->d_seq, ch->seq_bytes)) ||
            (rc = bzq_copy_to_host(ctx, id, ch->d_id, ch->id_bytes)) || (rc = bzq_copy_to_host(ctx, ends, ch->d_ends, ch->n_records * 8)) ||
            (rc = bzq_copy_to_host(ctx, id_ends, ch->d_id_ends, ch->n_records * 8)))
            die(ctx, "bzq_copy_to_host", rc);
    }
    for (uint64_t r = 0; r < ch->n_records; ++r) {
        const int64_t e0 = r ? ends[r - 1] : 0, i0 = r ? id_ends[r - 1] : 0;
        /* sequence bytes run in step with quality bytes except through an accepted unterminated last record */
        const int64_t s1 = (r + 1 == ch->n_records) ? (int64_t)ch->seq_bytes : ends[r];
        fwrite(id + i0, 1, (size_t)(id_ends[r] - i0), stdout); fputc('\t', stdout);
        fwrite(s + e0, 1, (size_t)(s1 - e0), stdout); fputc('\t', stdout);
        fwrite(q + e0, 1, (size_t)(ends[r] - e0), stdout); fputc('\n', stdout);
    }
    uint64_t g[3];
    if ((rc = bzq_global_counts(ctx, g)) != 0) die(ctx, "bzq_global_counts", rc);
    printf("# rank=%d records=%llu before=%llu global_records=%llu global_bases=%llu global_bytes=%llu status=%d stream_status=%d first_error=%lld error_rank=%d\n",
           rank, (unsigned long long)ch->n_records, (unsigned long long)res.records_before, (unsigned long long)g[0],
           (unsigned long long)g[1], (unsigned long long)g[2], ch->status, res.stream_status, (long long)res.first_error_record, res.error_rank);
    const int holds_error = res.first_error_record >= 0 ? res.error_rank == rank
                                                        : (res.stream_status != BZQ_EOF && res.plan.is_last);
    if (holds_error) {
        char msg[4096];
        const int64_t m = bzq_format_error(ctx, res.records_before, msg, sizeof msg);
        if (m >= 0) { printf("# error: "); fwrite(msg, 1, (size_t)(m < 4095 ? m : 4095), stdout); printf("\n"); }
    }
    bzq_comm_destroy(ctx);
    if (!(lib_read && atoi(lib_read) > 0)) bzq_device_free(ctx, d_shard);   /* (the library's own buffer goes with the ctx) */
    bzq_destroy(ctx);
    free(host); free(q); free(s); free(id); free(ends); free(id_ends);
    return 0;
}
