/* One rank of the FASTA byte-range shard protocol from plain C (include/blazeseq_hip.h: bzq_comm_init_shm / bzq_comm_init +
 * bzq_fasta_shard_stitch) -- no Python and no torch in the process; start one process per rank.
 *
 *   bzq_fasta_shard shm  RANK NRANKS NAME   FILE LO HI [check_ascii] [line_capacity]
 *   bzq_fasta_shard rccl RANK NRANKS IDFILE FILE LO HI [check_ascii] [line_capacity]   (world size 1 on a one-GPU box)
 *
 * The rank takes bytes [LO, HI) of FILE (cut anywhere), runs the protocol and prints its records as "hex(id) hex(seq)"
 * lines ("-" for an empty field), then
 *   # rank=R records=N before=B global_records=G status=T stream_status=U first_error=E error_rank=K head=H halo=L
 * and, on the rank that holds the stream's error, "# error " + hex of the reference's text. */
#define _POSIX_C_SOURCE 200809L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "blazeseq_hip.h"

static void hex(const uint8_t* p, int64_t n) {
    if (n == 0) { fputc('-', stdout); return; }
    for (int64_t i = 0; i < n; ++i) printf("%02x", p[i]);
}

int main(int argc, char** argv) {
    if (argc < 8) { fprintf(stderr, "usage: bzq_fasta_shard shm|rccl RANK NRANKS NAME|IDFILE FILE LO HI [check_ascii] [line_capacity]\n"); return 2; }
    const int rccl = strcmp(argv[1], "rccl") == 0;
    const int rank = atoi(argv[2]), nranks = atoi(argv[3]);
    const uint64_t lo = strtoull(argv[6], NULL, 10), hi = strtoull(argv[7], NULL, 10), n = hi - lo;
    const int check = argc > 8 ? atoi(argv[8]) : 0;
    const long long linecap = argc > 9 ? atoll(argv[9]) : 0;

    FILE* f = fopen(argv[5], "rb");
    if (!f) { perror(argv[5]); return 2; }
    uint8_t* host = malloc(n ? n : 1);
    fseek(f, (long)lo, SEEK_SET);
    if (n && fread(host, 1, n, f) != n) { fprintf(stderr, "short read\n"); return 2; }
    fclose(f);

    bzq_config cfg;
    bzq_config_default(&cfg);
    bzq_ctx* ctx = NULL;
    int rc = bzq_create(0, &cfg, &ctx);
    if (rc) { fprintf(stderr, "bzq_create failed (%d): %s\n", rc, bzq_last_error(NULL)); return 3; }
    if (rccl) {
        bzq_nccl_id id;
        if (nranks != 1) { fprintf(stderr, "this driver runs the RCCL transport at world size 1 only\n"); return 2; }
        if ((rc = bzq_comm_get_unique_id(&id)) != 0 || (rc = bzq_comm_init(ctx, rank, nranks, &id)) != 0) {
            fprintf(stderr, "bzq_comm_init failed (%d): %s\n", rc, bzq_last_error(ctx)); return 3;
        }
    } else if ((rc = bzq_comm_init_shm(ctx, rank, nranks, argv[4], 0)) != 0) {
        fprintf(stderr, "bzq_comm_init_shm failed (%d): %s\n", rc, bzq_last_error(ctx)); return 3;
    }
    if ((rc = bzq_comm_selftest(ctx)) != 0) { fprintf(stderr, "bzq_comm_selftest failed (%d): %s\n", rc, bzq_last_error(ctx)); return 3; }
    bzq_fasta_config fc;
    memset(&fc, 0, sizeof fc);
    fc.check_ascii = check; fc.line_capacity = linecap;
    bzq_fasta* fa = NULL;
    if ((rc = bzq_fasta_create(0, &fc, &fa)) != 0) { fprintf(stderr, "bzq_fasta_create failed (%d): %s\n", rc, bzq_fasta_last_error(NULL)); return 3; }

    const uint64_t capacity = n + (4u << 20);   /* room for the halo */
    void* d_shard = NULL;
    if ((rc = bzq_device_alloc(ctx, capacity, &d_shard)) != 0 || (rc = bzq_copy_to_device(ctx, d_shard, host, n)) != 0) {
        fprintf(stderr, "device buffer failed (%d): %s\n", rc, bzq_last_error(ctx)); return 3;
    }

    bzq_fasta_shard_result res;
    if ((rc = bzq_fasta_shard_stitch(ctx, fa, (uint8_t*)d_shard, n, capacity, &res)) != 0) {
        fprintf(stderr, "bzq_fasta_shard_stitch failed (%d): %s\n", rc, bzq_fasta_last_error(fa)); return 4;
    }
    const bzq_fasta_chunk* ch = &res.chunk;
    uint8_t *s = malloc((size_t)ch->seq_bytes + 1), *id = malloc((size_t)ch->id_bytes + 1);
    int64_t *se = malloc((size_t)ch->n_records * 8 + 8), *ie = malloc((size_t)ch->n_records * 8 + 8);
    if (ch->n_records > 0 &&
        ((rc = bzq_fasta_copy_to_host(fa, s, ch->d_seq_bytes, (size_t)ch->seq_bytes)) || (rc = bzq_fasta_copy_to_host(fa, id, ch->d_id_bytes, (size_t)ch->id_bytes)) ||
         (rc = bzq_fasta_copy_to_host(fa, se, ch->d_seq_ends, (size_t)ch->n_records * 8)) || (rc = bzq_fasta_copy_to_host(fa, ie, ch->d_id_ends, (size_t)ch->n_records * 8)))) {
        fprintf(stderr, "bzq_fasta_copy_to_host failed (%d): %s\n", rc, bzq_fasta_last_error(fa)); return 4;
    }
    for (int64_t r = 0; r < ch->n_records; ++r) {
        const int64_t s0 = r ? se[r - 1] : 0, i0 = r ? ie[r - 1] : 0;
        hex(id + i0, ie[r] - i0); fputc(' ', stdout); hex(s + s0, se[r] - s0); fputc('\n', stdout);
    }
    printf("# rank=%d records=%lld before=%llu global_records=%llu status=%d stream_status=%d first_error=%lld error_rank=%d head=%llu halo=%llu\n",
           rank, (long long)ch->n_records, (unsigned long long)res.records_before, (unsigned long long)res.global_records, ch->status,
           res.stream_status, (long long)res.first_error_record, res.error_rank, (unsigned long long)res.plan.head_bytes,
           (unsigned long long)res.plan.halo_bytes);
    if (res.error_rank == rank) {
        char msg[4096];
        const int m = bzq_fasta_format_error(fa, msg, sizeof msg);
        printf("# error "); hex((const uint8_t*)msg, m < 4095 ? m : 4095); printf("\n");
    }
    bzq_fasta_destroy(fa);
    bzq_comm_destroy(ctx);
    bzq_device_free(ctx, d_shard);
    bzq_destroy(ctx);
    free(host); free(s); free(id); free(se); free(ie);
    return 0;
}
