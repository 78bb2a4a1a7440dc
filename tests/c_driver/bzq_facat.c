/* The FASTA side of the drop-in boundary from plain C (no Python, no torch): what a foreign host does with
 * bzq_fasta_create / bzq_fasta_ingest_* / bzq_fasta_copy_to_host / bzq_fasta_format_error.
 *
 *   bzq_facat FILE [chunk_bytes] [check_ascii]   ->  one line per record on stdout:  id \t sequence
 *                                                    then "# records=N chunks=C status=S" and, on a failing stream,
 *                                                    the reference's error text after "# error: "
 * Compiled as C (gcc -std=c11). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "blazeseq_hip.h"

static void die(bzq_fasta* h, const char* what, int rc) {
    fprintf(stderr, "%s failed (%d): %s\n", what, rc, bzq_fasta_last_error(h));
    exit(2);
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: bzq_facat FILE [chunk_bytes] [check_ascii]\n"); return 2; }
    const uint64_t chunk_bytes = argc > 2 ? (uint64_t)atoll(argv[2]) : 0;
    bzq_fasta_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.check_ascii = argc > 3 ? atoi(argv[3]) : 0;
    bzq_fasta* h = NULL;
    int rc = bzq_fasta_create(0, &cfg, &h);
    if (rc) { fprintf(stderr, "bzq_fasta_create failed (%d): %s\n", rc, bzq_fasta_last_error(NULL)); return 3; }
    bzq_fasta_ingest* in = NULL;
    if ((rc = bzq_fasta_ingest_open(h, argv[1], chunk_bytes, 4, &in)) < 0) die(h, "bzq_fasta_ingest_open", rc);

    uint64_t total = 0, chunks = 0;
    int status = BZQ_OK;
    size_t cap_s = 0, cap_i = 0, cap_r = 0;
    uint8_t *seq = NULL, *id = NULL;
    int64_t *seq_ends = NULL, *id_ends = NULL;
    while (status == BZQ_OK) {
        bzq_fasta_chunk c;
        if ((rc = bzq_fasta_ingest_next(in, &c, NULL)) < 0) die(h, "bzq_fasta_ingest_next", rc);
        status = c.status;
        ++chunks;
        const size_t n = (size_t)c.n_records;
        if (n == 0) continue;
        if ((size_t)c.seq_bytes > cap_s) { cap_s = (size_t)c.seq_bytes * 2; seq = (uint8_t*)realloc(seq, cap_s); }
        if ((size_t)c.id_bytes > cap_i) { cap_i = (size_t)c.id_bytes * 2 + 16; id = (uint8_t*)realloc(id, cap_i); }
        if (n > cap_r) { cap_r = n * 2; seq_ends = (int64_t*)realloc(seq_ends, cap_r * 8); id_ends = (int64_t*)realloc(id_ends, cap_r * 8); }
        if ((rc = bzq_fasta_copy_to_host(h, seq, c.d_seq_bytes, (size_t)c.seq_bytes)) < 0 ||
            (rc = bzq_fasta_copy_to_host(h, id, c.d_id_bytes, (size_t)c.id_bytes)) < 0 ||
            (rc = bzq_fasta_copy_to_host(h, seq_ends, c.d_seq_ends, n * 8)) < 0 ||
            (rc = bzq_fasta_copy_to_host(h, id_ends, c.d_id_ends, n * 8)) < 0)
            die(h, "bzq_fasta_copy_to_host", rc);
        int64_t s0 = 0, i0 = 0;
        for (size_t r = 0; r < n; ++r) {
            fwrite(id + i0, 1, (size_t)(id_ends[r] - i0), stdout);
            fputc('\t', stdout);
            fwrite(seq + s0, 1, (size_t)(seq_ends[r] - s0), stdout);
            fputc('\n', stdout);
            s0 = seq_ends[r]; i0 = id_ends[r];
        }
        total += n;
    }
    printf("# records=%llu chunks=%llu status=%d\n", (unsigned long long)total, (unsigned long long)chunks, status);
    if (status != BZQ_EOF) {
        char msg[1024];
        bzq_fasta_format_error(h, msg, sizeof msg);
        printf("# error: %s\n", msg);
    }
    bzq_fasta_ingest_close(in);
    bzq_fasta_destroy(h);
    free(seq); free(id); free(seq_ends); free(id_ends);
    return 0;
}
