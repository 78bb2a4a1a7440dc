/* gunzip through the C ABI from plain C (include/blazeseq_hip.h: bzq_gzip_open / bzq_gzip_decode / bzq_gzip_close): the file is
 * read in pieces of PIECE bytes, every piece goes to the device decoder, what comes out is copied back and written to stdout --
 * no Python and no torch in the process.  What a Mojo host does in place of GZFile.read_to_buffer / RapidgzipReader.read_to_buffer
 * (blazeseq/io/readers.mojo:283-443), except that it would leave the bytes on the device for the parser.
 *
 *   bzq_gunzip FILE [piece_bytes] [out_capacity] [chunk_bytes]
 * Exit code 0 and the bytes on stdout, or 3 and the library's message on stderr when the stream is damaged. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "blazeseq_hip.h"

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: bzq_gunzip FILE [piece_bytes] [out_capacity] [chunk_bytes]\n"); return 2; }
    const size_t piece = argc > 2 ? (size_t)atoll(argv[2]) : (size_t)(8u << 20);
    const uint64_t cap = argc > 3 ? (uint64_t)atoll(argv[3]) : (uint64_t)(64u << 20);
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    bzq_config cfg;
    bzq_config_default(&cfg);
    bzq_ctx* ctx = NULL;
    int rc = bzq_create(0, &cfg, &ctx);
    if (rc) { fprintf(stderr, "bzq_create failed (%d): %s\n", rc, bzq_last_error(NULL)); return 2; }
    bzq_gzip* gz = NULL;
    if ((rc = bzq_gzip_open(ctx, &gz)) != 0) { fprintf(stderr, "bzq_gzip_open failed (%d): %s\n", rc, bzq_last_error(ctx)); return 2; }
    if (argc > 4 && (rc = bzq_gzip_set_option(gz, "chunk_bytes", atoll(argv[4]))) != 0) { fprintf(stderr, "chunk_bytes: %s\n", bzq_gzip_last_error(gz)); return 2; }
    void *pinned = NULL, *d_out = NULL;
    if (bzq_pinned_alloc(piece ? piece : 1, &pinned) || bzq_device_alloc(ctx, cap + 64, &d_out)) { fprintf(stderr, "allocation failed\n"); return 2; }
    uint8_t* host = malloc(cap ? cap : 1);
    for (int last = 0; !last;) {
        const size_t n = fread(pinned, 1, piece, f);
        last = n < piece || feof(f);
        if (!last) { int c = fgetc(f); if (c == EOF) last = 1; else ungetc(c, f); }
        size_t fed = n;
        for (int more = 1; more;) {
            uint64_t got = 0;
            int32_t m = 0;
            rc = bzq_gzip_decode(gz, (const uint8_t*)pinned, fed, last, (uint8_t*)d_out, cap, &got, &m);
            if (rc < 0) { fprintf(stderr, "bzq_gzip_decode failed (%d): %s\n", rc, bzq_gzip_last_error(gz)); return 3; }
            if (got) {
                if (bzq_copy_to_host(ctx, host, d_out, got)) { fprintf(stderr, "copy back failed: %s\n", bzq_last_error(ctx)); return 2; }
                fwrite(host, 1, got, stdout);
            }
            fed = 0;      /* the piece is inside the handle now: asking again (more) hands in nothing new */
            more = m;
        }
    }
    bzq_gzip_stats st;
    bzq_gzip_get_stats(gz, &st);
    fprintf(stderr, "# members=%llu bytes_out=%llu chain_jobs=%llu fallback_jobs=%llu finished=%d\n", (unsigned long long)st.members,
            (unsigned long long)st.bytes_out, (unsigned long long)st.chain_jobs, (unsigned long long)st.fallback_jobs, bzq_gzip_finished(gz));
    bzq_gzip_close(gz);
    bzq_device_free(ctx, d_out);
    bzq_pinned_free(pinned);
    bzq_destroy(ctx);
    free(host);
    fclose(f);
    return 0;
}
