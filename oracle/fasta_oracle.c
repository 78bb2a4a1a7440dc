/*
 * fasta_oracle.c -- TEST INFRASTRUCTURE ONLY (not shipped, never on the product path).
 *
 * Flat CPU restatement of the BlazeSeq FASTA record parser, the spec the
 * HIP kernels in blazeseq_amd/csrc/bzq_fasta.hpp implement.  Paths are relative to /root/reference.
 *
 *   FastaParser.next_record      blazeseq/fasta/parser.mojo:122-172
 *   FastaParser._read_header_line   parser.mojo:181-203
 *   Validator._validate          parser.mojo:41-45 (id first, then sequence)
 *   LineIterator.next_line       blazeseq/io/buffered.mojo:600-638 (lines end at '\n'; the last line may have none)
 *   _handle_line_exceeds_capacity   buffered.mojo:737-765 (a line of >= capacity bytes raises; capacity is the
 *                                LineIterator default, DEFAULT_CAPACITY = 256 KiB, CONSTS.mojo:26)
 *   _strip_spaces / is_posix_space  blazeseq/utils.mojo:221-289
 *   ParseError / ValidationError text   blazeseq/errors.mojo:178-234, 318-351
 *
 * Parity status: PINNED BY SOURCE + the reference's own known-answer tests (tests/fasta/test_fasta_parser.mojo,
 * tests/fasta/test_fasta_parser_correctness.mojo replayed in tests/test_oracle_fasta_kats.py), and cross-checked
 * against oracle/fasta.py's line-by-line streaming restatement of LineIterator + FastaParser on random streams.
 *
 * What "flat" means: every line is [start, '\n') (plus a last line without '\n'); a line is stripped of posix
 * spaces at both ends; a stripped line that begins with '>' opens a record whose id is the rest of the line,
 * stripped again; every other stripped line is appended to the open record's sequence; a record ends at the next
 * header or at the end of input.  Errors, in the order the reference meets them:
 *   - a line of >= line_cap bytes (before its '\n'): "Line exceeds buffer capacity of N bytes", raised while the
 *     record it sits in (or, for a header line, the record before it) is still being read;
 *   - the first non-blank line is not a header: ParseError "FASTA: sequence id line does not start with '>'";
 *   - a record with no sequence bytes: ParseError "FASTA record has empty sequence";
 *   - check_ascii: a byte >= 0x80 in id or sequence: ValidationError "Non ASCII letters found".
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

enum {
    FA_OK = 0,
    FA_NO_HEADER = 1,        /* same slot as ID_NO_AT */
    FA_ASCII_INVALID = 4,
    FA_EOF = 6,
    FA_LINE_TOO_LONG = 8,    /* same slot as BUFFER_EXCEEDED */
    FA_EMPTY_SEQUENCE = 11,
    FA_NEED_MORE = 12        /* chunk mode only: fewer than two header lines in the chunk */
};

typedef struct fa_flat {
    int64_t n_records;
    uint8_t* seq_bytes; int64_t seq_bytes_len;
    uint8_t* id_bytes;  int64_t id_bytes_len;
    int64_t* seq_ends;   /* inclusive running sums of sequence lengths */
    int64_t* id_ends;
    int64_t* hdr_pos;    /* offset of each record's '>' */
    int32_t status;      /* FA_OK (more input expected, chunk mode), FA_EOF (clean end) or the first error */
    int32_t _pad;
    int64_t err_record;  /* 0-based index of the record the error stopped at (= n_records), -1 before any header */
    int64_t err_record_number, err_line_number, err_file_position;   /* ParseContext as the reference prints it */
    int64_t consumed;    /* chunk mode: offset of the first byte the caller must carry into the next chunk */
    int64_t lines_consumed;   /* '\n' count in [0, consumed) */
    int64_t total_lines;
    char message[512];
} fa_flat;

static int fa_is_space(uint8_t c) { /* utils.mojo:267-289 */
    return c == 9 || c == 10 || c == 11 || c == 12 || c == 13 || c == 28 || c == 29 || c == 30 || c == 32;
}

static void fa_strip(const uint8_t* d, int64_t* lo, int64_t* hi) { /* utils.mojo:221-242 */
    while (*lo < *hi && fa_is_space(d[*lo])) ++*lo;
    while (*hi > *lo && fa_is_space(d[*hi - 1])) --*hi;
}

static void fa_parse_error(fa_flat* f, const char* msg, int64_t rec, int64_t line, int64_t pos) {
    /* ParseError.write_to, errors.mojo:178-192: fields printed only when > 0 */
    size_t k = (size_t)snprintf(f->message, sizeof f->message, "%s", msg);
    if (rec > 0) k += (size_t)snprintf(f->message + k, sizeof f->message - k, "\n  Record number: %lld", (long long)rec);
    if (line > 0) k += (size_t)snprintf(f->message + k, sizeof f->message - k, "\n  Line number: %lld", (long long)line);
    if (pos > 0) k += (size_t)snprintf(f->message + k, sizeof f->message - k, "\n  File position: %lld", (long long)pos);
    f->err_record_number = rec; f->err_line_number = line; f->err_file_position = pos;
}

void fa_flat_free(fa_flat* f) {
    free(f->seq_bytes); free(f->id_bytes); free(f->seq_ends); free(f->id_ends); free(f->hdr_pos);
    memset(f, 0, sizeof *f);
}

/* record_base / line_base / pos_base: what came before this chunk (0 for a whole file), so that messages carry
 * stream-global numbers exactly like the reference's single parser object. */
int fa_flat_parse(const uint8_t* d, int64_t n, int check_ascii, int64_t line_cap, int is_eof,
                  int64_t record_base, int64_t line_base, int64_t pos_base, fa_flat* f) {
    memset(f, 0, sizeof *f);
    int64_t cap_rec = n / 2 + 2;
    f->seq_bytes = (uint8_t*)malloc((size_t)n + 1);
    f->id_bytes = (uint8_t*)malloc((size_t)n + 1);
    f->seq_ends = (int64_t*)malloc((size_t)cap_rec * 8);
    f->id_ends = (int64_t*)malloc((size_t)cap_rec * 8);
    f->hdr_pos = (int64_t*)malloc((size_t)cap_rec * 8);
    if (!f->seq_bytes || !f->id_bytes || !f->seq_ends || !f->id_ends || !f->hdr_pos) return -1;

    int64_t seq_len = 0, id_len = 0;          /* committed bytes */
    int64_t cur = -1;                          /* open record: index, or -1 */
    int64_t cur_seq0 = 0, cur_id0 = 0;         /* column offsets where the open record began */
    int64_t cur_hdr_line = 0, cur_line_start = 0;
    int64_t w_seq = 0, w_id = 0;               /* write cursors (committed + open record) */
    int64_t i = 0, line_no = 0;
    int64_t blank_consumed = 0, blank_lines = 0;   /* chunk mode before any header: blank lines can be dropped */
    f->status = FA_OK;
    f->err_record = -1;

    while (i < n) {
        const int64_t line_start = i;
        const uint8_t* nl = (const uint8_t*)memchr(d + i, '\n', (size_t)(n - i));
        const int64_t j = nl ? (int64_t)(nl - d) : n;
        if (!nl && !is_eof) break;   /* chunk mode: a line without its '\n' yet is not looked at (it may still turn out too long) */
        ++line_no;
        if (j - line_start >= line_cap) {   /* buffered.mojo:634-636 */
            snprintf(f->message, sizeof f->message, "Line exceeds buffer capacity of %lld bytes", (long long)line_cap);
            f->status = FA_LINE_TOO_LONG;
            f->err_record = cur;
            break;
        }
        int64_t lo = line_start, hi = j;
        fa_strip(d, &lo, &hi);
        if (lo < hi && d[lo] == '>') {
            if (cur >= 0) {   /* the open record ends here: parser.mojo:157-170 */
                if (w_seq == cur_seq0) {
                    fa_parse_error(f, "FASTA record has empty sequence", record_base + cur + 1, line_base + cur_hdr_line + 1,
                                   pos_base + line_start);
                    f->status = FA_EMPTY_SEQUENCE; f->err_record = cur;
                    break;
                }
                if (check_ascii) {
                    int bad = 0;
                    for (int64_t k = cur_id0; k < w_id && !bad; ++k) bad = f->id_bytes[k] >= 0x80;
                    for (int64_t k = cur_seq0; k < w_seq && !bad; ++k) bad = f->seq_bytes[k] >= 0x80;
                    if (bad) {   /* ValidationError: only the record number, and it is the count so far (0 is omitted) */
                        size_t k = (size_t)snprintf(f->message, sizeof f->message, "Non ASCII letters found");
                        if (record_base + cur > 0)
                            snprintf(f->message + k, sizeof f->message - k, "\n  Record number: %lld", (long long)(record_base + cur));
                        f->err_record_number = record_base + cur;
                        f->status = FA_ASCII_INVALID; f->err_record = cur;
                        break;
                    }
                }
                f->seq_ends[cur] = w_seq; f->id_ends[cur] = w_id;
                seq_len = w_seq; id_len = w_id;
                f->n_records = cur + 1;
            }
            ++cur;
            cur_seq0 = w_seq; cur_id0 = w_id; cur_hdr_line = line_no; cur_line_start = line_start;
            f->hdr_pos[cur] = lo;
            int64_t a = lo + 1, b = hi;
            fa_strip(d, &a, &b);
            memcpy(f->id_bytes + w_id, d + a, (size_t)(b - a));
            w_id += b - a;
        } else if (lo < hi) {
            if (cur < 0) {   /* parser.mojo:196-200 */
                fa_parse_error(f, "FASTA: sequence id line does not start with '>'", record_base, line_base + line_no,
                               pos_base + line_start);
                f->status = FA_NO_HEADER; f->err_record = -1;
                break;
            }
            memcpy(f->seq_bytes + w_seq, d + lo, (size_t)(hi - lo));
            w_seq += hi - lo;
        } else if (cur < 0 && nl) {
            blank_consumed = j + 1; blank_lines = line_no;
        }
        i = j + 1;
        if (!nl) break;
    }
    f->total_lines = line_no;

    if (f->status == FA_OK && is_eof) {
        if (cur >= 0) {   /* the last record ends at the end of input; file position = n (next_line sets it first) */
            if (w_seq == cur_seq0) {
                fa_parse_error(f, "FASTA record has empty sequence", record_base + cur + 1, line_base + cur_hdr_line + 1, pos_base + n);
                f->status = FA_EMPTY_SEQUENCE; f->err_record = cur;
            } else {
                int bad = 0;
                if (check_ascii) {
                    for (int64_t k = cur_id0; k < w_id && !bad; ++k) bad = f->id_bytes[k] >= 0x80;
                    for (int64_t k = cur_seq0; k < w_seq && !bad; ++k) bad = f->seq_bytes[k] >= 0x80;
                }
                if (bad) {
                    size_t k = (size_t)snprintf(f->message, sizeof f->message, "Non ASCII letters found");
                    if (record_base + cur > 0)
                        snprintf(f->message + k, sizeof f->message - k, "\n  Record number: %lld", (long long)(record_base + cur));
                    f->err_record_number = record_base + cur;
                    f->status = FA_ASCII_INVALID; f->err_record = cur;
                } else {
                    f->seq_ends[cur] = w_seq; f->id_ends[cur] = w_id;
                    seq_len = w_seq; id_len = w_id;
                    f->n_records = cur + 1;
                }
            }
        }
        if (f->status == FA_OK) f->status = FA_EOF;
        f->consumed = n; f->lines_consumed = line_no;
    } else if (f->status == FA_OK) {
        /* chunk mode: carry from the start of the open record's header line */
        if (cur >= 0) {
            f->consumed = cur_line_start; f->lines_consumed = cur_hdr_line - 1;
        } else {
            f->consumed = blank_consumed; f->lines_consumed = blank_lines;
        }
        if (f->n_records == 0) f->status = FA_NEED_MORE;
    }
    f->seq_bytes_len = seq_len; f->id_bytes_len = id_len;
    return 0;
}

/* generate_synthetic_fasta_buffer, blazeseq/utils.mojo:1033-1139.  Returns the byte count; writes when out != NULL. */
int64_t fa_generate_synthetic(int64_t num_reads, int64_t min_len, int64_t max_len, int64_t line_width, double gc_bias,
                              uint8_t* out, int64_t cap) {
    if (num_reads <= 0) return 0;
    if (min_len < 0 || max_len < 0 || min_len > max_len || line_width <= 0) return -1;
    int gc_slots = (int)((float)gc_bias * 8.0f + 0.5f);
    if (gc_slots < 0) gc_slots = 0;
    if (gc_slots > 8) gc_slots = 8;
    uint8_t lut[8]; int m = 0;
    for (int k = 0; k < gc_slots; ++k) lut[m++] = (k % 2 == 0) ? 'G' : 'C';
    for (int k = 0; k < 8 - gc_slots; ++k) lut[m++] = (k % 2 == 0) ? 'A' : 'T';
    int digits = 1;
    if (num_reads > 1) { char t[32]; digits = snprintf(t, sizeof t, "%lld", (long long)(num_reads - 1)); }
    int64_t w = 0;
    for (int64_t i = 0; i < num_reads; ++i) {
        int64_t L = (max_len == min_len) ? min_len : min_len + ((i * 31 + 7) % (max_len - min_len + 1));
        int64_t need = 6 + digits + 1 + L + L / line_width + (L % line_width ? 1 : 0);
        if (out && w + need > cap) return -2;
        if (out) {
            char h[40];
            int hl = snprintf(h, sizeof h, ">read_%0*lld\n", digits, (long long)i);
            memcpy(out + w, h, (size_t)hl);
            w += hl;
            uint64_t s = ((uint64_t)i * 6364136223846793005ull + 1442695040888963407ull) & 0x7FFFFFFFFFFFFFFFull;
            int64_t col = 0;
            for (int64_t k = 0; k < L; ++k) {
                s = (s * 6364136223846793005ull + 1442695040888963407ull) & 0x7FFFFFFFFFFFFFFFull;
                out[w++] = lut[(s >> 33) % 8];
                if (++col == line_width) { out[w++] = '\n'; col = 0; }
            }
            if (col > 0) out[w++] = '\n';
        } else {
            w += need;
        }
    }
    return w;
}

/* compute_num_fasta_reads_for_size, utils.mojo:989-1030 */
int64_t fa_compute_num_reads_for_size(int64_t target, int64_t min_len, int64_t max_len, int64_t line_width) {
    if (target <= 0) return 0;
    int64_t avg = (min_len + max_len) / 2;
    int64_t nl = (avg + line_width - 1) / line_width;
    int64_t est = target / (15 + avg + nl);
    if (est <= 0) return 0;
    int digits = 1;
    if (est > 1) { char t[32]; digits = snprintf(t, sizeof t, "%lld", (long long)(est - 1)); }
    return target / (6 + digits + 1 + avg + nl);
}

/* Timed CPU leg for bench.py: count records and bases like benchmark/fasta-parser/run_blazeseq_fasta.mojo. */
int64_t fa_bench_run(const uint8_t* d, int64_t n, int64_t out[2]) {
    fa_flat f;
    if (fa_flat_parse(d, n, 0, 262144, 1, 0, 0, 0, &f)) return -1;
    out[0] = f.n_records; out[1] = f.seq_bytes_len;
    int64_t st = f.status;
    fa_flat_free(&f);
    return st;
}
