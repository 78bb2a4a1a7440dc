/*
 * bzq_oracle.c -- TEST INFRASTRUCTURE ONLY.  See bzq_oracle.h for the status header.
 *
 * CPU restatement of BlazeSeq's FASTQ batch-parse path.  Each function names the reference
 * lines it follows (paths relative to /root/reference).  Nothing here is copied: the reference
 * is Mojo, this is a from-scratch C statement of the same algorithm.
 */
#include "bzq_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ constants */
/* blazeseq/CONSTS.mojo:12-31 */
#define C_READ_HEADER 64   /* '@' */
#define C_QUAL_HEADER 43   /* '+' */
#define C_NEWLINE 10
#define C_CR 13
#define DEFAULT_CAPACITY (256 * 1024)
#define MAX_CAPACITY (1 << 30)
#define DEFAULT_BATCH_SIZE 4096

void orc_config_default(orc_config* c) {
    memset(c, 0, sizeof(*c));
    c->buffer_capacity = DEFAULT_CAPACITY;
    c->buffer_max_capacity = MAX_CAPACITY;
    c->buffer_growth_enabled = 0;
    c->check_ascii = 0;
    c->check_quality = 0;
    c->q_lower = 33; /* generic_schema, quality_schema.mojo:26 */
    c->q_upper = 126;
    c->q_offset = 33;
    c->simd_width = 0;
    c->batch_size = DEFAULT_BATCH_SIZE;
}

/* _parse_schema, utils.mojo:612-637; schema table quality_schema.mojo:26-31 */
int orc_schema_from_name(const char* name, uint8_t* lower, uint8_t* upper, uint8_t* offset) {
    static const struct { const char* n; uint8_t lo, up, off; } T[] = {
        {"sanger", 33, 126, 33},       {"solexa", 59, 126, 64},
        {"illumina_1.3", 64, 126, 64}, {"illumina_1.5", 66, 126, 64},
        {"illumina_1.8", 33, 126, 33}, {"generic", 33, 126, 33},
    };
    for (size_t i = 0; i < sizeof(T) / sizeof(T[0]); ++i) {
        if (name && strcmp(name, T[i].n) == 0) {
            *lower = T[i].lo; *upper = T[i].up; *offset = T[i].off;
            return 1;
        }
    }
    /* unknown: the reference prints a warning and parses with the generic schema */
    *lower = 33; *upper = 126; *offset = 33;
    return 0;
}

/* _message_for_code, errors.mojo:71-90 */
const char* orc_message_for_code(int code) {
    switch (code) {
        case ORC_ID_NO_AT: return "Sequence id line does not start with '@'";
        case ORC_SEP_NO_PLUS: return "Separator line does not start with '+'";
        case ORC_SEQ_QUAL_LEN_MISMATCH: return "Quality and sequence line do not match in length";
        case ORC_ASCII_INVALID: return "Non ASCII letters found";
        case ORC_QUALITY_OUT_OF_RANGE: return "Corrupt quality score according to provided schema";
        case ORC_UNEXPECTED_EOF: return "Unexpected end of file in FASTQ record";
        case ORC_BUFFER_EXCEEDED: return "FASTQ record exceeds buffer capacity";
        case ORC_BUFFER_AT_MAX: return "FASTQ record exceeds maximum buffer capacity";
        default: return "Parse or validation error";
    }
}

/* --------------------------------------------------------------- byte helpers */

/* is_posix_space, utils.mojo:267-289: {9,10,11,12,13,28,29,30,32} */
static int is_posix_space(uint8_t c) {
    if (c > 32) return 0;
    const uint64_t mask = (1ull << 9) | (1ull << 10) | (1ull << 11) | (1ull << 12) | (1ull << 13) |
                          (1ull << 28) | (1ull << 29) | (1ull << 30) | (1ull << 32);
    return (int)((mask >> c) & 1u);
}

/* _strip_spaces, utils.mojo:221-242 */
static void strip_spaces(const uint8_t* s, int64_t len, int64_t* out_start, int64_t* out_len) {
    if (len == 0) { *out_start = 0; *out_len = 0; return; }
    if (!is_posix_space(s[0]) && !is_posix_space(s[len - 1])) { *out_start = 0; *out_len = len; return; }
    int64_t start = 0;
    while (start < len && is_posix_space(s[start])) start++;
    int64_t end = len;
    while (end > start && is_posix_space(s[end - 1])) end--;
    *out_start = start;
    *out_len = end - start;
}

/* _check_ascii, utils.mojo:245-263 (the SIMD body and the scalar tail test the same bit) */
static int check_ascii(const uint8_t* s, int64_t len) {
    for (int64_t i = 0; i < len; ++i)
        if (s[i] & 0x80) return ORC_ASCII_INVALID;
    return ORC_OK;
}

/* Validator._validate_quality_range, fastq/record.mojo:76-104.  The SIMD body (first
 * floor(n/W)*W bytes) rejects (q-LOWER) >= span, the scalar tail rejects (q-LOWER) > span. */
static int validate_quality_range(const uint8_t* q, int64_t n, uint8_t lower, uint8_t upper, int W) {
    uint8_t span = (uint8_t)(upper - lower);
    int64_t i = 0;
    if (W > 0) {
        while (i + W <= n) {
            for (int k = 0; k < W; ++k)
                if ((uint8_t)(q[i + k] - lower) >= span) return ORC_QUALITY_OUT_OF_RANGE;
            i += W;
        }
    }
    while (i < n) {
        if ((uint8_t)(q[i] - lower) > span) return ORC_QUALITY_OUT_OF_RANGE;
        i++;
    }
    return ORC_OK;
}

/* Validator._validate, fastq/record.mojo:162-172: ascii (id, seq, qual) first, then quality */
static int validate_view(const orc_config* c, const orc_view* v) {
    if (c->check_ascii) {
        int code = check_ascii(v->id, v->id_len);
        if (code != ORC_OK) return code;
        code = check_ascii(v->seq, v->seq_len);
        if (code != ORC_OK) return code;
        code = check_ascii(v->qual, v->qual_len);
        if (code != ORC_OK) return code;
    }
    if (c->check_quality)
        return validate_quality_range(v->qual, v->qual_len, c->q_lower, c->q_upper, c->simd_width);
    return ORC_OK;
}

/* --------------------------------------------------------- message formatting */

typedef struct { char* p; size_t cap, len; } sbuf;
static void sb_init(sbuf* s, char* p, size_t cap) { s->p = p; s->cap = cap; s->len = 0; if (cap) p[0] = 0; }
static void sb_putn(sbuf* s, const void* src, size_t n) {
    if (s->len + n + 1 > s->cap) n = (s->cap > s->len + 1) ? s->cap - s->len - 1 : 0;
    memcpy(s->p + s->len, src, n);
    s->len += n;
    s->p[s->len] = 0;
}
static void sb_puts(sbuf* s, const char* z) { sb_putn(s, z, strlen(z)); }
static void sb_puti(sbuf* s, long long v) { char t[32]; snprintf(t, sizeof t, "%lld", v); sb_puts(s, t); }

/* ParseError.write_to, errors.mojo:178-192 */
static void format_parse_error(char* dst, size_t cap, const char* message, int64_t record_number,
                               int64_t line_number, int64_t file_position, const uint8_t* snip,
                               int64_t snip_len) {
    sbuf s; sb_init(&s, dst, cap);
    sb_puts(&s, message);
    if (record_number > 0) { sb_puts(&s, "\n  Record number: "); sb_puti(&s, record_number); }
    if (line_number > 0) { sb_puts(&s, "\n  Line number: "); sb_puti(&s, line_number); }
    if (file_position > 0) { sb_puts(&s, "\n  File position: "); sb_puti(&s, file_position); }
    if (snip_len > 0) { sb_puts(&s, "\n  Record snippet: "); sb_putn(&s, snip, (size_t)snip_len); }
}

/* ValidationError.write_to, errors.mojo:223-234 (field is "" on the parser path, parser.mojo:164-169) */
static void format_validation_error(char* dst, size_t cap, const char* message, int64_t record_number,
                                    const uint8_t* snip, int64_t snip_len) {
    sbuf s; sb_init(&s, dst, cap);
    sb_puts(&s, message);
    if (record_number > 0) { sb_puts(&s, "\n  Record number: "); sb_puti(&s, record_number); }
    if (snip_len > 0) { sb_puts(&s, "\n  Record snippet: "); sb_putn(&s, snip, (size_t)snip_len); }
}

/* FastqParser._get_record_snippet, parser.mojo:597-610 */
static int64_t validation_snippet(const orc_view* v, uint8_t* out /* >= 512 */) {
    int64_t n = 0;
    if (v->id_len > 0) {
        int64_t take = v->id_len > 400 ? 400 : v->id_len; /* only the first 200 can survive */
        memcpy(out, v->id, (size_t)take);
        n = v->id_len;            /* logical length (may exceed what we stored) */
        if (n < 200) { out[n++] = '\n'; }
    }
    if (n < 200 && v->seq_len > 0) {
        int64_t take = v->seq_len < 200 - n ? v->seq_len : 200 - n;
        memcpy(out + n, v->seq, (size_t)take);
        n += take;
    }
    if (n > 200) { memcpy(out + 197, "...", 3); n = 200; }
    return n;
}

/* _refill_error_message, parser.mojo:276-309 (non-structure branches) */
static void format_refill_error(char* dst, size_t cap, int code, int phase, int64_t capacity,
                                int64_t max_capacity) {
    sbuf s; sb_init(&s, dst, cap);
    if (code == ORC_UNEXPECTED_EOF) {
        sb_puts(&s, "Unexpected end of file in FASTQ record at phase ");
        sb_puti(&s, phase);
    } else if (code == ORC_BUFFER_EXCEEDED) {
        sb_puts(&s, "FASTQ record exceeds buffer capacity (");
        sb_puti(&s, capacity);
        sb_puts(&s, " bytes). Enable buffer growth or increase buffer_capacity.");
    } else {
        sb_puts(&s, "FASTQ record exceeds maximum buffer capacity (");
        sb_puti(&s, max_capacity);
        sb_puts(&s, " bytes). Enable buffer growth or increase max_capacity.");
    }
}

/* ============================================================== streaming path */

struct orc_parser {
    /* MemoryReader, io/readers.mojo:140-223 */
    const uint8_t* data;
    int64_t data_len;
    int64_t position;
    /* BufferedReader, io/buffered.mojo:115-149 */
    uint8_t* ptr;
    int64_t len;   /* capacity */
    int64_t head;
    int64_t end;
    int is_eof;
    int64_t stream_position;
    /* FastqParser, fastq/parser.mojo:77-145 */
    orc_config cfg;
    int64_t max_capacity;
    int64_t line_number;
    char errmsg[1400];
};

/* MemoryReader.read_to_buffer, io/readers.mojo:173-212 */
static int64_t reader_read(orc_parser* p, uint8_t* dst, int64_t amt) {
    if (p->position >= p->data_len) return 0;
    int64_t available = p->data_len - p->position;
    int64_t n = amt < available ? amt : available;
    if (n > 0) {
        memcpy(dst, p->data + p->position, (size_t)n);
        p->position += n;
    }
    return n;
}

/* BufferedReader._fill_buffer, io/buffered.mojo:262-281: EOF flag only on a zero-length read */
static int64_t fill_buffer(orc_parser* p) {
    if (p->is_eof) return 0;
    int64_t space = p->len - p->end;
    if (space == 0) return 0;
    int64_t amt = reader_read(p, p->ptr + p->end, space);
    p->end += amt;
    if (amt == 0) p->is_eof = 1;
    return amt;
}

/* BufferedReader._compact_from, io/buffered.mojo:239-260 */
static void compact_from(orc_parser* p, int64_t from_pos) {
    if (from_pos == 0) return;
    if (from_pos >= p->end) {
        p->stream_position += p->end;
        p->head = 0;
        p->end = 0;
        return;
    }
    p->stream_position += from_pos;
    int64_t remaining = p->end - from_pos;
    memmove(p->ptr, p->ptr + from_pos, (size_t)remaining);
    if (p->head < from_pos) p->head = 0; else p->head -= from_pos;
    p->end = remaining;
}

/* compact_and_fill, io/buffered.mojo:283-290 */
static int64_t compact_and_fill(orc_parser* p) {
    compact_from(p, p->head);
    return fill_buffer(p);
}

/* resize_buffer + _resize_internal, io/buffered.mojo:210-217, 292-299 */
static void resize_buffer(orc_parser* p, int64_t additional, int64_t max_capacity) {
    int64_t new_cap = p->len + additional;
    if (new_cap > max_capacity) new_cap = max_capacity;
    uint8_t* np = (uint8_t*)malloc((size_t)(new_cap > 0 ? new_cap : 1));
    memcpy(np, p->ptr, (size_t)(p->len < new_cap ? p->len : new_cap));
    free(p->ptr);
    p->ptr = np;
    p->len = new_cap;
}

static int64_t buf_available(const orc_parser* p) { return p->end - p->head; }

orc_parser* orc_parser_new(const uint8_t* data, int64_t n, const orc_config* cfg) {
    orc_parser* p = (orc_parser*)calloc(1, sizeof(orc_parser));
    p->data = data;
    p->data_len = n;
    p->position = 0;
    p->cfg = *cfg;
    if (p->cfg.batch_size <= 0) p->cfg.batch_size = DEFAULT_BATCH_SIZE;
    p->len = cfg->buffer_capacity;
    p->ptr = (uint8_t*)malloc((size_t)(p->len > 0 ? p->len : 1));
    p->max_capacity = cfg->buffer_max_capacity;
    (void)fill_buffer(p); /* BufferedReader.__init__ reads once, buffered.mojo:149 */
    return p;
}

void orc_parser_free(orc_parser* p) {
    if (!p) return;
    free(p->ptr);
    free(p);
}

int orc_parser_has_more(const orc_parser* p) { return buf_available(p) > 0 || !p->is_eof; }
const char* orc_parser_error(const orc_parser* p) { return p->errmsg; }
int64_t orc_parser_line_number(const orc_parser* p) { return p->line_number; }
int64_t orc_parser_stream_position(const orc_parser* p) { return p->stream_position + p->head; }
int64_t orc_parser_capacity(const orc_parser* p) { return p->len; }

typedef struct { int64_t header_start, seq_start, sep_start, qual_start, record_end; } rec_offsets;

/* _store_newline_offset, utils.mojo:408-432 */
static void store_newline_offset(rec_offsets* o, int found, int64_t abs_pos) {
    if (found == 1) o->seq_start = abs_pos;
    else if (found == 2) o->sep_start = abs_pos;
    else if (found == 3) o->qual_start = abs_pos;
    else o->record_end = abs_pos - 1;
}

/* _phase_start_offset, utils.mojo:332-353 */
static int64_t phase_start_offset(const rec_offsets* o, int phase) {
    switch (phase) {
        case 0: return o->header_start;
        case 1: return o->seq_start;
        case 2: return o->sep_start;
        default: return o->qual_start;
    }
}

/* _validate_fastq_structure, utils.mojo:448-462 */
static int validate_structure(const uint8_t* view, const rec_offsets* o) {
    if (view[o->header_start] != C_READ_HEADER) return ORC_ID_NO_AT;
    if (view[o->sep_start] != C_QUAL_HEADER) return ORC_SEP_NO_PLUS;
    int64_t seq_len = o->sep_start - o->seq_start - 1;
    int64_t qual_len = o->record_end - o->qual_start;
    if (seq_len != qual_len) return ORC_SEQ_QUAL_LEN_MISMATCH;
    return ORC_OK;
}

/* _scan_record, utils.mojo:470-551.  The reference sweeps W bytes per iteration with
 * pack_bits/count_trailing_zeros and finishes with a scalar tail; both visit newlines in
 * ascending order and stop at the fourth, which is what this loop does. */
static void scan_record(const uint8_t* view, int64_t view_len, rec_offsets* o, int* phase,
                        int* complete, int* code) {
    int64_t start_rel = phase_start_offset(o, *phase);
    int64_t avail = view_len - start_rel;
    *code = ORC_OK;
    if (avail <= 0) { *complete = 0; return; }
    int found = *phase; /* _phase_to_count, utils.mojo:361-372 */
    const uint8_t* ptr = view + start_rel;
    int64_t i = 0;
    while (i < avail && found < 4) {
        const uint8_t* hit = (const uint8_t*)memchr(ptr + i, C_NEWLINE, (size_t)(avail - i));
        if (!hit) break;
        i = (int64_t)(hit - ptr);
        found++;
        store_newline_offset(o, found, start_rel + i + 1);
        i++;
    }
    if (found == 4) {
        *code = validate_structure(view, o);
        *complete = 1;
        *phase = 0;
        return;
    }
    *complete = 0;
    *phase = found; /* _count_to_phase, utils.mojo:380-398 */
}

/* _record_snippet, utils.mojo:435-445 */
static int64_t record_snippet_len(int64_t view_len, const rec_offsets* o) {
    int64_t end = o->record_end + 1 < view_len ? o->record_end + 1 : view_len;
    if (end > 200) end = 200;
    if (end <= 0) return 0;
    return end;
}

/* _check_end_qual, utils.mojo:292-329 */
static int check_end_qual(const orc_parser* p, int64_t base, rec_offsets* o) {
    int64_t rest_start = base + o->qual_start;
    int64_t rest_len = p->end - rest_start;
    int all_blank = 1;
    for (int64_t i = 0; i < rest_len; ++i) {
        uint8_t b = p->ptr[rest_start + i];
        if (b != C_NEWLINE && b != C_CR && b != ' ' && b != '\t') { all_blank = 0; break; }
    }
    if (all_blank) return 0;
    o->record_end = p->end - base;
    return 1;
}

/* _next_ref_complete, parser.mojo:451-522 */
static void next_ref_complete(orc_parser* p, int64_t base, rec_offsets* o, int* phase, int* complete,
                              int* refill_code) {
    int current_phase = *phase;
    int64_t new_base = base;
    for (;;) {
        int64_t avail = buf_available(p);
        int64_t capacity = p->len;
        if (avail < capacity && p->is_eof) {
            if (current_phase == 3) {
                *complete = check_end_qual(p, new_base, o);
                *phase = (int)new_base; /* the reference returns SearchPhase(Int8(new_base)) */
                *refill_code = ORC_OK;
                return;
            }
            *complete = 0; *phase = current_phase; *refill_code = ORC_UNEXPECTED_EOF;
            return;
        }
        if (new_base == 0) {
            if (!p->cfg.buffer_growth_enabled) {
                *complete = 0; *phase = current_phase; *refill_code = ORC_BUFFER_EXCEEDED;
                return;
            }
            int64_t current_cap = p->len;
            int64_t max_cap = p->max_capacity;
            if (current_cap >= max_cap) {
                *complete = 0; *phase = current_phase; *refill_code = ORC_BUFFER_AT_MAX;
                return;
            }
            int64_t growth = current_cap < max_cap - current_cap ? current_cap : max_cap - current_cap;
            resize_buffer(p, growth, max_cap);
        } else {
            compact_from(p, new_base);
            new_base = 0;
        }
        int64_t filled = fill_buffer(p);
        if (filled == 0 && buf_available(p) == 0) {
            *complete = 0; *phase = current_phase; *refill_code = ORC_EOF;
            return;
        }
        int c, code;
        scan_record(p->ptr + new_base, p->end - new_base, o, &current_phase, &c, &code);
        if (c) {
            *complete = 1; *phase = current_phase; *refill_code = code;
            return;
        }
    }
}

/* _find_and_consume_ref_record, parser.mojo:311-379 */
static int find_and_consume(orc_parser* p, orc_view* out) {
    if (buf_available(p) == 0) (void)compact_and_fill(p);
    if (!orc_parser_has_more(p)) {
        snprintf(p->errmsg, sizeof p->errmsg, "EOF");
        return ORC_EOF;
    }
    int64_t base = p->head;
    rec_offsets o = {0, 0, 0, 0, 0};
    int phase = 0, complete = 0, parse_code = ORC_OK;
    scan_record(p->ptr + base, p->end - base, &o, &phase, &complete, &parse_code);
    if (parse_code != ORC_OK) {
        /* parser.mojo:332-338: record/line numbers are +1, position is the record start */
        int64_t rn = p->line_number / 4 + 1, ln = p->line_number + 1;
        format_parse_error(p->errmsg, sizeof p->errmsg, orc_message_for_code(parse_code), rn, ln,
                           orc_parser_stream_position(p), p->ptr + base,
                           record_snippet_len(p->end - base, &o));
        return parse_code;
    }
    if (!complete) {
        int refill_code = ORC_OK;
        next_ref_complete(p, base, &o, &phase, &complete, &refill_code);
        base = 0;
        if (refill_code == ORC_EOF && !complete) {
            snprintf(p->errmsg, sizeof p->errmsg, "EOF");
            return ORC_EOF;
        } else if (refill_code != ORC_OK) {
            if (refill_code == ORC_ID_NO_AT || refill_code == ORC_SEP_NO_PLUS ||
                refill_code == ORC_SEQ_QUAL_LEN_MISMATCH) {
                int64_t rn = p->line_number / 4 + 1, ln = p->line_number + 1;
                format_parse_error(p->errmsg, sizeof p->errmsg, orc_message_for_code(refill_code), rn,
                                   ln, orc_parser_stream_position(p), p->ptr + p->head,
                                   record_snippet_len(buf_available(p), &o));
            } else {
                format_refill_error(p->errmsg, sizeof p->errmsg, refill_code, phase, p->len,
                                    p->max_capacity);
            }
            return refill_code;
        }
        if (!complete) { /* parser.mojo:350-351: `raise Error()` with an empty message */
            p->errmsg[0] = 0;
            return ORC_OTHER;
        }
    }
    const uint8_t* view = p->ptr + p->head;
    const uint8_t* id_raw = view + o.header_start + 1;
    int64_t id_raw_len = o.seq_start - o.header_start - 2;
    int64_t s0, sl;
    strip_spaces(id_raw, id_raw_len, &s0, &sl);
    out->id = id_raw + s0;
    out->id_len = sl;
    out->seq = view + o.seq_start;
    out->seq_len = o.sep_start - o.seq_start - 1;
    out->qual = view + o.qual_start;
    out->qual_len = o.record_end - o.qual_start;
    out->rec_pos = orc_parser_stream_position(p);
    out->off[0] = o.header_start; out->off[1] = o.seq_start; out->off[2] = o.sep_start;
    out->off[3] = o.qual_start;   out->off[4] = o.record_end;
    out->id_pos = out->rec_pos + o.header_start + 1 + s0;

    int64_t to_consume = o.record_end + 1;
    int64_t lim = p->end - base;
    if (to_consume > lim) to_consume = lim;
    /* BufferedReader.consume clamps to available(), buffered.mojo:156-164 */
    if (to_consume > buf_available(p)) to_consume = buf_available(p);
    p->head += to_consume;
    p->line_number += 4;
    return ORC_OK;
}

/* next_view, parser.mojo:159-170 */
int orc_parser_next_view(orc_parser* p, orc_view* out) {
    int rc = find_and_consume(p, out);
    if (rc != ORC_OK) return rc;
    int code = validate_view(&p->cfg, out);
    if (code != ORC_OK) {
        uint8_t snip[1024];
        int64_t sn = validation_snippet(out, snip);
        format_validation_error(p->errmsg, sizeof p->errmsg, orc_message_for_code(code),
                                p->line_number / 4, snip, sn);
        return code;
    }
    return ORC_OK;
}

/* ------------------------------------------------------------------ FastqBatch */

void orc_batch_init(orc_batch* b) { memset(b, 0, sizeof(*b)); b->quality_offset = 33; }
void orc_batch_clear(orc_batch* b) {
    b->n = 0; b->id_bytes_len = 0; b->qual_bytes_len = 0; b->seq_bytes_len = 0;
}
void orc_batch_free(orc_batch* b) {
    free(b->id_bytes); free(b->qual_bytes); free(b->seq_bytes); free(b->id_ends); free(b->ends);
    memset(b, 0, sizeof(*b));
}
static void grow_bytes(uint8_t** p, int64_t* cap, int64_t need) {
    if (need <= *cap) return;
    int64_t nc = *cap ? *cap : 4096;
    while (nc < need) nc *= 2;
    *p = (uint8_t*)realloc(*p, (size_t)nc);
    *cap = nc;
}
/* FastqBatch.add(FastqView), record_batch.mojo:77-87: `_ends` accumulates QUALITY lengths */
void orc_batch_add(orc_batch* b, const orc_view* v) {
    grow_bytes(&b->qual_bytes, &b->cap_qual, b->qual_bytes_len + v->qual_len);
    memcpy(b->qual_bytes + b->qual_bytes_len, v->qual, (size_t)v->qual_len);
    b->qual_bytes_len += v->qual_len;
    grow_bytes(&b->seq_bytes, &b->cap_seq, b->seq_bytes_len + v->seq_len);
    memcpy(b->seq_bytes + b->seq_bytes_len, v->seq, (size_t)v->seq_len);
    b->seq_bytes_len += v->seq_len;
    grow_bytes(&b->id_bytes, &b->cap_id, b->id_bytes_len + v->id_len);
    memcpy(b->id_bytes + b->id_bytes_len, v->id, (size_t)v->id_len);
    b->id_bytes_len += v->id_len;
    if (b->n + 1 > b->cap_n) {
        int64_t nc = b->cap_n ? b->cap_n * 2 : 1024;
        b->id_ends = (int64_t*)realloc(b->id_ends, (size_t)nc * sizeof(int64_t));
        b->ends = (int64_t*)realloc(b->ends, (size_t)nc * sizeof(int64_t));
        b->cap_n = nc;
    }
    if (b->n == 0) {
        b->id_ends[0] = v->id_len;
        b->ends[0] = v->qual_len;
    } else {
        b->id_ends[b->n] = v->id_len + b->id_ends[b->n - 1];
        b->ends[b->n] = v->qual_len + b->ends[b->n - 1];
    }
    b->n++;
}

/* next_batch, parser.mojo:239-251 */
int orc_parser_next_batch(orc_parser* p, int64_t max_records, orc_batch* out) {
    int64_t limit = max_records ? max_records : p->cfg.batch_size;
    orc_batch_clear(out);
    out->quality_offset = 33;
    while (out->n < limit && orc_parser_has_more(p)) {
        orc_view v;
        int rc = orc_parser_next_view(p, &v);
        if (rc == ORC_OK) { orc_batch_add(out, &v); continue; }
        if (strncmp(p->errmsg, "EOF", 3) == 0) break; /* String(e).startswith(EOF) */
        return rc;
    }
    return ORC_OK;
}

int64_t orc_bench_run(const uint8_t* data, int64_t n, const orc_config* cfg, int mode,
                      int64_t* base_pairs) {
    orc_parser* p = orc_parser_new(data, n, cfg);
    int64_t records = 0, bp = 0;
    if (mode == 0) {
        orc_view v;
        while (orc_parser_next_view(p, &v) == ORC_OK) { records++; bp += v.seq_len; }
    } else {
        orc_batch b; orc_batch_init(&b);
        for (;;) {
            int rc = orc_parser_next_batch(p, cfg->batch_size, &b);
            if (rc != ORC_OK || b.n == 0) break;
            records += b.n;
            bp += b.seq_bytes_len;
        }
        orc_batch_free(&b);
    }
    if (base_pairs) *base_pairs = bp;
    orc_parser_free(p);
    return records;
}

/* ==================================================================== consumers of a FastqBatch (CPU twins)
 * Test infrastructure like the rest of this file: the reference's example consumers restated on the CPU, so that the device-side
 * consumers (blazeseq_amd/csrc/bzq_consumers.hpp) have a checker and bench.py's pipeline_mode has a host figure beside it. */

/* examples/nw_gpu/kernels.mojo:21-89 (nw_kernel): two DP rows, match +1, mismatch -1, gap -1; a query or reference longer than
 * 256 scores 0 (48-50); an empty query leaves the first row: the score is -ref_len. */
int32_t orc_nw_score(const uint8_t* ref, int64_t ref_len, const uint8_t* q, int64_t q_len) {
    if (q_len > 256 || ref_len > 256) return 0;
    int32_t row0[257], row1[257];
    int32_t* prev = row0; int32_t* curr = row1;
    for (int64_t i = 0; i <= ref_len; ++i) prev[i] = -(int32_t)i;          /* 60-62 */
    for (int64_t j = 1; j <= q_len; ++j) {                                 /* 65-86 */
        curr[0] = -(int32_t)j;
        const uint8_t qb = q[j - 1];
        for (int64_t i = 1; i <= ref_len; ++i) {
            int32_t best = prev[i - 1] + (ref[i - 1] == qb ? 1 : -1);
            const int32_t del = prev[i] - 1, ins = curr[i - 1] - 1;
            if (del > best) best = del;
            if (ins > best) best = ins;
            curr[i] = best;
        }
        int32_t* t = prev; prev = curr; curr = t;
    }
    return prev[ref_len];                                                  /* 88 */
}

/* The reference's GPU use case on the host (examples/nw_gpu/execution.mojo:100-130: next_batch -> consumer): the streaming parser in
 * batches(cfg->batch_size) mode and, per batch, (a) the NW score of every record against `ref` (scores summed into *score_sum so
 * that the work cannot be optimised away and two runs can be compared) and (b) the per-position quality distribution
 * counts[p * 128 + v] += 1 for every record's quality byte v at position p < max_pos (bytes >= 128 count in bin 127; the v0.1
 * quality_distribution example, CHANGELOG.md:73).  Returns the records parsed. */
int64_t orc_pipeline_run(const uint8_t* data, int64_t n, const orc_config* cfg, const uint8_t* ref, int64_t ref_len,
                         int64_t max_pos, uint64_t* counts, int64_t* score_sum, int64_t* base_pairs) {
    orc_parser* p = orc_parser_new(data, n, cfg);
    int64_t records = 0, bp = 0, ss = 0;
    orc_batch b; orc_batch_init(&b);
    for (;;) {
        int rc = orc_parser_next_batch(p, cfg->batch_size, &b);
        if (rc != ORC_OK || b.n == 0) break;
        for (int64_t r = 0; r < b.n; ++r) {
            const int64_t q0 = r ? b.ends[r - 1] : 0, q1 = b.ends[r];
            ss += orc_nw_score(ref, ref_len, b.seq_bytes + q0, q1 - q0);
            const int64_t m = (q1 - q0) < max_pos ? (q1 - q0) : max_pos;
            for (int64_t k = 0; k < m; ++k) {
                const uint8_t v = b.qual_bytes[q0 + k];
                counts[k * 128 + (v < 128 ? v : 127)] += 1;
            }
        }
        records += b.n;
        bp += b.seq_bytes_len;
    }
    orc_batch_free(&b);
    if (score_sum) *score_sum = ss;
    if (base_pairs) *base_pairs = bp;
    orc_parser_free(p);
    return records;
}

/* ==================================================================== flat path */

typedef struct {
    int64_t w;      /* stream offset of buffer[0]  (= _stream_position) */
    int64_t end;    /* absolute end of buffered data (= _stream_position + _end = reader position) */
    int64_t cap;    /* _len */
    int is_eof;
    int64_t N;
} win_state;

/* _fill_buffer on indices */
static int64_t win_fill(win_state* s) {
    if (s->is_eof) return 0;
    int64_t space = s->cap - (s->end - s->w);
    if (space == 0) return 0;
    int64_t amt = s->N - s->end;
    if (amt > space) amt = space;
    if (amt < 0) amt = 0;
    s->end += amt;
    if (amt == 0) s->is_eof = 1;
    return amt;
}

static void flat_reserve(orc_flat* f, int64_t* cap, int64_t need) {
    if (need <= *cap) return;
    int64_t nc = *cap ? *cap * 2 : 1024;
    while (nc < need) nc *= 2;
#define RS(a) f->a = (int64_t*)realloc(f->a, (size_t)nc * sizeof(int64_t))
    RS(header_start); RS(seq_start); RS(sep_start); RS(qual_start); RS(record_end);
    RS(id_start); RS(id_len); RS(ends); RS(id_ends);
#undef RS
    *cap = nc;
}

void orc_flat_free(orc_flat* f) {
    free(f->header_start); free(f->seq_start); free(f->sep_start); free(f->qual_start);
    free(f->record_end); free(f->id_start); free(f->id_len); free(f->ends); free(f->id_ends);
    free(f->seq_bytes); free(f->qual_bytes); free(f->id_bytes);
    memset(f, 0, sizeof(*f));
}

int orc_flat_parse(const uint8_t* data, int64_t n, const orc_config* cfg, int is_eof, orc_flat* f) {
    memset(f, 0, sizeof(*f));
    int64_t rcap = 0, cap_seq = 0, cap_qual = 0, cap_id = 0;
    win_state s;
    s.w = 0; s.end = 0; s.cap = cfg->buffer_capacity; s.is_eof = 0; s.N = n;
    if (is_eof) (void)win_fill(&s); /* BufferedReader.__init__ */

    /* total newline count (used by the shard-stitch tests) */
    {
        int64_t c = 0;
        const uint8_t* q = data; const uint8_t* e = data + n;
        while (q < e) {
            const uint8_t* h = (const uint8_t*)memchr(q, C_NEWLINE, (size_t)(e - q));
            if (!h) break;
            c++; q = h + 1;
        }
        f->n_newlines = c;
    }

    int64_t head = 0, line_number = 0;
    int64_t S = 0, Q = 0, I = 0;
    f->term_code = ORC_OK;
    f->term_record = -1;
    for (;;) {
        if (is_eof) {
            /* parser.mojo:314-317 */
            if (head == s.end) { s.w = head; (void)win_fill(&s); }
            if (head == s.end && s.is_eof) {
                f->term_code = ORC_EOF; snprintf(f->term_msg, sizeof f->term_msg, "EOF");
                break;
            }
        } else if (head >= n) {
            break;
        }
        /* newline rank: the k-th newline after `head` ends line k of this record (Q1) */
        int64_t nlp[4] = {-1, -1, -1, -1}; int k = 0;
        {
            int64_t pos = head;
            while (k < 4 && pos < n) {
                const uint8_t* h = (const uint8_t*)memchr(data + pos, C_NEWLINE, (size_t)(n - pos));
                if (!h) break;
                nlp[k++] = (int64_t)(h - data);
                pos = nlp[k - 1] + 1;
            }
        }
        int accepted_without_newline = 0;
        int64_t record_end;
        if (!is_eof) {
            if (k < 4) break; /* incomplete tail: carried over by the caller */
            record_end = nlp[3];
            /* chunk mode has no window to replay, but the refusal of a record that cannot fit the reader's buffer does
             * not depend on where the window sits (parser.mojo:484-492: a record longer than the capacity -- or, with
             * growth, than buffer_max_capacity -- never completes), so it is reported here exactly like at EOF */
            {
                const int64_t rlen = record_end - head + 1;
                if (!cfg->buffer_growth_enabled && rlen > cfg->buffer_capacity) {
                    f->term_code = ORC_BUFFER_EXCEEDED;
                    format_refill_error(f->term_msg, sizeof f->term_msg, ORC_BUFFER_EXCEEDED, 0, cfg->buffer_capacity,
                                        cfg->buffer_max_capacity);
                    f->term_record = f->n_records;
                    break;
                }
                if (cfg->buffer_growth_enabled && rlen > cfg->buffer_max_capacity) {
                    f->term_code = ORC_BUFFER_AT_MAX;
                    format_refill_error(f->term_msg, sizeof f->term_msg, ORC_BUFFER_AT_MAX, 0, cfg->buffer_max_capacity,
                                        cfg->buffer_max_capacity);
                    f->term_record = f->n_records;
                    break;
                }
            }
        } else {
            int complete = (k == 4 && nlp[3] < s.end);
            if (!complete) {
                /* _next_ref_complete replayed on indices (parser.mojo:451-522) */
                int term = 0;
                for (;;) {
                    int64_t avail = s.end - head;
                    int found_in_win = 0;
                    for (int i = 0; i < k; ++i) if (nlp[i] < s.end) found_in_win++;
                    if (avail < s.cap && s.is_eof) {
                        if (found_in_win == 3) {
                            int all_blank = 1;
                            for (int64_t i = nlp[2] + 1; i < s.end; ++i) {
                                uint8_t b = data[i];
                                if (b != C_NEWLINE && b != C_CR && b != ' ' && b != '\t') { all_blank = 0; break; }
                            }
                            if (all_blank) { f->term_code = ORC_OTHER; f->term_msg[0] = 0; term = 1; }
                            else accepted_without_newline = 1;
                        } else {
                            f->term_code = ORC_UNEXPECTED_EOF;
                            format_refill_error(f->term_msg, sizeof f->term_msg, ORC_UNEXPECTED_EOF,
                                                found_in_win, s.cap, cfg->buffer_max_capacity);
                            term = 1;
                        }
                        break;
                    }
                    if (head == s.w) {
                        if (!cfg->buffer_growth_enabled) {
                            f->term_code = ORC_BUFFER_EXCEEDED;
                            format_refill_error(f->term_msg, sizeof f->term_msg, ORC_BUFFER_EXCEEDED, 0,
                                                s.cap, cfg->buffer_max_capacity);
                            term = 1; break;
                        }
                        if (s.cap >= cfg->buffer_max_capacity) {
                            f->term_code = ORC_BUFFER_AT_MAX;
                            format_refill_error(f->term_msg, sizeof f->term_msg, ORC_BUFFER_AT_MAX, 0,
                                                s.cap, cfg->buffer_max_capacity);
                            term = 1; break;
                        }
                        int64_t growth = s.cap < cfg->buffer_max_capacity - s.cap
                                             ? s.cap : cfg->buffer_max_capacity - s.cap;
                        s.cap += growth;
                        if (s.cap > cfg->buffer_max_capacity) s.cap = cfg->buffer_max_capacity;
                    } else {
                        s.w = head;
                    }
                    int64_t filled = win_fill(&s);
                    if (filled == 0 && s.end - head == 0) {
                        f->term_code = ORC_EOF; snprintf(f->term_msg, sizeof f->term_msg, "EOF");
                        term = 1; break;
                    }
                    if (k == 4 && nlp[3] < s.end) break; /* rescan completes */
                }
                if (term) { f->term_record = f->n_records; break; }
            }
            record_end = accepted_without_newline ? s.end : nlp[3];
        }
        int64_t header_start = head, seq_start = nlp[0] + 1, sep_start = nlp[1] + 1,
                qual_start = nlp[2] + 1;
        if (!accepted_without_newline) {
            /* _validate_fastq_structure (skipped for the EOF-without-newline record, Q4) */
            int code = ORC_OK;
            if (data[header_start] != C_READ_HEADER) code = ORC_ID_NO_AT;
            else if (data[sep_start] != C_QUAL_HEADER) code = ORC_SEP_NO_PLUS;
            else if ((sep_start - seq_start - 1) != (record_end - qual_start)) code = ORC_SEQ_QUAL_LEN_MISMATCH;
            if (code != ORC_OK) {
                int64_t sn = record_end + 1 - header_start;
                int64_t viewlen = (is_eof ? s.end : n) - header_start;
                if (sn > viewlen) sn = viewlen;
                if (sn > 200) sn = 200;
                format_parse_error(f->term_msg, sizeof f->term_msg, orc_message_for_code(code),
                                   line_number / 4 + 1, line_number + 1, header_start,
                                   data + header_start, sn);
                f->term_code = code; f->term_record = f->n_records;
                break;
            }
        }
        orc_view v;
        int64_t s0, sl;
        strip_spaces(data + header_start + 1, seq_start - header_start - 2, &s0, &sl);
        v.id = data + header_start + 1 + s0; v.id_len = sl;
        v.seq = data + seq_start;            v.seq_len = sep_start - seq_start - 1;
        v.qual = data + qual_start;          v.qual_len = record_end - qual_start;
        line_number += 4;
        {
            int code = validate_view(cfg, &v);
            if (code != ORC_OK) {
                uint8_t snip[1024];
                int64_t sn = validation_snippet(&v, snip);
                format_validation_error(f->term_msg, sizeof f->term_msg, orc_message_for_code(code),
                                        line_number / 4, snip, sn);
                f->term_code = code; f->term_record = f->n_records;
                /* the record was consumed before validation ran (parser.mojo:375-377) */
                break;
            }
        }
        /* deliver */
        int64_t r = f->n_records;
        flat_reserve(f, &rcap, r + 1);
        f->header_start[r] = header_start; f->seq_start[r] = seq_start; f->sep_start[r] = sep_start;
        f->qual_start[r] = qual_start;     f->record_end[r] = record_end;
        f->id_start[r] = header_start + 1 + s0; f->id_len[r] = sl;
        grow_bytes(&f->seq_bytes, &cap_seq, S + v.seq_len);
        memcpy(f->seq_bytes + S, v.seq, (size_t)v.seq_len); S += v.seq_len;
        grow_bytes(&f->qual_bytes, &cap_qual, Q + v.qual_len);
        memcpy(f->qual_bytes + Q, v.qual, (size_t)v.qual_len); Q += v.qual_len;
        grow_bytes(&f->id_bytes, &cap_id, I + v.id_len);
        memcpy(f->id_bytes + I, v.id, (size_t)v.id_len); I += v.id_len;
        f->ends[r] = Q; f->id_ends[r] = I;
        f->n_records = r + 1;
        head = accepted_without_newline ? s.end : record_end + 1;
        if (head > n) head = n;
    }
    f->seq_bytes_len = S; f->qual_bytes_len = Q; f->id_bytes_len = I;
    f->consumed = head;
    return f->term_code;
}

/* =================================================================== generator */

/* _build_gc_biased_base_lut, utils.mojo:707-733 */
static void build_base_lut(double gc_bias, uint8_t lut[8]) {
    float g = (float)gc_bias; /* the reference takes Float32 */
    int gc_slots = (int)(g * 8.0f + 0.5f);
    if (gc_slots < 0) gc_slots = 0;
    if (gc_slots > 8) gc_slots = 8;
    int at_slots = 8 - gc_slots, n = 0;
    for (int k = 0; k < gc_slots; ++k) lut[n++] = (k % 2 == 0) ? 'G' : 'C';
    for (int k = 0; k < at_slots; ++k) lut[n++] = (k % 2 == 0) ? 'A' : 'T';
}

static int num_digits_for(int64_t num_reads) { /* utils.mojo:880-882 */
    if (num_reads <= 1) return 1;
    char t[32];
    return snprintf(t, sizeof t, "%lld", (long long)(num_reads - 1));
}

int64_t orc_generate_synthetic(int64_t num_reads, int64_t first, int64_t count, int64_t min_len,
                               int64_t max_len, int64_t min_phred, int64_t max_phred,
                               const char* schema, double gc_bias, uint8_t* out, int64_t cap) {
    if (num_reads <= 0) return 0;
    if (min_len < 0 || max_len < 0 || min_phred < 0 || max_phred < 0 || min_len > max_len ||
        min_phred > max_phred)
        return -1; /* _validate_synthetic_fastq_args, utils.mojo:682-704 */
    uint8_t lower, upper, offset;
    (void)orc_schema_from_name(schema, &lower, &upper, &offset);
    uint8_t lut[8];
    build_base_lut(gc_bias, lut);
    int nd = num_digits_for(num_reads);
    const int64_t q_start = max_phred, q_range = max_phred - min_phred, noise_amp = q_range / 6 + 1;
    const uint64_t MASK = 0x7FFFFFFFFFFFFFFFull;
    int64_t w = 0;
#define PUT(b) do { if (w < cap && out) out[w] = (uint8_t)(b); w++; } while (0)
    for (int64_t i = first; i < first + count && i < num_reads; ++i) {
        /* _build_synthetic_fastq_record, utils.mojo:736-828 */
        int64_t read_len = (max_len == min_len) ? min_len
                                                : min_len + (int64_t)(((uint64_t)i * 31u + 7u) % (uint64_t)(max_len - min_len + 1));
        char hdr[48];
        int hl = snprintf(hdr, sizeof hdr, "@read_%0*lld\n", nd, (long long)i);
        for (int j = 0; j < hl; ++j) PUT(hdr[j]);
        uint64_t st = ((uint64_t)i * 6364136223846793005ull + 1442695040888963407ull) & MASK;
        for (int64_t b = 0; b < read_len; ++b) {
            st = (st * 6364136223846793005ull + 1442695040888963407ull) & MASK;
            PUT(lut[(st >> 33) % 8]);
        }
        PUT('\n'); PUT('+'); PUT('\n');
        uint64_t qr = ((uint64_t)i * 2654435761ull + 1013904223ull) & MASK;
        int64_t lm1 = read_len - 1;
        for (int64_t pp = 0; pp < read_len; ++pp) {
            int64_t mean = (lm1 == 0) ? q_start : q_start - (q_range * pp + lm1 / 2) / lm1;
            qr = (qr * 1664525ull + 1013904223ull) & MASK;
            int64_t noise_raw = (int64_t)((qr >> 17) % (uint64_t)(2 * noise_amp + 1));
            int64_t phred = mean + noise_raw - noise_amp;
            if (phred < min_phred) phred = min_phred; else if (phred > max_phred) phred = max_phred;
            int64_t a = (int64_t)offset + phred;
            if (a < lower) a = lower; else if (a > upper) a = upper;
            PUT(a);
        }
        PUT('\n');
    }
#undef PUT
    return w;
}

/* compute_num_reads_for_size, utils.mojo:640-678 */
int64_t orc_compute_num_reads_for_size(int64_t target, int64_t min_len, int64_t max_len) {
    if (target <= 0) return 0;
    int64_t avg = (min_len + max_len) / 2;
    int64_t est = target / (15 + 2 * avg + 4);
    if (est <= 0) return 0;
    int nd = 1;
    if (est > 1) { char t[32]; nd = snprintf(t, sizeof t, "%lld", (long long)(est - 1)); }
    int64_t header = 6 + nd + 1;
    return target / (header + 2 * avg + 4);
}
