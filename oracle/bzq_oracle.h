/*
 * bzq_oracle.h -- TEST INFRASTRUCTURE ONLY (not shipped, never on the product path).
 *
 * CPU restatement of the BlazeSeq FASTQ batch-parse path, used as the parity
 * checker for the HIP kernels.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library.
 *
 * Parity status: PINNED BY SOURCE + the reference's own known-answer tests.
 * The reference (Mojo 0.26.2) can be neither compiled nor imported in this
 * image, so every function below cites the reference file:line it follows
 * (paths relative to /root/reference), and tests/test_oracle_reference_kats.py
 * replays the literal expectations of the reference's tests
 * (tests/fastq/test_parser.mojo, test_record_batch.mojo, test_fastq_record.mojo,
 * tests/test_error_context.mojo, tests/test_python_bindings.py) against it.
 * An independent C parser from the reference tree (benchmark/fastq-parser/
 * kseq_runner, built into oracle/_ref/) cross-checks record/base counts.
 *
 * Two restatements that must agree on every input:
 *   orc_parser_*  streaming: BufferedReader window + _find_and_consume_ref_record
 *                 + _next_ref_complete + _scan_record, line by line.
 *   orc_flat_*    flat: whole input in memory, "newline rank" formulation
 *                 (the spec the GPU kernels implement), plus an O(1)-per-record
 *                 replay of the BufferedReader window arithmetic so that
 *                 terminal errors (BUFFER_EXCEEDED / UNEXPECTED_EOF / accepted
 *                 last record without newline) come out identical.
 */
#ifndef BZQ_ORACLE_H
#define BZQ_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* FastxErrorCode, blazeseq/errors.mojo:33-68 */
enum {
    ORC_OK = 0,
    ORC_ID_NO_AT = 1,
    ORC_SEP_NO_PLUS = 2,
    ORC_SEQ_QUAL_LEN_MISMATCH = 3,
    ORC_ASCII_INVALID = 4,
    ORC_QUALITY_OUT_OF_RANGE = 5,
    ORC_EOF = 6,
    ORC_UNEXPECTED_EOF = 7,
    ORC_BUFFER_EXCEEDED = 8,
    ORC_BUFFER_AT_MAX = 9,
    ORC_OTHER = 10
};

/* ParserConfig (fastq/parser.mojo:33-74) + QualitySchema (fastq/quality_schema.mojo:9-31) */
typedef struct orc_config {
    int64_t buffer_capacity;      /* DEFAULT_CAPACITY = 256 KiB, CONSTS.mojo:26 */
    int64_t buffer_max_capacity;  /* MAX_CAPACITY = 2^30, CONSTS.mojo:28 */
    int32_t buffer_growth_enabled;
    int32_t check_ascii;
    int32_t check_quality;
    uint8_t q_lower, q_upper, q_offset, _pad;
    int32_t simd_width;           /* 0: scalar branch only (inclusive [LOWER,UPPER]);
                                     W>0: reproduce the SIMD body of
                                     Validator._validate_quality_range (record.mojo:90-97), Q9 */
    int32_t batch_size;           /* DEFAULT_BATCH_SIZE = 4096, CONSTS.mojo:31 */
} orc_config;

void orc_config_default(orc_config* c);
/* _parse_schema, utils.mojo:612-637.  Returns 1 if the name is known, 0 if it fell back to generic. */
int orc_schema_from_name(const char* name, uint8_t* lower, uint8_t* upper, uint8_t* offset);
/* _message_for_code, errors.mojo:71-90 */
const char* orc_message_for_code(int code);

/* ------------------------------------------------------------------ streaming */

typedef struct orc_parser orc_parser;

/* FastqView (fastq/record.mojo:431-472) plus the positions the GPU path is checked against. */
typedef struct orc_view {
    const uint8_t* id;
    const uint8_t* seq;
    const uint8_t* qual;
    int64_t id_len, seq_len, qual_len;
    int64_t rec_pos;   /* absolute stream position of header_start */
    int64_t off[5];    /* RecordOffsets relative to the record start: header_start(=0), seq_start,
                          sep_start, qual_start, record_end (utils.mojo:37-93) */
    int64_t id_pos;    /* absolute stream position of the stripped id */
} orc_view;

/* FastqBatch (fastq/record_batch.mojo:19-87): three byte columns + two inclusive running sums. */
typedef struct orc_batch {
    int64_t n;
    uint8_t* id_bytes;   int64_t id_bytes_len;
    uint8_t* qual_bytes; int64_t qual_bytes_len;
    uint8_t* seq_bytes;  int64_t seq_bytes_len;
    int64_t* id_ends;
    int64_t* ends;
    uint8_t quality_offset;   /* always 33 on the parser path (parser.mojo:243, Q10) */
    /* capacities (internal) */
    int64_t cap_n, cap_id, cap_qual, cap_seq;
} orc_batch;

/* FastqParser.__init__ over a MemoryReader (parser.mojo:89-145, io/readers.mojo:140-223,
 * io/buffered.mojo:137-149).  `data` is borrowed and must outlive the parser. */
orc_parser* orc_parser_new(const uint8_t* data, int64_t n, const orc_config* cfg);
void orc_parser_free(orc_parser* p);
int orc_parser_has_more(const orc_parser* p);                 /* parser.mojo:155-157 */
/* next_view, parser.mojo:159-170.  Returns a FastxErrorCode: 0 on success, 6 on EOF, anything
 * else is the raise; orc_parser_error() then holds the exact String(e). */
int orc_parser_next_view(orc_parser* p, orc_view* out);
/* next_batch, parser.mojo:239-251.  max_records==0 -> parser batch_size.  On a non-EOF error the
 * batch is left in whatever state the raise found it (the reference drops it). */
int orc_parser_next_batch(orc_parser* p, int64_t max_records, orc_batch* out);
const char* orc_parser_error(const orc_parser* p);
int64_t orc_parser_line_number(const orc_parser* p);
int64_t orc_parser_stream_position(const orc_parser* p);      /* buffered.mojo:177-182 */
int64_t orc_parser_capacity(const orc_parser* p);

void orc_batch_init(orc_batch* b);
void orc_batch_clear(orc_batch* b);
void orc_batch_free(orc_batch* b);
void orc_batch_add(orc_batch* b, const orc_view* v);          /* record_batch.mojo:77-87 */

/* ------------------------------------------------------------------------ flat */

typedef struct orc_flat {
    int64_t n_records;      /* records delivered before the terminal event */
    /* RecordOffsets as absolute positions, one entry per delivered record */
    int64_t* header_start;
    int64_t* seq_start;
    int64_t* sep_start;
    int64_t* qual_start;
    int64_t* record_end;
    int64_t* id_start;      /* stripped id */
    int64_t* id_len;
    /* FastqBatch columns over ALL delivered records (not split into batches; `ends`/`id_ends`
     * are cumulative over the whole input -- per-batch values are obtained by rebasing) */
    uint8_t* seq_bytes;  int64_t seq_bytes_len;
    uint8_t* qual_bytes; int64_t qual_bytes_len;
    uint8_t* id_bytes;   int64_t id_bytes_len;
    int64_t* ends;
    int64_t* id_ends;
    /* terminal event */
    int32_t term_code;      /* ORC_EOF for a clean end, otherwise the raise */
    int32_t _pad;
    int64_t term_record;    /* 0-based index of the record that raised (n_records), -1 for EOF */
    int64_t consumed;       /* bytes consumed (position after the last delivered record) */
    int64_t n_newlines;     /* total '\n' in the input (for the multi-GPU stitch tests) */
    char term_msg[1400];
} orc_flat;

/* Parse the whole buffer.  is_eof != 0: `data` is the complete stream and the terminal event is
 * classified exactly as the streaming parser would (window replay).  is_eof == 0: `data` is a
 * chunk that starts at a record start; only complete records are delivered, structure/validation
 * errors still terminate, and an incomplete tail is left for the caller to carry over
 * (term_code = ORC_OK, consumed = start of the tail). */
int orc_flat_parse(const uint8_t* data, int64_t n, const orc_config* cfg, int is_eof, orc_flat* out);
void orc_flat_free(orc_flat* f);

/* --------------------------------------------------------------------- generator */

/* generate_synthetic_fastq_buffer, utils.mojo:831-917 (+ helpers 707-828).  Returns the number of
 * bytes the full output has; writes at most `cap` of them into `out` (pass NULL/0 to size).
 * Records [first, first+count) of a num_reads-record file are produced (first=0,count=num_reads
 * for the whole file). */
int64_t orc_generate_synthetic(int64_t num_reads, int64_t first, int64_t count, int64_t min_len,
                               int64_t max_len, int64_t min_phred, int64_t max_phred,
                               const char* schema, double gc_bias, uint8_t* out, int64_t cap);
/* compute_num_reads_for_size, utils.mojo:640-678 */
int64_t orc_compute_num_reads_for_size(int64_t target, int64_t min_len, int64_t max_len);

/* ----------------------------------------------------------------- CPU baseline */

/* Timed loop for bench.py's cpu_baseline leg: the streaming restatement over `data`, mode 0 =
 * views(), mode 1 = batches(batch_size) (build + drop each FastqBatch like the reference's
 * runner, benchmark/throughput/run_throughput_memory_blazeseq.mojo:59-83).  Returns records
 * parsed; *base_pairs gets the sequence bytes. */
int64_t orc_bench_run(const uint8_t* data, int64_t n, const orc_config* cfg, int mode,
                      int64_t* base_pairs);
/* consumers of a FastqBatch on the CPU (examples/nw_gpu/kernels.mojo:21-89; CHANGELOG.md:73): see bzq_oracle.c */
int32_t orc_nw_score(const uint8_t* ref, int64_t ref_len, const uint8_t* q, int64_t q_len);
int64_t orc_pipeline_run(const uint8_t* data, int64_t n, const orc_config* cfg, const uint8_t* ref, int64_t ref_len,
                         int64_t max_pos, uint64_t* counts, int64_t* score_sum, int64_t* base_pairs);

#ifdef __cplusplus
}
#endif
#endif
