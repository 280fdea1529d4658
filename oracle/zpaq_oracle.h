/* oracle/zpaq_oracle.h — CPU oracle for the ZPAQ block-decompression hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under zpaqsharp_amd/ (the product) may
 * include, link or dlopen this.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it, and only as the checker.
 *
 * It is a plain-C restatement of the interpreter ("NOJIT") semantics of the
 * reference (mnadareski/ZPAQSharp, a transliteration of libzpaq 7.12):
 *   Predictor.cs:39-172,245-567   Decoder.cs:32-158   ZPAQL.cs:112-207,1010-1303
 *   PostProcessor.cs:27-86        Decompresser.cs:29-194
 *   Encoder.cs:26-103             Compressor.cs:27-299  (mirror, for fixtures)
 * with the transcription defects listed in SURVEY.md §8a corrected to the
 * intended libzpaq form that the same files keep in comments.
 *
 * Parity pin: the reference cannot be compiled or run (not valid C#, no .NET
 * toolchain), and ships no tests or vectors.  The oracle is pinned by the
 * constants the reference itself carries (table checksums Predictor.cs:71-77,
 * state table StateTable.cs:21-149, tag hashes Decompresser.cs:34-43, model
 * bytecodes Compressor.cs:48-74) — see tests/test_oracle_pins.py.
 * Compatibility with archives written by upstream zpaq: parity unpinned.
 */
#ifndef ZPAQ_ORACLE_H
#define ZPAQ_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- pins ------------------------------------------------------------- */
/* Fills the three checksums of Predictor.cs:71-77 / SURVEY §4; returns 0. */
int zo_table_pins(uint32_t *stsum, uint32_t *sqsum, uint32_t *sns_crc32);
/* Copies tables out for device-side comparison.  squash: 4096 u16,
 * stretch: 32768 i16, dt: 1024 i32, dt2k: 256 i32, ns: 1024 u8. */
void zo_tables(uint16_t *squash, int16_t *stretch, int32_t *dt, int32_t *dt2k,
               uint8_t *ns);

/* ---- decompression (Decompresser.cs) ---------------------------------- */
typedef struct zo_dec zo_dec;
zo_dec *zo_dec_new(void);
void zo_dec_free(zo_dec *);
const char *zo_dec_error(const zo_dec *);            /* "" if none */
void zo_dec_set_input(zo_dec *, const uint8_t *p, size_t n);
/* Each returns <0 on error() (message in zo_dec_error). */
int zo_dec_find_block(zo_dec *, double *mem);         /* 1 found, 0 EOF */
int zo_dec_find_filename(zo_dec *, char *buf, size_t cap); /* 1 seg, 0 end */
int zo_dec_read_comment(zo_dec *, char *buf, size_t cap);
/* decompress(n): n<0 = to end of segment.  Appends to out[*len..cap).
 * returns 1 = more data in segment, 0 = segment done. */
int zo_dec_decompress(zo_dec *, long n, uint8_t *out, size_t cap, size_t *len);
int zo_dec_read_segment_end(zo_dec *, uint8_t sha1[21]);
size_t zo_dec_tell(const zo_dec *);                   /* input bytes consumed */
/* hcomp()/pcomp() read-back (Decompresser.cs:60-63,155-158) */
long zo_dec_hcomp(zo_dec *, uint8_t *buf, size_t cap);
long zo_dec_pcomp(zo_dec *, uint8_t *buf, size_t cap);
/* Trace: fold (p<<1|y) of every modelled bit into a CRC-32; every 4096 bits
 * one digest is appended.  Lets a GPU divergence be localised. */
void zo_dec_set_trace(zo_dec *, uint32_t *digests, size_t cap, size_t *n);
/* Final coder/predictor state of the current block (after a segment). */
void zo_dec_state(const zo_dec *, uint32_t st[8]); /* low,high,curr,c8,hmap4,h0,h1,h2 */

/* One call: LibZPAQ.decompress (LibZPAQ.cs:65-79).  Returns plaintext length
 * or -1 (message in err). */
long zo_decompress(const uint8_t *in, size_t n, uint8_t *out, size_t cap,
                   char *err, size_t errcap);

/* ---- compression mirror (Encoder.cs / Compressor.cs), fixtures only ---- */
typedef struct zo_enc zo_enc;
zo_enc *zo_enc_new(uint8_t *out, size_t cap);
void zo_enc_free(zo_enc *);
const char *zo_enc_error(const zo_enc *);
size_t zo_enc_tell(const zo_enc *);
int zo_enc_write_tag(zo_enc *);
/* hdr = full block header bytes as stored in the stream: hsize[2] hh hm ph pm
 * n COMP 0 HCOMP 0 (i.e. what Compressor.startBlock(bytes) takes). */
int zo_enc_start_block(zo_enc *, const uint8_t *hdr, size_t hdrlen);
int zo_enc_start_segment(zo_enc *, const char *filename, const char *comment);
/* pcomp==NULL/len 0 → PASS (0); else PROG (1, len lo, len hi, bytes). */
int zo_enc_post_process(zo_enc *, const uint8_t *pcomp, size_t len);
/* test hook: open the segment's coder without writing a post-processor selector (the caller encodes it) */
int zo_enc_begin_raw(zo_enc *);
int zo_enc_compress(zo_enc *, const uint8_t *data, size_t n);
int zo_enc_end_segment(zo_enc *, const uint8_t sha1[20] /* or NULL */);
int zo_enc_end_block(zo_enc *);

/* ---- helpers ----------------------------------------------------------- */
void zo_sha1(const uint8_t *p, size_t n, uint8_t out[20]);
void zo_e8e9(uint8_t *buf, size_t n);                 /* LibZPAQ.cs:372-384 */
/* Test guard (not reference behaviour): ZPAQL instructions one run() may execute before the oracle reports
 * "ZPAQL instruction budget exhausted"; 0 = unlimited. */
void zo_set_zpaql_budget(uint64_t per_run);
/* Runs a ZPAQL program stand-alone as PCOMP: feeds in[0..n) then EOF. */
long zo_run_pcomp(const uint8_t *pcomp, size_t plen, int ph, int pm,
                  const uint8_t *in, size_t n, uint8_t *out, size_t cap);
/* Block-scanner rolling hash after feeding bytes (Decompresser.cs:34-45). */
void zo_tag_hash(const uint8_t *p, size_t n, uint32_t h[4]);

#ifdef __cplusplus
}
#endif
#endif
