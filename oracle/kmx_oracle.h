/*
 * kmx_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the COMBINE-lab/kmers hot path (reference = the Rust
 * crate mounted at /root/reference; every function below cites the file:line
 * it follows).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the shipped HIP path never does.
 *
 * Parity pin: the reference cannot be compiled here (no rustc/cargo), so the
 * oracle is pinned against every known-answer test the reference's own inline
 * test modules hold for this path (tests/golden/reference_kats.json, checked by
 * tests/test_oracle_golden.py).  Semantics the reference does not define
 * (B>1 canonical order / hash, bucket function, synthetic generator) are
 * marked BUILD-DEFINED and are "parity unpinned" by construction.
 */
#ifndef KMX_ORACLE_H
#define KMX_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KMO_OK 0
#define KMO_E_INVALID_BASE 1 /* reference panics: naive_impl/mod.rs:35 */
#define KMO_E_TOO_LONG 2     /* reference panics: naive_impl/kmer.rs:211-213,236-238 */
#define KMO_E_ARG 3

#define KMO_INVALID_CODE UINT64_MAX /* naive_impl/mod.rs:48 */

/* ---- naive_impl::prelude (src/naive_impl/mod.rs:19-90) ---- */
uint64_t kmo_encode_binary_u8(uint8_t c);            /* mod.rs:40-50 */
int kmo_encode_binary(uint8_t c, uint64_t *out);     /* mod.rs:27-37 (panic -> KMO_E_INVALID_BASE) */
uint64_t kmo_encode_complement_binary_u8(uint8_t c); /* mod.rs:54-65 */
uint64_t kmo_complement_base(uint64_t b);            /* mod.rs:81-84 */
int kmo_is_valid_nuc(uint64_t b);                    /* mod.rs:87-89 */

/* ---- naive_impl::Kmer (src/naive_impl/kmer.rs) ---- */
typedef struct {
    uint8_t k;
    uint64_t data;
} kmo_kmer;

uint64_t kmo_mask_table(unsigned k);                                   /* kmer.rs:584-618 (entry 32 == 0) */
kmo_kmer kmo_kmer_from_u64(uint64_t data, uint8_t k);                  /* kmer.rs:45-48 */
int kmo_kmer_from_bytes(const uint8_t *s, size_t len, kmo_kmer *out);  /* kmer.rs:234-251 */
uint64_t kmo_kmer_append_base(kmo_kmer *km, uint64_t c);               /* kmer.rs:98-102 */
uint64_t kmo_kmer_prepend_base(kmo_kmer *km, uint64_t c);              /* kmer.rs:91-95 */
uint64_t kmo_kmer_append_base_u8(kmo_kmer *km, uint8_t c);             /* kmer.rs:84-88 */
uint64_t kmo_kmer_prepend_base_u8(kmo_kmer *km, uint8_t c);            /* kmer.rs:77-81 */
uint64_t kmo_revcomp_word(uint64_t w, uint8_t k);                      /* kmer.rs:124-147 */
kmo_kmer kmo_kmer_to_reverse_complement(kmo_kmer km);                  /* kmer.rs:124-136 */
int kmo_kmer_cmp(kmo_kmer a, kmo_kmer b);                              /* derived Ord kmer.rs:6 (k, then data) */
int kmo_kmer_eq(kmo_kmer a, kmo_kmer b);                               /* kmer.rs:12-16 */
int kmo_kmer_is_canonical(kmo_kmer km);                                /* kmer.rs:55-58 */
kmo_kmer kmo_kmer_to_canonical(kmo_kmer km);                           /* kmer.rs:68-74 */
int kmo_sub_kmer_word(uint64_t word, size_t k, size_t pos, size_t width, uint64_t *out); /* kmer.rs:156-162 */
size_t kmo_kmer_to_string(kmo_kmer km, char *out);                     /* kmer.rs:196-207 (lower-case) */

/* ---- naive_impl::CanonicalKmer (src/naive_impl/canonical_kmer.rs) ---- */
typedef struct {
    kmo_kmer fw;
    kmo_kmer rc;
} kmo_canonical_kmer;

#define KMO_NO_MATCH 0       /* canonical_kmer.rs:8-12 */
#define KMO_IDENTITY_MATCH 1
#define KMO_TWIN_MATCH 2

kmo_canonical_kmer kmo_ck_blank_of_size(uint8_t k);                       /* canonical_kmer.rs:22-29 */
kmo_canonical_kmer kmo_ck_from_u64(uint64_t data, uint8_t k);             /* canonical_kmer.rs:42-51 */
kmo_canonical_kmer kmo_ck_from_kmer(kmo_kmer km);                         /* canonical_kmer.rs:164-172 */
int kmo_ck_from_bytes(const uint8_t *s, size_t len, kmo_canonical_kmer *out); /* canonical_kmer.rs:188-196 */
void kmo_ck_swap(kmo_canonical_kmer *ck);                                 /* canonical_kmer.rs:62-64 */
int kmo_ck_is_fw_canonical(const kmo_canonical_kmer *ck);                 /* canonical_kmer.rs:67-69 */
uint64_t kmo_ck_append_base(kmo_canonical_kmer *ck, uint64_t b);          /* canonical_kmer.rs:90-94 */
uint64_t kmo_ck_prepend_base(kmo_canonical_kmer *ck, uint64_t b);         /* canonical_kmer.rs:97-101 */
uint64_t kmo_ck_append_base_u8(kmo_canonical_kmer *ck, uint8_t c);        /* canonical_kmer.rs:72-78 */
uint64_t kmo_ck_prepend_base_u8(kmo_canonical_kmer *ck, uint8_t c);       /* canonical_kmer.rs:81-87 */
uint64_t kmo_ck_get_canonical_word(const kmo_canonical_kmer *ck);         /* canonical_kmer.rs:113-119 */
int kmo_ck_get_word_equivalency(const kmo_canonical_kmer *ck, uint64_t other); /* canonical_kmer.rs:152-161 */

/* ---- naive_impl::CanonicalKmerIterator (src/naive_impl/canonical_kmer_iterator.rs) ---- */
typedef struct {
    const uint8_t *seq;
    int32_t seq_len;
    kmo_canonical_kmer km; /* value_pair.km */
    int32_t pos;           /* value_pair.pos */
    int invalid;
    int32_t last_invalid;
    int32_t k;
} kmo_iter;

void kmo_iter_from_u8_slice(kmo_iter *it, const uint8_t *s, size_t len, uint8_t k); /* :72-83 */
int kmo_iter_exhausted(const kmo_iter *it);                                         /* :89-91 */
int kmo_iter_inc(kmo_iter *it);                                                     /* :94-101 */
int kmo_iter_inc_by(kmo_iter *it, size_t count);                                    /* :104-111 */

/* ---- naive_impl::hash (src/naive_impl/hash.rs) ---- */
uint64_t kmo_lex_hash_u64(uint64_t word, size_t hasher_k); /* hash.rs:60-71; Hash for Kmer = write_u64(data), :4-8 */
/* hash_one with std's DefaultHasher / RandomState (hash.rs:10-20): SipHash-c-d (paper: Aumasson, Bernstein 2012); DefaultHasher = 1-3 */
uint64_t kmo_siphash(unsigned c, unsigned d, uint64_t k0, uint64_t k1, const uint8_t *msg, size_t len);
uint64_t kmo_siphash13_u64(uint64_t word, uint64_t k0, uint64_t k1);

/* ---- encoding::Naive / Xor10 (src/encoding/naive.rs, xor10.rs) ----
 * Words [P;B] are handled as their flat little-endian bit array (bit_field 0.10:
 * flat bit i lives in word i/BITS, bit i%BITS; pinned by naive.rs:297-445), i.e.
 * `nbytes` = B * sizeof(P) bytes, byte i/8 bit i%8.                          */
uint8_t kmo_nuc2internal(uint8_t nuc);                       /* naive.rs:14-16 */
uint8_t kmo_rev_encoding(uint8_t enc);                       /* naive.rs:28-39 */
uint8_t kmo_naive_nuc2bits(uint8_t enc, uint8_t nuc);        /* naive.rs:78-85 */
uint8_t kmo_naive_bits2nuc(uint8_t enc, uint8_t bits);       /* naive.rs:88-95 */
uint8_t kmo_naive_complement(uint8_t enc, uint8_t bits);     /* naive.rs:98-109 */
int kmo_naive_encode(uint8_t enc, const uint8_t *seq, size_t len, uint8_t *array, size_t nbytes);  /* naive.rs:116-124 */
void kmo_naive_decode(uint8_t enc, const uint8_t *array, size_t nbytes, uint8_t *seq_out);         /* naive.rs:126-136 */
int kmo_naive_rev_comp(uint8_t enc, size_t K, uint8_t *array, size_t nbytes);                      /* naive.rs:138-154 */
int kmo_xor10_encode(const uint8_t *seq, size_t len, uint8_t *array, size_t nbytes);               /* xor10.rs:52-60 */
void kmo_xor10_decode(const uint8_t *array, size_t nbytes, uint8_t *seq_out);                      /* xor10.rs:62-72 */
int kmo_xor10_rev_comp(size_t K, uint8_t *array, size_t nbytes);  /* xor10.rs:86-103 (B>1 swap loop) */
uint64_t kmo_xor10_rev_comp_b1_quirk(uint64_t word, unsigned p_bits); /* xor10.rs:75-85 (NOT a reverse complement) */

/* ---- generic kmer::Kmer<P,K,B> (src/kmer.rs) ---- */
size_t kmo_word_for_k(size_t p_bytes, size_t K);                                /* kmer.rs:67-69 */
uint64_t kmo_generic_get(const uint8_t *array, size_t nbytes, size_t index);    /* kmer.rs:46-48 */
int kmo_generic_get_prefix(const uint8_t *array, size_t nbytes, unsigned p_bits, size_t len, uint64_t *out); /* kmer.rs:50-52: bits 0..=2*len */
void kmo_bitmer_to_bytes(uint64_t mer, size_t len, uint8_t *out);               /* kmer.rs:71-91 */

/* ---- batch drivers (the shapes the GPU path is checked against) ---- */
typedef struct {
    uint64_t n_valid;    /* number of windows the iterator yields */
    uint64_t sum_canon;  /* wrapping sum of get_canonical_word() (benches/simple_benchmark.rs:14-22 shape) */
    uint64_t xor_hash;   /* xor of LexHasher(hasher_k)(canonical word); 0 when hasher_k == 0 */
    uint64_t sum_fw;     /* wrapping sum of get_fw_word() (== compute_naive on valid input) */
} kmo_summary;

/* reads: n_reads contiguous reads. offsets == NULL -> uniform read_len, read r at reads + r*read_len;
 * else read r = reads[offsets[r] .. offsets[r+1]).                                                     */
int kmo_canonical_reduce(const uint8_t *reads, size_t n_reads, size_t read_len, const uint64_t *offsets,
                         uint8_t k, size_t hasher_k, kmo_summary *out);
/* dense per-window materialisation: window slot for read r, pos p is (win_off(r) + p) where
 * uniform: win_off(r) = r*(read_len-k+1); ragged: caller passes win_offsets[r].
 * flags: bit0 = valid (iterator yields this pos), bit1 = fw is canonical (fw<rc). Invalid slots are zeroed. */
int kmo_canonical_windows(const uint8_t *reads, size_t n_reads, size_t read_len, const uint64_t *offsets,
                          const uint64_t *win_offsets, uint8_t k,
                          uint64_t *out_fw, uint64_t *out_rc, uint64_t *out_canon, uint8_t *out_flags);
/* benches/simple_benchmark.rs:14-22 compute_naive: sum over b.windows(K) of Kmer::from(x).into_u64();
 * returns KMO_E_INVALID_BASE where the reference would panic. */
int kmo_compute_naive(const uint8_t *b, size_t len, size_t K, uint64_t *out_sum);
/* benches/simple_benchmark.rs:36-44 rc_naive shape but keeping the result: sum of min(fw, rc) per window */
int kmo_compute_naive_canonical(const uint8_t *b, size_t len, size_t K, uint64_t *out_sum);

/* ---- SeqVector (src/naive_impl/seq_vector.rs) ----
 * `words`: little-endian u64 array, base i at flat bits [2i,2i+1] (RawVector is LSB-first; From<&[u8]> stores
 * Kmer::from(32-base chunk).into_u64() per word, seq_vector.rs:230-242). */
/* push_chars (seq_vector.rs:141-161): first len%32 bases, then 32-base chunks, each via Kmer::from + push_int.
 * n_before = bases already stored; returns KMO_E_INVALID_BASE (with *bad_index) where Kmer::from would panic. */
int kmo_seqvec_push_chars(uint64_t *words, size_t n_before, const uint8_t *bytes, size_t n, size_t *bad_index);
/* get_kmer_u64 (seq_vector.rs:96-99): RawVector::int(pos*2, k*2); KMO_E_ARG where the reference asserts (pos >= len)
 * or where the field would leave the vector */
int kmo_seqvec_get_kmer_u64(const uint64_t *words, size_t n_bases, size_t pos, size_t k, uint64_t *out);
/* String::from(&SeqVector) (seq_vector.rs:171-182) */
void kmo_seqvec_to_bytes(const uint64_t *words, size_t n_bases, uint8_t *out);
/* iter_kmers over slice [start,end) (seq_vector.rs:64-71,341-357): end-start-k+1 forward words; returns the count */
size_t kmo_seqvec_iter_kmers(const uint64_t *words, size_t n_bases, size_t start, size_t end, size_t k, uint64_t *out);
/* canonical scan of reads stored back to back (read r = slice [r*L,(r+1)*L)): every k-mer of every slice via
 * get_kmer_u64 + Kmer::to_canonical (kmer.rs:68-74); same summary as kmo_canonical_reduce on the decoded letters */
int kmo_seqvec_canonical_reduce(const uint64_t *words, size_t n_reads, size_t read_len, uint8_t k, size_t hasher_k,
                                kmo_summary *out);

/* ---- minimizers ----
 * hasher: 0 = identity (hash == l-mer word: write_u64(data), hash.rs:4-8), else LexHasher(hasher_k) (hash.rs:60-71);
 * the std RandomState/SipHash hashers the reference also accepts have no pinned outputs. */
uint64_t kmo_mm_hash(uint64_t lmer, size_t hasher_k);
/* Kmer::minimizer_word (kmer.rs:170-192): leftmost minimum over the k-w+1 sub-words (strict `<` against u64::MAX) */
int kmo_minimizer_word(uint64_t word, size_t k, size_t width, size_t hasher_k, uint64_t *out_mmer, size_t *out_offset);
/* SeqVecMinimizerIter (seq_vector/minimizers.rs:39-141) with its monotone deque, over slice [start, end) */
#define KMO_DQ_CAP 96
typedef struct {
    uint64_t lmer;
    size_t pos;
    uint64_t hash;
} kmo_dqmer;                        /* minimizers.rs:8-13 */
typedef struct {
    kmo_dqmer dq[KMO_DQ_CAP];       /* VecDeque: [head, head+len) */
    size_t head, len;
    size_t k, w, curr_km_i;
    const uint64_t *words;
    size_t n_bases, start, slice_len;
    size_t hasher_k;
} kmo_mmiter;
void kmo_mmiter_enqueue(kmo_mmiter *it, kmo_dqmer m);                                   /* enqueue_dqmer, :60-80 */
int kmo_mmiter_new(kmo_mmiter *it, const uint64_t *words, size_t n_bases, size_t start, size_t end,
                   size_t k, size_t w, size_t hasher_k);                               /* new, :97-122 (assert len >= k) */
int kmo_mmiter_next(kmo_mmiter *it, uint64_t *out_word, size_t *out_pos);              /* next, :127-141; 0 = exhausted */
/* batch: every read (slice [r*L,(r+1)*L)) -> (word, pos) per k-mer, slot r*(L-k+1)+i */
int kmo_seqvec_minimizers(const uint64_t *words, size_t n_reads, size_t read_len, size_t k, size_t w, size_t hasher_k,
                          uint64_t *out_word, uint32_t *out_pos);

/* ---- BUILD-DEFINED extensions (no reference counterpart) ---- */
typedef struct {
    uint64_t n_valid;
    uint64_t sum_lo, sum_hi; /* wrapping sums of word 0 / word 1 of the canonical [u64;2] */
    uint64_t xor_lo, xor_hi; /* xor of the 2K-bit lexicographic-rank hash words */
} kmo_summary2;
/* k in [33,64]: [u64;2] k-mers, rolling = A.3 carried across words, order = 2K-bit little-endian integer */
int kmo_canonical_reduce2(const uint8_t *reads, size_t n_reads, size_t read_len, const uint64_t *offsets,
                          uint8_t k, int with_hash, kmo_summary2 *out);
int kmo_canonical_windows2(const uint8_t *reads, size_t n_reads, size_t read_len, const uint64_t *offsets,
                           const uint64_t *win_offsets, uint8_t k,
                           uint64_t *out_fw2, uint64_t *out_rc2, uint64_t *out_canon2, uint8_t *out_flags);
/* FASTA/FASTQ record splitting -- BUILD-DEFINED: the reference has no parser, so nothing pins
 * this; the Python restatement (oracle.fastx_parse, written from the prose spec at the top of kmers_amd/csrc/kmx_fastx.hip with bytes.split) and
 * this byte-at-a-time state machine are checked against each other.
 * format: 1 = FASTQ (strict 4-line records: line i is a read iff i % 4 == 1), 2 = FASTA ('>' lines start a record,
 * all other lines up to the next '>' line are its sequence), 0 = by the first byte ('@' / '>').  Lines end at '\n';
 * every '\r' on a sequence line is dropped; the last line may lack its '\n'.  bases/offsets may be NULL (count only).
 * Returns KMO_E_ARG if the text does not start with '@' (FASTQ) / '>' (FASTA). */
int kmo_fastx_parse(const uint8_t *text, size_t n, unsigned format, uint8_t *bases, uint64_t *offsets,
                    size_t *n_reads, size_t *n_bases);
/* synthetic reads: word w of the stream = splitmix64(seed + w); base j of the 32 in it = "ACGT"[(z>>2j)&3] */
uint64_t kmo_splitmix64(uint64_t x);
void kmo_gen_reads(uint64_t seed, uint64_t first_byte, uint8_t *out, size_t nbytes);
/* bucket of a hash value: top log2_buckets bits of the 32-bit sum lo32(h) * 0x9E3779B1 + hi32(h) * 0x85EBCA6B (log2_buckets <= 30) */
uint64_t kmo_bucket_of(uint64_t h, unsigned log2_buckets);
int kmo_histogram(const uint8_t *reads, size_t n_reads, size_t read_len, const uint64_t *offsets,
                  uint8_t k, size_t hasher_k, unsigned log2_buckets, uint64_t *counts);

#ifdef __cplusplus
}
#endif
#endif
