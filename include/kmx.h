/*
 * kmx.h -- C ABI of libkmx: the MI355X (gfx950) implementation of the streaming
 * 2-bit k-mer encode / canonicalise / hash hot path of COMBINE-lab/kmers.
 *
 * This header is the drop-in boundary.  The reference is a pure-Rust crate with
 * scalar per-k-mer calls; a GPU is a drop-in only at batch granularity, so every
 * entry point below is the batch restatement of one reference interface (cited
 * as file:line under /root/reference) with identical per-element results.
 * INTEGRATION.md shows the Rust `extern "C"` binding a maintainer would add.
 *
 * Conventions
 *   - plain C types only: pointers + sizes, no C++/torch types.
 *   - every `d_*` pointer is a DEVICE pointer valid on the ctx's GPU (from
 *     kmx_malloc, hipMalloc, or a torch CUDA tensor's data_ptr()).
 *   - all calls are asynchronous on the ctx's HIP stream unless stated; results
 *     are visible after kmx_ctx_synchronize() (or stream sync by the owner).
 *   - return value: KMX_OK or a KMX_E_* code; nothing ever unwinds across the ABI.
 *   - k-mer word layout (reference A.2): base i at bits [2i,2i+1], first base
 *     lowest; multi-word k-mers are little-endian by u64 word.
 */
#ifndef KMX_H
#define KMX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KMX_VERSION 2

/* ---- status codes (reference panics become codes; SURVEY 8b "Errors") ---- */
#define KMX_OK 0
#define KMX_E_ARG 1          /* null / inconsistent argument */
#define KMX_E_K_RANGE 2      /* k outside the supported domain of the call */
#define KMX_E_HIP 3          /* HIP runtime error; see kmx_last_error() */
#define KMX_E_INVALID_BASE 4 /* naive_impl::encode_binary panic (src/naive_impl/mod.rs:35) */
#define KMX_E_TOO_LONG 5     /* Kmer::from len>32 panic (src/naive_impl/kmer.rs:211-213,236-238); bit_field OOB in Encoding::encode */
#define KMX_E_NOMEM 6

/* ---- hashers (src/naive_impl/hash.rs) ---- */
#define KMX_HASH_NONE 0     /* no hash folded into the summary */
#define KMX_HASH_LEX 1      /* LexHasher{k}: hash.rs:22-72 */
#define KMX_HASH_IDENTITY 2 /* impl Hash for Kmer feeds write_u64(data) (hash.rs:4-8) into an identity hasher */

/* ---- MatchType (src/naive_impl/canonical_kmer.rs:7-12) ---- */
#define KMX_NO_MATCH 0
#define KMX_IDENTITY_MATCH 1
#define KMX_TWIN_MATCH 2

/* ---- per-window flags written by kmx_canonical_windows ---- */
#define KMX_WIN_VALID 1u        /* the iterator yields this position (no non-ACGTacgt byte in the window) */
#define KMX_WIN_FW_CANONICAL 2u /* CanonicalKmer::is_fw_canonical: fw < rc (canonical_kmer.rs:67-69) */

typedef struct kmx_ctx kmx_ctx;

/* A batch of reads resident in device memory.
 * offsets == NULL: uniform layout, read r = d_bases[r*read_len .. (r+1)*read_len).
 * offsets != NULL: ragged layout, read r = d_bases[offsets[r] .. offsets[r+1]) (n_reads+1 device u64); read_len may
 *   then carry an upper bound of the read lengths (0 = unknown): a bound <= 160 selects the smaller, faster frame of the
 *   tiled kernels, and the tighter it is the fewer windows a lane carries (150 bp reads: 7 % faster at k = 31 with 150
 *   than with 160 or 0).  Reads that are all of ONE length up to the bound (no bound: 160) -- untrimmed FASTQ -- are recognised on the
 *   device and take the uniform kernels whatever the bound says (round 5) in kmx_canonical_reduce (13 <= k <= 31, 16-byte aligned
 *   d_bases, a bound of at most 256, no KMX_REDUCE_SUM_FW) and in kmx_canonical_reduce2 (16-byte aligned d_bases, a bound of at most
 *   256); kmx_histogram, kmx_canonical_windows(2), kmx_minimizers and the Sum-fw reduce take such reads through their ragged kernels.
 *   It is only a hint -- tiles with a longer read take the exact per-read path, and the
 *   histogram's work buffer, sized from it, overflows into exact (slow) global atomics.  A bound ABOVE 256 says "long reads"
 *   (PacBio / ONT reads, contigs): kmx_canonical_reduce (13 <= k <= 31), kmx_canonical_reduce2, kmx_canonical_windows and
 *   kmx_canonical_windows2 (16-byte aligned d_bases) then cut every read into overlapping segments on the device and scan those (two
 *   host round trips: the batch's first and last offset, the number of segments; the segment arrays live in the context's work
 *   buffer); with 0 or a bound <= 256 a long read costs its tile the per-read path.  (kmx_canonical_reduce2 cuts ragged reads with
 *   any bound above 160 this way: the two-word ragged kernel holds 160 bases; with a bound of 161..256 it first reads the batch's
 *   first and last offset back -- one host round trip -- and skips the cut when they say "one length, k .. bound" -- untrimmed reads at
 *   the bound or below it; the device-side gate then confirms it, and the lane-per-read kernel counts if it does not.)
 *   EVERY call that takes one of these routes SYNCHRONISES the context's stream on the host (once or twice) and cannot be captured
 *   in a HIP graph; all other scan calls only enqueue work.
 * d_bases must be a device pointer whenever n_reads > 0, also when every read is empty (KMX_E_ARG otherwise). */
typedef struct {
    const uint8_t *d_bases;
    uint64_t n_reads;
    uint32_t read_len;
    const uint64_t *d_offsets;
} kmx_reads;
/* Limits: a single read is shorter than 2^31 bases (the iterator's positions are i32 in the reference as well,
 * canonical_kmer_iterator.rs:15); offsets and totals are 64-bit.  A ragged read of 2^31 bases or more -- e.g. a whole
 * chromosome out of kmx_fastx_parse's FASTA mode -- is DIAGNOSED, not scanned: the scan kernels skip it (it contributes no
 * window) and raise a sticky flag on the context, the next kmx_ctx_synchronize returns KMX_E_ARG with the text in
 * kmx_last_error and clears the flag; kmx_reads_length_range returns KMX_E_ARG for such a batch right away.  Cut such
 * records into overlapping pieces first (k - 1 bases of overlap, or hand the bytes over as uniform reads: reads longer
 * than 256 bases are scanned as overlapping segments by the tiled kernel, every k from 13 to 64).  Uniform reads may start at any byte address; ragged reads need a
 * 16-byte-aligned d_bases for the tiled kernels (any address is served, by the per-read kernel). */

/* Result of a streaming reduce pass (device-resident, 32 bytes).
 * == what a consumer loop over CanonicalKmerIterator accumulates
 * (src/naive_impl/canonical_kmer_iterator.rs:42-116; benches/simple_benchmark.rs:14-22 `.sum()` shape). */
typedef struct {
    uint64_t n_valid;   /* windows yielded */
    uint64_t sum_canon; /* wrapping sum of get_canonical_word() (canonical_kmer.rs:113-119) */
    uint64_t xor_hash;  /* xor of hash_one(hasher, canonical kmer) (hash.rs:10-20); 0 for KMX_HASH_NONE */
    uint64_t sum_fw;    /* wrapping sum of get_fw_word() == compute_naive on valid input; 0 unless KMX_REDUCE_SUM_FW */
} kmx_summary;

/* [u64;2] k-mers (k in 33..64), BUILD-DEFINED extension of the above */
typedef struct {
    uint64_t n_valid;
    uint64_t sum_lo, sum_hi;
    uint64_t xor_lo, xor_hi;
} kmx_summary2;

#define KMX_REDUCE_SUM_FW 1u /* also accumulate sum_fw */

/* ------------------------------------------------------------ context ---- */
int kmx_ctx_create(int device, kmx_ctx **out);                 /* owns a new non-blocking stream */
int kmx_ctx_create_on_stream(int device, void *hip_stream, kmx_ctx **out); /* borrows the caller's hipStream_t (NULL = default stream) */
void kmx_ctx_destroy(kmx_ctx *ctx);
int kmx_ctx_synchronize(kmx_ctx *ctx);                         /* also reports what the asynchronous scans could not: KMX_E_ARG after a read of >= 2^31 bases was skipped */
int kmx_ctx_device(const kmx_ctx *ctx);
/* The context owns ONE grow-only device work buffer (bucket-id streams of kmx_histogram above 2^14 buckets, chunk prefixes of
 * kmx_fastx_parse, the segment arrays of reads longer than 256 bases: 16-24 bytes per segment of <= 226 windows); the histogram
 * processes its reads in as many chunks as it takes, a batch of long reads whose segment arrays do not fit under a limit set
 * here takes the per-read kernels instead (same results, a tenth of the rate).  Its size is chosen per call -- an eighth of the device
 * memory, at most half of what is free -- unless a limit is set here (bytes; 0 = automatic again).  A server that shares the
 * device caps it; a small limit forces the chunked paths. */
int kmx_ctx_set_work_buffer_limit(kmx_ctx *ctx, size_t bytes);
/* what the context holds now, and how often the buffer has been (re)allocated since kmx_ctx_create (either pointer may be NULL) */
int kmx_ctx_work_buffer_info(const kmx_ctx *ctx, size_t *bytes_held, uint64_t *n_allocations);
const char *kmx_strerror(int status);
const char *kmx_last_error(const kmx_ctx *ctx); /* text of the last HIP failure on this ctx */
int kmx_version(void);

/* device memory helpers so a host language needs no HIP binding of its own */
int kmx_malloc(kmx_ctx *ctx, size_t nbytes, void **d_out);
int kmx_free(kmx_ctx *ctx, void *d_ptr);
int kmx_memcpy_h2d(kmx_ctx *ctx, void *d_dst, const void *h_src, size_t nbytes); /* synchronous */
int kmx_memcpy_d2h(kmx_ctx *ctx, void *h_dst, const void *d_src, size_t nbytes); /* synchronous */
int kmx_memset(kmx_ctx *ctx, void *d_dst, int value, size_t nbytes);

/* ------------------------------------------- the streaming hot path ---- */

/* Replaces: CanonicalKmerIterator::{from_u8_slice,find_next,inc,get} over every read
 * (src/naive_impl/canonical_kmer_iterator.rs:42-116) + CanonicalKmer::append_base /
 * get_canonical_word (canonical_kmer.rs:90-94,113-119) + Kmer::{append,prepend}_base
 * (kmer.rs:91-102) + encode_binary_u8 (mod.rs:40-50) + optional hash_one (hash.rs:10-20).
 * k in [1,31] (the reference's MASK_TABLE[32]==0, kmer.rs:617, breaks its own rolling at k=32).
 * hasher: KMX_HASH_*; hasher_k: LexHasher's k (usually == k), ignored otherwise.
 * d_out is OVERWRITTEN with this batch's summary. */
int kmx_canonical_reduce(kmx_ctx *ctx, const kmx_reads *reads, uint32_t k, uint32_t hasher, uint32_t hasher_k,
                         uint32_t flags, kmx_summary *d_out);

/* The same call with its answer in HOST memory when it returns (round 6): what a caller that reduces one small
 * batch at a time -- the reference's iterator over one read set, canonical_kmer_iterator.rs:42-116 -- pays is
 * the launch and the wait, not the bytes.  Uniform reads of up to 256 bases, k in [9,31], hasher NONE or
 * LEX(hasher_k == k): ONE kernel launch whose last block writes the summary to pinned host words this call
 * watches (1e5 reads of 150 bases: ~25 us against ~70 us for kmx_canonical_reduce + kmx_memcpy_d2h); a batch
 * with invalid bytes adds the sweep and a copy.  Every other input: kmx_canonical_reduce + the copy.
 * `reads` in device memory as always; *h_out is OVERWRITTEN.  Synchronous on the context's stream. */
int kmx_canonical_reduce_host(kmx_ctx *ctx, const kmx_reads *reads, uint32_t k, uint32_t hasher, uint32_t hasher_k,
                              uint32_t flags, kmx_summary *h_out);

/* Same pass, materialising per-window state: slot(read r, pos p) = win_off(r) + p with
 * win_off(r) = r*(read_len-k+1) (uniform) or d_win_offsets[r] (ragged; n_reads+1 device u64,
 * exclusive prefix sum of max(len-k+1,0)).  Any of the four outputs may be NULL.
 * d_fw/d_rc/d_canon: get_fw_word / get_rc_word / get_canonical_word of the iterator state at
 * that pos (canonical_kmer.rs:113-139); invalid slots are written as 0 with flags 0. */
int kmx_canonical_windows(kmx_ctx *ctx, const kmx_reads *reads, const uint64_t *d_win_offsets, uint32_t k,
                          uint64_t *d_fw, uint64_t *d_rc, uint64_t *d_canon, uint8_t *d_flags);

/* BUILD-DEFINED [u64;2] variants, k in [33,64] (word_for_k::<u64,K>() == 2, src/kmer.rs:67-69):
 * rolling = kmer.rs:91-102 carried across words, order = 2K-bit little-endian integer,
 * hash = 2-bit-group reversal of the 2K-bit value.  Outputs are 2 u64 per slot. */
int kmx_canonical_reduce2(kmx_ctx *ctx, const kmx_reads *reads, uint32_t k, uint32_t with_hash, kmx_summary2 *d_out);
int kmx_canonical_windows2(kmx_ctx *ctx, const kmx_reads *reads, const uint64_t *d_win_offsets, uint32_t k,
                           uint64_t *d_fw2, uint64_t *d_rc2, uint64_t *d_canon2, uint8_t *d_flags);

/* Per-bucket occupancy of hash(canonical k-mer): d_counts[bucket] += 1 for every yielded window;
 * bucket = (uint32_t)(lo32(hash) * 0x9E3779B1 + hi32(hash) * 0x85EBCA6B) >> (32 - log2_buckets)  (BUILD-DEFINED bucket
 * function: the top bits of a 32-bit multiplicative mix of the two halves; log2_buckets <= 30).
 * d_counts (2^log2_buckets device u64) is ACCUMULATED into; the caller zeroes it and, across
 * GPUs, all-reduces it (RCCL ncclSum/uint64).
 * With 2^15..2^22 buckets the reads (uniform or ragged) go through a grow-only work buffer owned by the context (sized to
 * the call: 3 bytes per window, at most an eighth of the device memory -- 8 GiB if that is more -- and at most half of
 * what is free; kmx_ctx_set_work_buffer_limit overrides): growing it synchronises the stream once. */
int kmx_histogram(kmx_ctx *ctx, const kmx_reads *reads, uint32_t k, uint32_t hasher, uint32_t hasher_k,
                  uint32_t log2_buckets, uint64_t *d_counts);

/* Deterministic synthetic reads (BUILD-DEFINED; the reference bench input is unseeded,
 * benches/simple_benchmark.rs:59-65): byte g of the stream = "ACGT"[(splitmix64(seed + g/32) >> 2*(g%32)) & 3].
 * Writes nbytes bytes for stream positions [first_byte, first_byte+nbytes). */
int kmx_gen_reads(kmx_ctx *ctx, uint64_t seed, uint64_t first_byte, uint8_t *d_out, uint64_t nbytes);

/* -------------------------------------- element-wise batch operations ---- */

/* naive_impl::Kmer::from(&[u8]) for n sequences of k bytes each, contiguous (kmer.rs:234-251).
 * Strict semantics: returns KMX_E_INVALID_BASE if any byte is not ACGTacgt (the reference panics,
 * mod.rs:35); *h_first_bad (host, may be NULL) receives the lowest offending byte index.
 * k in [1,32].  SYNCHRONOUS (it has to report the status). */
int kmx_kmers_from_bytes(kmx_ctx *ctx, const uint8_t *d_seqs, uint64_t n, uint32_t k, uint64_t *d_words,
                         uint64_t *h_first_bad);

/* Kmer::to_reverse_complement / get_reverse_complement_word (kmer.rs:124-147), k in [1,32] */
int kmx_revcomp_words(kmx_ctx *ctx, const uint64_t *d_in, uint64_t n, uint32_t k, uint64_t *d_out);

/* Kmer::to_canonical + is_canonical (kmer.rs:55-74): d_canon[i] = min(w, rc(w)); d_is_canonical[i] = (w <= rc(w)).
 * Either output may be NULL. */
int kmx_canonical_words(kmx_ctx *ctx, const uint64_t *d_in, uint64_t n, uint32_t k, uint64_t *d_canon,
                        uint8_t *d_is_canonical);

/* hash_one(&state, Kmer) with state = LexHasherState::new(hasher_k) (hash.rs:10-20,60-71) or identity */
int kmx_hash_words(kmx_ctx *ctx, const uint64_t *d_in, uint64_t n, uint32_t hasher, uint32_t hasher_k, uint64_t *d_out);

/* hash_one(&state, Kmer) with one of std's BuildHashers (hash.rs:10-20; the reference uses DefaultHasher at kmer.rs:546-557 and
 * RandomState at :564-575): std's DefaultHasher is SipHash-1-3 (one compression round, three finalisation rounds), `Hash for Kmer`
 * feeds it ONE write_u64(data) (hash.rs:4-8) -- so d_out[i] = SipHash-1-3(key0, key1; the 8 little-endian bytes of d_in[i]).
 * DefaultHasher::new() / BuildHasherDefault: key0 = key1 = 0; RandomState: its pair of random keys, which the caller holds.
 * The algorithm lives in Rust's standard library, not in the crate: restated from the SipHash paper (Aumasson, Bernstein 2012);
 * the restatement reproduces the paper's SipHash-2-4 test vectors through the same round function (tests/test_oracle_golden.py),
 * its 1-3 instance the first row of Rust's own SipHasher13 test table (library/core/tests/hash/sip.rs); no reference value for
 * 1-3 exists in the crate itself (its two tests check properties): parity is pinned to that extent. */
int kmx_hash_words_sip13(kmx_ctx *ctx, const uint64_t *d_in, uint64_t n, uint64_t key0, uint64_t key1, uint64_t *d_out);

/* CanonicalKmer::get_word_equivalency (canonical_kmer.rs:152-161): out[i] in KMX_{NO,IDENTITY,TWIN}_MATCH */
int kmx_match_words(kmx_ctx *ctx, const uint64_t *d_fw, const uint64_t *d_rc, const uint64_t *d_other, uint64_t n,
                    uint8_t *d_out);

/* CanonicalKmer::append_base / prepend_base on n independent (fw, rc) states, in place
 * (canonical_kmer.rs:90-101; Kmer::append_base/prepend_base kmer.rs:91-102).  d_bases are 2-bit
 * codes (A0 C1 G2 T3); d_dropped (may be NULL) receives the shifted-off base.  k in [1,31]. */
int kmx_ck_append_bases(kmx_ctx *ctx, uint64_t *d_fw, uint64_t *d_rc, const uint8_t *d_bases, uint64_t n, uint32_t k,
                        uint8_t *d_dropped);
int kmx_ck_prepend_bases(kmx_ctx *ctx, uint64_t *d_fw, uint64_t *d_rc, const uint8_t *d_bases, uint64_t n, uint32_t k,
                         uint8_t *d_dropped);

/* Encoding::encode for encoding::Naive (src/encoding/naive.rs:116-124) / Xor10 (xor10.rs:52-60):
 * n sequences of seq_len bytes each (contiguous) -> words_per_kmer u64 each, i.e.
 * Kmer::<u64,K,B>::new(seq, &enc).  enc_byte is the Naive discriminant (naive.rs:48-74);
 * Xor10 == 0x1B (Naive::ACTG).  No validity check (naive.rs:14-16 maps every byte).
 * KMX_E_TOO_LONG if seq_len > 32*words_per_kmer (bit_field would panic). */
int kmx_encode_kmers(kmx_ctx *ctx, const uint8_t *d_seqs, uint64_t n, uint32_t seq_len, uint8_t enc_byte,
                     uint32_t words_per_kmer, uint64_t *d_words);
/* Same for every length-k window of every read: the benches' `b.windows(K).map(Kmer::new)` shape
 * (benches/simple_benchmark.rs:24-34); slot layout as kmx_canonical_windows (uniform reads only). */
int kmx_encode_windows(kmx_ctx *ctx, const kmx_reads *reads, uint32_t k, uint8_t enc_byte, uint32_t words_per_kmer,
                       uint64_t *d_words);
/* Encoding::rev_comp::<K> (naive.rs:138-154; xor10.rs:86-103 B>1 loop): result base i =
 * complement(base K-1-i) for i<K, bits >= 2K unchanged.  K in [2, 32*words_per_kmer]. */
int kmx_encoding_rev_comp(kmx_ctx *ctx, const uint64_t *d_in, uint64_t n, uint32_t K, uint8_t enc_byte,
                          uint32_t words_per_kmer, uint64_t *d_out);
/* Encoding::decode (naive.rs:126-136): emits ALL 32*words_per_kmer letters per k-mer */
int kmx_encoding_decode(kmx_ctx *ctx, const uint64_t *d_in, uint64_t n, uint8_t enc_byte, uint32_t words_per_kmer,
                        uint8_t *d_seqs);

/* ---------------------------------------------------------------- decode / display direction ---- */
/* Kmer::sub_kmer_word(word, k, pos, width) (src/naive_impl/kmer.rs:156-162) for n words: (word >> 2*pos) & MASK_TABLE[width].
 * The reference asserts pos < k and pos + width <= k: KMX_E_ARG otherwise.  1 <= k <= 32 (width 32 would meet
 * MASK_TABLE[32] == 0, kmer.rs:617, and give 0 like the reference). */
int kmx_sub_kmer_words(kmx_ctx *ctx, const uint64_t *d_words, uint64_t n, uint32_t k, uint32_t pos, uint32_t width,
                       uint64_t *d_out);
/* String::from(Kmer) (src/naive_impl/kmer.rs:196-207, BASE_TABLE :24): n words -> n*k LOWER-case letters, base 0 first. */
int kmx_kmers_to_strings(kmx_ctx *ctx, const uint64_t *d_words, uint64_t n, uint32_t k, uint8_t *d_out);
/* kmer::bitmer_to_bytes(mer, len) (src/kmer.rs:71-91): n words -> n*len UPPER-case letters, base 0 first.  len <= 32. */
int kmx_bitmers_to_bytes(kmx_ctx *ctx, const uint64_t *d_mers, uint64_t n, uint32_t len, uint8_t *d_out);

/* ---------------------------------------------------------------- Encoding<P, B> for every utils::Data word type ----
 * src/utils.rs:4-24 implements Data for u8, u16, u32, u64 and u128; encoding::Naive is generic over it
 * (src/encoding/naive.rs:112-154; Xor10 == Naive::ACTG for u64 / u128, xor10.rs:50).  bit_field's BitArray puts flat bit i
 * into word i / BITS, bit i % BITS, so the little-endian byte image of a [P; B] is the same flat bit string for every P:
 * the arrays are passed as bytes, word_bits * words_per_kmer / 8 per k-mer, and P only decides the capacity
 * (KMX_E_TOO_LONG where bit_field would panic) and the number of letters decode() emits (word_bits * words_per_kmer / 2).
 * word_bits in {8, 16, 32, 64, 128}; at most 64 bytes per k-mer. */
int kmx_encode_kmers_p(kmx_ctx *ctx, const uint8_t *d_seqs, uint64_t n, uint32_t seq_len, uint8_t enc_byte,
                       uint32_t word_bits, uint32_t words_per_kmer, void *d_arrays);
int kmx_encoding_rev_comp_p(kmx_ctx *ctx, const void *d_in, uint64_t n, uint32_t K, uint8_t enc_byte, uint32_t word_bits,
                            uint32_t words_per_kmer, void *d_out);
int kmx_encoding_decode_p(kmx_ctx *ctx, const void *d_in, uint64_t n, uint8_t enc_byte, uint32_t word_bits,
                          uint32_t words_per_kmer, uint8_t *d_seqs);

/* ----------------------------------------------------------------------------------------------------------------
 * SeqVector -- the reference's 2-bit packed sequence container (src/naive_impl/seq_vector.rs).
 * Layout: base i at flat bits [2i, 2i+1] of a little-endian u64 word array, codes A0 C1 G2 T3 (it is built from
 * Kmer::from of 32-base chunks, seq_vector.rs:230-242); a vector of n bases owns ceil(n/32) words, bits past 2n are 0.
 * The caller owns `d_words`; reads stored back to back are slices [r*L, (r+1)*L) of one vector (SeqVector::slice).
 * -------------------------------------------------------------------------------------------------------------- */
/* SeqVector::push_chars (seq_vector.rs:141-161) / From<&[u8]> (:230-242, n_bases_before = 0): append `n` ASCII bases
 * to a vector that holds `n_bases_before`.  Strict like Kmer::from (kmer.rs:234-251 panics on a bad base):
 * KMX_E_INVALID_BASE with *h_first_bad = index into d_bytes of the first offending byte (the words are then
 * unspecified).  d_words must have room for ceil((n_bases_before + n)/32) words. */
int kmx_seqvec_push_chars(kmx_ctx *ctx, uint64_t *d_words, uint64_t n_bases_before, const uint8_t *d_bytes, uint64_t n,
                          uint64_t *h_first_bad);
/* String::from(&SeqVector) (seq_vector.rs:171-182): n upper-case letters */
int kmx_seqvec_to_bytes(kmx_ctx *ctx, const uint64_t *d_words, uint64_t n_bases, uint8_t *d_bytes);
/* SeqVector::get_kmer_u64(pos, k) (seq_vector.rs:96-99; get_base = k 1) for n positions; k in [1,32].
 * A position whose k-mer does not lie inside the vector (the reference asserts pos < len) gives KMX_E_ARG. */
int kmx_seqvec_get_kmers(kmx_ctx *ctx, const uint64_t *d_words, uint64_t n_bases, const uint64_t *d_pos, uint64_t n,
                         uint32_t k, uint64_t *d_out);
/* SeqVectorSlice::iter_kmers(k) (seq_vector.rs:64-71, 117-124) over the slice [start, end): end-start-k+1 forward
 * words (not canonicalised), in order. */
int kmx_seqvec_iter_kmers(kmx_ctx *ctx, const uint64_t *d_words, uint64_t n_bases, uint64_t start, uint64_t end,
                          uint32_t k, uint64_t *d_out);
/* kmx_canonical_reduce over reads held in a SeqVector: read r = slice [r*read_len, (r+1)*read_len).  Every 2-bit code
 * is a base, so every window counts.  0.25 B per base read from HBM instead of 1.  d_words 16-byte aligned for the
 * fast kernel (k in [13,31]); any k in [1,31] is served. */
int kmx_seqvec_canonical_reduce(kmx_ctx *ctx, const uint64_t *d_words, uint64_t n_reads, uint32_t read_len, uint32_t k,
                                uint32_t hasher, uint32_t hasher_k, uint32_t flags, kmx_summary *d_out);

/* ---------------------------------------------------------------- minimizers ----
 * hasher: KMX_HASH_LEX with hasher_k (LexHasherState::new(hasher_k): the k of the hasher is independent of the l-mer
 * length, minimizers.rs:240 uses 6 for 3-mers) or KMX_HASH_IDENTITY.  std's RandomState has no pinned outputs. */
/* Kmer::minimizer_word(word, k, width, state) (kmer.rs:170-192) for n k-mer words: the leftmost minimum-hash
 * sub-word of `width` bases and its offset.  1 <= width <= k <= 32. */
int kmx_minimizer_words(kmx_ctx *ctx, const uint64_t *d_words, uint64_t n, uint32_t k, uint32_t width, uint32_t hasher,
                        uint32_t hasher_k, uint64_t *d_mmer, uint32_t *d_offset);
/* SeqVectorSlice::iter_minimizers(k, w, hasher) (seq_vector.rs:73-80; SeqVecMinimizerIter, minimizers.rs:39-141) for
 * every read slice [r*read_len, (r+1)*read_len) of a SeqVector: MappedMinimizer{word, pos} per k-mer, slot
 * r*(read_len-k+1) + i, pos relative to the slice.  read_len >= k (the iterator asserts it), 1 <= w <= k, w <= 32. */
int kmx_seqvec_minimizers(kmx_ctx *ctx, const uint64_t *d_words, uint64_t n_reads, uint32_t read_len, uint32_t k,
                          uint32_t w, uint32_t hasher, uint32_t hasher_k, uint64_t *d_word, uint32_t *d_pos);

/* The same iterator over READS (round 6): what SeqVector::from(read).slice(..).iter_minimizers(k, w, hasher) yields
 * (seq_vector.rs:73-80, 230-242; minimizers.rs:39-141) for every read of a batch -- ASCII, one length or behind offsets, as
 * kmx_fastx_parse hands them over -- without building the SeqVector.  MappedMinimizer{word, pos} of k-mer i of read r at slot
 * r*(read_len-k+1) + i (uniform; read_len >= k) or d_win_offsets[r] + i (ragged: n_reads + 1 entries as for kmx_canonical_windows;
 * a read shorter than k owns no slot), pos relative to the read.  1 <= w <= k, w <= 32.
 * A byte outside ACGTacgt: the reference panics (Kmer::from, kmer.rs:45-60).  With h_first_bad != NULL the call synchronises, stores
 * the index of the first read that holds one (else ~0) and returns KMX_E_INVALID_BASE; that read's slots then hold what the codes
 * (byte >> 1) & 3 spell.  With NULL nothing is checked on the host and the call stays asynchronous for uniform reads (reads behind
 * offsets cost one host round trip: their longest length selects the kernel).
 * Reads of up to 256 bases, and uniform reads of any length (cut into pieces of 256 bases), take the sliding-minimum kernel when the
 * hash fits 56 bits (w <= 28; LexHasher: hasher_k <= 28); anything else a lane-per-k-mer kernel. */
int kmx_minimizers(kmx_ctx *ctx, const kmx_reads *reads, const uint64_t *d_win_offsets, uint32_t k, uint32_t w, uint32_t hasher,
                   uint32_t hasher_k, uint64_t *d_word, uint32_t *d_pos, uint64_t *h_first_bad);

/* ---------------------------------------------------------------- FASTA / FASTQ ingestion ----
 * BUILD-DEFINED: the reference has no parser (its iterators take `&[u8]` reads, canonical_kmer_iterator.rs:72-83);
 * this call produces, from a file image in device memory, the ragged-reads input of the calls above:
 * d_bases = the reads back to back, d_offsets[r] = first base of read r, d_offsets[n_reads] = n_bases.
 *   KMX_FASTX_FASTQ: strict 4-line records, line i is a read iff i % 4 == 1.
 *   KMX_FASTX_FASTA: a line starting with '>' opens a record; every other line up to the next '>' line is its sequence.
 *   KMX_FASTX_AUTO:  by the first byte ('@' / '>').
 * Lines end at '\n'; every '\r' on a sequence line is dropped; the last line may lack its '\n'.  Bases are copied as
 * they are (lower case, N, ...: the k-mer calls treat them as the reference's iterator does).  A text that does not
 * start with '@' / '>' gives KMX_E_ARG.
 * d_text 16-byte aligned.  d_bases: room for n_bases bytes (n_bytes always suffices); d_offsets: max_reads+1 words.
 * With d_bases == d_offsets == NULL only the counts are computed; if n_reads > max_reads nothing is written, the counts
 * are returned and the status is KMX_E_NOMEM.  Synchronises the context's stream (the counts come back to the host). */
#define KMX_FASTX_AUTO 0
#define KMX_FASTX_FASTQ 1
#define KMX_FASTX_FASTA 2
/* OR-ed into `format` on the second call of the usual pair (counts first, then the emit into buffers of exactly that
 * size): "d_text / n_bytes are the image the previous kmx_fastx_parse call on this context counted, unchanged, and no
 * kmx_histogram call came in between" -- the emit then reuses the chunk summaries of the counting call instead of
 * reading the text a third time (the pair: 1.06 -> 1.4 TB/s of text).  Ignored when the context has nothing to reuse. */
#define KMX_FASTX_SAME_TEXT 0x100u
int kmx_fastx_parse(kmx_ctx *ctx, const uint8_t *d_text, uint64_t n_bytes, uint32_t format, uint8_t *d_bases,
                    uint64_t *d_offsets, uint64_t max_reads, uint64_t *h_n_reads, uint64_t *h_n_bases);
/* The longest and the shortest read of a ragged batch (d_offsets: n_reads+1 device u64, as kmx_fastx_parse writes them):
 * what kmx_reads.read_len wants as its bound for ragged input (the tiled kernels size their frame and the windows per
 * lane from it; a tight bound is up to 7 % faster than none), and whether the batch is uniform after all (min == max:
 * hand it over with d_offsets = NULL, or pack it with kmx_seqvec_push_chars).  Lengths of 2^32 or more come back as
 * UINT32_MAX.  BUILD-DEFINED helper; synchronises the context's stream (the two values come back to the host). */
int kmx_reads_length_range(kmx_ctx *ctx, const uint64_t *d_offsets, uint64_t n_reads, uint32_t *h_min_len, uint32_t *h_max_len);

/* ---------------------------------------------------------------- multi-GPU exchange ----
 * The reference has no distributed code; reads shard embarrassingly (k-mers never span reads,
 * canonical_kmer_iterator.rs:72-83), so the scans need no collective.  What is exchanged is the optional bucket
 * histogram -- ncclAllReduce(ncclUint64, ncclSum) over RCCL/xGMI -- and the 32-byte summaries.  One kmx_comm per
 * kmx_ctx (one process or thread per GPU); the collectives run on the context's stream.
 * Bootstrap: rank 0 calls kmx_comm_get_unique_id and hands the KMX_COMM_ID_BYTES bytes to the other ranks by any
 * means (file, environment, socket, MPI, a torch.distributed broadcast); then every rank calls kmx_comm_create. */
#define KMX_COMM_ID_BYTES 128
typedef struct kmx_comm kmx_comm;
int kmx_comm_get_unique_id(uint8_t *h_id /* KMX_COMM_ID_BYTES, host */);
int kmx_comm_create(kmx_ctx *ctx, const uint8_t *h_id, int n_ranks, int rank, kmx_comm **out); /* collective over the ranks */
void kmx_comm_destroy(kmx_comm *comm);
int kmx_comm_size(const kmx_comm *comm);
int kmx_comm_rank(const kmx_comm *comm);
/* d_counts[i] = sum over ranks of d_counts[i], in place, n_counts u64 (2^log2_buckets of kmx_histogram) */
int kmx_histogram_allreduce(kmx_comm *comm, uint64_t *d_counts, uint64_t n_counts);
/* the per-shard kmx_summary of every rank combined in place: wrapping sums of n_valid / sum_canon / sum_fw, xor of xor_hash */
int kmx_summary_allreduce(kmx_comm *comm, kmx_summary *d_summary);

/* ---------------------------------------------------------------- measurement helper ----
 * Read-only pass over [d_buf, d_buf + nbytes) with the load shape of the scan kernels (16 bytes per lane, non-temporal,
 * one 9600-byte tile = 64 reads x 150 bytes per wave and step, the next tile requested before the current one is
 * consumed), xor-folded into *d_out (8 bytes, overwritten): the same-run ceiling of the HBM read stream next to which
 * bench.py reports kmx_canonical_reduce, and the known-byte-count kernel the FETCH_SIZE counter is calibrated on.
 * d_buf 16-byte aligned. */
int kmx_calib_stream_read(kmx_ctx *ctx, const uint8_t *d_buf, uint64_t nbytes, uint64_t *d_out);

#ifdef __cplusplus
}
#endif
#endif /* KMX_H */
