/*
 * kmx_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 * See kmx_oracle.h for the contract.  Each function restates the cited lines of
 * the reference crate (paths relative to /root/reference); nothing here is
 * optimised, it is meant to be obviously the same algorithm.
 */
#include "kmx_oracle.h"

#include <string.h>

typedef unsigned __int128 u128;

/* ------------------------------------------------------------------ prelude */

/* src/naive_impl/mod.rs:40-50 */
uint64_t kmo_encode_binary_u8(uint8_t c) {
    switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return KMO_INVALID_CODE;
    }
}

/* src/naive_impl/mod.rs:27-37 -- the char version panics on anything else */
int kmo_encode_binary(uint8_t c, uint64_t *out) {
    uint64_t b = kmo_encode_binary_u8(c);
    if (b == KMO_INVALID_CODE) return KMO_E_INVALID_BASE;
    *out = b;
    return KMO_OK;
}

/* src/naive_impl/mod.rs:54-65 */
uint64_t kmo_encode_complement_binary_u8(uint8_t c) {
    switch (c) {
    case 'A': case 'a': return 3;
    case 'C': case 'c': return 2;
    case 'G': case 'g': return 1;
    case 'T': case 't': return 0;
    default: return KMO_INVALID_CODE;
    }
}

/* src/naive_impl/mod.rs:81-84 */
uint64_t kmo_complement_base(uint64_t b) { return 3 - b; }

/* src/naive_impl/mod.rs:87-89 */
int kmo_is_valid_nuc(uint64_t b) { return b < 4; }

/* --------------------------------------------------------------------- Kmer */

/* src/naive_impl/kmer.rs:30-32 bitmask(pos) = (1<<pos)-1 and :584-618: entries
 * 0..31 are bitmask(2k); entry 32 is the literal 0 (kmer.rs:617). */
uint64_t kmo_mask_table(unsigned k) {
    if (k >= 32) return 0;
    return (((uint64_t)1) << (2 * k)) - 1;
}

/* src/naive_impl/kmer.rs:45-48 */
kmo_kmer kmo_kmer_from_u64(uint64_t data, uint8_t k) {
    kmo_kmer r;
    r.k = k;
    r.data = data & kmo_mask_table(k);
    return r;
}

/* src/naive_impl/kmer.rs:234-251: iterate the bytes reversed, w = (w<<2)|code */
int kmo_kmer_from_bytes(const uint8_t *s, size_t len, kmo_kmer *out) {
    if (len > 32) return KMO_E_TOO_LONG;
    uint64_t w = 0;
    for (size_t i = len; i-- > 0;) {
        uint64_t b;
        if (kmo_encode_binary(s[i], &b) != KMO_OK) return KMO_E_INVALID_BASE;
        w <<= 2;
        w |= b;
    }
    out->k = (uint8_t)len;
    out->data = w;
    return KMO_OK;
}

/* src/naive_impl/kmer.rs:98-102 */
uint64_t kmo_kmer_append_base(kmo_kmer *km, uint64_t c) {
    uint64_t r = km->data & 0x03;
    km->data = (km->data >> 2) | (c << (2 * km->k - 2));
    return r;
}

/* src/naive_impl/kmer.rs:91-95 */
uint64_t kmo_kmer_prepend_base(kmo_kmer *km, uint64_t c) {
    uint64_t r = (km->data >> (2 * km->k - 2)) & 0x03;
    km->data = kmo_mask_table(km->k) & ((km->data << 2) | c);
    return r;
}

/* src/naive_impl/kmer.rs:84-88 */
uint64_t kmo_kmer_append_base_u8(kmo_kmer *km, uint8_t c) {
    return kmo_kmer_append_base(km, kmo_encode_binary_u8(c));
}

/* src/naive_impl/kmer.rs:77-81 */
uint64_t kmo_kmer_prepend_base_u8(kmo_kmer *km, uint8_t c) {
    return kmo_kmer_prepend_base(km, kmo_encode_binary_u8(c));
}

/* the five swap stages shared by kmer.rs:124-147 and hash.rs:60-71 */
static uint64_t swap_2bit_groups(uint64_t res) {
    res = ((res >> 2) & 0x3333333333333333ULL) | ((res & 0x3333333333333333ULL) << 2);
    res = ((res >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((res & 0x0F0F0F0F0F0F0F0FULL) << 4);
    res = ((res >> 8) & 0x00FF00FF00FF00FFULL) | ((res & 0x00FF00FF00FF00FFULL) << 8);
    res = ((res >> 16) & 0x0000FFFF0000FFFFULL) | ((res & 0x0000FFFF0000FFFFULL) << 16);
    res = ((res >> 32) & 0x00000000FFFFFFFFULL) | ((res & 0x00000000FFFFFFFFULL) << 32);
    return res;
}

/* src/naive_impl/kmer.rs:138-147 (and :124-136) */
uint64_t kmo_revcomp_word(uint64_t w, uint8_t k) {
    uint64_t res = swap_2bit_groups(~w);
    unsigned sh = 2 * (32 - (unsigned)k);
    return sh >= 64 ? 0 : res >> sh; /* k == 0 would overflow the shift in Rust */
}

kmo_kmer kmo_kmer_to_reverse_complement(kmo_kmer km) {
    kmo_kmer r;
    r.k = km.k;
    r.data = kmo_revcomp_word(km.data, km.k);
    return r;
}

/* derived Ord (kmer.rs:6): field order k, then data */
int kmo_kmer_cmp(kmo_kmer a, kmo_kmer b) {
    if (a.k != b.k) return a.k < b.k ? -1 : 1;
    if (a.data != b.data) return a.data < b.data ? -1 : 1;
    return 0;
}

/* src/naive_impl/kmer.rs:12-16 */
int kmo_kmer_eq(kmo_kmer a, kmo_kmer b) { return a.data == b.data && a.k == b.k; }

/* src/naive_impl/kmer.rs:55-58: *self <= rc */
int kmo_kmer_is_canonical(kmo_kmer km) {
    return kmo_kmer_cmp(km, kmo_kmer_to_reverse_complement(km)) <= 0;
}

/* src/naive_impl/kmer.rs:68-74 */
kmo_kmer kmo_kmer_to_canonical(kmo_kmer km) {
    return kmo_kmer_is_canonical(km) ? km : kmo_kmer_to_reverse_complement(km);
}

/* src/naive_impl/kmer.rs:156-162 (asserts -> KMO_E_ARG) */
int kmo_sub_kmer_word(uint64_t word, size_t k, size_t pos, size_t width, uint64_t *out) {
    if (!(pos < k)) return KMO_E_ARG;
    if (!(pos + width <= k)) return KMO_E_ARG;
    if (width > 32) return KMO_E_ARG;
    word = word >> (pos * 2);
    *out = word & kmo_mask_table((unsigned)width);
    return KMO_OK;
}

/* src/naive_impl/kmer.rs:196-207, BASE_TABLE :24 */
size_t kmo_kmer_to_string(kmo_kmer km, char *out) {
    static const char base_table[4] = {'a', 'c', 'g', 't'};
    uint64_t w = km.data;
    for (unsigned i = 0; i < km.k; ++i) {
        out[i] = base_table[w & 3];
        w >>= 2;
    }
    out[km.k] = 0;
    return km.k;
}

/* ------------------------------------------------------------ CanonicalKmer */

/* src/naive_impl/canonical_kmer.rs:22-29 */
kmo_canonical_kmer kmo_ck_blank_of_size(uint8_t k) {
    kmo_canonical_kmer r;
    r.fw.k = k;
    r.fw.data = 0;
    r.rc.k = k;
    r.rc.data = UINT64_MAX;
    return r;
}

/* src/naive_impl/canonical_kmer.rs:42-51 */
kmo_canonical_kmer kmo_ck_from_u64(uint64_t data, uint8_t k) {
    kmo_canonical_kmer r;
    r.fw = kmo_kmer_from_u64(data, k);
    r.rc = kmo_kmer_to_reverse_complement(r.fw);
    return r;
}

/* src/naive_impl/canonical_kmer.rs:164-172 */
kmo_canonical_kmer kmo_ck_from_kmer(kmo_kmer km) {
    kmo_canonical_kmer r;
    r.rc = kmo_kmer_to_reverse_complement(km);
    r.fw = km;
    return r;
}

/* src/naive_impl/canonical_kmer.rs:188-196 */
int kmo_ck_from_bytes(const uint8_t *s, size_t len, kmo_canonical_kmer *out) {
    kmo_kmer fw;
    int st = kmo_kmer_from_bytes(s, len, &fw);
    if (st != KMO_OK) return st;
    out->fw = fw;
    out->rc = kmo_kmer_to_reverse_complement(fw);
    return KMO_OK;
}

/* src/naive_impl/canonical_kmer.rs:62-64 */
void kmo_ck_swap(kmo_canonical_kmer *ck) {
    uint64_t t = ck->fw.data;
    ck->fw.data = ck->rc.data;
    ck->rc.data = t;
}

/* src/naive_impl/canonical_kmer.rs:67-69 */
int kmo_ck_is_fw_canonical(const kmo_canonical_kmer *ck) { return ck->fw.data < ck->rc.data; }

/* src/naive_impl/canonical_kmer.rs:90-94 */
uint64_t kmo_ck_append_base(kmo_canonical_kmer *ck, uint64_t b) {
    uint64_t r = kmo_kmer_append_base(&ck->fw, b);
    kmo_kmer_prepend_base(&ck->rc, kmo_complement_base(b));
    return r;
}

/* src/naive_impl/canonical_kmer.rs:97-101 */
uint64_t kmo_ck_prepend_base(kmo_canonical_kmer *ck, uint64_t b) {
    uint64_t r = kmo_kmer_prepend_base(&ck->fw, b);
    kmo_kmer_append_base(&ck->rc, kmo_complement_base(b));
    return r;
}

/* src/naive_impl/canonical_kmer.rs:72-78 */
uint64_t kmo_ck_append_base_u8(kmo_canonical_kmer *ck, uint8_t c) {
    uint64_t b = kmo_encode_binary_u8(c);
    uint64_t cb = kmo_complement_base(b);
    uint64_t r = kmo_kmer_append_base(&ck->fw, b);
    kmo_kmer_prepend_base(&ck->rc, cb);
    return r;
}

/* src/naive_impl/canonical_kmer.rs:81-87 */
uint64_t kmo_ck_prepend_base_u8(kmo_canonical_kmer *ck, uint8_t c) {
    uint64_t b = kmo_encode_binary_u8(c);
    uint64_t cb = kmo_complement_base(b);
    uint64_t r = kmo_kmer_prepend_base(&ck->fw, b);
    kmo_kmer_append_base(&ck->rc, cb);
    return r;
}

/* src/naive_impl/canonical_kmer.rs:113-119 */
uint64_t kmo_ck_get_canonical_word(const kmo_canonical_kmer *ck) {
    return ck->fw.data < ck->rc.data ? ck->fw.data : ck->rc.data;
}

/* src/naive_impl/canonical_kmer.rs:152-161 */
int kmo_ck_get_word_equivalency(const kmo_canonical_kmer *ck, uint64_t other) {
    if (ck->fw.data == other) return KMO_IDENTITY_MATCH;
    if (ck->rc.data == other) return KMO_TWIN_MATCH;
    return KMO_NO_MATCH;
}

/* ---------------------------------------------------- CanonicalKmerIterator */

/* src/naive_impl/canonical_kmer_iterator.rs:42-70 */
static void kmo_iter_find_next(kmo_iter *it, int32_t ii, int32_t jj) {
    int32_t i = ii + 1;
    int32_t j = jj + 1;
    int32_t seq_len = it->seq_len;
    for (int32_t l = j; l < seq_len; ++l) {
        uint64_t b = kmo_encode_binary_u8(it->seq[l]);
        if (b < 4) {
            kmo_ck_append_base(&it->km, b);
            if ((l - it->last_invalid) >= it->k) {
                it->pos = i;
                return;
            }
        } else {
            it->last_invalid = l;
            i = l + 1;
        }
    }
    it->invalid = 1;
}

/* src/naive_impl/canonical_kmer_iterator.rs:72-83 (+ CanonicalKmerPos::new :19-24) */
void kmo_iter_from_u8_slice(kmo_iter *it, const uint8_t *s, size_t len, uint8_t k) {
    it->seq = s;
    it->seq_len = (int32_t)len;
    it->km = kmo_ck_blank_of_size(k);
    it->pos = -1;
    it->invalid = 0;
    it->last_invalid = -1;
    it->k = (int32_t)k;
    kmo_iter_find_next(it, -1, -1);
}

/* src/naive_impl/canonical_kmer_iterator.rs:89-91 */
int kmo_iter_exhausted(const kmo_iter *it) { return it->invalid; }

/* src/naive_impl/canonical_kmer_iterator.rs:94-101 */
int kmo_iter_inc(kmo_iter *it) {
    int32_t lpos = it->pos + it->k;
    it->invalid = it->invalid || (lpos >= it->seq_len);
    if (!it->invalid) kmo_iter_find_next(it, it->pos, lpos - 1);
    return !it->invalid;
}

/* src/naive_impl/canonical_kmer_iterator.rs:104-111 */
int kmo_iter_inc_by(kmo_iter *it, size_t count) {
    int v = !it->invalid;
    while (count > 0 && v) {
        v = kmo_iter_inc(it);
        count -= 1;
    }
    return v;
}

/* --------------------------------------------------------------------- hash */

/* src/naive_impl/hash.rs:60-71: swap stages WITHOUT the complement, >> (32-k)*2 */
uint64_t kmo_lex_hash_u64(uint64_t word, size_t hasher_k) {
    uint64_t res = swap_2bit_groups(word);
    size_t sh = (32 - hasher_k) * 2;
    return sh >= 64 ? 0 : res >> sh;
}

/* std's DefaultHasher behind hash_one (src/naive_impl/hash.rs:10-20; used at kmer.rs:546-557 with DefaultHasher, :564-575 with
 * RandomState): SipHash-c-d of a byte string, restated from the SipHash paper (Aumasson, Bernstein 2012, section 2 / appendix A: the
 * algorithm is in Rust's standard library, not in the crate).  c = 2, d = 4 reproduces the paper's test vectors
 * (tests/test_oracle_golden.py); DefaultHasher is c = 1, d = 3 over the 8 little-endian bytes of write_u64(data) (hash.rs:4-8). */
static uint64_t rotl64(uint64_t x, int b) { return (x << b) | (x >> (64 - b)); }
uint64_t kmo_siphash(unsigned c, unsigned d, uint64_t k0, uint64_t k1, const uint8_t *msg, size_t len) {
    uint64_t v0 = k0 ^ 0x736f6d6570736575ull, v1 = k1 ^ 0x646f72616e646f6dull, v2 = k0 ^ 0x6c7967656e657261ull, v3 = k1 ^ 0x7465646279746573ull;
#define KMO_SIPROUND                                                                                   \
    do {                                                                                               \
        v0 += v1; v1 = rotl64(v1, 13); v1 ^= v0; v0 = rotl64(v0, 32);                                  \
        v2 += v3; v3 = rotl64(v3, 16); v3 ^= v2;                                                       \
        v0 += v3; v3 = rotl64(v3, 21); v3 ^= v0;                                                       \
        v2 += v1; v1 = rotl64(v1, 17); v1 ^= v2; v2 = rotl64(v2, 32);                                  \
    } while (0)
    size_t i = 0;
    for (; i + 8 <= len; i += 8) {
        uint64_t m = 0;
        for (int j = 0; j < 8; ++j) m |= (uint64_t)msg[i + j] << (8 * j);
        v3 ^= m;
        for (unsigned r = 0; r < c; ++r) KMO_SIPROUND;
        v0 ^= m;
    }
    uint64_t b = (uint64_t)(len & 0xff) << 56;
    for (int j = 0; i + j < len; ++j) b |= (uint64_t)msg[i + j] << (8 * j);
    v3 ^= b;
    for (unsigned r = 0; r < c; ++r) KMO_SIPROUND;
    v0 ^= b;
    v2 ^= 0xff;
    for (unsigned r = 0; r < d; ++r) KMO_SIPROUND;
#undef KMO_SIPROUND
    return v0 ^ v1 ^ v2 ^ v3;
}
uint64_t kmo_siphash13_u64(uint64_t word, uint64_t k0, uint64_t k1) {
    uint8_t m[8];
    for (int j = 0; j < 8; ++j) m[j] = (uint8_t)(word >> (8 * j));
    return kmo_siphash(1, 3, k0, k1, m, 8);
}

/* ----------------------------------------------------------- encoding::Naive */

static unsigned flat_get2(const uint8_t *array, size_t bit) { /* bit_field get_bits(bit..bit+2), bit even */
    return (array[bit >> 3] >> (bit & 7)) & 3u;
}
static void flat_set2(uint8_t *array, size_t bit, unsigned v) { /* bit_field set_bits(bit..bit+2, v) */
    array[bit >> 3] = (uint8_t)((array[bit >> 3] & ~(3u << (bit & 7))) | ((v & 3u) << (bit & 7)));
}

/* src/encoding/naive.rs:14-16 */
uint8_t kmo_nuc2internal(uint8_t nuc) { return (nuc >> 1) & 3; }

/* src/encoding/naive.rs:19 INTERNAL2NUC */
static const uint8_t INTERNAL2NUC[4] = {'A', 'C', 'T', 'G'};

/* src/encoding/naive.rs:28-39 */
uint8_t kmo_rev_encoding(uint8_t enc) {
    uint8_t rev = 0;
    rev ^= (uint8_t)(0u << (6 - ((enc >> 6) * 2)));
    rev ^= (uint8_t)(1u << (6 - (((enc >> 4) & 3) * 2)));
    rev ^= (uint8_t)(2u << (6 - (((enc >> 2) & 3) * 2)));
    rev ^= (uint8_t)(3u << (6 - ((enc & 3) * 2)));
    return rev;
}

/* src/encoding/naive.rs:78-85 */
uint8_t kmo_naive_nuc2bits(uint8_t enc, uint8_t nuc) {
    unsigned index = 6 - kmo_nuc2internal(nuc) * 2;
    return (enc >> index) & 3;
}

/* src/encoding/naive.rs:88-95 */
uint8_t kmo_naive_bits2nuc(uint8_t enc, uint8_t bits) {
    uint8_t rev = kmo_rev_encoding(enc);
    return INTERNAL2NUC[(rev >> (6 - (bits & 3) * 2)) & 3];
}

/* src/encoding/naive.rs:98-109 */
uint8_t kmo_naive_complement(uint8_t enc, uint8_t bits) {
    uint8_t rev = kmo_rev_encoding(enc);
    uint8_t internal = (rev >> (6 - (bits & 3) * 2)) & 3;
    uint8_t comp_internal = (internal ^ 2) & 3;
    return (enc >> (6 - comp_internal * 2)) & 3;
}

/* src/encoding/naive.rs:116-124: zeroed array, set_bits(2idx..=2idx+1); bit_field
 * panics when the index runs past the array -> KMO_E_TOO_LONG */
int kmo_naive_encode(uint8_t enc, const uint8_t *seq, size_t len, uint8_t *array, size_t nbytes) {
    memset(array, 0, nbytes);
    for (size_t idx = 0; idx < len; ++idx) {
        if (idx * 2 + 1 >= nbytes * 8) return KMO_E_TOO_LONG;
        flat_set2(array, idx * 2, kmo_naive_nuc2bits(enc, seq[idx]));
    }
    return KMO_OK;
}

/* src/encoding/naive.rs:126-136: emits ALL nbytes*4 slots */
void kmo_naive_decode(uint8_t enc, const uint8_t *array, size_t nbytes, uint8_t *seq_out) {
    for (size_t idx = 0; idx < nbytes * 4; ++idx)
        seq_out[idx] = kmo_naive_bits2nuc(enc, (uint8_t)flat_get2(array, idx * 2));
}

/* src/encoding/naive.rs:138-154: two-pointer swap+complement; K<2 underflows usize */
int kmo_naive_rev_comp(uint8_t enc, size_t K, uint8_t *array, size_t nbytes) {
    if (K < 2 || K * 2 > nbytes * 8) return KMO_E_ARG;
    size_t i = 0;
    size_t j = K * 2 - 2;
    while (i <= j) {
        uint8_t comp_i = kmo_naive_complement(enc, (uint8_t)flat_get2(array, i));
        uint8_t comp_j = kmo_naive_complement(enc, (uint8_t)flat_get2(array, j));
        flat_set2(array, i, comp_j);
        flat_set2(array, j, comp_i);
        i += 2;
        if (j < 2) break; /* K >= 2 never reaches here before i > j; guard only */
        j -= 2;
    }
    return KMO_OK;
}

/* ----------------------------------------------------------- encoding::Xor10 */

/* src/encoding/xor10.rs:9 BITS2NUC, :17-22 nuc2bits, :35-40 complement */
static const uint8_t XOR10_BITS2NUC[4] = {'A', 'C', 'T', 'G'};

/* src/encoding/xor10.rs:52-60 */
int kmo_xor10_encode(const uint8_t *seq, size_t len, uint8_t *array, size_t nbytes) {
    memset(array, 0, nbytes);
    for (size_t idx = 0; idx < len; ++idx) {
        if (idx * 2 + 1 >= nbytes * 8) return KMO_E_TOO_LONG;
        flat_set2(array, idx * 2, (seq[idx] >> 1) & 3u);
    }
    return KMO_OK;
}

/* src/encoding/xor10.rs:62-72 */
void kmo_xor10_decode(const uint8_t *array, size_t nbytes, uint8_t *seq_out) {
    for (size_t idx = 0; idx < nbytes * 4; ++idx) seq_out[idx] = XOR10_BITS2NUC[flat_get2(array, idx * 2)];
}

/* src/encoding/xor10.rs:86-103 (the B>1 branch) */
int kmo_xor10_rev_comp(size_t K, uint8_t *array, size_t nbytes) {
    if (K < 2 || K * 2 > nbytes * 8) return KMO_E_ARG;
    size_t i = 0;
    size_t j = K * 2 - 2;
    while (i <= j) {
        unsigned comp_i = flat_get2(array, i) ^ 2u;
        unsigned comp_j = flat_get2(array, j) ^ 2u;
        flat_set2(array, i, comp_j);
        flat_set2(array, j, comp_i);
        i += 2;
        if (j < 2) break;
        j -= 2;
    }
    return KMO_OK;
}

/* src/encoding/xor10.rs:75-85: the B==1 branch as written -- group-reverse then
 * `(8*size_of::<P>()) as u64 - kmer*2` (wrapping in a release build).  It is not a
 * reverse complement; restated only so the quirk is documented and testable. */
uint64_t kmo_xor10_rev_comp_b1_quirk(uint64_t word, unsigned p_bits) {
    uint64_t kmer = swap_2bit_groups(word);
    return (uint64_t)p_bits - kmer * 2u;
}

/* ------------------------------------------------------------- generic Kmer */

/* src/kmer.rs:67-69 */
size_t kmo_word_for_k(size_t p_bytes, size_t K) {
    return (p_bytes * 8 / 2 + K - 1) / (p_bytes * 8 / 2);
}

/* src/kmer.rs:46-48 */
uint64_t kmo_generic_get(const uint8_t *array, size_t nbytes, size_t index) {
    (void)nbytes;
    return flat_get2(array, index * 2);
}

/* src/kmer.rs:50-52: get_bits(0..=(len*2)) -- an INCLUSIVE range, 2*len+1 bits, returned as P */
int kmo_generic_get_prefix(const uint8_t *array, size_t nbytes, unsigned p_bits, size_t len, uint64_t *out) {
    size_t nbits = len * 2 + 1;
    if (nbits > p_bits || nbits > 64 || nbits > nbytes * 8) return KMO_E_ARG;
    uint64_t v = 0;
    for (size_t b = 0; b < nbits; ++b) v |= (uint64_t)((array[b >> 3] >> (b & 7)) & 1u) << b;
    *out = v;
    return KMO_OK;
}

/* src/kmer.rs:71-91 */
void kmo_bitmer_to_bytes(uint64_t mer, size_t len, uint8_t *out) {
    static const uint8_t tbl[4] = {'A', 'C', 'G', 'T'};
    for (size_t i = 0; i < len; ++i) {
        out[i] = tbl[mer & 3];
        mer >>= 2;
    }
}

/* ------------------------------------------------------------ batch drivers */

static void read_span(const uint8_t *reads, size_t r, size_t read_len, const uint64_t *offsets,
                      const uint8_t **s, size_t *len) {
    if (offsets) {
        *s = reads + offsets[r];
        *len = (size_t)(offsets[r + 1] - offsets[r]);
    } else {
        *s = reads + r * read_len;
        *len = read_len;
    }
}

int kmo_canonical_reduce(const uint8_t *reads, size_t n_reads, size_t read_len, const uint64_t *offsets,
                         uint8_t k, size_t hasher_k, kmo_summary *out) {
    if (k < 1 || k > 32 || hasher_k > 32) return KMO_E_ARG;
    kmo_summary s = {0, 0, 0, 0};
    for (size_t r = 0; r < n_reads; ++r) {
        const uint8_t *p;
        size_t len;
        read_span(reads, r, read_len, offsets, &p, &len);
        kmo_iter it;
        kmo_iter_from_u8_slice(&it, p, len, k);
        while (!kmo_iter_exhausted(&it)) {
            uint64_t w = kmo_ck_get_canonical_word(&it.km);
            s.n_valid += 1;
            s.sum_canon += w;
            s.sum_fw += it.km.fw.data;
            if (hasher_k) s.xor_hash ^= kmo_lex_hash_u64(w, hasher_k);
            kmo_iter_inc(&it);
        }
    }
    *out = s;
    return KMO_OK;
}

int kmo_canonical_windows(const uint8_t *reads, size_t n_reads, size_t read_len, const uint64_t *offsets,
                          const uint64_t *win_offsets, uint8_t k,
                          uint64_t *out_fw, uint64_t *out_rc, uint64_t *out_canon, uint8_t *out_flags) {
    if (k < 1 || k > 32) return KMO_E_ARG;
    for (size_t r = 0; r < n_reads; ++r) {
        const uint8_t *p;
        size_t len;
        read_span(reads, r, read_len, offsets, &p, &len);
        size_t nwin = len >= k ? len - k + 1 : 0;
        size_t base = win_offsets ? (size_t)win_offsets[r] : r * (read_len >= k ? read_len - k + 1 : 0);
        for (size_t w = 0; w < nwin; ++w) {
            if (out_fw) out_fw[base + w] = 0;
            if (out_rc) out_rc[base + w] = 0;
            if (out_canon) out_canon[base + w] = 0;
            if (out_flags) out_flags[base + w] = 0;
        }
        kmo_iter it;
        kmo_iter_from_u8_slice(&it, p, len, k);
        while (!kmo_iter_exhausted(&it)) {
            size_t slot = base + (size_t)it.pos;
            if (out_fw) out_fw[slot] = it.km.fw.data;
            if (out_rc) out_rc[slot] = it.km.rc.data;
            if (out_canon) out_canon[slot] = kmo_ck_get_canonical_word(&it.km);
            if (out_flags) out_flags[slot] = (uint8_t)(1u | (kmo_ck_is_fw_canonical(&it.km) ? 2u : 0u));
            kmo_iter_inc(&it);
        }
    }
    return KMO_OK;
}

/* benches/simple_benchmark.rs:14-22 */
int kmo_compute_naive(const uint8_t *b, size_t len, size_t K, uint64_t *out_sum) {
    uint64_t sum = 0;
    if (K == 0) return KMO_E_ARG; /* slice::windows(0) panics */
    for (size_t o = 0; o + K <= len; ++o) {
        kmo_kmer km;
        int st = kmo_kmer_from_bytes(b + o, K, &km);
        if (st != KMO_OK) return st;
        sum += km.data;
    }
    *out_sum = sum;
    return KMO_OK;
}

/* benches/simple_benchmark.rs:36-44 shape, result kept: canonical word per window */
int kmo_compute_naive_canonical(const uint8_t *b, size_t len, size_t K, uint64_t *out_sum) {
    uint64_t sum = 0;
    if (K == 0) return KMO_E_ARG;
    for (size_t o = 0; o + K <= len; ++o) {
        kmo_kmer km;
        int st = kmo_kmer_from_bytes(b + o, K, &km);
        if (st != KMO_OK) return st;
        sum += kmo_kmer_to_canonical(km).data;
    }
    *out_sum = sum;
    return KMO_OK;
}

/* ------------------------------------------- BUILD-DEFINED: [u64;2] k-mers */

static u128 mask128(unsigned k) { return k >= 64 ? ~(u128)0 : ((((u128)1) << (2 * k)) - 1); }

static u128 swap_2bit_groups128(u128 v) {
    uint64_t lo = (uint64_t)v, hi = (uint64_t)(v >> 64);
    return ((u128)swap_2bit_groups(lo) << 64) | (u128)swap_2bit_groups(hi);
}

typedef struct {
    const uint8_t *seq;
    int32_t seq_len, pos, last_invalid, k;
    int invalid;
    u128 fw, rc;
} iter2;

/* same control flow as canonical_kmer_iterator.rs:42-70, arithmetic of kmer.rs:91-102 on 128 bits */
static void iter2_find_next(iter2 *it, int32_t ii, int32_t jj) {
    int32_t i = ii + 1, j = jj + 1;
    for (int32_t l = j; l < it->seq_len; ++l) {
        uint64_t b = kmo_encode_binary_u8(it->seq[l]);
        if (b < 4) {
            it->fw = (it->fw >> 2) | ((u128)b << (2 * it->k - 2));
            it->rc = mask128((unsigned)it->k) & ((it->rc << 2) | (u128)(3 - b));
            if ((l - it->last_invalid) >= it->k) {
                it->pos = i;
                return;
            }
        } else {
            it->last_invalid = l;
            i = l + 1;
        }
    }
    it->invalid = 1;
}
static void iter2_init(iter2 *it, const uint8_t *s, size_t len, uint8_t k) {
    it->seq = s;
    it->seq_len = (int32_t)len;
    it->pos = -1;
    it->last_invalid = -1;
    it->k = k;
    it->invalid = 0;
    it->fw = 0;
    it->rc = ~(u128)0;
    iter2_find_next(it, -1, -1);
}
static void iter2_inc(iter2 *it) {
    int32_t lpos = it->pos + it->k;
    it->invalid = it->invalid || (lpos >= it->seq_len);
    if (!it->invalid) iter2_find_next(it, it->pos, lpos - 1);
}

/* ------------------------------------------------------------------ SeqVector (seq_vector.rs) ---- */

/* RawVector::push_int(value, width): append `width` bits LSB-first at bit offset `bit_len` */
static void rawvec_push_int(uint64_t *words, size_t *bit_len, uint64_t value, size_t width) {
    if (width == 0) return;
    if (width < 64) value &= ((uint64_t)1 << width) - 1;
    size_t w = *bit_len >> 6, off = *bit_len & 63;
    if (off == 0) {
        words[w] = value;
    } else {
        words[w] |= value << off;
        if (off + width > 64) words[w + 1] = value >> (64 - off);
    }
    *bit_len += width;
}

/* RawVector::int(bit_offset, width), width in [1,64] */
static uint64_t rawvec_int(const uint64_t *words, size_t n_words, size_t bit_offset, size_t width) {
    size_t w = bit_offset >> 6, off = bit_offset & 63;
    uint64_t v = words[w] >> off;
    if (off != 0 && off + width > 64 && w + 1 < n_words) v |= words[w + 1] << (64 - off);
    if (width < 64) v &= ((uint64_t)1 << width) - 1;
    return v;
}

int kmo_seqvec_push_chars(uint64_t *words, size_t n_before, const uint8_t *bytes, size_t n, size_t *bad_index) {
    size_t bit_len = 2 * n_before;
    if ((bit_len & 63) != 0) words[bit_len >> 6] &= (((uint64_t)1 << (bit_len & 63)) - 1);   /* RawVector keeps the tail zero */
    size_t first_word_len = n % 32;   /* seq_vector.rs:142-143 */
    size_t done = 0;
    while (done < n) {
        size_t chunk = (done == 0 && first_word_len) ? first_word_len : 32;
        kmo_kmer km;
        int st = kmo_kmer_from_bytes(bytes + done, chunk, &km);   /* Kmer::from(chunk) */
        if (st != KMO_OK) {
            if (bad_index) {
                for (size_t i = 0; i < chunk; ++i)
                    if (kmo_encode_binary_u8(bytes[done + i]) > 3) { *bad_index = done + i; break; }
            }
            return st;
        }
        rawvec_push_int(words, &bit_len, km.data, chunk * 2);
        done += chunk;
    }
    return KMO_OK;
}

int kmo_seqvec_get_kmer_u64(const uint64_t *words, size_t n_bases, size_t pos, size_t k, uint64_t *out) {
    if (k < 1 || k > 32 || pos >= n_bases || pos + k > n_bases) return KMO_E_ARG;
    *out = rawvec_int(words, (n_bases + 31) / 32, pos * 2, k * 2);
    return KMO_OK;
}

void kmo_seqvec_to_bytes(const uint64_t *words, size_t n_bases, uint8_t *out) {
    static const char bases[4] = {'A', 'C', 'G', 'T'};   /* seq_vector.rs:174 */
    for (size_t i = 0; i < n_bases; ++i) {
        uint64_t b = 0;
        kmo_seqvec_get_kmer_u64(words, n_bases, i, 1, &b);
        out[i] = (uint8_t)bases[b];
    }
}

size_t kmo_seqvec_iter_kmers(const uint64_t *words, size_t n_bases, size_t start, size_t end, size_t k, uint64_t *out) {
    if (end > n_bases || start > end || end - start < k) return 0;
    size_t len = end - start - k + 1;
    for (size_t p = 0; p < len; ++p) kmo_seqvec_get_kmer_u64(words, n_bases, start + p, k, &out[p]);
    return len;
}

int kmo_seqvec_canonical_reduce(const uint64_t *words, size_t n_reads, size_t read_len, uint8_t k, size_t hasher_k,
                                kmo_summary *out) {
    if (k < 1 || k > 32 || hasher_k > 32) return KMO_E_ARG;
    kmo_summary s = {0, 0, 0, 0};
    size_t n_bases = n_reads * read_len;
    for (size_t r = 0; r < n_reads; ++r) {
        if (read_len < k) break;
        for (size_t p = 0; p + k <= read_len; ++p) {
            uint64_t w = 0;
            kmo_seqvec_get_kmer_u64(words, n_bases, r * read_len + p, k, &w);
            kmo_kmer canon = kmo_kmer_to_canonical(kmo_kmer_from_u64(w, k));
            s.n_valid += 1;
            s.sum_canon += canon.data;
            s.sum_fw += w;
            if (hasher_k) s.xor_hash ^= kmo_lex_hash_u64(canon.data, hasher_k);
        }
    }
    *out = s;
    return KMO_OK;
}


/* ------------------------------------------------------------------ minimizers ---- */

uint64_t kmo_mm_hash(uint64_t lmer, size_t hasher_k) { return hasher_k ? kmo_lex_hash_u64(lmer, hasher_k) : lmer; }

int kmo_minimizer_word(uint64_t word, size_t k, size_t width, size_t hasher_k, uint64_t *out_mmer, size_t *out_offset) {
    if (k < 1 || k > 32 || width < 1 || width > k) return KMO_E_ARG;   /* sub_kmer_word asserts pos + width <= k */
    uint64_t min_mmer = 0, min_hash = UINT64_MAX;
    size_t offset = 0;
    kmo_sub_kmer_word(word, k, 0, width, &min_mmer);
    for (size_t pos = 0; pos < k - width + 1; ++pos) {
        uint64_t mmer = 0;
        kmo_sub_kmer_word(word, k, pos, width, &mmer);
        uint64_t h = kmo_mm_hash(mmer, hasher_k);
        if (h < min_hash) {
            min_mmer = mmer;
            min_hash = h;
            offset = pos;
        }
    }
    *out_mmer = min_mmer;
    *out_offset = offset;
    return KMO_OK;
}

void kmo_mmiter_enqueue(kmo_mmiter *it, kmo_dqmer m) {
    if (it->len && it->dq[it->head].pos < it->curr_km_i) {   /* at most one falls out of the window */
        it->head = (it->head + 1) % KMO_DQ_CAP;
        it->len -= 1;
    }
    while (it->len) {
        const kmo_dqmer *back = &it->dq[(it->head + it->len - 1) % KMO_DQ_CAP];
        if (back->hash <= m.hash) break;
        it->len -= 1;
    }
    it->dq[(it->head + it->len) % KMO_DQ_CAP] = m;
    it->len += 1;
}

static kmo_dqmer mm_dqmer_at(const kmo_mmiter *it, size_t pos) {
    kmo_dqmer m;
    m.pos = pos;
    m.lmer = 0;
    kmo_seqvec_get_kmer_u64(it->words, it->n_bases, it->start + pos, it->w, &m.lmer);
    m.hash = kmo_mm_hash(m.lmer, it->hasher_k);
    return m;
}

int kmo_mmiter_new(kmo_mmiter *it, const uint64_t *words, size_t n_bases, size_t start, size_t end,
                   size_t k, size_t w, size_t hasher_k) {
    if (end > n_bases || start > end || end - start < k || w < 1 || w > k || w > 32 || k - w + 2 > KMO_DQ_CAP) return KMO_E_ARG;
    it->head = it->len = 0;
    it->k = k;
    it->w = w;
    it->curr_km_i = 0;
    it->words = words;
    it->n_bases = n_bases;
    it->start = start;
    it->slice_len = end - start;
    it->hasher_k = hasher_k;
    for (size_t i = 0; i < k - w; ++i) kmo_mmiter_enqueue(it, mm_dqmer_at(it, i));
    return KMO_OK;
}

int kmo_mmiter_next(kmo_mmiter *it, uint64_t *out_word, size_t *out_pos) {
    if (it->curr_km_i >= it->slice_len - it->k + 1) return 0;
    kmo_mmiter_enqueue(it, mm_dqmer_at(it, it->curr_km_i + it->k - it->w));
    *out_word = it->dq[it->head].lmer;
    *out_pos = it->dq[it->head].pos;
    it->curr_km_i += 1;
    return 1;
}

int kmo_seqvec_minimizers(const uint64_t *words, size_t n_reads, size_t read_len, size_t k, size_t w, size_t hasher_k,
                          uint64_t *out_word, uint32_t *out_pos) {
    if (read_len < k) return KMO_OK;
    const size_t W = read_len - k + 1;
    for (size_t r = 0; r < n_reads; ++r) {
        kmo_mmiter it;
        int st = kmo_mmiter_new(&it, words, n_reads * read_len, r * read_len, (r + 1) * read_len, k, w, hasher_k);
        if (st != KMO_OK) return st;
        uint64_t word;
        size_t pos, i = 0;
        while (kmo_mmiter_next(&it, &word, &pos)) {
            out_word[r * W + i] = word;
            out_pos[r * W + i] = (uint32_t)pos;
            ++i;
        }
    }
    return KMO_OK;
}


int kmo_canonical_reduce2(const uint8_t *reads, size_t n_reads, size_t read_len, const uint64_t *offsets,
                          uint8_t k, int with_hash, kmo_summary2 *out) {
    if (k < 33 || k > 64) return KMO_E_ARG;
    kmo_summary2 s = {0, 0, 0, 0, 0};
    for (size_t r = 0; r < n_reads; ++r) {
        const uint8_t *p;
        size_t len;
        read_span(reads, r, read_len, offsets, &p, &len);
        iter2 it;
        iter2_init(&it, p, len, k);
        while (!it.invalid) {
            u128 c = it.fw < it.rc ? it.fw : it.rc;
            s.n_valid += 1;
            s.sum_lo += (uint64_t)c;
            s.sum_hi += (uint64_t)(c >> 64);
            if (with_hash) {
                u128 h = swap_2bit_groups128(c) >> (2 * (64 - (unsigned)k));
                s.xor_lo ^= (uint64_t)h;
                s.xor_hi ^= (uint64_t)(h >> 64);
            }
            iter2_inc(&it);
        }
    }
    *out = s;
    return KMO_OK;
}

int kmo_canonical_windows2(const uint8_t *reads, size_t n_reads, size_t read_len, const uint64_t *offsets,
                           const uint64_t *win_offsets, uint8_t k,
                           uint64_t *out_fw2, uint64_t *out_rc2, uint64_t *out_canon2, uint8_t *out_flags) {
    if (k < 33 || k > 64) return KMO_E_ARG;
    for (size_t r = 0; r < n_reads; ++r) {
        const uint8_t *p;
        size_t len;
        read_span(reads, r, read_len, offsets, &p, &len);
        size_t nwin = len >= k ? len - k + 1 : 0;
        size_t base = win_offsets ? (size_t)win_offsets[r] : r * (read_len >= k ? read_len - k + 1 : 0);
        for (size_t w = 0; w < nwin; ++w) {
            if (out_fw2) out_fw2[2 * (base + w)] = out_fw2[2 * (base + w) + 1] = 0;
            if (out_rc2) out_rc2[2 * (base + w)] = out_rc2[2 * (base + w) + 1] = 0;
            if (out_canon2) out_canon2[2 * (base + w)] = out_canon2[2 * (base + w) + 1] = 0;
            if (out_flags) out_flags[base + w] = 0;
        }
        iter2 it;
        iter2_init(&it, p, len, k);
        while (!it.invalid) {
            size_t slot = base + (size_t)it.pos;
            u128 c = it.fw < it.rc ? it.fw : it.rc;
            if (out_fw2) { out_fw2[2 * slot] = (uint64_t)it.fw; out_fw2[2 * slot + 1] = (uint64_t)(it.fw >> 64); }
            if (out_rc2) { out_rc2[2 * slot] = (uint64_t)it.rc; out_rc2[2 * slot + 1] = (uint64_t)(it.rc >> 64); }
            if (out_canon2) { out_canon2[2 * slot] = (uint64_t)c; out_canon2[2 * slot + 1] = (uint64_t)(c >> 64); }
            if (out_flags) out_flags[slot] = (uint8_t)(1u | (it.fw < it.rc ? 2u : 0u));
            iter2_inc(&it);
        }
    }
    return KMO_OK;
}

/* --------------------------------- BUILD-DEFINED: generator, buckets, histogram */

uint64_t kmo_splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}

void kmo_gen_reads(uint64_t seed, uint64_t first_byte, uint8_t *out, size_t nbytes) {
    static const char acgt[4] = {'A', 'C', 'G', 'T'};
    for (size_t i = 0; i < nbytes; ++i) {
        uint64_t g = first_byte + i;
        uint64_t z = kmo_splitmix64(seed + (g >> 5));
        out[i] = (uint8_t)acgt[(z >> (2 * (g & 31))) & 3];
    }
}

uint64_t kmo_bucket_of(uint64_t h, unsigned log2_buckets) {
    if (log2_buckets == 0) return 0;
    uint32_t x = (uint32_t)h * 0x9E3779B1u + (uint32_t)(h >> 32) * 0x85EBCA6Bu;
    return (uint64_t)(x >> (32 - log2_buckets));
}

int kmo_histogram(const uint8_t *reads, size_t n_reads, size_t read_len, const uint64_t *offsets,
                  uint8_t k, size_t hasher_k, unsigned log2_buckets, uint64_t *counts) {
    if (k < 1 || k > 32 || hasher_k > 32 || log2_buckets > 30) return KMO_E_ARG;
    for (size_t r = 0; r < n_reads; ++r) {
        const uint8_t *p;
        size_t len;
        read_span(reads, r, read_len, offsets, &p, &len);
        kmo_iter it;
        kmo_iter_from_u8_slice(&it, p, len, k);
        while (!kmo_iter_exhausted(&it)) {
            uint64_t w = kmo_ck_get_canonical_word(&it.km);
            uint64_t h = hasher_k ? kmo_lex_hash_u64(w, hasher_k) : w; /* hasher_k==0: identity (write_u64(data)) */
            counts[kmo_bucket_of(h, log2_buckets)] += 1;
            kmo_iter_inc(&it);
        }
    }
    return KMO_OK;
}

/* BUILD-DEFINED (no reference counterpart): see kmx_oracle.h */
int kmo_fastx_parse(const uint8_t *text, size_t n, unsigned format, uint8_t *bases, uint64_t *offsets,
                    size_t *n_reads, size_t *n_bases) {
    size_t nr = 0, nb = 0;
    if (format > 2) return KMO_E_ARG;
    if (n == 0) {
        if (offsets) offsets[0] = 0;
        *n_reads = 0;
        *n_bases = 0;
        return KMO_OK;
    }
    if (format == 0) format = text[0] == '@' ? 1 : text[0] == '>' ? 2 : 3;
    if (format == 3 || (format == 1 && text[0] != '@') || (format == 2 && text[0] != '>')) return KMO_E_ARG;
    size_t line = 0;       /* index of the current line */
    int at_start = 1;      /* the next byte is the first of its line */
    int seq_line = 0;      /* the current line is a sequence line */
    for (size_t i = 0; i < n; ++i) {
        uint8_t b = text[i];
        if (at_start) {    /* a line exists as soon as it has a byte (or is terminated: an empty line still has its \n) */
            at_start = 0;
            if (format == 1) {
                seq_line = (line & 3) == 1;
                if (seq_line) { if (offsets) offsets[nr] = nb; nr += 1; }
            } else {
                seq_line = b != '>';
                if (!seq_line) { if (offsets) offsets[nr] = nb; nr += 1; }
            }
        }
        if (b == '\n') {
            line += 1;
            at_start = 1;
        } else if (seq_line && b != '\r') {
            if (bases) bases[nb] = b;
            nb += 1;
        }
    }
    if (offsets) offsets[nr] = nb;
    *n_reads = nr;
    *n_bases = nb;
    return KMO_OK;
}
