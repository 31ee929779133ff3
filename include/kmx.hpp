// kmx.hpp -- header-only C++17 host layer above the libkmx C ABI (include/kmx.h).
//
// It mirrors the reference crate's interface for the hot path -- same names, argument meaning and
// error behaviour -- so that code (and tests) written against COMBINE-lab/kmers read the same:
//   kmx::naive_impl::{Kmer, CanonicalKmer, CanonicalKmerIterator, MatchType, A,C,G,T, hash::*}
//       <-> src/naive_impl/{kmer,canonical_kmer,canonical_kmer_iterator,hash,mod}.rs
//   kmx::encoding::{Naive, Xor10}, kmx::kmer::{Kmer<K,B>, word_for_k, bitmer_to_bytes}
//       <-> src/encoding/{naive,xor10}.rs, src/kmer.rs
// The reference panics become kmx::Panic exceptions (thrown on the host, never across the C ABI).
// Every k-mer computation is done by a libkmx HIP kernel: scalar calls are 1-element device batches
// (the GPU is a drop-in at batch granularity; use the *_batch entry points / kmx.h for throughput).
// There is no CPU fallback: constructing a Context without an MI355X throws.
#pragma once

#include <array>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "kmx.h"

namespace kmx {

struct Panic : std::runtime_error {  // where the reference would panic
    int status;
    Panic(int st, const std::string& what) : std::runtime_error(what + ": " + kmx_strerror(st)), status(st) {}
};

class Context {
  public:
    explicit Context(int device = 0) {
        int st = kmx_ctx_create(device, &h_);
        if (st != KMX_OK) throw Panic(st, "kmx_ctx_create");
    }
    ~Context() { kmx_ctx_destroy(h_); }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    kmx_ctx* get() const { return h_; }
    void check(int st, const char* what) const {
        if (st != KMX_OK) throw Panic(st, std::string(what) + (st == KMX_E_HIP ? std::string(" [") + kmx_last_error(h_) + "]" : ""));
    }
    void sync() const { check(kmx_ctx_synchronize(h_), "sync"); }
    static Context& instance() {  // lazily created default context for the scalar convenience API
        static Context c(0);
        return c;
    }

  private:
    kmx_ctx* h_ = nullptr;
};

template <typename T>
class DeviceBuffer {  // RAII device array
  public:
    DeviceBuffer(const Context& c, size_t n) : c_(c), n_(n) { c_.check(kmx_malloc(c_.get(), n * sizeof(T), reinterpret_cast<void**>(&p_)), "kmx_malloc"); }
    DeviceBuffer(const Context& c, const T* host, size_t n) : DeviceBuffer(c, n) { upload(host, n); }
    ~DeviceBuffer() { kmx_free(c_.get(), p_); }
    DeviceBuffer(const DeviceBuffer&) = delete;
    T* data() const { return p_; }
    size_t size() const { return n_; }
    void upload(const T* host, size_t n) { c_.check(kmx_memcpy_h2d(c_.get(), p_, host, n * sizeof(T)), "h2d"); }
    std::vector<T> download() const {
        std::vector<T> v(n_);
        c_.check(kmx_memcpy_d2h(c_.get(), v.data(), p_, n_ * sizeof(T)), "d2h");
        return v;
    }

  private:
    const Context& c_;
    T* p_ = nullptr;
    size_t n_;
};

// ---------------------------------------------------------------------------------------------------
namespace naive_impl {

using Base = uint64_t;                       // src/naive_impl/mod.rs:20
constexpr Base A = 0, C = 1, G = 2, T = 3;   // :21-24

inline Base complement_base(Base b) { return 3 - b; }  // mod.rs:81-84
inline bool is_valid_nuc(Base b) { return b < 4; }     // mod.rs:87-89

// mod.rs:40-50: invalid -> u64::MAX.  Computed on the device (Kmer::from of a 1-mer, strict) like everything else.
inline Base encode_binary_u8(uint8_t c, Context& ctx = Context::instance()) {
    DeviceBuffer<uint8_t> s(ctx, &c, 1);
    DeviceBuffer<uint64_t> w(ctx, 1);
    uint64_t bad = 0;
    int st = kmx_kmers_from_bytes(ctx.get(), s.data(), 1, 1, w.data(), &bad);
    if (st == KMX_E_INVALID_BASE) return UINT64_MAX;
    ctx.check(st, "encode_binary_u8");
    return w.download()[0];
}
// mod.rs:27-37: panics on anything but ACGTacgt
inline Base encode_binary(char c, Context& ctx = Context::instance()) {
    Base b = encode_binary_u8(static_cast<uint8_t>(c), ctx);
    if (b == UINT64_MAX) throw Panic(KMX_E_INVALID_BASE, std::string("cannot decode ") + c + " into 2 bit encoding");
    return b;
}

// mod.rs:54-65: the complement code of a letter, u64::MAX for anything but ACGTacgt
inline Base encode_complement_binary_u8(uint8_t c, Context& ctx = Context::instance()) {
    const Base b = encode_binary_u8(c, ctx);
    return b == UINT64_MAX ? UINT64_MAX : complement_base(b);
}
// mod.rs:67-78: panics on anything but ACGTacgt
inline Base encode_complement_binary(char c, Context& ctx = Context::instance()) { return complement_base(encode_binary(c, ctx)); }

enum class MatchType { NoMatch = KMX_NO_MATCH, IdentityMatch = KMX_IDENTITY_MATCH, TwinMatch = KMX_TWIN_MATCH };  // canonical_kmer.rs:7-12
enum class Orientation { IsCanonical, NotCanononical };  // kmer.rs:18-22 (spelling as in the reference)

struct Kmer {  // src/naive_impl/kmer.rs:6-10
    uint8_t k = 0;
    uint64_t data = 0;

    size_t len() const { return k; }
    bool is_empty() const { return k == 0; }
    uint64_t into_u64() const { return data; }
    // kmer.rs:45-48 (MASK_TABLE[32] == 0, kmer.rs:617, reproduced)
    static Kmer from_u64(uint64_t data, uint8_t k) {
        const uint64_t mask = k >= 32 ? 0 : ((1ull << (2 * k)) - 1);
        return Kmer{k, data & mask};
    }
    // kmer.rs:209-257 From<&[u8]> / From<&str>: panics if len > 32 or on a non-ACGTacgt byte
    static Kmer from(const std::string& s, Context& ctx = Context::instance()) { return from(reinterpret_cast<const uint8_t*>(s.data()), s.size(), ctx); }
    static Kmer from(const uint8_t* s, size_t n, Context& ctx = Context::instance()) {
        if (n > 32) throw Panic(KMX_E_TOO_LONG, "kmers longer than 32 bases not supported");
        if (n == 0) return Kmer{0, 0};
        DeviceBuffer<uint8_t> d(ctx, s, n);
        DeviceBuffer<uint64_t> w(ctx, 1);
        uint64_t bad = 0;
        int st = kmx_kmers_from_bytes(ctx.get(), d.data(), 1, static_cast<uint32_t>(n), w.data(), &bad);
        if (st != KMX_OK) throw Panic(st, "Kmer::from");
        return Kmer{static_cast<uint8_t>(n), w.download()[0]};
    }
    // kmer.rs:124-136
    Kmer to_reverse_complement(Context& ctx = Context::instance()) const {
        DeviceBuffer<uint64_t> in(ctx, &data, 1), out(ctx, 1);
        ctx.check(kmx_revcomp_words(ctx.get(), in.data(), 1, k, out.data()), "to_reverse_complement");
        return Kmer{k, out.download()[0]};
    }
    static uint64_t get_reverse_complement_word(uint64_t w, uint8_t k, Context& ctx = Context::instance()) { return Kmer{k, w}.to_reverse_complement(ctx).data; }
    // kmer.rs:55-74
    bool is_canonical(Context& ctx = Context::instance()) const {
        DeviceBuffer<uint64_t> in(ctx, &data, 1);
        DeviceBuffer<uint8_t> f(ctx, 1);
        ctx.check(kmx_canonical_words(ctx.get(), in.data(), 1, k, nullptr, f.data()), "is_canonical");
        return f.download()[0] != 0;
    }
    Orientation orientation(Context& ctx = Context::instance()) const { return is_canonical(ctx) ? Orientation::IsCanonical : Orientation::NotCanononical; }
    Kmer to_canonical(Context& ctx = Context::instance()) const {
        DeviceBuffer<uint64_t> in(ctx, &data, 1), out(ctx, 1);
        ctx.check(kmx_canonical_words(ctx.get(), in.data(), 1, k, out.data(), nullptr), "to_canonical");
        return Kmer{k, out.download()[0]};
    }
    // kmer.rs:91-102: return the shifted-off base.  (The lone-Kmer ops run the paired kernel with a scratch twin.)
    Base append_base(Base c, Context& ctx = Context::instance()) { return shift(true, c, ctx); }
    Base prepend_base(Base c, Context& ctx = Context::instance()) { return shift(false, c, ctx); }
    Base append_base_u8(uint8_t c, Context& ctx = Context::instance()) { return shift(true, encode_binary_u8(c, ctx), ctx); }
    Base prepend_base_u8(uint8_t c, Context& ctx = Context::instance()) { return shift(false, encode_binary_u8(c, ctx), ctx); }
    // kmer.rs:12-16 and the derived Ord (k, then data; numeric, kmer.rs:319-322)
    bool operator==(const Kmer& o) const { return data == o.data && k == o.k; }
    bool operator!=(const Kmer& o) const { return !(*this == o); }
    bool operator<(const Kmer& o) const { return k != o.k ? k < o.k : data < o.data; }
    bool operator<=(const Kmer& o) const { return !(o < *this); }
    // kmer.rs:196-207: lower-case letters, first base in the lowest bits
    std::string to_string() const {
        static const char base_table[4] = {'a', 'c', 'g', 't'};
        std::string s(k, 'a');
        uint64_t w = data;
        for (unsigned i = 0; i < k; ++i, w >>= 2) s[i] = base_table[w & 3];
        return s;
    }
    // kmer.rs:151-162: panics (assert) unless pos < k and pos + width <= k
    static uint64_t sub_kmer_word(uint64_t word, size_t k, size_t pos, size_t width) {
        if (!(pos < k) || !(pos + width <= k)) throw Panic(KMX_E_ARG, "sub_kmer_word: assertion failed");
        word >>= 2 * pos;
        return width >= 32 ? word : (word & ((uint64_t{1} << (2 * width)) - 1));
    }
    Kmer sub_kmer(size_t pos, size_t width) const { return from_u64(sub_kmer_word(data, k, pos, width), static_cast<uint8_t>(width)); }
    // kmer.rs:164-192 with the in-crate deterministic hasher: (minimizer, offset), leftmost of equal hashes
    std::pair<Kmer, size_t> minimizer(size_t width, size_t lex_hasher_k, Context& ctx = Context::instance()) const {
        DeviceBuffer<uint64_t> in(ctx, &data, 1), mm(ctx, 1);
        DeviceBuffer<uint32_t> off(ctx, 1);
        int st = kmx_minimizer_words(ctx.get(), in.data(), 1, k, static_cast<uint32_t>(width), KMX_HASH_LEX,
                                     static_cast<uint32_t>(lex_hasher_k), mm.data(), off.data());
        if (st == KMX_E_K_RANGE) throw Panic(st, "Kmer::minimizer: assertion failed: pos + width <= k");
        ctx.check(st, "Kmer::minimizer");
        return {from_u64(mm.download()[0], static_cast<uint8_t>(width)), off.download()[0]};
    }

  private:
    Base shift(bool append, Base c, Context& ctx) {
        uint64_t twin = 0;
        const uint8_t code = static_cast<uint8_t>(c & 3);
        DeviceBuffer<uint64_t> f(ctx, &data, 1), r(ctx, &twin, 1);
        DeviceBuffer<uint8_t> b(ctx, &code, 1), d(ctx, 1);
        // the twin is shifted the opposite way; lone append == the fw half of CanonicalKmer::append_base
        int st = append ? kmx_ck_append_bases(ctx.get(), f.data(), r.data(), b.data(), 1, k, d.data())
                        : kmx_ck_prepend_bases(ctx.get(), f.data(), r.data(), b.data(), 1, k, d.data());
        ctx.check(st, append ? "append_base" : "prepend_base");
        data = f.download()[0];
        return d.download()[0];
    }
};

struct CanonicalKmer {  // src/naive_impl/canonical_kmer.rs:14-18
    Kmer fw, rc;

    static CanonicalKmer blank_of_size(uint8_t k) { return CanonicalKmer{Kmer{k, 0}, Kmer{k, UINT64_MAX}}; }  // :22-29
    static CanonicalKmer from_u64(uint64_t data, uint8_t k, Context& ctx = Context::instance()) {              // :42-51
        Kmer f = Kmer::from_u64(data, k);
        return CanonicalKmer{f, f.to_reverse_complement(ctx)};
    }
    static CanonicalKmer from(const Kmer& km, Context& ctx = Context::instance()) { return CanonicalKmer{km, km.to_reverse_complement(ctx)}; }  // :164-172
    static CanonicalKmer from(const std::string& s, Context& ctx = Context::instance()) { return from(Kmer::from(s, ctx), ctx); }              // :175-196
    bool is_empty() const { return fw.is_empty(); }
    size_t len() const { return fw.len(); }
    void swap() { std::swap(fw.data, rc.data); }                         // :62-64
    bool is_fw_canonical() const { return fw.data < rc.data; }           // :67-69
    Base append_base(Base b, Context& ctx = Context::instance()) { return shift(true, b, ctx); }     // :90-94
    Base prepend_base(Base b, Context& ctx = Context::instance()) { return shift(false, b, ctx); }   // :97-101
    Base append_base_u8(uint8_t c, Context& ctx = Context::instance()) { return shift(true, encode_binary_u8(c, ctx), ctx); }   // :72-78
    Base prepend_base_u8(uint8_t c, Context& ctx = Context::instance()) { return shift(false, encode_binary_u8(c, ctx), ctx); } // :81-87
    Kmer get_canonical_kmer() const { return fw.data < rc.data ? fw : rc; }        // :104-110
    uint64_t get_canonical_word() const { return fw.data < rc.data ? fw.data : rc.data; }  // :113-119 (a select of two device results)
    Kmer get_fw_mer() const { return fw; }
    Kmer get_rc_mer() const { return rc; }
    uint64_t get_fw_word() const { return fw.data; }
    uint64_t get_rc_word() const { return rc.data; }
    MatchType get_word_equivalency(uint64_t other, Context& ctx = Context::instance()) const {   // :152-161
        DeviceBuffer<uint64_t> f(ctx, &fw.data, 1), r(ctx, &rc.data, 1), o(ctx, &other, 1);
        DeviceBuffer<uint8_t> m(ctx, 1);
        ctx.check(kmx_match_words(ctx.get(), f.data(), r.data(), o.data(), 1, m.data()), "get_word_equivalency");
        return static_cast<MatchType>(m.download()[0]);
    }
    MatchType get_kmer_equivalency(const Kmer& other, Context& ctx = Context::instance()) const { return get_word_equivalency(other.data, ctx); }  // :142-150
    bool operator==(const CanonicalKmer& o) const { return fw == o.fw && rc == o.rc; }

  private:
    Base shift(bool append, Base b, Context& ctx) {
        const uint8_t code = static_cast<uint8_t>(b & 3);
        DeviceBuffer<uint64_t> f(ctx, &fw.data, 1), r(ctx, &rc.data, 1);
        DeviceBuffer<uint8_t> bb(ctx, &code, 1), d(ctx, 1);
        int st = append ? kmx_ck_append_bases(ctx.get(), f.data(), r.data(), bb.data(), 1, fw.k, d.data())
                        : kmx_ck_prepend_bases(ctx.get(), f.data(), r.data(), bb.data(), 1, fw.k, d.data());
        ctx.check(st, "CanonicalKmer shift");
        fw.data = f.download()[0];
        rc.data = r.download()[0];
        return d.download()[0];
    }
};

struct CanonicalKmerPos {  // canonical_kmer_iterator.rs:13-16
    CanonicalKmer km;
    int32_t pos = -1;
};

// canonical_kmer_iterator.rs:32-117.  The whole read is scanned by ONE kmx_canonical_windows launch at
// construction; inc()/inc_by()/get()/exhausted() then walk the device results exactly like the reference
// iterator walks the read (windows containing a non-ACGTacgt byte are skipped).
class CanonicalKmerIterator {
  public:
    static CanonicalKmerIterator from_u8_slice(const uint8_t* s, size_t n, uint8_t k, Context& ctx = Context::instance()) {
        CanonicalKmerIterator it;
        it.k_ = k;
        it.value_.km = CanonicalKmer::blank_of_size(k);
        const size_t nwin = n >= k ? n - k + 1 : 0;
        if (nwin) {
            DeviceBuffer<uint8_t> d(ctx, s, n);
            DeviceBuffer<uint64_t> fw(ctx, nwin), rc(ctx, nwin);
            DeviceBuffer<uint8_t> fl(ctx, nwin);
            kmx_reads r{d.data(), 1, static_cast<uint32_t>(n), nullptr};
            ctx.check(kmx_canonical_windows(ctx.get(), &r, nullptr, k, fw.data(), rc.data(), nullptr, fl.data()), "CanonicalKmerIterator");
            ctx.sync();
            it.fw_ = fw.download();
            it.rc_ = rc.download();
            it.flags_ = fl.download();
        }
        it.cursor_ = -1;
        it.advance();
        return it;
    }
    static CanonicalKmerIterator from_u8_slice(const std::string& s, uint8_t k, Context& ctx = Context::instance()) {
        return from_u8_slice(reinterpret_cast<const uint8_t*>(s.data()), s.size(), k, ctx);
    }
    bool exhausted() const { return invalid_; }  // :89-91
    bool inc() {                                 // :94-101
        if (!invalid_) advance();
        return !invalid_;
    }
    bool inc_by(size_t count) {                  // :104-111
        bool v = !invalid_;
        while (count > 0 && v) {
            v = inc();
            --count;
        }
        return v;
    }
    const CanonicalKmerPos& get() const { return value_; }  // :114-116

  private:
    void advance() {
        int64_t p = cursor_ + 1;
        while (p < static_cast<int64_t>(flags_.size()) && !(flags_[p] & KMX_WIN_VALID)) ++p;
        if (p >= static_cast<int64_t>(flags_.size())) {
            invalid_ = true;  // ran off the end: the last yielded state stays in place, like the reference
            return;
        }
        cursor_ = p;
        value_.km.fw = Kmer{k_, fw_[p]};
        value_.km.rc = Kmer{k_, rc_[p]};
        value_.pos = static_cast<int32_t>(p);
    }
    uint8_t k_ = 0;
    std::vector<uint64_t> fw_, rc_;
    std::vector<uint8_t> flags_;
    int64_t cursor_ = -1;
    bool invalid_ = false;
    CanonicalKmerPos value_;
};

namespace hash {
struct LexHasherState {  // hash.rs:22-36
    size_t k;
    explicit LexHasherState(size_t k_) : k(k_) {}
};
// hash_one(&state, kmer) with impl Hash for Kmer = write_u64(data) (hash.rs:4-20, 60-71)
inline uint64_t hash_one(const LexHasherState& st, const Kmer& km, Context& ctx = Context::instance()) {
    DeviceBuffer<uint64_t> in(ctx, &km.data, 1), out(ctx, 1);
    ctx.check(kmx_hash_words(ctx.get(), in.data(), 1, KMX_HASH_LEX, static_cast<uint32_t>(st.k), out.data()), "hash_one");
    return out.download()[0];
}
// hash_one with one of std's BuildHashers (hash.rs:10-20; kmer.rs:546-575): std's DefaultHasher is SipHash-1-3; DefaultHasher::new() has
// the keys (0, 0), a RandomState its own random pair
struct SipHasher13State {
    uint64_t key0 = 0, key1 = 0;
};
inline uint64_t hash_one(const SipHasher13State& st, const Kmer& km, Context& ctx = Context::instance()) {
    DeviceBuffer<uint64_t> in(ctx, &km.data, 1), out(ctx, 1);
    ctx.check(kmx_hash_words_sip13(ctx.get(), in.data(), 1, st.key0, st.key1, out.data()), "hash_one");
    return out.download()[0];
}
}  // namespace hash

// Batch form of the streaming loop (what the GPU is for): summary over many reads resident on the device.
inline kmx_summary canonical_reduce(Context& ctx, const kmx_reads& reads, uint32_t k, uint32_t hasher = KMX_HASH_NONE,
                                    uint32_t hasher_k = 0, uint32_t flags = 0) {
    // (the summary straight into host memory: one kernel launch for clean uniform reads, no device allocation, no copy -- kmx.h, round 6)
    kmx_summary out{};
    ctx.check(kmx_canonical_reduce_host(ctx.get(), &reads, k, hasher, hasher_k, flags, &out), "canonical_reduce");
    return out;
}

// FASTA / FASTQ ingestion (build-defined, the reference has no parser): a file image -> the reads
// back to back on the device + their offsets, i.e. the ragged `kmx_reads` the calls above take.
class FastxReads {
  public:
    FastxReads(const std::string& text, Context& ctx = Context::instance(), uint32_t format = KMX_FASTX_AUTO) : ctx_(&ctx) {
        DeviceBuffer<uint8_t> d(ctx, text.size() ? text.size() : 1);
        if (!text.empty()) d.upload(reinterpret_cast<const uint8_t*>(text.data()), text.size());
        uint64_t nr = 0, nb = 0;
        ctx.check(kmx_fastx_parse(ctx.get(), d.data(), text.size(), format, nullptr, nullptr, 0, &nr, &nb), "fastx_parse (count)");
        bases_.reset(new DeviceBuffer<uint8_t>(ctx, nb ? nb : 1));
        offsets_.reset(new DeviceBuffer<uint64_t>(ctx, nr + 1));
        // (the emit reuses the chunk summaries of the counting call just made on the same image)
        ctx.check(kmx_fastx_parse(ctx.get(), d.data(), text.size(), format | KMX_FASTX_SAME_TEXT, bases_->data(), offsets_->data(), nr, &nr, &nb),
                  "fastx_parse");
        n_reads_ = nr;
        n_bases_ = nb;
        uint32_t mn = 0, mx = 0;
        ctx.check(kmx_reads_length_range(ctx.get(), offsets_->data(), n_reads_, &mn, &mx), "reads_length_range");
        min_len_ = mn;
        max_len_ = mx;
    }
    // shortest / longest read; uniform(): every read has the same length (hand such a batch over with d_offsets = NULL)
    uint32_t min_len() const { return min_len_; }
    uint32_t max_len() const { return max_len_; }
    bool uniform() const { return n_reads_ != 0 && min_len_ == max_len_; }
    // the batch the way the scan kernels like it best: uniform layout if it is uniform, else ragged with the tight length bound
    kmx_reads best_reads() const {
        kmx_reads r = reads(max_len_);
        if (uniform()) r.d_offsets = nullptr;
        return r;
    }
    size_t len() const { return n_reads_; }
    size_t n_bases() const { return n_bases_; }
    // read_len_bound: 0 = unknown; an upper bound of the read lengths lets the bit-sliced kernel size its frame
    kmx_reads reads(uint32_t read_len_bound = 0) const {
        kmx_reads r{};
        r.d_bases = bases_->data();
        r.n_reads = n_reads_;
        r.read_len = read_len_bound;
        r.d_offsets = offsets_->data();
        return r;
    }
    std::vector<uint64_t> offsets() const { return offsets_->download(); }
    std::string read(size_t i) const {
        auto off = offsets();
        auto all = bases_->download();
        return std::string(all.begin() + off.at(i), all.begin() + off.at(i + 1));
    }

  private:
    Context* ctx_;
    std::unique_ptr<DeviceBuffer<uint8_t>> bases_;
    std::unique_ptr<DeviceBuffer<uint64_t>> offsets_;
    size_t n_reads_ = 0, n_bases_ = 0;
    uint32_t min_len_ = 0, max_len_ = 0;
};

// src/naive_impl/seq_vector.rs: the 2-bit packed sequence container, resident on the device.
class SeqVector {
  public:
    explicit SeqVector(size_t capacity_bases = 0, Context& ctx = Context::instance()) : ctx_(&ctx) { reserve(capacity_bases); }   // with_capacity, :231-235
    explicit SeqVector(const std::string& s, Context& ctx = Context::instance()) : ctx_(&ctx) { push_chars(s); }                 // From<&String>, :323-329
    size_t len() const { return n_; }              // :208-210
    bool is_empty() const { return n_ == 0; }      // :212-214
    void push_chars(const std::string& bytes) {    // :141-161 (panics on a non-ACGTacgt byte like Kmer::from)
        if (bytes.empty()) return;
        reserve(n_ + bytes.size());
        DeviceBuffer<uint8_t> d(*ctx_, reinterpret_cast<const uint8_t*>(bytes.data()), bytes.size());
        uint64_t bad = 0;
        int st = kmx_seqvec_push_chars(ctx_->get(), words_->data(), n_, d.data(), bytes.size(), &bad);
        if (st == KMX_E_INVALID_BASE) throw Panic(st, "SeqVector::push_chars: invalid base at " + std::to_string(bad));
        ctx_->check(st, "SeqVector::push_chars");
        n_ += bytes.size();
    }
    uint64_t get_kmer_u64(size_t pos, size_t k) const {   // :96-99 (assert!(pos < len))
        DeviceBuffer<uint64_t> p(*ctx_, 1), o(*ctx_, 1);
        const uint64_t pp = pos;
        p.upload(&pp, 1);
        int st = kmx_seqvec_get_kmers(ctx_->get(), words_->data(), n_, p.data(), 1, static_cast<uint32_t>(k), o.data());
        if (st == KMX_E_ARG) throw Panic(st, "SeqVector::get_kmer_u64: assertion failed: pos < self.len()");
        ctx_->check(st, "SeqVector::get_kmer_u64");
        return o.download()[0];
    }
    Kmer get_kmer(size_t pos, size_t k) const { return Kmer::from_u64(get_kmer_u64(pos, k), static_cast<uint8_t>(k)); }   // :212-215
    uint64_t get_base(size_t pos) const { return get_kmer_u64(pos, 1); }                                                 // :222-224
    std::vector<uint64_t> iter_kmers(size_t k, size_t start = 0, size_t end = SIZE_MAX) const {   // :117-124 (+ slice, :226-234)
        if (end == SIZE_MAX) end = n_;
        const size_t cnt = end - start >= k ? end - start - k + 1 : 0;
        if (cnt == 0) return {};
        DeviceBuffer<uint64_t> o(*ctx_, cnt);
        ctx_->check(kmx_seqvec_iter_kmers(ctx_->get(), words_->data(), n_, start, end, static_cast<uint32_t>(k), o.data()), "SeqVector::iter_kmers");
        return o.download();
    }
    std::string to_string() const {   // String::from(&SeqVector), :171-182
        if (n_ == 0) return {};
        DeviceBuffer<uint8_t> o(*ctx_, n_);
        ctx_->check(kmx_seqvec_to_bytes(ctx_->get(), words_->data(), n_, o.data()), "SeqVector::to_string");
        auto v = o.download();
        return std::string(v.begin(), v.end());
    }
    // canonical k-mer summary of the reads stored back to back as read_len-base slices (the batch form of the hot path)
    kmx_summary canonical_reduce(size_t read_len, uint32_t k, uint32_t hasher = KMX_HASH_NONE, uint32_t hasher_k = 0, uint32_t flags = 0) const {
        DeviceBuffer<kmx_summary> out(*ctx_, 1);
        ctx_->check(kmx_seqvec_canonical_reduce(ctx_->get(), words_ ? words_->data() : nullptr, read_len ? n_ / read_len : 0,
                                                static_cast<uint32_t>(read_len), k, hasher, hasher_k, flags, out.data()), "SeqVector::canonical_reduce");
        return out.download()[0];
    }
    struct MappedMinimizer {   // seq_vector/minimizers.rs:21-37
        uint64_t word;
        size_t pos;
        uint64_t as_u64() const { return word; }
        bool operator==(const MappedMinimizer& o) const { return word == o.word && pos == o.pos; }
    };
    // iter_minimizers(k, w, LexHasherState::new(lex_hasher_k)) collected (seq_vector.rs:126-133; minimizers.rs:39-141)
    std::vector<MappedMinimizer> iter_minimizers(size_t k, size_t w, size_t lex_hasher_k) const {
        if (n_ < k) throw Panic(KMX_E_ARG, "SeqVecMinimizerIter::new: assertion failed: sv.len() >= k");
        const size_t cnt = n_ - k + 1;
        DeviceBuffer<uint64_t> mw(*ctx_, cnt);
        DeviceBuffer<uint32_t> mp(*ctx_, cnt);
        ctx_->check(kmx_seqvec_minimizers(ctx_->get(), words_->data(), 1, static_cast<uint32_t>(n_), static_cast<uint32_t>(k),
                                          static_cast<uint32_t>(w), KMX_HASH_LEX, static_cast<uint32_t>(lex_hasher_k), mw.data(), mp.data()),
                    "SeqVector::iter_minimizers");
        auto a = mw.download();
        auto b = mp.download();
        std::vector<MappedMinimizer> out(cnt);
        for (size_t i = 0; i < cnt; ++i) out[i] = MappedMinimizer{a[i], b[i]};
        return out;
    }
    const uint64_t* device_words() const { return words_ ? words_->data() : nullptr; }

  private:
    void reserve(size_t bases) {
        const size_t need = (bases + 31) / 32;
        if (words_ && words_->size() >= need) return;
        const size_t cap = need > 2 * (words_ ? words_->size() : 0) ? need : 2 * words_->size();
        auto nw = std::make_unique<DeviceBuffer<uint64_t>>(*ctx_, cap ? cap : 1);
        std::vector<uint64_t> host = words_ ? words_->download() : std::vector<uint64_t>();
        host.resize(cap ? cap : 1, 0);
        nw->upload(host.data(), host.size());
        words_ = std::move(nw);
    }
    Context* ctx_;
    std::unique_ptr<DeviceBuffer<uint64_t>> words_;
    size_t n_ = 0;
};

}  // namespace naive_impl

// ---------------------------------------------------------------------------------------------------
namespace encoding {

// src/encoding/naive.rs:48-74: the discriminant byte IS the base->code map (A bits 7:6, C 5:4, T 3:2, G 1:0)
enum class Naive : uint8_t {
    ACTG = 0b00011011, ACGT = 0b00011110, ATCG = 0b00100111, ATGC = 0b00110110, AGCT = 0b00101101, AGTC = 0b00111001,
    CATG = 0b01001011, CAGT = 0b01001110, CTAG = 0b10000111, CTGA = 0b11000110, CGAT = 0b10001101, CGTA = 0b11001001,
    TACG = 0b01100011, TAGC = 0b01110010, TCAG = 0b10010011, TCGA = 0b11010010, TGAC = 0b10110001, TGCA = 0b11100001,
    GACT = 0b01101100, GATC = 0b01111000, GCAT = 0b10011100, GCTA = 0b11011000, GTAC = 0b10110100, GTCA = 0b11100100,
};
struct Xor10 {};  // src/encoding/xor10.rs: the same map as Naive::ACTG (xor10.rs:17-22 vs naive.rs:50)

inline uint8_t enc_byte(Naive e) { return static_cast<uint8_t>(e); }
inline uint8_t enc_byte(Xor10) { return static_cast<uint8_t>(Naive::ACTG); }

// trait Encoding<u64, B> (src/encoding/mod.rs:14-23) for P = u64
template <size_t B, typename E>
std::array<uint64_t, B> encode(const E& enc, const std::string& seq, Context& ctx = Context::instance()) {  // naive.rs:116-124
    std::array<uint64_t, B> out{};
    if (seq.size() > 32 * B) throw Panic(KMX_E_TOO_LONG, "Encoding::encode: sequence longer than the k-mer storage");
    if (seq.empty()) return out;
    DeviceBuffer<uint8_t> d(ctx, reinterpret_cast<const uint8_t*>(seq.data()), seq.size());
    DeviceBuffer<uint64_t> w(ctx, B);
    ctx.check(kmx_encode_kmers(ctx.get(), d.data(), 1, static_cast<uint32_t>(seq.size()), enc_byte(enc), B, w.data()), "Encoding::encode");
    auto v = w.download();
    std::copy(v.begin(), v.end(), out.begin());
    return out;
}
template <size_t B, typename E>
std::string decode(const E& enc, const std::array<uint64_t, B>& a, Context& ctx = Context::instance()) {  // naive.rs:126-136 (ALL 32*B letters)
    DeviceBuffer<uint64_t> w(ctx, a.data(), B);
    DeviceBuffer<uint8_t> s(ctx, 32 * B);
    ctx.check(kmx_encoding_decode(ctx.get(), w.data(), 1, enc_byte(enc), B, s.data()), "Encoding::decode");
    auto v = s.download();
    return std::string(v.begin(), v.end());
}
template <size_t K, size_t B, typename E>
std::array<uint64_t, B> rev_comp(const E& enc, const std::array<uint64_t, B>& a, Context& ctx = Context::instance()) {  // naive.rs:138-154
    static_assert(K >= 2 && K <= 32 * B, "rev_comp::<K>: K=1 underflows usize in the reference; K must fit the storage");
    DeviceBuffer<uint64_t> w(ctx, a.data(), B), o(ctx, B);
    ctx.check(kmx_encoding_rev_comp(ctx.get(), w.data(), 1, K, enc_byte(enc), B, o.data()), "Encoding::rev_comp");
    std::array<uint64_t, B> out{};
    auto v = o.download();
    std::copy(v.begin(), v.end(), out.begin());
    return out;
}
}  // namespace encoding

// ---------------------------------------------------------------------------------------------------
namespace kmer {

// src/kmer.rs:67-69 (P given by its size in bytes)
template <size_t PBytes, size_t K>
constexpr size_t word_for_k() { return (PBytes * 8 / 2 + K - 1) / (PBytes * 8 / 2); }

// src/kmer.rs:12-53 for P = u64
template <size_t K, size_t B = word_for_k<8, K>()>
struct Kmer {
    std::array<uint64_t, B> array{};  // Default = zeroed (kmer.rs:55-64)

    template <typename E>
    static Kmer new_(const std::string& seq, const E& enc, Context& ctx = Context::instance()) { return Kmer{encoding::encode<B>(enc, seq, ctx)}; }  // :21-28
    static Kmer with_data(const std::array<uint64_t, B>& d) { return Kmer{d}; }                                                                      // :31-33
    size_t k() const { return K; }                                                                                                                 // :36-38
    size_t num_bytes() const { return sizeof(uint64_t) * word_for_k<8, K>(); }                                                                      // :41-43
    uint64_t get(size_t index) const { return (array[(index * 2) / 64] >> ((index * 2) % 64)) & 3; }                                                // :46-48 (2-bit field read)
    uint64_t get_prefix(size_t len) const {                                                                                                        // :50-52: bits 0..=2*len (INCLUSIVE)
        const size_t nbits = 2 * len + 1;
        if (nbits > 64) throw Panic(KMX_E_ARG, "get_prefix: range longer than P");
        return nbits == 64 ? array[0] : (array[0] & ((1ull << nbits) - 1));
    }
};

// src/kmer.rs:71-91
inline std::string bitmer_to_bytes(uint64_t mer, size_t len) {
    static const char t[4] = {'A', 'C', 'G', 'T'};
    std::string s(len, 'A');
    for (size_t i = 0; i < len; ++i, mer >>= 2) s[i] = t[mer & 3];
    return s;
}
}  // namespace kmer

}  // namespace kmx
