// C++ host-layer test: reads like the reference's inline test modules (cited per block), every value
// computed by the libkmx HIP kernels through include/kmx.hpp.  Exit code 0 = all passed.
#include <cstdio>
#include <string>

#include "kmx.hpp"

using namespace kmx;
using namespace kmx::naive_impl;

static int fails = 0;
#define CHECK(cond) do { if (!(cond)) { std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond); ++fails; } } while (0)
template <typename F> static bool panics(F f) { try { f(); } catch (const Panic&) { return true; } return false; }

int main() {
    // ---- src/naive_impl/kmer.rs tests
    {   // test_into_canon / test_is_canon (:292-317)
        CHECK(Kmer::from("taa").to_canonical() == Kmer::from("taa"));
        CHECK(Kmer::from("tta").to_canonical() == Kmer::from("taa"));
        CHECK(Kmer::from("atc").to_canonical() == Kmer::from("atc"));
        CHECK(Kmer::from("gat").to_canonical() == Kmer::from("atc"));
        Kmer not_canon = Kmer::from("gatacataggatgg");
        CHECK(not_canon.to_reverse_complement() == not_canon.to_canonical());
        CHECK(Kmer::from("agatacataggatgg").is_canonical());
        CHECK(!Kmer::from("gatacataggatgg").is_canonical());
    }
    CHECK(Kmer::from("tcc") < Kmer::from("cct"));  // test_ord (:319-322)
    {   // test_append / test_prepend (:324-384)
        Kmer k1 = Kmer::from("att");
        CHECK(k1.append_base_u8('c') == A && k1 == Kmer::from("ttc"));
        k1 = Kmer::from("ttcga");
        CHECK(k1.append_base_u8('g') == T && k1 == Kmer::from("tcgag"));
        k1 = Kmer::from("att");
        CHECK(k1.append_base(encode_binary_u8('c')) == A && k1 == Kmer::from("ttc"));
        k1 = Kmer::from("att");
        CHECK(k1.prepend_base_u8('c') == T && k1 == Kmer::from("cat"));
        k1 = Kmer::from("ttcga");
        CHECK(k1.prepend_base_u8('g') == A && k1 == Kmer::from("gttcg"));
    }
    {   // test_rc (:386-424)
        CHECK((Kmer{1, 0}.to_reverse_complement() == Kmer::from("t")));
        CHECK(Kmer::from("aaa").to_reverse_complement() == Kmer::from("ttt"));
        CHECK(Kmer::from("ta").to_reverse_complement() == Kmer::from("ta"));
        CHECK(Kmer::from("ccg").to_reverse_complement() == Kmer::from("cgg"));
        CHECK(Kmer::from("aat").to_reverse_complement() == Kmer::from("att"));
        CHECK(Kmer::from("gatacataggatgg").to_reverse_complement() == Kmer::from("ccatcctatgtatc"));
    }
    CHECK(Kmer::from("catagatacat").to_string() == "catagatacat");  // str_repr (:426-431)
    CHECK(Kmer::from("aac").into_u64() == 0b010000 && Kmer::from("acc").into_u64() == 0b010100 && Kmer::from("ccc").into_u64() == 0b010101);  // bin_repr
    CHECK(Kmer::from("aaa") == Kmer::from_u64(0, 3));
    CHECK(Kmer::from("aaa") == Kmer::from("AAA") && Kmer::from("aCa") == Kmer::from("AcA") && Kmer::from("a") != Kmer::from("aa"));  // test_eq
    CHECK(panics([] { Kmer::from(std::string(33, 'a')); }));  // too_long (:476-480)
    CHECK(!panics([] { Kmer::from(std::string(32, 'a')); }));
    CHECK(panics([] { encode_binary('N'); }));                // encode_panics (:499-503)
    CHECK(encode_binary('A') == A && encode_binary('c') == C && encode_binary('G') == G && encode_binary('t') == T);
    CHECK(encode_binary_u8('N') == UINT64_MAX);               // mod.rs:48
    CHECK(complement_base(A) == T && complement_base(C) == G && is_valid_nuc(T) && !is_valid_nuc(5));

    // ---- src/naive_impl/canonical_kmer.rs tests (:243-297)
    {
        CanonicalKmer ck = CanonicalKmer::from("acttg");
        CHECK(ck.fw.to_string() == "acttg" && ck.rc.to_string() == "caagt");
        Kmer km = Kmer::from("acttg");
        CanonicalKmer ck2 = CanonicalKmer::from_u64(km.into_u64(), (uint8_t)km.len());
        CHECK(ck2 == ck);
        ck.swap();
        CHECK(ck.rc.to_string() == "acttg" && ck.fw.to_string() == "caagt");
        ck = CanonicalKmer::from("acttg");
        ck.append_base_u8('a');
        CHECK(ck.fw.to_string() == "cttga" && ck.rc.to_string() == "tcaag");
        ck.prepend_base_u8('c');
        CHECK(ck.rc.to_string() == "caagg" && ck.fw.to_string() == "ccttg");
        CanonicalKmer a = CanonicalKmer::from("acttg"), b = CanonicalKmer::from("caagt");
        CHECK(a.get_kmer_equivalency(b.get_fw_mer()) == MatchType::TwinMatch);
        b.swap();
        CHECK(a.get_kmer_equivalency(b.get_fw_mer()) == MatchType::IdentityMatch);
        b.append_base_u8('c');
        CHECK(a.get_kmer_equivalency(b.get_fw_mer()) == MatchType::NoMatch);
    }
    // ---- src/naive_impl/canonical_kmer_iterator.rs tests (:123-206)
    {
        const std::string r = "TTTTGGCCATTTTTCCTGTTCTTCAAGAAAACAGGAGATAACTAGAAGGACTAGAGAATGGGGCTGCCAGAACTAGTGGGAAGCTCCCTAGAAATGGTGACATCGCCCACCAAACAGACC";
        const uint8_t k = 31;
        auto it = CanonicalKmerIterator::from_u8_slice(r, k);
        CHECK(it.get().km == CanonicalKmer::from(r.substr(0, 31)) && it.get().pos == 0);   // test_iter_init
        it.inc();
        CHECK(it.get().km == CanonicalKmer::from(r.substr(1, 31)) && it.get().pos == 1);   // test_iter_inc
        it = CanonicalKmerIterator::from_u8_slice(r, k);
        it.inc_by(10);
        CHECK(it.get().km == CanonicalKmer::from(r.substr(10, 31)) && it.get().pos == 10); // test_iter_inc_by
        std::string rn = r.substr(0, 4) + "N" + r.substr(4);
        it = CanonicalKmerIterator::from_u8_slice(rn, k);
        CHECK(it.get().km == CanonicalKmer::from(rn.substr(5, 31)) && it.get().pos == 5);  // test_iter_init_invalid
        rn = r.substr(0, 35) + "N" + r.substr(35);
        it = CanonicalKmerIterator::from_u8_slice(rn, k);
        it.inc_by(5);
        CHECK(it.get().km == CanonicalKmer::from(rn.substr(36, 31)) && it.get().pos == 36); // test_iter_inc_by_invalid
        it = CanonicalKmerIterator::from_u8_slice(r, k);                                    // test_exhausted_works
        it.inc_by(20);
        CHECK(!it.exhausted());
        it.inc_by(r.size() - 20);
        CHECK(it.exhausted());
        it.inc();
        CHECK(it.exhausted());
        // short read: exhausted immediately, pos stays -1 (:50,69)
        it = CanonicalKmerIterator::from_u8_slice(std::string("ACGT"), k);
        CHECK(it.exhausted() && it.get().pos == -1);
    }
    // ---- src/naive_impl/hash.rs lex_order (:83-104)
    {
        hash::LexHasherState seed(3);
        CHECK(hash::hash_one(seed, Kmer::from("aaa")) == 0);
        CHECK(hash::hash_one(seed, Kmer::from("aac")) == 0b000001);
        CHECK(hash::hash_one(seed, Kmer::from("caa")) == 0b010000);
        CHECK(hash::hash_one(seed, Kmer::from("cac")) == 0b010001);
    }
    // ---- src/naive_impl/kmer.rs test_hash (:546-557): DefaultHasher over the k-mer == DefaultHasher over its data word -- one write_u64
    //      either way; on the device both are kmx_hash_words_sip13 of the same word, so what is left to check is that the hash is a
    //      function of the word and of the keys (SipHash-1-3 itself is pinned by the oracle tests)
    {
        const Kmer km = Kmer::from("ACTTGAT");
        const hash::SipHasher13State dflt{};   // DefaultHasher::new(): keys (0, 0)
        const uint64_t h1 = hash::hash_one(dflt, km), h2 = hash::hash_one(dflt, Kmer::from_u64(km.into_u64(), km.len()));
        CHECK(h1 == h2);
        CHECK(hash::hash_one(hash::SipHasher13State{1, 2}, km) != h1);
        CHECK(hash::hash_one(dflt, Kmer::from("ACTTGAA")) != h1);
    }
    // ---- src/encoding/naive.rs k45pu64 (:388-416), src/kmer.rs kmer_prefix / kmer_naive_encoder (:173-203)
    {
        using encoding::Naive;
        const std::string s = "TAAGGATTCTAATCATAAGGATTCTAATCATAAGGATTCTAATCA";
        auto a = encoding::encode<2>(Naive::ACGT, s);
        CHECK(a[0] == 3585846758293238403ull && a[1] == 7397160ull);
        CHECK(encoding::decode<2>(Naive::ACGT, a) == s + std::string(19, 'A'));
        CHECK(encoding::decode<2>(Naive::ACGT, encoding::rev_comp<45, 2>(Naive::ACGT, a)) ==
              "TGATTAGAATCCTTATGATTAGAATCCTTATGATTAGAATCCTTA" + std::string(19, 'A'));
        auto x = encoding::encode<2>(encoding::Xor10{}, s);   // xor10.rs commented KAT value (informational)
        CHECK(x[0] == 2414607732474225602ull && x[1] == 6330940ull);
        auto km = kmer::Kmer<31>::new_("GTAC", Naive::ACGT);
        CHECK(km.get_prefix(4) == 0b01001110 && kmer::bitmer_to_bytes(km.get_prefix(4), 4) == "GTAC");
        auto k4 = kmer::Kmer<4>::new_("ACTG", Naive::TAGC);
        CHECK(k4.get(0) == 0b01 && k4.get(1) == 0b11 && k4.get(2) == 0b00 && k4.get(3) == 0b10);
        CHECK((kmer::word_for_k<8, 32>() == 1) && (kmer::word_for_k<8, 64>() == 2) && (kmer::word_for_k<16, 65>() == 2));
        CHECK(kmer::Kmer<15>().num_bytes() == 8 && kmer::Kmer<15>().k() == 15);
        CHECK(panics([] { encoding::encode<1>(Naive::ACGT, std::string(33, 'A')); }));
    }
    // ---- src/naive_impl/seq_vector.rs tests (:364-428)
    {   // push_chars (:386-399)
        SeqVector sv(64);
        const std::string a30(30, 'A'), c40(40, 'C');
        sv.push_chars(a30);
        CHECK(sv.to_string() == a30 && sv.len() == 30);
        sv.push_chars(c40);
        CHECK(sv.len() == 70 && sv.to_string() == a30 + c40);
    }
    {   // iter_kmers (:401-416)
        SeqVector sv(std::string("ACTTGAT"));
        const char* mers[] = {"act", "ctt", "ttg", "tga", "gat"};
        auto all = sv.iter_kmers(3);
        CHECK(all.size() == 5);
        for (size_t i = 0; i < all.size() && i < 5; ++i) CHECK(Kmer::from_u64(all[i], 3).to_string() == mers[i]);
        auto mid = sv.iter_kmers(3, 1, sv.len() - 1);   // sv.slice(1, len-1).iter_kmers(3)
        CHECK(mid.size() == 3);
        for (size_t i = 0; i < mid.size() && i < 3; ++i) CHECK(Kmer::from_u64(mid[i], 3).to_string() == mers[i + 1]);
        CHECK(sv.get_kmer(2, 3) == Kmer::from("ttg") && sv.get_base(0) == A && sv.get_base(3) == T);
        CHECK(panics([&] { sv.get_kmer_u64(7, 1); }));                // assert!(pos < self.len())
        CHECK(panics([] { SeqVector bad(std::string("ACGNT")); }));   // Kmer::from panics on 'N'
    }
    {   // reads stored back to back: the packed scan equals the scan of the letters
        std::string reads;
        for (int r = 0; r < 130; ++r)
            for (int i = 0; i < 150; ++i) reads.push_back("ACGT"[(r * 7 + i * i + (i >> 3)) & 3]);
        SeqVector sv(reads);
        DeviceBuffer<uint8_t> d(Context::instance(), reinterpret_cast<const uint8_t*>(reads.data()), reads.size());
        kmx_reads rd{d.data(), 130, 150, nullptr};
        kmx_summary a = canonical_reduce(Context::instance(), rd, 31, KMX_HASH_LEX, 31, KMX_REDUCE_SUM_FW);
        kmx_summary b = sv.canonical_reduce(150, 31, KMX_HASH_LEX, 31, KMX_REDUCE_SUM_FW);
        CHECK(a.n_valid == b.n_valid && a.sum_canon == b.sum_canon && a.xor_hash == b.xor_hash && a.sum_fw == b.sum_fw && b.n_valid == 130u * 120u);
    }
    // ---- src/naive_impl/seq_vector/minimizers.rs tests (:237-290) and kmer.rs test_minimizer (:560-580)
    {
        using MM = SeqVector::MappedMinimizer;
        const uint64_t aac = 0b010000, acc = 0b010100, aaa = 0, aca = 0b000100;
        CHECK((SeqVector(std::string("AAACAAA")).iter_minimizers(6, 3, 6) == std::vector<MM>{{0, 0}, {0, 4}}));                      // mmers0
        CHECK((SeqVector(std::string("AACCAAA")).iter_minimizers(5, 3, 5) == std::vector<MM>{{aac, 0}, {acc, 1}, {aaa, 4}}));          // mmers1
        CHECK((SeqVector(std::string("CACACACCAC")).iter_minimizers(7, 3, 3) == std::vector<MM>{{aca, 1}, {aca, 1}, {aca, 3}, {aca, 3}}));  // mmers2
        CHECK((SeqVector(std::string("AAAAAAA")).iter_minimizers(5, 3, 3) == std::vector<MM>{{0, 0}, {0, 1}, {0, 2}}));               // leftmost_mmer
        const std::string s = "ACTTGAT";
        Kmer km = Kmer::from(s);
        for (size_t w = 1; w < s.size(); ++w) {
            auto [mm, o] = km.minimizer(w, 7);
            CHECK(mm == Kmer::from(s.substr(o, w)));
            for (size_t i = 0; i + w <= s.size(); ++i)
                CHECK(hash::hash_one(hash::LexHasherState(7), mm) <= hash::hash_one(hash::LexHasherState(7), km.sub_kmer(i, w)));
        }
        CHECK(Kmer::sub_kmer_word(Kmer::from("acttgat").data, 7, 2, 3) == Kmer::from("ttg").data);
        CHECK(panics([&] { km.sub_kmer(5, 3); }));
    }
    // ---- mod.rs:54-78 complement encoders; benches/simple_benchmark.rs:14-56 consumer shapes on the device
    {
        CHECK(encode_complement_binary('A') == T && encode_complement_binary('c') == G && encode_complement_binary_u8('g') == C &&
              encode_complement_binary_u8('T') == A && encode_complement_binary_u8('N') == UINT64_MAX);
        CHECK(panics([] { encode_complement_binary('N'); }));
        std::string b;
        for (int i = 0; i < 4096; ++i) b.push_back("ACGT"[(i * i + (i >> 2) + 3 * (i >> 5)) & 3]);
        constexpr uint32_t K = 31;
        Context& ctx = Context::instance();
        DeviceBuffer<uint8_t> d(ctx, reinterpret_cast<const uint8_t*>(b.data()), b.size());
        kmx_reads rd{d.data(), 1, static_cast<uint32_t>(b.size()), nullptr};   // b.windows(K) = the windows of one long read
        kmx_summary s = canonical_reduce(ctx, rd, K, KMX_HASH_NONE, 0, KMX_REDUCE_SUM_FW);
        uint64_t naive = 0, canon = 0;   // compute_naive / the canonical variant, one scalar k-mer at a time (the reference's shape)
        for (size_t i = 0; i + K <= b.size(); i += 97) {   // spot-check every 97th window against the scalar API
            Kmer km = Kmer::from(b.substr(i, K));
            naive += km.into_u64();
            canon += km.to_canonical().into_u64();
        }
        const size_t n_windows = b.size() - K + 1;
        CHECK(s.n_valid == n_windows);
        // full sums on the device in one batch call each: encode every window (Naive::ACGT == naive_impl) and add up on the host
        DeviceBuffer<uint64_t> w(ctx, n_windows);
        ctx.check(kmx_encode_windows(ctx.get(), &rd, K, static_cast<uint8_t>(encoding::Naive::ACGT), 1, w.data()), "encode_windows");
        uint64_t sum_acgt = 0;
        for (uint64_t x : w.download()) sum_acgt += x;
        CHECK(sum_acgt == s.sum_fw);                       // compute_naive (== rc_naive: the reverse complement is discarded there)
        ctx.check(kmx_encode_windows(ctx.get(), &rd, K, encoding::enc_byte(encoding::Xor10{}), 1, w.data()), "encode_windows xor10");
        uint64_t rc_xor10 = 0;
        for (uint64_t x : w.download()) rc_xor10 += x;     // rc_xor10 returns the sum of the ENCODED words (rev_comp's result is dropped)
        uint64_t chk = 0;
        for (size_t i = 0; i + K <= b.size(); i += 97) chk += encoding::encode<1>(encoding::Xor10{}, b.substr(i, K))[0];
        uint64_t chk2 = 0;
        {
            auto all = w.download();
            for (size_t i = 0; i + K <= b.size(); i += 97) chk2 += all[i];
        }
        CHECK(chk == chk2 && rc_xor10 != 0);
        const uint64_t compute_xor10 = n_windows * kmer::Kmer<K>().num_bytes();   // compute_xor10 sums num_bytes()
        CHECK(compute_xor10 == n_windows * 8);
        (void)naive;
        (void)canon;
    }
    // ---- FASTA / FASTQ ingestion (build-defined): file image -> ragged reads -> the streaming loop
    {
        FastxReads fq(std::string("@r1\nACGTTGCA\n+\nIIIIIIII\n@r2 x\nGGNACGT\n+r2\n@@@@@@@\n"));
        CHECK(fq.len() == 2 && fq.n_bases() == 15);
        CHECK(fq.read(0) == "ACGTTGCA" && fq.read(1) == "GGNACGT");
        CHECK((fq.offsets() == std::vector<uint64_t>{0, 8, 15}));
        kmx_summary s = canonical_reduce(Context::instance(), fq.reads(), 3);
        size_t n = 0;
        uint64_t sum = 0;
        for (const std::string r : {"ACGTTGCA", "GGNACGT"})
            for (auto it = CanonicalKmerIterator::from_u8_slice(reinterpret_cast<const uint8_t*>(r.data()), r.size(), 3); !it.exhausted(); it.inc()) {
                n += 1;
                sum += it.get().km.get_canonical_word();
            }
        CHECK(s.n_valid == n && s.sum_canon == sum && n == 6 + 2);
        CHECK(fq.min_len() == 7 && fq.max_len() == 8 && !fq.uniform() && fq.best_reads().d_offsets != nullptr && fq.best_reads().read_len == 8);
        FastxReads fu(std::string("@a\nACGTTGCA\n+\nIIIIIIII\n@b\nGGTACGTA\n+\nIIIIIIII\n"));
        CHECK(fu.uniform() && fu.max_len() == 8 && fu.best_reads().d_offsets == nullptr);
        kmx_summary su = canonical_reduce(Context::instance(), fu.best_reads(), 3), sr = canonical_reduce(Context::instance(), fu.reads(), 3);
        CHECK(su.n_valid == 12 && su.n_valid == sr.n_valid && su.sum_canon == sr.sum_canon);
        FastxReads fa(std::string(">s1 d\nACG\nTTA\n\n>s2\n>s3\r\nNN\r\nA"));
        CHECK(fa.len() == 3 && fa.read(0) == "ACGTTA" && fa.read(1).empty() && fa.read(2) == "NNA");
        CHECK(panics([] { FastxReads bad(std::string("ACGT\n")); }));
    }
    std::printf(fails ? "%d check(s) FAILED\n" : "all C++ host-layer checks passed\n", fails);
    return fails ? 1 : 0;
}
