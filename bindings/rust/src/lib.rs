//! Safe layer over libkmx that keeps the `kmers` crate's own surface.
//!
//! A GPU is a drop-in only at batch granularity, so every type here offers the crate's scalar interface (one k-mer
//! per call = a one-element batch: the correctness path, bit-identical to the CPU implementation) next to a batch
//! form (the throughput path).  Reference lines (`kmers` @ COMBINE-lab) are cited per item.
//!
//! * [`HipEncoder`]             — `impl Encoding<P, B>` for every `utils::Data` word type P = u8 .. u128 (src/encoding/mod.rs:14-23,
//!                                src/encoding/naive.rs:112-115, src/utils.rs:4-24) and every `Naive` map (+ Xor10 = `Naive::ACTG`)
//! * [`HipCanonicalKmerBatch`]  — `CanonicalKmerIterator` over many reads (src/naive_impl/canonical_kmer_iterator.rs:42-116);
//!                                [`HipCanonicalKmerIter`] walks one read of it with the crate's own `inc` / `get` / `exhausted`
//! * [`canonical_sum`]          — the consumer shape of benches/simple_benchmark.rs:14-22 on device-resident reads
//! * [`HipSeqVector`]           — `SeqVector` (src/naive_impl/seq_vector.rs) with the words on the device
//! * [`HipComm`]                — the RCCL exchange of the optional bucket histogram (no counterpart: the crate is single-process)
pub mod ffi;

use ffi::*;
use kmers::encoding::{Encoding, Naive};
use kmers::utils::Data;
use std::os::raw::c_void;
use std::ptr;

/// Error = a libkmx status code (`KMX_E_*`) with its text.
#[derive(Debug, Clone, PartialEq, Eq)]
pub struct KmxError {
    pub status: i32,
    pub message: String,
}

fn check(ctx: *const kmx_ctx, status: i32) -> Result<(), KmxError> {
    if status == KMX_OK {
        return Ok(());
    }
    let mut message = unsafe { std::ffi::CStr::from_ptr(kmx_strerror(status)) }.to_string_lossy().into_owned();
    if status == KMX_E_HIP && !ctx.is_null() {
        message.push_str(" -- ");
        message.push_str(&unsafe { std::ffi::CStr::from_ptr(kmx_last_error(ctx)) }.to_string_lossy());
    }
    Err(KmxError { status, message })
}

/// One `kmx_ctx`: a device, a HIP stream, scratch.  Single-threaded like the ABI says (one per host thread / GPU).
pub struct HipContext(*mut kmx_ctx);

impl HipContext {
    pub fn new(device: i32) -> Result<Self, KmxError> {
        let mut p = ptr::null_mut();
        check(ptr::null(), unsafe { kmx_ctx_create(device, &mut p) })?;
        Ok(Self(p))
    }
    pub fn synchronize(&self) -> Result<(), KmxError> {
        check(self.0, unsafe { kmx_ctx_synchronize(self.0) })
    }
    pub fn raw(&self) -> *mut kmx_ctx {
        self.0
    }
    fn ck(&self, status: i32) -> Result<(), KmxError> {
        check(self.0, status)
    }
    /// `n` bytes of device memory, freed on drop
    pub fn alloc(&self, n: usize) -> Result<DeviceBuf<'_>, KmxError> {
        let mut p = ptr::null_mut();
        self.ck(unsafe { kmx_malloc(self.0, n, &mut p) })?;
        Ok(DeviceBuf { ctx: self, ptr: p, len: n })
    }
    pub fn upload(&self, host: &[u8]) -> Result<DeviceBuf<'_>, KmxError> {
        let b = self.alloc(host.len())?;
        self.ck(unsafe { kmx_memcpy_h2d(self.0, b.ptr, host.as_ptr() as *const c_void, host.len()) })?;
        Ok(b)
    }
}

impl Drop for HipContext {
    fn drop(&mut self) {
        unsafe { kmx_ctx_destroy(self.0) }
    }
}

/// Device allocation owned by a context.
pub struct DeviceBuf<'c> {
    ctx: &'c HipContext,
    ptr: *mut c_void,
    len: usize,
}

impl<'c> DeviceBuf<'c> {
    pub fn as_ptr<T>(&self) -> *const T {
        self.ptr as *const T
    }
    pub fn as_mut_ptr<T>(&self) -> *mut T {
        self.ptr as *mut T
    }
    pub fn len(&self) -> usize {
        self.len
    }
    pub fn is_empty(&self) -> bool {
        self.len == 0
    }
    pub fn download<T: Copy + Default>(&self, n: usize) -> Result<Vec<T>, KmxError> {
        let mut v = vec![T::default(); n];
        assert!(n * std::mem::size_of::<T>() <= self.len);
        self.ctx.ck(unsafe { kmx_memcpy_d2h(self.ctx.0, v.as_mut_ptr() as *mut c_void, self.ptr, n * std::mem::size_of::<T>()) })?;
        Ok(v)
    }
}

impl<'c> Drop for DeviceBuf<'c> {
    fn drop(&mut self) {
        unsafe {
            kmx_free(self.ctx.0, self.ptr);
        }
    }
}

// ------------------------------------------------------------------------------------------------ Encoding

/// Drop-in for `impl Encoding<u64, B> for Naive` (src/encoding/naive.rs:112-154): the scalar trait methods run a
/// one-element batch; `encode_batch` / `decode_batch` / `rev_comp_batch` are the throughput forms.
/// `Xor10` (src/encoding/xor10.rs) is the same map as `Naive::ACTG`, so `HipEncoder { enc: Naive::ACTG, .. }` covers it
/// (its B == 1 `rev_comp`, xor10.rs:75-85, is not a reverse complement in the crate and is deliberately not reproduced).
pub struct HipEncoder<'c> {
    pub ctx: &'c HipContext,
    pub enc: Naive,
}

impl<'c> HipEncoder<'c> {
    pub fn new(ctx: &'c HipContext, enc: Naive) -> Self {
        Self { ctx, enc }
    }

    /// the little-endian byte image of a `[P; B]` slice (what the `_p` calls of the C ABI take: bit_field puts flat bit i into
    /// word i / BITS, bit i % BITS, so that image is the same flat bit string for every P -- include/kmx.h)
    fn bytes_of<P: Data>(words: &[P]) -> &[u8] {
        unsafe { std::slice::from_raw_parts(words.as_ptr() as *const u8, std::mem::size_of_val(words)) }
    }

    /// `seqs`: n sequences of `seq_len` bytes, contiguous -> n * B words of P.  Panics where the crate panics
    /// (`bit_field` index out of bounds when `2 * seq_len > P::BITS * B`, naive.rs:120).
    pub fn encode_batch<P: Data + Default, const B: usize>(&self, seqs: &[u8], seq_len: usize) -> Result<Vec<P>, KmxError> {
        let n = if seq_len == 0 { 0 } else { seqs.len() / seq_len };
        let wb = std::mem::size_of::<P>();
        let d_s = self.ctx.upload(seqs)?;
        let d_w = self.ctx.alloc(wb * B * n.max(1))?;
        let st = unsafe { kmx_encode_kmers_p(self.ctx.0, d_s.as_ptr(), n as u64, seq_len as u32, self.enc as u8, 8 * wb as u32, B as u32, d_w.as_mut_ptr()) };
        assert_ne!(st, KMX_E_TOO_LONG, "index out of bounds: the sequence is longer than the k-mer storage");
        self.ctx.ck(st)?;
        d_w.download::<P>(B * n)
    }

    /// `Encoding::decode` per k-mer: ALL `P::BITS * B / 2` letters each (naive.rs:126-136)
    pub fn decode_batch<P: Data, const B: usize>(&self, words: &[P]) -> Result<Vec<u8>, KmxError> {
        let n = words.len() / B;
        let wb = std::mem::size_of::<P>();
        let letters = 4 * wb * B;
        let d_w = self.ctx.upload(Self::bytes_of(words))?;
        let d_s = self.ctx.alloc(letters * n.max(1))?;
        self.ctx.ck(unsafe { kmx_encoding_decode_p(self.ctx.0, d_w.as_ptr(), n as u64, self.enc as u8, 8 * wb as u32, B as u32, d_s.as_mut_ptr()) })?;
        d_s.download::<u8>(letters * n)
    }

    /// `Encoding::rev_comp::<K>` per k-mer (naive.rs:138-154): base i = complement(base K-1-i) for i < K, bits >= 2K unchanged
    pub fn rev_comp_batch<P: Data + Default, const B: usize>(&self, words: &[P], big_k: usize) -> Result<Vec<P>, KmxError> {
        let n = words.len() / B;
        let wb = std::mem::size_of::<P>();
        let d_in = self.ctx.upload(Self::bytes_of(words))?;
        let d_out = self.ctx.alloc((wb * words.len()).max(1))?;
        let st = unsafe { kmx_encoding_rev_comp_p(self.ctx.0, d_in.as_ptr(), n as u64, big_k as u32, self.enc as u8, 8 * wb as u32, B as u32, d_out.as_mut_ptr()) };
        assert_ne!(st, KMX_E_K_RANGE, "attempt to subtract with overflow"); // K == 1 in the crate (naive.rs:140,150)
        self.ctx.ck(st)?;
        d_out.download::<P>(words.len())
    }
}

/// `impl<P, const B: usize> Encoding<P, B> for Naive where P: utils::Data` (src/encoding/naive.rs:112-115), on the device:
/// P = u8, u16, u32, u64, u128 (src/utils.rs:24).
impl<'c, P: Data + Default, const B: usize> Encoding<P, B> for HipEncoder<'c> {
    fn encode(&self, seq: &[u8]) -> [P; B] {
        let v = self.encode_batch::<P, B>(seq, seq.len().max(1)).expect("Encoding::encode");
        let mut out = [P::default(); B];
        if !seq.is_empty() {
            out.copy_from_slice(&v[..B]);
        }
        out
    }

    fn decode(&self, array: [P; B]) -> Vec<u8> {
        self.decode_batch::<P, B>(&array).expect("Encoding::decode")
    }

    fn rev_comp<const K: usize>(&self, array: [P; B]) -> [P; B] {
        let v = self.rev_comp_batch::<P, B>(&array, K).expect("Encoding::rev_comp");
        let mut out = [P::default(); B];
        out.copy_from_slice(&v[..B]);
        out
    }
}

// ------------------------------------------------------------------------------------------------ CanonicalKmerIterator

/// Batch form of `for km in CanonicalKmerIterator::from_u8_slice(read, k)` over many reads
/// (src/naive_impl/canonical_kmer_iterator.rs:72-116): holds, per window slot, what the iterator's `get()` would show.
pub struct HipCanonicalKmerBatch {
    pub fw: Vec<u64>,
    pub rc: Vec<u64>,
    pub flags: Vec<u8>,
    pub windows_per_read: usize,
}

impl HipCanonicalKmerBatch {
    /// `reads`: n_reads x read_len ASCII bytes, contiguous.  k in 1..=31 (the crate's MASK_TABLE[32] == 0, kmer.rs:617).
    pub fn scan(ctx: &HipContext, reads: &[u8], read_len: usize, k: u8) -> Result<Self, KmxError> {
        let n_reads = if read_len == 0 { 0 } else { reads.len() / read_len };
        let w = (read_len + 1).saturating_sub(k as usize);
        let total = n_reads * w;
        let d_b = ctx.upload(reads)?;
        let (d_fw, d_rc, d_fl) = (ctx.alloc(8 * total.max(1))?, ctx.alloc(8 * total.max(1))?, ctx.alloc(total.max(1))?);
        let r = kmx_reads { d_bases: d_b.as_ptr(), n_reads: n_reads as u64, read_len: read_len as u32, d_offsets: ptr::null() };
        ctx.ck(unsafe { kmx_canonical_windows(ctx.0, &r, ptr::null(), k as u32, d_fw.as_mut_ptr(), d_rc.as_mut_ptr(), ptr::null_mut(), d_fl.as_mut_ptr()) })?;
        Ok(Self { fw: d_fw.download(total)?, rc: d_rc.download(total)?, flags: d_fl.download(total)?, windows_per_read: w })
    }

    /// (read, pos, fw word, rc word) for every window the crate's iterator yields, in its order;
    /// `CanonicalKmer::get_canonical_word()` (canonical_kmer.rs:113-119) is `if fw < rc { fw } else { rc }`.
    pub fn iter(&self) -> impl Iterator<Item = (usize, i32, u64, u64)> + '_ {
        let w = self.windows_per_read.max(1);
        self.flags.iter().enumerate().filter(|(_, f)| **f & KMX_WIN_VALID != 0).map(move |(i, _)| (i / w, (i % w) as i32, self.fw[i], self.rc[i]))
    }
}

/// What `CanonicalKmerIterator::get()` shows (canonical_kmer_iterator.rs:13-16,113-116): the k-mer pair and its offset on the read.
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub struct HipCanonicalKmerPos {
    pub fw: u64,
    pub rc: u64,
    pub pos: i32,
}

impl HipCanonicalKmerPos {
    /// `CanonicalKmer::get_canonical_word()` (canonical_kmer.rs:113-119)
    pub fn canonical_word(&self) -> u64 {
        if self.fw < self.rc { self.fw } else { self.rc }
    }
}

/// One read of a [`HipCanonicalKmerBatch`] behind the crate's iterator protocol
/// (canonical_kmer_iterator.rs:89-116): `from_u8_slice` positions on the first valid k-mer, `inc()` moves to the next and
/// returns whether there is one, `exhausted()` tells when there is none, `get()` shows the current pair -- so a loop written
/// as `while !it.exhausted() { use(it.get()); it.inc(); }` against the crate runs unchanged over a scanned batch.
pub struct HipCanonicalKmerIter<'b> {
    batch: &'b HipCanonicalKmerBatch,
    base: usize,      // first window slot of the read
    slot: usize,      // current window (valid only while !invalid)
    invalid: bool,
}

impl<'b> HipCanonicalKmerIter<'b> {
    fn seek(&mut self, from: usize) {
        let w = self.batch.windows_per_read;
        let mut i = from;
        while i < w && self.batch.flags[self.base + i] & KMX_WIN_VALID == 0 {
            i += 1;
        }
        self.invalid = i >= w;
        self.slot = i;
    }
    /// `exhausted()` (canonical_kmer_iterator.rs:89-92)
    pub fn exhausted(&self) -> bool {
        self.invalid
    }
    /// `inc()` (canonical_kmer_iterator.rs:94-102)
    pub fn inc(&mut self) -> bool {
        if !self.invalid {
            let next = self.slot + 1;
            self.seek(next);
        }
        !self.invalid
    }
    /// `inc_by(count)` (canonical_kmer_iterator.rs:104-112)
    pub fn inc_by(&mut self, mut count: usize) -> bool {
        let mut v = !self.invalid;
        while count > 0 && v {
            v = self.inc();
            count -= 1;
        }
        v
    }
    /// `get()` (canonical_kmer_iterator.rs:113-116); like the crate's, meaningful while `!exhausted()`
    pub fn get(&self) -> HipCanonicalKmerPos {
        let i = self.base + self.slot.min(self.batch.windows_per_read.saturating_sub(1));
        HipCanonicalKmerPos { fw: self.batch.fw[i], rc: self.batch.rc[i], pos: self.slot as i32 }
    }
}

impl HipCanonicalKmerBatch {
    /// the iterator of read `read` (`CanonicalKmerIterator::from_u8_slice(read_bytes, k)`, canonical_kmer_iterator.rs:72-83)
    pub fn read_iter(&self, read: usize) -> HipCanonicalKmerIter<'_> {
        let mut it = HipCanonicalKmerIter { batch: self, base: read * self.windows_per_read, slot: 0, invalid: self.windows_per_read == 0 };
        if !it.invalid {
            it.seek(0);
        }
        it
    }
}

/// The consumer shape of benches/simple_benchmark.rs:14-22 (`.sum()` over the words of all windows) on device-resident
/// reads: count, wrapping sum of canonical words, xor of `hash_one(LexHasherState::new(k), ..)`, wrapping sum of fw words.
pub fn canonical_sum(ctx: &HipContext, d_reads: &DeviceBuf<'_>, n_reads: u64, read_len: u32, k: u8) -> Result<kmx_summary, KmxError> {
    assert!(n_reads as u128 * read_len as u128 <= d_reads.len() as u128, "reads past the end of the device buffer");
    let r = kmx_reads { d_bases: d_reads.as_ptr(), n_reads, read_len, d_offsets: ptr::null() };
    // (the summary straight into host memory: one kernel launch for clean reads of up to 256 bases, no device allocation, no copy)
    let mut out = std::mem::MaybeUninit::<kmx_summary>::zeroed();
    ctx.ck(unsafe { kmx_canonical_reduce_host(ctx.0, &r, k as u32, KMX_HASH_LEX, k as u32, KMX_REDUCE_SUM_FW, out.as_mut_ptr()) })?;
    Ok(unsafe { out.assume_init() })
}

/// `hash_one(&state, kmer)` for a batch of k-mer words with one of std's `BuildHasher`s (src/naive_impl/hash.rs:10-20): std's
/// `DefaultHasher` is SipHash-1-3 and `impl Hash for Kmer` feeds it one `write_u64(data)` (hash.rs:4-8).  `keys` = (0, 0) for
/// `DefaultHasher::new()` / `BuildHasherDefault<DefaultHasher>`; a `RandomState`'s keys are private to std, so a caller who wants
/// device hashes equal to host hashes builds its hasher from keys it knows (`SipHasher13::new_with_keys` semantics).
pub fn hash_words_sip13(ctx: &HipContext, words: &[u64], keys: (u64, u64)) -> Result<Vec<u64>, KmxError> {
    // (the device takes the words as they lie in memory: little-endian u64, what `write_u64` hashes on every target this library runs beside)
    let bytes = unsafe { std::slice::from_raw_parts(words.as_ptr() as *const u8, words.len() * 8) };
    let d_in = ctx.upload(bytes)?;
    let d_out = ctx.alloc(words.len() * 8)?;
    ctx.ck(unsafe { kmx_hash_words_sip13(ctx.0, d_in.as_ptr::<u64>(), words.len() as u64, keys.0, keys.1, d_out.as_mut_ptr::<u64>()) })?;
    d_out.download::<u64>(words.len())
}

/// `SeqVector::from(read).iter_minimizers(k, w, LexHasherState::new(lex_hasher_k))` (seq_vector.rs:73-80, minimizers.rs:39-141)
/// for every read of a uniform batch already on the device, without building the `SeqVector`s: `(word, pos)` of k-mer `i` of read
/// `r` at index `r * (read_len - k + 1) + i`.  A byte outside ACGTacgt is the panic of `Kmer::from` (kmer.rs:45-60): `Err` with
/// `KMX_E_INVALID_BASE`.
pub fn minimizers_of_reads(ctx: &HipContext, d_reads: &DeviceBuf<'_>, n_reads: u64, read_len: u32, k: u8, w: u8, lex_hasher_k: u8)
                           -> Result<Vec<(u64, u32)>, KmxError> {
    assert!(n_reads as u128 * read_len as u128 <= d_reads.len() as u128, "reads past the end of the device buffer");
    assert!(read_len >= k as u32, "SeqVecMinimizerIter::new: assertion failed: sv.len() >= k");
    let n = (n_reads * (read_len - k as u32 + 1) as u64) as usize;
    let d_word = ctx.alloc(n * 8)?;
    let d_pos = ctx.alloc(n * 4)?;
    let r = kmx_reads { d_bases: d_reads.as_ptr(), n_reads, read_len, d_offsets: ptr::null() };
    let mut first_bad = u64::MAX;
    ctx.ck(unsafe { kmx_minimizers(ctx.0, &r, ptr::null(), k as u32, w as u32, KMX_HASH_LEX, lex_hasher_k as u32, d_word.as_mut_ptr::<u64>(),
                                   d_pos.as_mut_ptr::<u32>(), &mut first_bad) })?;
    let (a, b) = (d_word.download::<u64>(n)?, d_pos.download::<u32>(n)?);
    Ok(a.into_iter().zip(b).collect())
}

// ------------------------------------------------------------------------------------------------ SeqVector

/// `SeqVector` (src/naive_impl/seq_vector.rs) with its words on the device: same bit layout as the crate's `RawVector`
/// (base i at flat bits [2i, 2i+1]), so a host vector's words can be uploaded as they are.
pub struct HipSeqVector<'c> {
    ctx: &'c HipContext,
    d_words: DeviceBuf<'c>,
    len: usize,
}

impl<'c> HipSeqVector<'c> {
    /// `SeqVector::from(&[u8])` (seq_vector.rs:230-242); panics on a non-ACGTacgt byte like `Kmer::from` does
    pub fn from_bytes(ctx: &'c HipContext, bytes: &[u8]) -> Result<Self, KmxError> {
        let d_words = ctx.alloc(8 * ((bytes.len() + 31) / 32).max(1) + 16)?;
        ctx.ck(unsafe { kmx_memset(ctx.0, d_words.as_mut_ptr(), 0, d_words.len()) })?;
        let d_b = ctx.upload(bytes)?;
        let mut bad = 0u64;
        let st = unsafe { kmx_seqvec_push_chars(ctx.0, d_words.as_mut_ptr(), 0, d_b.as_ptr(), bytes.len() as u64, &mut bad) };
        assert_ne!(st, KMX_E_INVALID_BASE, "cannot decode character at index {} into nucleotide", bad);
        ctx.ck(st)?;
        Ok(Self { ctx, d_words, len: bytes.len() })
    }
    pub fn len(&self) -> usize {
        self.len
    }
    pub fn is_empty(&self) -> bool {
        self.len == 0
    }
    /// `SeqVector::get_kmer_u64(pos, k)` (seq_vector.rs:96-99) for many positions
    pub fn get_kmers(&self, pos: &[u64], k: u8) -> Result<Vec<u64>, KmxError> {
        let bytes = unsafe { std::slice::from_raw_parts(pos.as_ptr() as *const u8, 8 * pos.len()) };
        let d_pos = self.ctx.upload(bytes)?;
        let d_out = self.ctx.alloc(8 * pos.len().max(1))?;
        self.ctx.ck(unsafe { kmx_seqvec_get_kmers(self.ctx.0, self.d_words.as_ptr(), self.len as u64, d_pos.as_ptr(), pos.len() as u64, k as u32, d_out.as_mut_ptr()) })?;
        d_out.download(pos.len())
    }
    /// `iter_kmers(k)` (seq_vector.rs:64-71, 117-124): the forward words of every window, in order
    pub fn iter_kmers(&self, k: u8) -> Result<Vec<u64>, KmxError> {
        let cnt = (self.len + 1).saturating_sub(k as usize);
        let d_out = self.ctx.alloc(8 * cnt.max(1))?;
        self.ctx.ck(unsafe { kmx_seqvec_iter_kmers(self.ctx.0, self.d_words.as_ptr(), self.len as u64, 0, self.len as u64, k as u32, d_out.as_mut_ptr()) })?;
        d_out.download(cnt)
    }
    /// `String::from(&SeqVector)` (seq_vector.rs:171-182)
    pub fn to_string(&self) -> Result<String, KmxError> {
        let d_b = self.ctx.alloc(self.len.max(1))?;
        self.ctx.ck(unsafe { kmx_seqvec_to_bytes(self.ctx.0, self.d_words.as_ptr(), self.len as u64, d_b.as_mut_ptr()) })?;
        Ok(String::from_utf8(d_b.download::<u8>(self.len)?).expect("ACGT"))
    }
    /// canonical k-mer scan of the reads stored back to back (read r = slice [r * read_len, (r + 1) * read_len))
    pub fn canonical_sum(&self, read_len: u32, k: u8) -> Result<kmx_summary, KmxError> {
        let d_out = self.ctx.alloc(std::mem::size_of::<kmx_summary>())?;
        let n_reads = if read_len == 0 { 0 } else { self.len as u64 / read_len as u64 };
        self.ctx.ck(unsafe { kmx_seqvec_canonical_reduce(self.ctx.0, self.d_words.as_ptr(), n_reads, read_len, k as u32, KMX_HASH_LEX, k as u32, KMX_REDUCE_SUM_FW, d_out.as_mut_ptr()) })?;
        Ok(d_out.download::<kmx_summary>(1)?[0])
    }
}

// ------------------------------------------------------------------------------------------------ multi-GPU exchange

/// The RCCL communicator of libkmx (one per context; one process or thread per GPU).  Reads shard embarrassingly, so the
/// scans need no collective: this carries the optional bucket-histogram all-reduce and the 32-byte summaries.
pub struct HipComm<'c> {
    ctx: &'c HipContext,
    comm: *mut kmx_comm,
}

impl<'c> HipComm<'c> {
    /// rank 0: the id to hand to the other ranks (file, socket, MPI, ...)
    pub fn unique_id() -> Result<[u8; KMX_COMM_ID_BYTES], KmxError> {
        let mut id = [0u8; KMX_COMM_ID_BYTES];
        check(ptr::null(), unsafe { kmx_comm_get_unique_id(id.as_mut_ptr()) })?;
        Ok(id)
    }
    /// collective over the ranks
    pub fn new(ctx: &'c HipContext, id: &[u8; KMX_COMM_ID_BYTES], n_ranks: i32, rank: i32) -> Result<Self, KmxError> {
        let mut comm = ptr::null_mut();
        ctx.ck(unsafe { kmx_comm_create(ctx.0, id.as_ptr(), n_ranks, rank, &mut comm) })?;
        Ok(Self { ctx, comm })
    }
    /// in place: counts[i] = sum over ranks (ncclAllReduce, ncclUint64 / ncclSum, on the context's stream)
    pub fn histogram_allreduce(&self, d_counts: &DeviceBuf<'_>, n_counts: u64) -> Result<(), KmxError> {
        assert!(8 * n_counts as u128 <= d_counts.len() as u128, "counters past the end of the device buffer");
        self.ctx.ck(unsafe { kmx_histogram_allreduce(self.comm, d_counts.as_mut_ptr(), n_counts) })
    }
    /// in place: wrapping sums / xor of the per-shard summaries (`d_summary`: one device-resident `kmx_summary`)
    pub fn summary_allreduce(&self, d_summary: &DeviceBuf<'_>) -> Result<(), KmxError> {
        assert!(std::mem::size_of::<kmx_summary>() <= d_summary.len());
        self.ctx.ck(unsafe { kmx_summary_allreduce(self.comm, d_summary.as_mut_ptr()) })
    }
    /// # Safety
    /// `d_counts` must be a device pointer to `n_counts` u64 owned by the caller for the duration of the call.
    pub unsafe fn histogram_allreduce_raw(&self, d_counts: *mut u64, n_counts: u64) -> Result<(), KmxError> {
        self.ctx.ck(kmx_histogram_allreduce(self.comm, d_counts, n_counts))
    }
}

impl<'c> Drop for HipComm<'c> {
    fn drop(&mut self) {
        unsafe { kmx_comm_destroy(self.comm) }
    }
}
