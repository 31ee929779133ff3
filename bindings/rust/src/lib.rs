//! Safe layer over libkmx that keeps the `kmers` crate's own surface.
//!
//! A GPU is a drop-in only at batch granularity, so every type here offers the crate's scalar interface (one k-mer
//! per call = a one-element batch: the correctness path, bit-identical to the CPU implementation) next to a batch
//! form (the throughput path).  Reference lines (`kmers` @ COMBINE-lab) are cited per item.
//!
//! * [`HipEncoder`]             — `impl Encoding<u64, B>` (src/encoding/mod.rs:14-23) for every `Naive` map (+ Xor10 = `Naive::ACTG`)
//! * [`HipCanonicalKmerBatch`]  — `CanonicalKmerIterator` over many reads (src/naive_impl/canonical_kmer_iterator.rs:42-116)
//! * [`canonical_sum`]          — the consumer shape of benches/simple_benchmark.rs:14-22 on device-resident reads
//! * [`HipSeqVector`]           — `SeqVector` (src/naive_impl/seq_vector.rs) with the words on the device
//! * [`HipComm`]                — the RCCL exchange of the optional bucket histogram (no counterpart: the crate is single-process)
pub mod ffi;

use ffi::*;
use kmers::encoding::{Encoding, Naive};
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

    /// `seqs`: n sequences of `seq_len` bytes, contiguous -> n * B words.  Panics where the crate panics
    /// (`bit_field` index out of bounds when `seq_len > 32 * B`, naive.rs:120).
    pub fn encode_batch<const B: usize>(&self, seqs: &[u8], seq_len: usize) -> Result<Vec<u64>, KmxError> {
        let n = if seq_len == 0 { 0 } else { seqs.len() / seq_len };
        let d_s = self.ctx.upload(seqs)?;
        let d_w = self.ctx.alloc(8 * B * n.max(1))?;
        let st = unsafe { kmx_encode_kmers(self.ctx.0, d_s.as_ptr(), n as u64, seq_len as u32, self.enc as u8, B as u32, d_w.as_mut_ptr()) };
        assert_ne!(st, KMX_E_TOO_LONG, "index out of bounds: the sequence is longer than the k-mer storage");
        self.ctx.ck(st)?;
        d_w.download::<u64>(B * n)
    }

    /// `Encoding::decode` per k-mer: ALL 32 * B letters each (naive.rs:126-136)
    pub fn decode_batch<const B: usize>(&self, words: &[u64]) -> Result<Vec<u8>, KmxError> {
        let n = words.len() / B;
        let bytes = unsafe { std::slice::from_raw_parts(words.as_ptr() as *const u8, 8 * words.len()) };
        let d_w = self.ctx.upload(bytes)?;
        let d_s = self.ctx.alloc(32 * B * n.max(1))?;
        self.ctx.ck(unsafe { kmx_encoding_decode(self.ctx.0, d_w.as_ptr(), n as u64, self.enc as u8, B as u32, d_s.as_mut_ptr()) })?;
        d_s.download::<u8>(32 * B * n)
    }

    /// `Encoding::rev_comp::<K>` per k-mer (naive.rs:138-154): base i = complement(base K-1-i) for i < K, bits >= 2K unchanged
    pub fn rev_comp_batch<const B: usize>(&self, words: &[u64], big_k: usize) -> Result<Vec<u64>, KmxError> {
        let n = words.len() / B;
        let bytes = unsafe { std::slice::from_raw_parts(words.as_ptr() as *const u8, 8 * words.len()) };
        let d_in = self.ctx.upload(bytes)?;
        let d_out = self.ctx.alloc(8 * words.len().max(1))?;
        let st = unsafe { kmx_encoding_rev_comp(self.ctx.0, d_in.as_ptr(), n as u64, big_k as u32, self.enc as u8, B as u32, d_out.as_mut_ptr()) };
        assert_ne!(st, KMX_E_K_RANGE, "attempt to subtract with overflow"); // K == 1 in the crate (naive.rs:140,150)
        self.ctx.ck(st)?;
        d_out.download::<u64>(words.len())
    }
}

impl<'c, const B: usize> Encoding<u64, B> for HipEncoder<'c> {
    fn encode(&self, seq: &[u8]) -> [u64; B] {
        let v = self.encode_batch::<B>(seq, seq.len().max(1)).expect("Encoding::encode");
        let mut out = [0u64; B];
        if !seq.is_empty() {
            out.copy_from_slice(&v[..B]);
        }
        out
    }

    fn decode(&self, array: [u64; B]) -> Vec<u8> {
        self.decode_batch::<B>(&array).expect("Encoding::decode")
    }

    fn rev_comp<const K: usize>(&self, array: [u64; B]) -> [u64; B] {
        let v = self.rev_comp_batch::<B>(&array, K).expect("Encoding::rev_comp");
        let mut out = [0u64; B];
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

/// The consumer shape of benches/simple_benchmark.rs:14-22 (`.sum()` over the words of all windows) on device-resident
/// reads: count, wrapping sum of canonical words, xor of `hash_one(LexHasherState::new(k), ..)`, wrapping sum of fw words.
pub fn canonical_sum(ctx: &HipContext, d_reads: *const u8, n_reads: u64, read_len: u32, k: u8) -> Result<kmx_summary, KmxError> {
    let d_out = ctx.alloc(std::mem::size_of::<kmx_summary>())?;
    let r = kmx_reads { d_bases: d_reads, n_reads, read_len, d_offsets: ptr::null() };
    ctx.ck(unsafe { kmx_canonical_reduce(ctx.0, &r, k as u32, KMX_HASH_LEX, k as u32, KMX_REDUCE_SUM_FW, d_out.as_mut_ptr()) })?;
    Ok(d_out.download::<kmx_summary>(1)?[0])
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
    /// `SeqVector::from(&[u8])` (seq_vector.rs:346-358); panics on a non-ACGTacgt byte like `Kmer::from` does
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
    /// `SeqVector::get_kmer_u64(pos, k)` (seq_vector.rs:217-220) for many positions
    pub fn get_kmers(&self, pos: &[u64], k: u8) -> Result<Vec<u64>, KmxError> {
        let bytes = unsafe { std::slice::from_raw_parts(pos.as_ptr() as *const u8, 8 * pos.len()) };
        let d_pos = self.ctx.upload(bytes)?;
        let d_out = self.ctx.alloc(8 * pos.len().max(1))?;
        self.ctx.ck(unsafe { kmx_seqvec_get_kmers(self.ctx.0, self.d_words.as_ptr(), self.len as u64, d_pos.as_ptr(), pos.len() as u64, k as u32, d_out.as_mut_ptr()) })?;
        d_out.download(pos.len())
    }
    /// `iter_kmers(k)` (seq_vector.rs:56-63, 236-243): the forward words of every window, in order
    pub fn iter_kmers(&self, k: u8) -> Result<Vec<u64>, KmxError> {
        let cnt = (self.len + 1).saturating_sub(k as usize);
        let d_out = self.ctx.alloc(8 * cnt.max(1))?;
        self.ctx.ck(unsafe { kmx_seqvec_iter_kmers(self.ctx.0, self.d_words.as_ptr(), self.len as u64, 0, self.len as u64, k as u32, d_out.as_mut_ptr()) })?;
        d_out.download(cnt)
    }
    /// `String::from(&SeqVector)` (seq_vector.rs:272-284)
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
    pub fn histogram_allreduce(&self, d_counts: *mut u64, n_counts: u64) -> Result<(), KmxError> {
        self.ctx.ck(unsafe { kmx_histogram_allreduce(self.comm, d_counts, n_counts) })
    }
    /// in place: wrapping sums / xor of the per-shard summaries
    pub fn summary_allreduce(&self, d_summary: *mut kmx_summary) -> Result<(), KmxError> {
        self.ctx.ck(unsafe { kmx_summary_allreduce(self.comm, d_summary) })
    }
}

impl<'c> Drop for HipComm<'c> {
    fn drop(&mut self) {
        unsafe { kmx_comm_destroy(self.comm) }
    }
}
