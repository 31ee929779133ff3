"""bindings/rust: the Rust host side of the boundary, as files.  There is no rustc in the build image, so the check is
textual: every `extern "C"` declaration of src/ffi.rs is compared, parameter by parameter, with the prototype of the same
name in include/kmx.h (an independent parser, not the generator's), the committed ffi.rs is what the generator emits
today, and the safe layer (src/lib.rs) only calls declared functions with the declared number of arguments and has no
unimplemented!() / todo!() left -- in particular `impl Encoding<P, B> for HipEncoder` (every utils::Data word type) has all three trait methods
(/root/reference/src/encoding/mod.rs:14-23)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RS = os.path.join(ROOT, "bindings", "rust")

C2RUST = {"int": "c_int", "uint8_t": "u8", "uint32_t": "u32", "uint64_t": "u64", "size_t": "usize", "void": "c_void",
          "char": "c_char", "kmx_ctx": "kmx_ctx", "kmx_comm": "kmx_comm", "kmx_reads": "kmx_reads", "kmx_summary": "kmx_summary",
          "kmx_summary2": "kmx_summary2"}


def _c_protos():
    src = open(os.path.join(ROOT, "include", "kmx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"^\s*#.*$", "", src, flags=re.M)
    protos = {}
    for stmt in src.split(";"):
        stmt = " ".join(stmt.split())
        m = re.search(r"(.*?)\b(kmx_[a-z0-9_]+)\s*\((.*)\)$", stmt)
        if not m or "typedef" in stmt or "{" in stmt:
            continue
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        params = [] if args in ("", "void") else [a.strip() for a in args.split(",")]
        protos[name] = (ret, params)
    return protos


def _c_to_rust(ctype_and_name: str, with_name: bool) -> str:
    t = ctype_and_name
    if with_name:
        t = re.sub(r"\b\w+$", "", t).strip()        # drop the parameter name
    const = t.startswith("const ")
    t = t[6:] if const else t
    n_ptr = t.count("*")
    base = C2RUST[t.replace("*", "").strip()]
    for i in range(n_ptr):
        base = ("*const " if (const and i == 0) else "*mut ") + base
    return base


def _rust_protos():
    src = open(os.path.join(RS, "src", "ffi.rs")).read()
    block = src[src.index('extern "C" {'):]
    protos = {}
    for m in re.finditer(r"pub fn (kmx_[a-z0-9_]+)\((.*?)\)( -> ([^;]+))?;", block, flags=re.S):
        params = [p.split(":", 1)[1].strip() for p in m.group(2).split(",") if p.strip()]
        protos[m.group(1)] = ((m.group(4) or "").strip(), params)
    return protos


def test_every_header_prototype_is_bound_with_the_same_signature():
    c, r = _c_protos(), _rust_protos()
    assert len(c) >= 50
    assert sorted(c) == sorted(r), set(c) ^ set(r)
    for name, (ret, params) in c.items():
        rret, rparams = r[name]
        assert len(params) == len(rparams), name
        for cp, rp in zip(params, rparams):
            assert _c_to_rust(cp, True) == rp, (name, cp, rp)
        assert ("" if ret == "void" else _c_to_rust(ret, False)) == rret, (name, ret, rret)


def test_ffi_rs_is_what_the_generator_emits():
    rc = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_rust_ffi.py"), "--check"]).returncode
    assert rc == 0, "bindings/rust/src/ffi.rs is stale: run python tools/gen_rust_ffi.py"


def test_struct_layouts_match_the_header():
    src = open(os.path.join(RS, "src", "ffi.rs")).read()
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "kmx.h")).read(), flags=re.S)
    for name in ("kmx_reads", "kmx_summary", "kmx_summary2"):
        cm = re.search(r"typedef struct \{([^{}]*)\} %s;" % name, hdr)
        cfields = []
        for decl in cm.group(1).split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            ctype, names = decl.rsplit(" ", 1)[0], decl
            first = re.match(r"^(.*?)(\**\w+(?:\s*,\s*\**\w+)*)$", decl)
            base = first.group(1).strip()
            for nm in first.group(2).split(","):
                nm = nm.strip()
                cfields.append((nm.lstrip("*"), _c_to_rust((base + " " + "*" * nm.count("*")).strip(), False)))
        rm = re.search(r"#\[repr\(C\)\][^\n]*\n(?:#\[derive[^\n]*\n)?pub struct %s \{(.*?)\n\}" % name, src, flags=re.S)
        rfields = [(f.split(":")[0].replace("pub", "").strip(), f.split(":")[1].strip()) for f in rm.group(1).split(",") if ":" in f]
        assert cfields == rfields, (name, cfields, rfields)


def test_safe_layer_is_complete_and_calls_only_what_exists():
    lib = open(os.path.join(RS, "src", "lib.rs")).read()
    assert "unimplemented!" not in lib and "todo!" not in lib
    r = _rust_protos()
    calls = re.findall(r"\b(kmx_[a-z0-9_]+)\s*\(", lib)
    assert len(set(calls)) >= 20
    for name in set(calls):
        assert name in r, f"lib.rs calls {name}, which ffi.rs does not declare"
    # argument counts (top-level commas of each call)
    for m in re.finditer(r"\b(kmx_[a-z0-9_]+)\s*\(", lib):
        depth, i, commas = 1, m.end(), 0
        empty = True
        while depth:
            ch = lib[i]
            if ch in "([{":
                depth += 1
            elif ch in ")]}":
                depth -= 1
            elif ch == "," and depth == 1:
                commas += 1
            if depth and not ch.isspace():
                empty = False
            i += 1
        n_args = 0 if empty else commas + 1
        assert n_args == len(r[m.group(1)][1]), (m.group(1), n_args, len(r[m.group(1)][1]))
    # the trait of /root/reference/src/encoding/mod.rs:14-23, all three methods, generic over the word type like
    # `impl<P, const B: usize> Encoding<P, B> for Naive where P: utils::Data` (/root/reference/src/encoding/naive.rs:112-115)
    impl = lib[lib.index("impl<'c, P: Data + Default, const B: usize> Encoding<P, B> for HipEncoder<'c>"):]
    impl = impl[:impl.index("\n}\n") + 3]
    for sig in ("fn encode(&self, seq: &[u8]) -> [P; B]", "fn decode(&self, array: [P; B]) -> Vec<u8>",
                "fn rev_comp<const K: usize>(&self, array: [P; B]) -> [P; B]"):
        assert sig in impl, sig
    # ... through the word-size-generic calls of the C ABI
    for call in ("kmx_encode_kmers_p", "kmx_encoding_decode_p", "kmx_encoding_rev_comp_p"):
        assert call in lib, call
    # the iterator protocol of /root/reference/src/naive_impl/canonical_kmer_iterator.rs:89-116 over a scanned batch
    it = lib[lib.index("impl<'b> HipCanonicalKmerIter<'b>"):]
    for sig in ("pub fn exhausted(&self) -> bool", "pub fn inc(&mut self) -> bool", "pub fn inc_by(&mut self, mut count: usize) -> bool",
                "pub fn get(&self) -> HipCanonicalKmerPos"):
        assert sig in it, sig
    assert "pub fn read_iter(&self, read: usize) -> HipCanonicalKmerIter" in lib


def test_safe_functions_do_not_take_raw_device_pointers():
    """a safe `pub fn` taking a raw device pointer is unsound -- they take a DeviceBuf, or are `unsafe fn`"""
    lib = open(os.path.join(RS, "src", "lib.rs")).read()
    for m in re.finditer(r"pub (unsafe )?fn (\w+)\s*(<[^>]*>)?\(([^)]*)\)", lib):
        is_unsafe, name, args = bool(m.group(1)), m.group(2), m.group(4)
        if name in ("raw", "as_ptr", "as_mut_ptr"):
            continue
        if "*const" in args or "*mut" in args:
            assert is_unsafe, f"pub fn {name} takes a raw pointer but is not unsafe"


def test_every_header_constant_is_in_ffi_rs():
    """the generator used to drop hex / U-suffixed literals (KMX_FASTX_SAME_TEXT 0x100u)"""
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "kmx.h")).read(), flags=re.S)
    ffi = open(os.path.join(RS, "src", "ffi.rs")).read()
    names = re.findall(r"^#define\s+(KMX_[A-Z0-9_]+)\s+\S", hdr, flags=re.M)
    assert len(names) >= 20
    for nm in names:
        if nm in ("KMX_H",):
            continue
        m = re.search(r"^#define\s+%s\s+(0[xX][0-9A-Fa-f]+|[0-9]+)[uU]?\s*$" % nm, hdr, flags=re.M)
        assert m, f"{nm}: not an integer literal the generator understands"
        r = re.search(r"pub const %s: \w+ = (\d+);" % nm, ffi)
        assert r, f"{nm} missing from ffi.rs"
        assert int(r.group(1)) == int(m.group(1), 0), nm


def test_crate_files_exist():
    for f in ("Cargo.toml", "build.rs", "src/ffi.rs", "src/lib.rs"):
        assert os.path.getsize(os.path.join(RS, f)) > 100, f
    assert "rustc-link-lib=dylib=kmx" in open(os.path.join(RS, "build.rs")).read()
