"""Shared generators of FASTA/FASTQ test texts (build-defined, no reference fixtures exist)."""
import numpy as np

SEQ_ALPHA = list(b"ACGTacgtN")
QUAL_ALPHA = list(b"@>+IIFF#5")   # quality strings may hold the characters that open header lines


def fastq_text(rng, n, lo=0, hi=200, crlf=False, trail=True, fixed=None):
    out = []
    for i in range(n):
        L = fixed if fixed is not None else int(rng.integers(lo, hi + 1))
        s = bytes(rng.choice(SEQ_ALPHA, L).astype(np.uint8))
        q = bytes(rng.choice(QUAL_ALPHA, L).astype(np.uint8))
        out += [b"@read%d len=%d" % (i, L), s, b"+", q]
    nl = b"\r\n" if crlf else b"\n"
    return nl.join(out) + (nl if trail and out else b"")


def fasta_text(rng, n, lo=0, hi=400, width=60, crlf=False, trail=True, blank=0.1):
    out = []
    for i in range(n):
        L = int(rng.integers(lo, hi + 1))
        s = bytes(rng.choice(SEQ_ALPHA, L).astype(np.uint8))
        out.append(b">seq%d some description" % i)
        out += [s[j:j + width] for j in range(0, L, width)]
        if rng.random() < blank:
            out.append(b"")
    nl = b"\r\n" if crlf else b"\n"
    return nl.join(out) + (nl if trail and out else b"")


EDGE_TEXTS = [b"", b"@x", b"@x\n", b"@x\nACGT", b"@x\nACGT\n", b"@x\nACGT\n+\nIIII", b"@x\nACGT\n+\nIIII\n", b"@x\n\n+\n\n@y\nA\n+\nI\n",
              b">x", b">x\n", b">x\nAC", b">a\n>b\nAC\n\nGT\n>c", b">a\n\n\n>b\n\nA\n", b">a\nAC\r\nGT\r\n>b\r\nTT", b"@a\r\nAC\r\n+\r\nII\r\n",
              # bytes that look like a newline or a '\r' to a SWAR test: 0x0B right after a '\n' (the borrow of the zero-byte test), tabs and
              # other control bytes (the '\r' suspects), '\r' in header and quality lines only, "\n\r"
              b"@a\tb\x0b\nAC\x0bGT\n\x0b+\n\x0b\x0b\x01\x0b\n@c\x08\x0c\x0e\x0f\nA\x01C\n+\n\x0b\x0a", b"@a\rb\nACGT\n+\nI\rI\r\n@b\n\rAC\r\rG\n\r+\n\r\r\n",
              b">a\x0b\n\x0bAC\n\x0b\n>\x0b\n\rA\x0b\rC\n", b">h\n" + b"ACGT\x0b\n\x0bACGT\r\n" * 300,
              b">" + b"h" * 5000 + b"\n" + b"ACGT" * 3000 + b"\n", b"@" + b"h" * 70000 + b"\nAC\n+\nII\n", b">\n" * 3000, b"@\n\n+\n\n" * 3000]
