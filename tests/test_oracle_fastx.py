"""FASTA/FASTQ record splitting -- BUILD-DEFINED, parity unpinned: the reference has no parser and
no fixtures.  What can be checked on the CPU: the two independent restatements of the prose spec (bytes.split in
Python, a byte-at-a-time state machine in C) agree, and hand-written cases give the hand-derived answers."""
import numpy as np
import pytest

from fastx_cases import EDGE_TEXTS, fasta_text, fastq_text


def _same(a, b):
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_hand_cases(orc):
    b, o = orc.fastx_parse(b"@r1\nACGT\n+\nIIII\n@r2\nGG\n+r2\n@@\n")
    assert bytes(b) == b"ACGTGG" and list(o) == [0, 4, 6]
    b, o = orc.fastx_parse(b">s1 x\nACG\nTTA\n\n>s2\n>s3\r\nNN\r\nA")
    assert bytes(b) == b"ACGTTANNA" and list(o) == [0, 6, 6, 9]
    b, o = orc.fastx_parse(b"")
    assert len(b) == 0 and list(o) == [0]
    with pytest.raises(ValueError):
        orc.fastx_parse(b"ACGT\n")
    with pytest.raises(ValueError):
        orc.fastx_parse_c(b"ACGT\n")
    with pytest.raises(ValueError):
        orc.fastx_parse(b">x\nAC\n", 1)      # FASTA text, FASTQ requested
    with pytest.raises(ValueError):
        orc.fastx_parse_c(b"@x\nAC\n", 2)


@pytest.mark.parametrize("i", range(len(EDGE_TEXTS)))
def test_edge_texts(orc, i):
    _same(orc.fastx_parse(EDGE_TEXTS[i]), orc.fastx_parse_c(EDGE_TEXTS[i]))


@pytest.mark.parametrize("crlf", [False, True])
@pytest.mark.parametrize("trail", [False, True])
def test_random_texts(orc, crlf, trail):
    rng = np.random.default_rng(11 + 2 * crlf + trail)
    for n in (1, 7, 300):
        t = fastq_text(rng, n, crlf=crlf, trail=trail)
        a = orc.fastx_parse(t)
        assert len(a[1]) == n + 1
        _same(a, orc.fastx_parse_c(t))
        _same(a, orc.fastx_parse_c(t, 1))
        t = fasta_text(rng, n, width=int(rng.integers(1, 90)), crlf=crlf, trail=trail)
        a = orc.fastx_parse(t)
        assert len(a[1]) == n + 1
        _same(a, orc.fastx_parse_c(t))
        _same(a, orc.fastx_parse_c(t, 2))
