"""What hipcc reported for the hot kernels when libkmx was built (kmers_amd/build.py keeps -Rpass-analysis=kernel-resource-usage
per translation unit): no scratch beyond a few spilled dwords, no dynamic stack (= a lambda that stopped being inlined: its
closure then lives in scratch and the kernel loses ~20 %), and the occupancy each variant is laid out for."""
import glob
import os
import re

import pytest

OBJ = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "kmers_amd", "csrc", "_obj")


def _kernels():
    out = {}
    for f in glob.glob(os.path.join(OBJ, "*.usage.txt")):
        for ln in open(f):
            parts = [p.strip() for p in ln.strip().split("|")]
            if len(parts) < 2:
                continue
            d = {}
            for p in parts[1:]:
                k, _, v = p.rpartition(":")
                d[k.strip()] = v.strip()
            out[parts[0]] = d
    return out


@pytest.fixture(scope="module")
def kernels():
    k = _kernels()
    if not k:
        pytest.skip("no *.usage.txt next to the objects (library not built by kmers_amd.build in this tree)")
    return k


def test_bitsliced_kernels_stay_out_of_scratch(kernels):
    seen = 0
    for name, d in kernels.items():
        if "scan_bitsliced_kernel" not in name:
            continue
        seen += 1
        assert d["Dynamic Stack"] == "False", name
        # Round 5: with the lane id and the half-wave selects rematerialised at their cold use sites and the wave's LDS bases scalar,
        # no variant keeps more than 40 bytes in scratch (round 4: up to 80; 28 before the queue's end was read at a glance -- the
        # two-word k at four windows per lane parks 8 bytes more since), and what is left is parked across the tile loop for the
        # epilogue or a rare path -- the loop's main path touches no scratch (tools/asm_loop_scratch.py on a --keep-asm build; a
        # reload there waits with vmcnt(0) behind the next tile's rows: that, not the spill itself, is what three waves used to cost)
        assert int(d["ScratchSize [bytes/lane]"]) <= 40, (name, d["ScratchSize [bytes/lane]"])
        # the ragged variants (three waves on the 7- / 10-word frame since round 5) and the segments of long uniform reads: <= 16 bytes.
        # Template flags: PACKED, RAGGED, SEG.
        if re.search(r"ELb0ELb1ELb0EEEv|ELb0ELb0ELb1EEEv", name):
            assert int(d["ScratchSize [bytes/lane]"]) <= 16, (name, d["ScratchSize [bytes/lane]"])
    assert seen >= 100   # every k of the three families, every frame


def test_headline_kernel_occupancy(kernels):
    """k = 31, 150 bp (10-word frame, 4 windows per lane): 3 waves per SIMD since pass 2 runs on the matrix pipe (64 fp32
    accumulators, the first and the last block pair sharing one block; four waves of a 32-accumulator form measured slower);
    the single-word ragged variants on the 7- / 10-word frame 3 since round 5 (bs_waves), the 16-word ragged frame and the two-word
    ragged variants 2, two-word k = 63: 3"""
    def occ(pattern):
        hits = [d for n, d in kernels.items() if re.search(pattern, n)]
        assert len(hits) == 1, (pattern, len(hits))
        return int(hits[0]["Occupancy [waves/SIMD]"])
    assert occ(r"scan_bitsliced_kernelILi31ELi10ELi4ELb0ELb0ELb0EEEv") == 3
    assert occ(r"scan_bitsliced_kernelILi31ELi10ELi4ELb0ELb1ELb0EEEv") == 3
    assert occ(r"scan_bitsliced_kernelILi31ELi10ELi5ELb0ELb1ELb0EEEv") == 3
    assert occ(r"scan_bitsliced_kernelILi31ELi7ELi3ELb0ELb1ELb0EEEv") == 3
    assert occ(r"scan_bitsliced_kernelILi31ELi16ELi5ELb0ELb1ELb0EEEv") == 2
    assert occ(r"scan_bitsliced_kernelILi63ELi10ELi4ELb0ELb1ELb0EEEv") == 2
    assert occ(r"scan_bitsliced_kernelILi63ELi10ELi4ELb0ELb0ELb0EEEv") == 3
    assert occ(r"scan_bitsliced_kernelILi31ELi10ELi4ELb0ELb0ELb1EEEv") == 3   # segments of long uniform reads
    assert occ(r"scan_bitsliced_kernelILi63ELi10ELi4ELb0ELb0ELb1EEEv") == 2


def test_scan_and_histogram_kernels_do_not_call(kernels):
    for name, d in kernels.items():
        if "scan_uniform_kernel" in name or "sweep_flagged_kernel" in name or "hist_part_reduce_kernel" in name:
            assert d["Dynamic Stack"] == "False", name


def test_hot_tile_loops_do_not_touch_scratch(tmp_path):
    """Round 5: what a spill costs in the bit-sliced scan is WHERE its reload sits -- inside the tile loop it waits with
    s_waitcnt vmcnt(0) behind the next tile's rows (the ragged scan at three waves lost 10 % to three such reloads per tile).  The
    hot instantiations are compiled to assembly here (device only, two translation units in parallel, ~1 minute) and every scratch
    access inside a loop is counted: none on any main path -- a ragged kernel may keep the one of its "read of 2^31 bases" path, and
    what the block that marks a dirty tile's reads reloads (round 6) is counted apart."""
    import shutil
    import subprocess
    import sys
    from concurrent.futures import ThreadPoolExecutor

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import asm_loop_scratch

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    sys.path.insert(0, root)
    from kmers_amd import build as kb

    def asm(src):
        out = str(tmp_path / (src + ".s"))
        r = subprocess.run([hipcc, *kb.CXXFLAGS, "--cuda-device-only", "-S", os.path.join(kb.CSRC, src), "-o", out], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        return out

    with ThreadPoolExecutor(max_workers=2) as ex:
        head, ragged = ex.map(asm, ["kmx_bitslice.hip", "kmx_bitslice_ragged_k29_31.hip"])
    seen = 0
    for path in (head, ragged):
        for name, (_, in_loads, in_stores, _in_dirty) in asm_loop_scratch.scan(path, "scan_bitsliced_kernel", split_dirty=True).items():
            seen += 1
            # (the ragged variants: at most the one reload of a 64-bit constant on the path that flags a read of 2^31 bases or more; round 6:
            # the reloads of the block that marks a dirty tile's reads are counted apart -- a clean tile does not pay them -- and the headline
            # kernel <31,10,4>, which until round 5 kept two reloads on its invalid-byte path, keeps nothing in scratch at all)
            allowed = 1 if "ELb0ELb1ELb0E" in name else 0
            if re.search(r"ELi16ELi\dELb1ELb0ELb0E", name):   # the packed 16-word frame (SeqVector reads of 161..256 bases): not a headline path
                continue
            assert in_loads <= allowed and in_stores == 0, (name, in_loads, in_stores)
    assert seen >= 40
