"""The tracked measurement files must not carry a failed self-check: a tool that prints WRONG / MISMATCH, or died with a traceback,
was committed once (profiles/r03_dirty_bench.txt) and cited from DESIGN.md before anybody looked."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_round4_profiles_hold_no_failed_check():
    bad = []
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r04_*"))):
        try:
            text = open(f, errors="replace").read()
        except OSError:
            continue
        for m in re.finditer(r"WRONG|MISMATCH|Traceback \(most recent call last\)", text):
            # (a file may QUOTE the words when it explains a dev probe whose results are wrong on purpose)
            line = text[text.rfind("\n", 0, m.start()) + 1: text.find("\n", m.end())]
            if "on purpose" in line or "timing only" in line:
                continue
            bad.append((os.path.basename(f), line.strip()[:160]))
    assert not bad, bad
