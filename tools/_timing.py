"""dev tools: the warm-up every timing loop here needs.  From an idle device (and the tools idle between cases: they build their
inputs on the host) the first ten or so launches of any kernel run up to 40 % slow (profiles/r03_step_times_cold.txt); a loop of
"one call, then five timed ones" measures that ramp, not the kernel -- 3.87 instead of 3.25 ms for the ragged scan at 1e8 reads."""
import time
import torch

import devlib

devlib.from_env()   # KMX_DEV_LIB=NAME: a development build (tools/dev_variant.py) instead of kmers_amd/libkmx.so


def warm(f, seconds=0.12, at_least=12):
    """call f until `seconds` of wall time and `at_least` calls have passed (the calls are queued asynchronously: synchronise)"""
    t0 = time.perf_counter()
    n = 0
    while n < at_least or time.perf_counter() - t0 < seconds:
        f()
        n += 1
        if n % 4 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
