"""Development only: make kmers_amd load a library built by tools/dev_variant.py.

    import devlib; devlib.use("NAME")      # tools/_variants/NAME/libkmx.so, before the first Context
    KMX_DEV_LIB=NAME python tools/bench_x.py   # the same for the bench tools (they import _timing, which calls from_env())

The product (kmers_amd/_lib.py) reads nothing from the environment; this module does, and lives in tools/."""
from __future__ import annotations

import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def path_of(name: str) -> str:
    return name if os.path.sep in name else os.path.join(ROOT, "tools", "_variants", name, "libkmx.so")


def use(name: str) -> str:
    from kmers_amd import _lib

    p = os.path.abspath(path_of(name))
    if not os.path.exists(p):
        raise FileNotFoundError(p)
    if _lib._LIB is not None:
        raise RuntimeError("libkmx is already loaded")
    _lib.LIB_PATH = p
    return p


def from_env() -> None:
    name = os.environ.get("KMX_DEV_LIB", "")
    if name and name != "default":
        print(f"# development library: {use(name)}", file=sys.stderr)
