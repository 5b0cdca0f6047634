"""Identity of the kernel sources a measurement belongs to: profiles/hbm_traffic.json carries this hash and
bench.py only reports PMC traffic measured on the very sources it is running."""
import hashlib
import os

_HERE = os.path.dirname(os.path.abspath(__file__))


def kernel_source_sha() -> str:
    h = hashlib.sha256()
    csrc = os.path.join(_HERE, "csrc")
    files = sorted(f for f in os.listdir(csrc) if f.endswith((".hip", ".h")))
    for f in files:
        with open(os.path.join(csrc, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    for f in ("suma_types.h", "suma_detmath.h"):
        with open(os.path.join(_HERE, "..", "include", f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]
