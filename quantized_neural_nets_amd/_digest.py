"""sha256 over the kernel sources the library is built from.  Standalone (no torch, no library load) so that
tools/pmc_traffic.py can stamp a profile with it on any machine."""
import hashlib
import os

_HERE = os.path.dirname(os.path.abspath(__file__))


def kernel_source_digest():
    """Profiles that quote per-kernel numbers are stamped with this digest (csrc/*.h, *.hip and the Makefile's
    flags), and bench.py only quotes a profile whose stamp matches the sources it runs."""
    h = hashlib.sha256()
    src = os.path.join(_HERE, "csrc")
    for name in sorted(os.listdir(src)):
        if name.endswith((".h", ".hip")) or name == "Makefile":
            h.update(name.encode())
            h.update(open(os.path.join(src, name), "rb").read())
    return h.hexdigest()
