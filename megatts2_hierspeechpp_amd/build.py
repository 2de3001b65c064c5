"""Build libhsp.so for gfx950 (hipcc cross-compiles without a GPU).

    python -m megatts2_hierspeechpp_amd.build
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def build(verbose: bool = False) -> str:
    """`make all` = libhsp.so AND the ISA lint of every hand-scheduled translation unit (tools/check_isa.py over the
    device assembly; csrc/Makefile `build/isa/.checked`): a library whose fragment pipelines fail the lint does not build."""
    cmd = ["make", "-C", os.path.join(HERE, "csrc"), "-j4"]
    res = subprocess.run(cmd, capture_output=not verbose, text=True)
    if res.returncode != 0:
        raise RuntimeError("libhsp.so build failed:\n" + (res.stdout or "") + (res.stderr or ""))
    return os.path.join(HERE, "libhsp.so")


if __name__ == "__main__":
    print(build(verbose=True))


def source_id() -> str:
    """16 hex digits of the sha256 over the kernel sources libhsp.so is built from (csrc/*.hip, csrc/*.h, the
    Makefile, include/hsp.h): what a committed profile is keyed by -- unlike a hash of the .so it does not depend on
    where or when the library was compiled."""
    import glob
    import hashlib
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    files = sorted(glob.glob(os.path.join(here, "csrc", "*.hip")) + glob.glob(os.path.join(here, "csrc", "*.h")))
    files += [os.path.join(here, "csrc", "Makefile"), os.path.join(here, "csrc", "hsp.map"), os.path.join(root, "include", "hsp.h")]
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]
