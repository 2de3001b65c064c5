"""Build libhsp.so for gfx950 (hipcc cross-compiles without a GPU).

    python -m megatts2_hierspeechpp_amd.build
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def build(verbose: bool = False) -> str:
    cmd = ["make", "-C", os.path.join(HERE, "csrc"), "-j4"]
    res = subprocess.run(cmd, capture_output=not verbose, text=True)
    if res.returncode != 0:
        raise RuntimeError("libhsp.so build failed:\n" + (res.stdout or "") + (res.stderr or ""))
    return os.path.join(HERE, "libhsp.so")


if __name__ == "__main__":
    print(build(verbose=True))
