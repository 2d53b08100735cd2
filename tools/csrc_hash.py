"""sha256 over the sources libraygraph.so is built from: ties a stored rocprofv3 summary to the kernels it profiled
(tools/summarize_profile.py stores it, bench.py compares it before quoting the summary's traffic)."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_sha256():
    d = os.path.join(ROOT, "ray_amd", "csrc")
    h = hashlib.sha256()
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h", ".cpp")) or name == "Makefile":
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()


if __name__ == "__main__":
    print(csrc_sha256())
