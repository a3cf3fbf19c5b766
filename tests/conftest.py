"""pytest config: registers the `gpu` marker and puts the repo root on sys.path."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The shared library is a build product (git-ignored): build it in-tree if it is not there yet
    (hipcc cross-compiles gfx950 without a GPU).  Never a fallback path: without the library the ABI
    tests fail, and without a GPU nothing is decoded."""
    lib = os.path.join(ROOT, "lzma_rs_amd", "libmilzma.so")
    if not os.path.exists(lib):
        import subprocess
        subprocess.call(["make", "-C", os.path.join(ROOT, "lzma_rs_amd", "csrc"), "-s"])
