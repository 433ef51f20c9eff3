"""Build of libinsider_hip.so and the identity of what was built.

The library carries a hash of the sources it was compiled from (-DINSIDER_SOURCE_SHA, returned by insider_hip_version());
`needs_build()` compares it with the hash of the sources on disk — by CONTENT, not by mtime — so a stale binary that
travelled with a snapshot is rebuilt instead of run, and bench.py can say which sources the numbers belong to.
"""
import fcntl
import hashlib
import os
import re
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
CSRC = os.path.join(_HERE, "csrc")
INCLUDE = os.path.join(ROOT, "include")
HIP_SRC = os.path.join(CSRC, "insider_hip.hip")
HIP_LIB = os.path.join(_HERE, "libinsider_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-pass-failed", "-mllvm", "-amdgpu-mfma-vgpr-form=1"]


def source_files():
    fs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".hpp"))]
    fs += [os.path.join(INCLUDE, f) for f in sorted(os.listdir(INCLUDE)) if f.endswith(".h")]
    return fs


def source_sha(extra_flags=()):
    """sha256 (first 16 hex digits) over insider_amd/csrc/*.{hip,hpp}, include/*.h and the compiler flags.  A VARIANT build
    (extra_flags: -D switches of the A/B tools and of tests/test_gpu_period.py) hashes its extra flags too, so it never
    carries the hash of the canonical build: written to the default path it is rebuilt before use, and bench.py does not
    attach the canonical library's counter figures to it."""
    h = hashlib.sha256()
    for f in source_files():
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    h.update(" ".join(FLAGS).encode())
    if extra_flags:
        h.update(b"\0variant\0" + " ".join(extra_flags).encode())
    return h.hexdigest()[:16]


def library_sha(path=HIP_LIB):
    """The source hash a built library carries (None: no library, or one from before the hash existed).  Read from the
    file's bytes (the version string in .rodata), not through dlopen: a stale library must not be mapped into this process,
    where a later dlopen of the rebuilt file under the same name would return the old handle."""
    try:
        m = re.search(rb"insider_hip [0-9.]+ \(gfx950\) src:([0-9a-f]{16})", open(path, "rb").read())
        return m.group(1).decode() if m else None
    except OSError:
        return None


def needs_build(path=HIP_LIB):
    return library_sha(path) != source_sha()


def build_library(force=False, extra_flags=(), out=HIP_LIB):
    """Compile the library for gfx950 (hipcc cross-compiles without a GPU).  Serialised across processes by a file lock:
    several ranks / test workers may find the same stale binary at once."""
    lock = open(os.path.join(_HERE, ".build.lock"), "w")      # (git-ignored: a runtime artefact)
    fcntl.flock(lock, fcntl.LOCK_EX)
    try:
        if not force and not extra_flags and library_sha(out) == source_sha():
            return False
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        tmp = out + ".tmp.so"
        cmd = [hipcc] + FLAGS + list(extra_flags) + [f'-DINSIDER_SOURCE_SHA="{source_sha(extra_flags)}"', "-o", tmp, HIP_SRC,
                                                     "-L/opt/rocm/lib", "-lrccl"]
        subprocess.check_call(cmd, cwd=ROOT)
        os.replace(tmp, out)
        return True
    finally:
        fcntl.flock(lock, fcntl.LOCK_UN)
        lock.close()
