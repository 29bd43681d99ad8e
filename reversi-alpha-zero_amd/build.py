"""Build csrc/libraz.so for gfx950 with hipcc (cross-compiles without a GPU)."""
import glob
import os
import shutil
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libraz.so")
HIPCC = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
# -ffp-contract=off: the search arithmetic (PUCT, backup, softmax) is specified operation by
# operation so that it is bit-identical to the CPU oracle; the compiler must not fuse a*b+c.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
         "-Wall", "-Wno-unused-function"] + os.environ.get("RAZ_EXTRA_FLAGS", "").split()


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + \
        glob.glob(os.path.join(os.path.dirname(os.path.dirname(CSRC)), "include", "*.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build_native(force=False, verbose=False):
    """Compile every .hip under csrc/ into libraz.so.  Returns the library path."""
    if not force and not _stale():
        return LIB
    tmp = f"{LIB}.{os.getpid()}.tmp"   # several ranks may build at once: private output, atomic rename
    cmd = [HIPCC] + FLAGS + sources() + ["-o", tmp]
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True)
    if r.returncode != 0:
        if os.path.exists(tmp):
            os.remove(tmp)
        raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
    os.replace(tmp, LIB)
    return LIB


if __name__ == "__main__":
    print(build_native(force=True, verbose=True))
