"""build_ref.py — stage the reference for the GPU box: oracle/_ref/ (git-ignored BUILD OUTPUT, never committed).

TEST / MEASUREMENT INFRASTRUCTURE ONLY (see orc.h): the `cpu_baseline` leg of bench.py times the reference's own
pure-Python self-play on the bench box's host cores, and /root/reference does not exist there.  The reference is
Python, so "building" it means byte-compiling: every module under /root/reference/src/reversi_zero is compiled from
where it lies into a SOURCELESS .pyc under oracle/_ref/src/ (py_compile, the interpreter of this image = the GPU
box's), its two Cython modules (lib/alt/*.pyx, imported by agent/player.py:15) into extension modules (cython + gcc,
what the reference's own pyximport call does at import time), and the yml files the path reads (config/*.yml: data, not code) are placed beside them.  No reference source
file is copied into the repository or its history; oracle/_ref/ travels with the tree to the GPU box exactly like
the other built artefacts (libraz.so, liboracle.so).

    python oracle/build_ref.py        (also run by __graft_entry__.build() when /root/reference is present)

oracle/ref_harness.py falls back to oracle/_ref when /root/reference is absent; nothing under
reversi-alpha-zero_amd/ imports either."""
import os
import py_compile
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")
SRC_ROOT = os.environ.get("RAZ_REFERENCE_SRC_ROOT", "/root/reference")


def stale():
    stamp = os.path.join(OUT, ".built")
    if not os.path.exists(stamp):
        return True
    t = os.path.getmtime(stamp)
    with open(stamp) as f:
        if f.read().strip() != sys.version:
            return True
    for base, _, files in os.walk(os.path.join(SRC_ROOT, "src", "reversi_zero")):
        if any(os.path.getmtime(os.path.join(base, n)) > t for n in files if n.endswith(".py")):
            return True
    return False


def _compile_pyx(pyx, out_dir, scratch):
    """lib/alt/*.pyx (agent/player.py:15 imports the Cython solver through pyximport at import time) -> extension module
    beside the byte-compiled package, so the import resolves without pyximport finding a source to build."""
    import subprocess
    import sysconfig
    mod = os.path.splitext(os.path.basename(pyx))[0]
    c_file = os.path.join(scratch, mod + ".c")
    subprocess.run([sys.executable, "-m", "cython", pyx, "-o", c_file], check=True, capture_output=True)
    so = os.path.join(out_dir, mod + sysconfig.get_config_var("EXT_SUFFIX"))
    subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-w", "-I" + sysconfig.get_paths()["include"], c_file, "-o", so], check=True,
                   capture_output=True)
    os.remove(c_file)


def build(force=False):
    """Returns the staged root, or None when the reference tree is not present (the GPU box: the prebuilt
    oracle/_ref is used as it is)."""
    pkg = os.path.join(SRC_ROOT, "src", "reversi_zero")
    if not os.path.isdir(pkg):
        return OUT if os.path.isdir(os.path.join(OUT, "src", "reversi_zero")) else None
    if not force and not stale():
        return OUT
    tmp = f"{OUT}.{os.getpid()}.tmp"
    shutil.rmtree(tmp, ignore_errors=True)
    n = 0
    for base, dirs, files in os.walk(pkg):
        dirs[:] = [d for d in dirs if d != "__pycache__"]
        rel = os.path.relpath(base, os.path.join(SRC_ROOT, "src"))
        os.makedirs(os.path.join(tmp, "src", rel), exist_ok=True)
        for name in files:
            if name.endswith(".pyx"):   # the reference's compiled code: cython -> C -> gcc, straight from where the source lies
                _compile_pyx(os.path.join(base, name), os.path.join(tmp, "src", rel), tmp)
            if name.endswith(".py"):
                py_compile.compile(os.path.join(base, name), cfile=os.path.join(tmp, "src", rel, name + "c"), doraise=True,
                                   dfile=os.path.join("reversi_zero", os.path.relpath(os.path.join(base, name), pkg)),
                                   invalidation_mode=py_compile.PycInvalidationMode.UNCHECKED_HASH)
                n += 1
    os.makedirs(os.path.join(tmp, "config"), exist_ok=True)
    for name in os.listdir(os.path.join(SRC_ROOT, "config")):
        if name.endswith(".yml"):
            shutil.copyfile(os.path.join(SRC_ROOT, "config", name), os.path.join(tmp, "config", name))
    with open(os.path.join(tmp, ".built"), "w") as f:
        f.write(sys.version)
    shutil.rmtree(OUT, ignore_errors=True)
    os.replace(tmp, OUT)
    return OUT


if __name__ == "__main__":
    print(build(force=True))
