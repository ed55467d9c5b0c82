"""Build libdaspeech_hip.so (gfx950) in-tree with hipcc.  Cross-compiles without a GPU.

    python -m daspeech_amd.build [--force]
"""
import glob
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
SO = os.path.join(LIBDIR, "libdaspeech_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value"]


def _file_flags(src):
    """Extra hipcc flags a source asks for in a `// HIPCC_FLAGS: ...` line of its header comment."""
    out = []
    with open(src) as f:
        for _ in range(40):
            line = f.readline()
            if line.startswith("// HIPCC_FLAGS:"):
                out += line.split(":", 1)[1].split()
    return out


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _deps():
    return sources() + glob.glob(os.path.join(CSRC, "*.h")) + \
        glob.glob(os.path.join(os.path.dirname(PKG), "include", "*.h"))


def is_stale():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    return any(os.path.getmtime(s) > t for s in _deps())


def build(force=False, verbose=False):
    """Compile every HIP source into one shared object.  Returns the .so path."""
    if not force and not is_stale():
        return SO
    os.makedirs(LIBDIR, exist_ok=True)
    # one object per source (parallel), then link: keeps incremental rebuilds at seconds
    objs, procs = [], []
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        hdr_t = max(os.path.getmtime(h) for h in _deps() if h.endswith(".h"))
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_t):
            cmd = [HIPCC] + [f for f in FLAGS if f != "-shared"] + _file_flags(src) + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        if verbose and out.strip():
            print(out)
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
