"""Build the native parts of libh264_amd in-tree.

  libh264_amd/lib/libh264_amd_engine.so   HIP kernels + C-ABI shim (hipcc, gfx950)
  libh264_amd/lib/libh264.so              drop-in libh264 C API (plain C host
                                          code linked against the engine)
  libh264_amd/lib/libh264gen.so           synthetic stream generator (gcc)

hipcc cross-compiles for gfx950 without a GPU.  The .so files are git-ignored
but travel to the GPU box with the source snapshot.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib")
INC = os.path.join(ROOT, "include")


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc",
                 shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    for s in sources:
        if os.path.getmtime(s) > t:
            return True
    return False


def _tree(*dirs):
    out = []
    for d in dirs:
        for base, _dirs, files in os.walk(d):
            for f in files:
                if f.endswith((".c", ".h", ".hip", ".inc", ".cpp")):
                    out.append(os.path.join(base, f))
    return out


def source_id():
    """sha256 over every source file the engine and the drop-in library are
    built from (path-sorted, names included): h264_amd_build_id() of a library
    built from this tree returns it."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(_tree(CSRC, INC)):
        h.update(os.path.relpath(f, ROOT).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def built_id(lib_path):
    """h264_amd_build_id() of a built library, read without loading it into
    this process (the string sits in .rodata behind a marker)."""
    with open(lib_path, "rb") as fh:
        blob = fh.read()
    tag = b"h264_amd_build_id="
    i = blob.find(tag)
    if i < 0:
        return None
    j = blob.find(b"\0", i)
    return blob[i + len(tag):j].decode()


def _run(cmd):
    print("+", " ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)


def build_gen(force=False):
    out = os.path.join(LIB, "libh264gen.so")
    srcs = _tree(os.path.join(CSRC, "gen"), os.path.join(CSRC, "common"))
    if force or _newer(out, srcs):
        _run(["gcc", "-std=gnu99", "-O2", "-fPIC", "-shared", "-Wall",
              os.path.join(CSRC, "gen", "h264gen.c"), "-o", out, "-lpthread"])
    return out


def build_engine(force=False):
    out = os.path.join(LIB, "libh264_amd_engine.so")
    srcs = _tree(CSRC, INC)
    sid = source_id()
    if force or _newer(out, srcs) or built_id(out) != sid:
        # -structurizecfg-skip-uniform-regions: the decode kernel's control
        # flow is wave-uniform by construction; without the flag the backend
        # still threads uniform regions through the structurizer's flow blocks
        # (s_mov -1 / s_andn2 vcc, exec / s_cbranch_vccnz chains): 17 % more
        # scalar instructions, 4 % more time (profiles/r02_decode_cost_model.md).
        # -O2 with the unroll threshold of -O3: the pick of six optimiser
        # settings timed at the end of round 3 (profiles/r03_decode_ab_history.jsonl);
        # they lie within 2 % of each other and trade I slices against P / B:
        # this one is the fastest on I slices (4096: 50.2 ms against 51.0 with
        # -O3, a lone wave 40.8 against 42.0) and 3 % slower than -O3 on the
        # mixed I / P stream of config 4.
        opt = os.environ.get(
            "H264_AMD_HIPCC_OPT",
            "-O2 -mllvm -unroll-threshold=300 -mllvm -structurizecfg-skip-uniform-regions").split()
        _run([_hipcc(), "--offload-arch=gfx950"] + opt + ["-std=c++17", "-fPIC",
              "-shared", "-Wno-unused-value", "-Wno-inline-asm", "-I" + INC, "-I" + CSRC,
              '-DH264_AMD_BUILD_ID="%s"' % sid,
              os.path.join(CSRC, "engine.hip"), "-o", out])
    return out


def build_host(force=False):
    """The drop-in C library: plain C objects linked with the engine."""
    out = os.path.join(LIB, "libh264.so")
    hostdir = os.path.join(CSRC, "host")
    if not os.path.isdir(hostdir):
        return None
    csrcs = sorted(os.path.join(hostdir, f) for f in os.listdir(hostdir)
                   if f.endswith(".c"))
    if not csrcs:
        return None
    srcs = _tree(hostdir, INC)
    sid = source_id()
    if force or _newer(out, srcs + [build_engine()]) or built_id(out) != sid:
        objs = []
        objdir = os.path.join(LIB, "obj")
        os.makedirs(objdir, exist_ok=True)
        for c in csrcs:
            o = os.path.join(objdir, os.path.basename(c)[:-2] + ".o")
            _run(["gcc", "-std=gnu99", "-D_GNU_SOURCE", "-DH264_API_EXPORTS",
                  "-fvisibility=hidden", "-O2", "-fPIC", "-Wall",
                  '-DH264_AMD_BUILD_ID="%s"' % sid,
                  "-I" + INC, "-I" + CSRC, "-c", c, "-o", o])
            objs.append(o)
        _run(["gcc", "-shared", "-o", out] + objs +
             ["-L" + LIB, "-lh264_amd_engine", "-Wl,-rpath,$ORIGIN",
              "-lpthread"])
    return out


def build_tools(force=False):
    """tools/h264_dump/h264-dump: the dump tool of BASELINE config 1 over the
    drop-in library (a plain C program against include/h264/h264.h)."""
    src = os.path.join(ROOT, "tools", "h264_dump", "h264_dump.c")
    out = os.path.join(ROOT, "tools", "h264_dump", "h264-dump")
    host = build_host()
    if host is None or not os.path.exists(src):
        return None
    if force or _newer(out, [src, host]):
        _run(["gcc", "-std=gnu99", "-O2", "-Wall", "-I" + INC, src, "-o", out,
              "-L" + LIB, "-lh264", "-lh264_amd_engine",
              "-Wl,-rpath,$ORIGIN/../../libh264_amd/lib"])
    return out


def build_all(force=False):
    os.makedirs(LIB, exist_ok=True)
    return {"gen": build_gen(force), "engine": build_engine(force),
            "host": build_host(force), "tools": build_tools(force)}


if __name__ == "__main__":
    print(build_all(force="--force" in sys.argv))
