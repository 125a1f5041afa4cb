"""Builds the native libraries in-tree.

  raytracer_amd/libmrt_hip.so   HIP kernels + C-ABI (hipcc, gfx950; cross-compiles without a GPU)
  raytracer_amd/libmrt_host.so  C++ host side (g++), links libmrt_hip.so
  raytracer_amd/mrt_main        examples/main.cpp: the reference's main.cpp against the host library
  oracle/libmrt_oracle.so       CPU oracle (test infrastructure)
  oracle/_ref/libmrt_ref.so     the reference's own translation units (only where /root/reference exists)

A library is rebuilt when any of its sources is newer than it.
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "raytracer_amd")
CSRC = os.path.join(PKG, "csrc")
INC = os.path.join(ROOT, "include")
ORACLE = os.path.join(ROOT, "oracle")
REFERENCE = "/root/reference"

HIP_SO = os.path.join(PKG, "libmrt_hip.so")
HOST_SO = os.path.join(PKG, "libmrt_host.so")
ORACLE_SO = os.path.join(ORACLE, "libmrt_oracle.so")
REF_SO = os.path.join(ORACLE, "_ref", "libmrt_ref.so")

# -ffp-contract=off is a correctness flag here: primitive ids are bit-identical to the CPU
# reference only with unfused fp32 arithmetic.
# -fno-slp-vectorize: the SLP vectorizer pairs the kernels' fp32 arithmetic into packed instructions,
# which costs register pairs and moves: without it the traversal kernels need 55-66 instead of 70-78
# registers and the frame is 4 % faster (same bits: packed and scalar fp32 operations round alike).
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-slp-vectorize", "-std=c++17", "-fPIC", "-shared"]
HOST_FLAGS = ["-std=c++17", "-O2", "-fPIC", "-shared", "-fopenmp", "-ffp-contract=off", "-fno-fast-math"]


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources if os.path.exists(s))


def _files(d, exts):
    out = []
    for base, _, names in os.walk(d):
        for n in names:
            if n.endswith(exts):
                out.append(os.path.join(base, n))
    return sorted(out)


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout)
        raise RuntimeError("build failed: " + " ".join(cmd))
    return r.stdout


def build_hip(force=False):
    srcs = _files(CSRC, (".hip", ".hpp", ".inc")) + _files(INC, (".h",))
    srcs = [s for s in srcs if os.sep + "host" + os.sep not in s] + [os.path.abspath(__file__)]  # the flags live in this file
    if force or _newer(HIP_SO, srcs):
        _run(["hipcc"] + HIP_FLAGS + ["-I" + INC, "-o", HIP_SO, os.path.join(CSRC, "mrt_hip.hip")])
    return HIP_SO


def build_host(force=False):
    host_dir = os.path.join(CSRC, "host")
    srcs = _files(host_dir, (".cpp",))
    deps = srcs + _files(host_dir, (".hpp",)) + _files(INC, (".h", ".hpp")) + [HIP_SO, os.path.abspath(__file__)]
    if force or _newer(HOST_SO, deps):
        _run(["g++"] + HOST_FLAGS + ["-I" + INC, "-I" + host_dir, "-o", HOST_SO] + srcs +
             ["-L" + PKG, "-lmrt_hip", "-Wl,-rpath,$ORIGIN"])
    return HOST_SO


def build_oracle(force=False):
    if force or _newer(ORACLE_SO, _files(ORACLE, (".cpp", ".h", ".inc")) + _files(INC, (".h",))):
        _run(["make", "-C", ORACLE, "oracle"] + (["-B"] if force else []))
    return ORACLE_SO


def build_ref(force=False):
    """Only in the build container: the reference sources do not travel."""
    if not os.path.isdir(os.path.join(REFERENCE, "src")):
        return None
    if force or _newer(REF_SO, _files(os.path.join(ORACLE, "ref"), (".cpp",)) + [os.path.join(ORACLE, "Makefile")]):
        _run(["make", "-C", ORACLE, "ref"] + (["-B"] if force else []))
    return REF_SO


EXAMPLE_BIN = os.path.join(PKG, "mrt_main")


def build_example(force=False):
    """examples/main.cpp, the reference's main.cpp against the host library."""
    src = os.path.join(ROOT, "examples", "main.cpp")
    if force or _newer(EXAMPLE_BIN, [src, HOST_SO] + _files(INC, (".h", ".hpp"))):
        _run(["g++", "-std=c++17", "-O2", "-I" + INC, "-o", EXAMPLE_BIN, src, "-L" + PKG, "-lmrt_host", "-lmrt_hip", "-Wl,-rpath,$ORIGIN"])
    return EXAMPLE_BIN


SANITIZE_BIN = os.path.join(ROOT, "tests", "sanitize", "host_sanitize")


def build_sanitize(force=False):
    """The host library's sources, the host side of the HIP library (mrt_hip.hip: the C-ABI's
    validation and set-up code) and tests/sanitize/host_sanitize.cpp compiled with AddressSanitizer +
    UndefinedBehaviorSanitizer into one executable.  Host code only: GPU sanitizers are not available,
    so the .hip file gets its -fsanitize options through -Xarch_host.  Everything is built by the
    ROCm clang so that one sanitizer runtime serves all objects."""
    host_dir = os.path.join(CSRC, "host")
    clang = "/opt/rocm/lib/llvm/bin/clang++"
    srcs = _files(host_dir, (".cpp",)) + [os.path.join(ROOT, "tests", "sanitize", "host_sanitize.cpp")]
    hip_deps = [f for f in _files(CSRC, (".hip", ".hpp", ".inc")) if os.sep + "host" + os.sep not in f]
    deps = srcs + hip_deps + _files(host_dir, (".hpp",)) + _files(INC, (".h", ".hpp"))
    if force or _newer(SANITIZE_BIN, deps):
        import tempfile
        san = ["-fsanitize=address", "-fsanitize=undefined", "-fno-sanitize-recover=undefined"]
        with tempfile.TemporaryDirectory() as tmp:
            objs = [os.path.join(tmp, "mrt_hip.o")]
            _run(["hipcc", "--offload-arch=gfx950", "-O1", "-g", "-ffp-contract=off", "-std=c++17"] +
                 [a for f in san for a in ("-Xarch_host", f)] + ["-I" + INC, "-c", "-o", objs[0], os.path.join(CSRC, "mrt_hip.hip")])
            for src in srcs:
                objs.append(os.path.join(tmp, os.path.basename(src)[:-4] + ".o"))
                _run([clang, "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer"] + san +
                     ["-fopenmp", "-ffp-contract=off", "-I" + INC, "-I" + host_dir, "-c", "-o", objs[-1], src])
            _run(["hipcc", "-fno-gpu-sanitize"] + san + ["-fopenmp", "-o", SANITIZE_BIN] + objs + ["-Wl,-rpath,/opt/rocm/lib/llvm/lib"])
    return SANITIZE_BIN


GATHER_BENCH = os.path.join(PKG, "gather_bench")


def build_tools(force=False):
    """tools/micro/gather_bench.hip: the random-gather ceilings the traversal kernels are priced against."""
    src = os.path.join(ROOT, "tools", "micro", "gather_bench.hip")
    if force or _newer(GATHER_BENCH, [src]):
        _run(["hipcc", "--offload-arch=gfx950", "-O3", "-o", GATHER_BENCH, src])
    return GATHER_BENCH


def build_all(force=False):
    build_hip(force)
    build_host(force)
    build_example(force)
    build_oracle(force)
    build_ref(force)
    build_tools(force)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
    print("built:", HIP_SO, HOST_SO, ORACLE_SO, REF_SO if os.path.exists(REF_SO) else "(no reference build)")
