"""Build helpers: compile liborbx.so (HIP, gfx950 only) and the CPU checkers under oracle/.

`python -m` is awkward with the hyphenated package name, so __graft_entry__.build() imports
this file by path.  Everything is built IN-TREE so the .so files travel to the GPU box with
the gpurun snapshot.
"""
import os
import shutil
import subprocess
from pathlib import Path

PKG = Path(__file__).resolve().parent
ROOT = PKG.parent
CSRC = PKG / "csrc"
LIB = PKG / "lib" / "liborbx.so"
HIP_SOURCES = ["orbx_kernels.hip", "orbx_extractor.hip", "orbx_match.hip", "orbx_match_proj.hip", "orbx_lba.hip", "orbx_bow.hip", "orbx_frame.hip"]
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
             "-Wall", "-Wno-unused-function"]


def _newer(target: Path, deps) -> bool:
    if not target.exists():
        return False
    t = target.stat().st_mtime
    return all(Path(d).stat().st_mtime <= t for d in deps if Path(d).exists())


def hipcc_path():
    for c in ("hipcc", "/opt/rocm/bin/hipcc"):
        p = shutil.which(c)
        if p:
            return p
    return None


def build_liborbx(force=False, verbose=True):
    """Several ranks of one node may import the package at the same moment (bench.py --gpus N): the staleness check and the
    compilation run under an exclusive file lock, and the library is linked to a temporary name and renamed into place, so
    no rank ever dlopens a half-written file."""
    import fcntl
    LIB.parent.mkdir(parents=True, exist_ok=True)
    with open(LIB.parent / ".build.lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return _build_liborbx_locked(force, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _flags_stamp(hipcc):
    import hashlib
    try:
        ver = subprocess.run([hipcc, "--version"], capture_output=True, text=True).stdout
    except OSError:
        ver = ""
    return hashlib.sha256((" ".join(HIP_FLAGS) + "\n" + ver).encode()).hexdigest()


def _stamp_matches(hipcc):
    f = LIB.parent / "obj" / "flags.stamp"
    return f.exists() and f.read_text() == _flags_stamp(hipcc)


def _build_liborbx_locked(force, verbose):
    """One object per source (compiled in parallel, only the stale ones), then one link: a kernel edit costs the compile of its own
    file, not of all eight."""
    from concurrent.futures import ThreadPoolExecutor
    srcs = [CSRC / s for s in HIP_SOURCES if (CSRC / s).exists()]
    common = list(CSRC.glob("*.h")) + list(CSRC.glob("*.inc")) + [ROOT / "include" / "orbx.h"]
    hipcc = hipcc_path()
    if hipcc is None:
        if LIB.exists():
            return LIB   # GPU box without a compiler: use the prebuilt library from the snapshot
        raise RuntimeError("hipcc not found and no prebuilt liborbx.so")
    objdir = LIB.parent / "obj"
    objdir.mkdir(parents=True, exist_ok=True)
    # objects are only as good as the flags and the compiler they came from (bit-exactness hangs on -ffp-contract=off): both are
    # hashed into a stamp next to the objects, and a different (or missing) stamp recompiles everything
    stamp = _flags_stamp(hipcc)
    stamp_file = objdir / "flags.stamp"
    if not _stamp_matches(hipcc):
        force = True
    if not force and _newer(LIB, srcs + common):
        return LIB
    cflags = [f for f in HIP_FLAGS if f != "-shared"]
    jobs = []
    for s in srcs:
        o = objdir / (s.name + ".o")
        if force or not _newer(o, [s] + common):
            jobs.append((s, o))

    def compile_one(job):
        s, o = job
        cmd = [hipcc] + cflags + ["-c", "-o", str(o), str(s)]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 1))) as ex:
        list(ex.map(compile_one, jobs))
    tmp = LIB.with_suffix(".so.tmp%d" % os.getpid())
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(tmp)] + [str(objdir / (s.name + ".o")) for s in srcs]
    if verbose:
        print("[build]", " ".join(cmd).replace(str(tmp), str(LIB)), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(tmp, LIB)
    stamp_file.write_text(stamp)
    return LIB


SYNTH_LIB = PKG / "lib" / "liborbx_synth.so"


def build_synth(verbose=True):
    """Synthetic test / bench frames (synth/orbx_synth.cc): a host-only helper library of its own, NOT part of liborbx.so."""
    src = [PKG / "synth" / "orbx_synth.cc", PKG / "synth" / "orbx_synth.h"]
    if _newer(SYNTH_LIB, src):
        return SYNTH_LIB
    cxx = shutil.which("g++") or shutil.which("c++")
    if cxx is None:
        if SYNTH_LIB.exists():
            return SYNTH_LIB
        raise RuntimeError("no C++ compiler and no prebuilt liborbx_synth.so")
    SYNTH_LIB.parent.mkdir(parents=True, exist_ok=True)
    tmp = SYNTH_LIB.with_suffix(".so.tmp%d" % os.getpid())
    cmd = [cxx, "-O2", "-std=c++11", "-fPIC", "-shared", "-o", str(tmp), str(src[0])]
    if verbose:
        print("[build]", " ".join(cmd).replace(str(tmp), str(SYNTH_LIB)), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(tmp, SYNTH_LIB)
    return SYNTH_LIB


def build_oracle(verbose=True):
    """Compile the CPU checkers (test infrastructure).  oracle/_ref needs /root/reference."""
    mk = ROOT / "oracle" / "Makefile"
    if shutil.which("make") is None or shutil.which("g++") is None:
        return
    out = None if verbose else subprocess.DEVNULL
    jobs = str(max(1, min(16, os.cpu_count() or 1)))      # ~45 translation units since the vendored g2o joined the reference build
    subprocess.run(["make", "-s", "-j", jobs, "-f", str(mk), "all"], check=True, stdout=out)


def build_all(force=False, verbose=True):
    build_liborbx(force=force, verbose=verbose)
    build_synth(verbose=verbose)
    build_oracle(verbose=verbose)
    return LIB
