"""Build libpdeopt_hip.so for gfx950 (in-tree, next to the Python package).

    python -m pde_opt_amd.csrc.build          # or: python pde_opt_amd/csrc/build.py

hipcc cross-compiles without a GPU.  Translation units are compiled in parallel and the
objects are cached by source mtime, so a rebuild after touching one file takes seconds.
"""

from __future__ import annotations

import glob
import hashlib
import json
import os
import re
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
ROOT = os.path.dirname(PKG)
LIB = os.path.join(PKG, "libpdeopt_hip.so")
OBJ_DIR = os.path.join(HERE, "build")

SOURCES = ["api.hip", "stencil.hip", "reduce.hip", "spectral.hip", "halo.hip", "strang_fused.hip", "comm.hip", "jit.hip"]


def _embed_jit_source() -> None:
    """csrc/jit_device.hpp -- the device source hiprtc compiles at run time for closures outside the in-kernel family
    (jit.hip) -- as a C++ raw string literal under build/, #included by jit.hip"""
    src = open(os.path.join(HERE, "jit_device.hpp")).read()
    assert ')PDEOPTJIT"' not in src
    out = os.path.join(OBJ_DIR, "jit_device_source.inc")
    text = 'R"PDEOPTJIT(' + src + ')PDEOPTJIT"\n'
    if not os.path.exists(out) or open(out).read() != text:
        with open(out, "w") as f:
            f.write(text)


def _headers() -> list[str]:
    """every header a translation unit may include: all of csrc/*.hpp + the public C header"""
    hs = sorted(glob.glob(os.path.join(HERE, "*.hpp")))
    return hs + [os.path.join(ROOT, "include", "pdeopt_hip.h")]


ARCH = "gfx950"
CXXFLAGS = [
    "-O3",
    "-std=c++17",
    "-fPIC",
    f"--offload-arch={ARCH}",
    "-fno-gpu-rdc",
    # no SLP packing of fp32 pairs into v_pk_* instructions: on gfx950 a packed fp32 FMA issues in 2.4 ns per
    # wave against 2 x 1.45 ns for the two scalar ones it replaces only when nothing else is waiting, and the
    # register pairing it needs costs moves (tools/valubench.hip).  Same-box A/B with the flag: CH RK4 pair
    # kernel +6.5 %, Strang +5 %, IMEX +6 %.
    "-fno-slp-vectorize",
    "-Wall",
    "-Wno-unused-function",
    "-I/opt/rocm/include",
]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP toolchain is required to build libpdeopt_hip.so")
    return exe


def _newest_header() -> float:
    return max(os.path.getmtime(p) for p in _headers() + [os.path.abspath(__file__)])


def _flags_tag(extra: list[str]) -> str:
    """objects built with different flags never share a cache entry"""
    return hashlib.sha1(" ".join(CXXFLAGS + extra).encode()).hexdigest()[:10]


_REMARK = re.compile(r"remark: (?:Function Name: (?P<name>\S+)|\s+(?P<key>VGPRs|AGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|"
                     r"Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (?P<val>\d+))")
_KEYS = {"VGPRs": "vgpr", "AGPRs": "agpr", "TotalSGPRs": "sgpr", "ScratchSize [bytes/lane]": "scratch",
         "Occupancy [waves/SIMD]": "occupancy", "LDS Size [bytes/block]": "lds_static"}


def _parse_resources(stderr: str):
    """hipcc -Rpass-analysis=kernel-resource-usage remarks -> {mangled kernel name: {vgpr, sgpr, scratch, ...}}, and
    the rest of the compiler's output when it holds a warning"""
    res, cur, rest = {}, None, []
    for line in stderr.splitlines():
        m = _REMARK.search(line)
        if m and m.group("name"):
            cur = res.setdefault(m.group("name"), {})
        elif m and cur is not None:
            cur[_KEYS[m.group("key")]] = int(m.group("val"))
        elif "remark:" not in line:
            rest.append(line)
    noise = not any("warning:" in ln or "error:" in ln for ln in rest)  # source excerpts under the remarks
    return res, "" if noise else "\n".join(rest)


LAST_BUILD = {"compiled": [], "reused": [], "linked": False}  # what the last build() did (printed by it)


def _compile(src: str, force: bool, extra: list[str]) -> str:
    obj = os.path.join(OBJ_DIR, src.replace(".hip", f".{_flags_tag(extra)}.o"))
    sp = os.path.join(HERE, src)
    stamp = max(os.path.getmtime(sp), _newest_header())
    if not force and os.path.exists(obj) and os.path.exists(obj + ".resources.json") and os.path.getmtime(obj) >= stamp:
        LAST_BUILD["reused"].append(src)
        return obj
    LAST_BUILD["compiled"].append(src)
    # the remarks cost nothing at run time: registers / scratch / occupancy of every kernel land next to the object,
    # where tests/test_kernel_budgets.py holds the occupancy-critical kernels to their budgets
    cmd = [_hipcc(), *CXXFLAGS, *extra, "-Rpass-analysis=kernel-resource-usage", "-c", sp, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
    res, rest = _parse_resources(r.stderr)
    with open(obj + ".resources.json", "w") as f:
        json.dump(res, f)
    if rest.strip():
        sys.stderr.write(rest + "\n")
    return obj


def kernel_resources(extra_flags: list[str] | None = None) -> dict:
    """{demangled kernel name: {vgpr, sgpr, scratch, occupancy, lds_static}} of the library as built with these
    flags (builds what is stale)"""
    os.makedirs(OBJ_DIR, exist_ok=True)
    _embed_jit_source()
    extra = list(extra_flags or [])
    merged = {}
    for src in SOURCES:
        with open(_compile(src, False, extra) + ".resources.json") as f:
            merged.update(json.load(f))
    names = list(merged)
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    clean = lambda n: re.sub(r"\(.*", "", re.sub(r"pdeopt::|\(anonymous namespace\)::|^void ", "", n))
    return {clean(d): merged[m] for m, d in zip(names, out)}


def build(force: bool = False, verbose: bool = True, extra_flags: list[str] | None = None) -> str:
    os.makedirs(OBJ_DIR, exist_ok=True)
    _embed_jit_source()
    extra = list(extra_flags or [])
    LAST_BUILD.update(compiled=[], reused=[], linked=False)
    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(lambda s: _compile(s, force, extra), SOURCES))
    # the link stamp records which objects (flag sets included) the library was made of
    stamp = os.path.join(OBJ_DIR, "link.stamp")
    want = "\n".join(objs)
    have = open(stamp).read() if os.path.exists(stamp) else ""
    need_link = force or not os.path.exists(LIB) or have != want or any(
        os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs
    )
    if need_link:
        cmd = [_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB, *objs,
               "-L/opt/rocm/lib", "-lrocfft", "-ldl", "-Wl,-rpath,/opt/rocm/lib"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        with open(stamp, "w") as f:
            f.write(want)
        LAST_BUILD["linked"] = True
    if verbose:
        # objects are cached by source / header mtime and flag set: say which were rebuilt and which reused, so a
        # log shows whether the library on a box is this checkout's (tests/test_kernel_budgets.py reads the
        # resource reports written next to the objects by THIS function)
        print(f"[pde_opt_amd] compiled {LAST_BUILD['compiled'] or 'nothing'}, reused {len(LAST_BUILD['reused'])} cached "
              f"object(s), {'linked' if LAST_BUILD['linked'] else 'library up to date'}")
        print(f"[pde_opt_amd] {LIB} ({os.path.getsize(LIB) / 1e6:.1f} MB)")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
