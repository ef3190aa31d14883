"""Build the HIP shared library (gfx950 only) in-tree.

    python accelerated-tinympc_amd/build.py [--force]

Produces accelerated-tinympc_amd/lib/libtinympc_hip.so with `hipcc --offload-arch=gfx950`.
hipcc cross-compiles without a GPU.  The .so is git-ignored but travels with the tree.
"""
from __future__ import annotations

import os
import subprocess
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIB = PKG / "lib" / "libtinympc_hip.so"
WRAPPER_LIB = PKG / "lib" / "libtinympc_wrapper.so"  # same-name twin of the reference's generated wrapper library
WRAPPER64_LIB = PKG / "lib" / "libtinympc_wrapper64.so"  # the native names (tiny_solve, forward_pass, ...) for tinytype = double
SOURCES = ["tinympc_batch.hip", "tinympc_batch64.hip", "admm_stream.hip", "admm_generic.hip", "admm_rowlane.hip", "admm_rowloop.hip", "admm_quadlane.hip", "admm_tile16.hip", "admm_tile16_pi.hip", "admm_wave.hip", "admm_waveres.hip", "admm_tile48.hip", "admm_steps.hip", "dispatch_order.hip", "riccati.cpp"]
WRAPPER_SRCS = [CSRC / "wrapper_compat.cpp", CSRC / "admm_compat.cpp"]
HEADERS = [CSRC / "tinympc_internal.h", CSRC / "rowlane_math.h", CSRC / "tile_math.h", CSRC / "wave_math.h", CSRC / "dpp_ops_gen.h", PKG.parent / "include" / "tinympc_batch.h", PKG.parent / "include" / "tinympc_batch64.h"]
# -ffp-contract=off : exact arithmetic must not fuse a*b+c; the fast paths call fma explicitly
# -fno-slp-vectorize: hipcc otherwise pairs scalar fp32 adds into v_pk_add_f32 (+ v_mov to build the pairs), which on
#                     gfx950 is slower than two plain v_add_f32 (measured, tools/micro/*.hip)
# per-file extras.  admm_tile16.hip: the max-ILP scheduling strategy (one wave per SIMD: nothing else hides a latency; measured
# 1.88 -> 1.79 ms, the other kernels do not react to it); MFMA results go to VGPRs (the sums that consume the exact products are VALU instructions,
# which cannot read the accumulator half of the register file: left to its heuristics hipcc parks the products there and
# copies every one of them back, 110 v_accvgpr_read per horizon step)
_T16_FLAGS = ["-mllvm", "-amdgpu-mfma-vgpr-form=1", "-mllvm", "-amdgpu-sched-strategy=max-ilp"] + os.environ.get("TINYMPC_T16_FLAGS", "").split()
EXTRA_FLAGS = {"admm_tile16.hip": _T16_FLAGS, "admm_tile16_pi.hip": _T16_FLAGS,
               "admm_rowlane.hip": os.environ.get("TINYMPC_ROWLANE_FLAGS", "").split(),
               "admm_waveres.hip": os.environ.get("TINYMPC_WAVERES_FLAGS", "").split(),
               "admm_tile48.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"] + os.environ.get("TINYMPC_T48_FLAGS", "").split()}
# translation units that #include another kernel source: rebuilt with it
INCLUDED_SOURCES = {"admm_tile16_pi.hip": ["admm_tile16.hip"]}
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize", "-fno-gpu-rdc"]


def _objs():
    return [(CSRC / s, PKG / "lib" / (Path(s).stem + ".o")) for s in SOURCES if (CSRC / s).exists()]


def needs_build() -> bool:
    if not LIB.exists() or not WRAPPER_LIB.exists() or not WRAPPER64_LIB.exists():
        return True
    t = min(LIB.stat().st_mtime, WRAPPER_LIB.stat().st_mtime, WRAPPER64_LIB.stat().st_mtime)
    inc = PKG.parent / "include"
    deps = [s for s, _ in _objs()] + HEADERS + [Path(__file__), *WRAPPER_SRCS, inc / "tinympc_wrapper.h", inc / "tinympc_admm.h", inc / "tinympc_batch.h"]
    return any(d.stat().st_mtime > t for d in deps if d.exists())


def build(force: bool = False, verbose: bool = False) -> Path:
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    LIB.parent.mkdir(parents=True, exist_ok=True)
    hdr_t = max(h.stat().st_mtime for h in HEADERS + [Path(__file__)] if h.exists())
    procs = []
    for src, obj in _objs():
        src_t = max([src.stat().st_mtime] + [(CSRC / i).stat().st_mtime for i in INCLUDED_SOURCES.get(src.name, [])])
        if not force and obj.exists() and obj.stat().st_mtime > max(src_t, hdr_t):
            continue
        cmd = [hipcc, *FLAGS, *EXTRA_FLAGS.get(src.name, []), "-c", str(src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB)] + [str(o) for _, o in _objs()]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    # libtinympc_wrapper.so: plain host C++ on top of the C-ABI, finds libtinympc_hip.so next to itself
    cmd = ["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-o", str(WRAPPER_LIB), *[str(w) for w in WRAPPER_SRCS], f"-L{LIB.parent}",
           "-ltinympc_hip", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    # libtinympc_wrapper64.so: the native names over TinySolver with double members (include/tinympc_admm.h, TINYMPC_TINYTYPE_DOUBLE)
    cmd = ["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-DTINYMPC_TINYTYPE_DOUBLE", "-o", str(WRAPPER64_LIB), str(CSRC / "admm_compat.cpp"),
           f"-L{LIB.parent}", "-ltinympc_hip", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB


def device_asm(source: str) -> Path:
    """gfx950 assembly listing of one kernel source, compiled with exactly the flags of build() (`hipcc -S --cuda-device-only`);
    cached under lib/asm/ and redone when the source or a header is newer.  Used by tests/test_isa.py (static hazard and
    register-spill guards over what the compiler actually emitted) and by the developer tools under tools/."""
    src = CSRC / source
    out = PKG / "lib" / "asm" / (Path(source).stem + ".s")
    out.parent.mkdir(parents=True, exist_ok=True)
    newest = max(d.stat().st_mtime for d in [src, Path(__file__), *HEADERS, *[CSRC / i for i in INCLUDED_SOURCES.get(source, [])]] if d.exists())
    if not out.exists() or out.stat().st_mtime < newest:
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        cmd = [hipcc, *[f for f in FLAGS if f != "-fPIC"], *EXTRA_FLAGS.get(source, []), "-S", "--cuda-device-only", str(src), "-o", str(out)]
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    return out


# kernel family (the name tiny_batch_kernel_name() reports, up to the '<') -> the translation unit that holds its device code
KERNEL_SOURCES = {"rowlane": "admm_rowlane.hip", "rowloop": "admm_rowloop.hip", "rowstream": "admm_steps.hip", "quadlane": "admm_quadlane.hip",
                  "tile16": "admm_tile16.hip", "tile48": "admm_tile48.hip", "waveres": "admm_waveres.hip", "wavestream": "admm_wave.hip",
                  "stream": "admm_stream.hip", "generic": "admm_generic.hip", "rows64": "tinympc_batch64.hip", "thread64": "tinympc_batch64.hip"}


def _elf_sections(b: bytes, base: int = 0) -> dict:
    import struct
    assert b[base:base + 4] == b"\x7fELF", "not an ELF image"
    shoff = struct.unpack_from("<Q", b, base + 0x28)[0]
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", b, base + 0x3A)
    secs = [struct.unpack_from("<IIQQQQ", b, base + shoff + i * shentsize) for i in range(shnum)]
    stro = base + secs[shstrndx][4]
    out = {}
    for name, typ, _flags, _addr, off, size in secs:
        out[b[stro + name:b.index(b"\0", stro + name)].decode()] = (base + off, size, typ)
    return out


def device_isa_sha(source: str) -> str:
    """sha256 (16 hex digits) over the gfx950 machine code (.text) and kernel descriptors (.rodata) of one translation unit, read from
    the object file build() left in lib/ — the clang offload bundle inside its .hip_fatbin section, parsed here (no tool is executed:
    bench.py calls this with the GPU initialised).  It binds a measured figure (profiles/hbm_traffic.json) to the CODE it was taken
    on: unlike a hash of the source text it does not move when a comment does."""
    import hashlib
    import struct
    obj = PKG / "lib" / (Path(source).stem + ".o")
    b = obj.read_bytes()
    off, size, _ = _elf_sections(b)[".hip_fatbin"]
    fb = b[off:off + size]
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    assert fb.startswith(magic), "uncompressed clang offload bundle expected"
    n = struct.unpack_from("<Q", fb, len(magic))[0]
    p = len(magic) + 8
    h = hashlib.sha256()
    found = False
    for _ in range(n):
        eo, es, ts = struct.unpack_from("<QQQ", fb, p)
        p += 24
        triple = fb[p:p + ts].decode()
        p += ts
        if "gfx950" in triple and es:
            found = True
            secs = _elf_sections(fb, eo)
            for name in (".text", ".rodata"):
                if name in secs and secs[name][2] != 8:  # (8 = SHT_NOBITS)
                    o, sz, _ = secs[name]
                    h.update(fb[o:o + sz])
    assert found, f"no gfx950 code object in {obj}"
    return h.hexdigest()[:16]


def kernel_isa_sha(kernel_name: str) -> str | None:
    """device_isa_sha of the translation unit behind a kernel name such as 'tile16<12,4,30,exact>'; None for an unknown family or a missing object."""
    src = KERNEL_SOURCES.get(kernel_name.split("<", 1)[0])
    if src == "admm_tile16.hip" and kernel_name.endswith(",pi>"):
        src = "admm_tile16_pi.hip"  # the per-instance instantiations are a translation unit of their own
    try:
        return device_isa_sha(src) if src else None
    except (OSError, AssertionError, KeyError):
        return None


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
