"""Build every native library of the stack in-tree.

HIP device code is compiled for gfx950 only (``hipcc --offload-arch=gfx950``; it
cross-compiles without a GPU present); host-only C++ is built with g++.  The shared
objects land next to this file so that they travel with the source tree and are loaded
with ``ctypes`` (no Python C-API, no torch headers: a full rebuild takes seconds).

Usage:  python -m vllm_production_stack_amd._native.build [--force] [target ...]
"""

from __future__ import annotations

import os
import shutil
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
CSRC = ROOT / "csrc"
ROCM = Path(os.environ.get("ROCM_PATH", "/opt/rocm"))

HIPCC = shutil.which("hipcc") or str(ROCM / "bin" / "hipcc")
CXX = shutil.which("g++") or "g++"

# target name -> (compiler kind, sources relative to csrc/, extra flags)
TARGETS = {
    "vpsa_kernels": (
        "hip",
        ["kernels/elementwise.hip", "kernels/attention.hip", "kernels/kvcache.hip", "kernels/moe.hip",
         "kernels/lora.hip", "kernels/gemm.hip", "kernels/rans.hip", "kernels/sampling.hip",
         "kernels/grouped_gemm.hip", "kernels/packed_gemm.hip", "kernels/dense_attention.hip",
         "kernels/tile_gemm.hip"],
        [],
    ),
    "vpsa_kvio": ("hip", ["kvio/kvio.cpp"], []),
    "vpsa_comm": ("hip", ["comm/comm.cpp"], ["-lrccl"]),
    "vpsa_routercore": ("cxx", ["routercore/routercore.cpp"], []),
    # measurement tool, not part of the serving path: the card's HBM read roof (benchmarks/roof_probe.py)
    "vpsa_roof_probe": ("hip", ["tools/roof_probe.hip"], []),
}

# extra compile flags for single sources.  attention.hip: the one-wave-per-SIMD prefill kernel keeps
# its score tiles in arch VGPRs (MFMA "VGPR form") and its output accumulators in AGPRs (inline
# asm); without the flag every MFMA result of a 512-register kernel lands in AGPRs and has to be
# copied out one register at a time before the softmax can read it.
SOURCE_FLAGS = {
    "kernels/attention.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
}

# standalone executables: name -> (kind, sources, flags)
BINARIES = {
    "vpsa-picker": ("cxx", ["routercore/picker_service.cpp", "routercore/routercore.cpp"], ["-lpthread"]),
    "vpsa-operator": ("cxx", ["operator/operator_main.cpp"], ["-DVHTTP_TLS", "-lssl", "-lcrypto", "-lpthread"]),
}


def _stale(out: Path, srcs: list[Path]) -> bool:
    if not out.exists():
        return True
    t = out.stat().st_mtime
    deps = list(srcs) + list((CSRC / "common").glob("*.hpp"))
    for s in srcs:
        deps.extend(s.parent.glob("*.h"))
        deps.extend(s.parent.glob("*.hpp"))
    return any(d.stat().st_mtime > t for d in deps if d.exists())


def _run(cmd: list[str]) -> None:
    print("[build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)


def build_target(name: str, force: bool = False) -> Path | None:
    if name in TARGETS:
        kind, srcs, flags = TARGETS[name]
        out = HERE / f"lib{name}.so"
        shared = ["-shared", "-fPIC"]
    else:
        kind, srcs, flags = BINARIES[name]
        out = HERE / name
        shared = []
    paths = [CSRC / s for s in srcs]
    missing = [p for p in paths if not p.exists()]
    if missing:
        print(f"[build] skip {name}: missing {[str(m) for m in missing]}")
        return None
    if not force and not _stale(out, paths):
        return out
    if kind == "hip" and any(src in SOURCE_FLAGS for src in srcs):
        # sources with flags of their own: compile every source to an object, then link
        objdir = HERE / "_obj" / name
        objdir.mkdir(parents=True, exist_ok=True)
        objs = []
        for src, path in zip(srcs, paths):
            obj = objdir / (path.stem + ".o")
            if force or _stale(obj, [path]):
                _run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
                      "-I", str(CSRC), *SOURCE_FLAGS.get(src, []), "-c", str(path), "-o", str(obj)])
            objs.append(str(obj))
        cmd = [HIPCC, "--offload-arch=gfx950", *shared, *objs, "-o", str(out), *flags]
    elif kind == "hip":
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", *shared,
               "-fvisibility=hidden", "-I", str(CSRC), *map(str, paths), "-o", str(out), *flags]
    else:
        cmd = [CXX, "-O2", "-std=c++17", *shared, "-I", str(CSRC), *map(str, paths), "-o",
               str(out), *flags]
    _run(cmd)
    return out


def build_all(force: bool = False, targets: list[str] | None = None) -> dict[str, Path | None]:
    names = targets or (list(TARGETS) + list(BINARIES))
    return {n: build_target(n, force) for n in names}


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    res = build_all(force="--force" in sys.argv, targets=args or None)
    for k, v in res.items():
        print(f"{k}: {v}")
