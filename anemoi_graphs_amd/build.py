"""Builds ``lib/libanemoi_graphs_hip.so`` in-tree with hipcc for gfx950.

    python -m anemoi_graphs_amd.build [--force]

The library is compiled ahead of time (hipcc cross-compiles without a GPU) and travels to the
GPU machine as a file; nothing is compiled at import time.
"""
from __future__ import annotations

import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIB_DIR = PKG / "lib"
LIB_PATH = LIB_DIR / "libanemoi_graphs_hip.so"
OBJ_DIR = PKG / "csrc" / "_obj"

HIPCC = "/opt/rocm/bin/hipcc"
FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-fPIC",
    # scikit-learn's float64 ranking formula is evaluated with separate roundings
    "-ffp-contract=off",
    "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops",
    "-Wno-unused-command-line-argument",
]


def _sources() -> list[Path]:
    return sorted(CSRC.glob("*.hip"))


def _stale(target: Path, deps: list[Path]) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(d.stat().st_mtime > t for d in deps)


def _run(cmd: list[str]) -> None:
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        sys.stderr.write(proc.stdout + proc.stderr)
        raise RuntimeError("command failed: " + " ".join(cmd))


def build(force: bool = False, verbose: bool = True) -> Path:
    headers = sorted(CSRC.glob("*.hpp")) + sorted(CSRC.glob("*.inc")) + sorted((PKG.parent / "include").glob("*.h"))
    OBJ_DIR.mkdir(parents=True, exist_ok=True)
    LIB_DIR.mkdir(parents=True, exist_ok=True)
    jobs = []
    objs = []
    for src in _sources():
        obj = OBJ_DIR / (src.stem + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            jobs.append([HIPCC, *FLAGS, "-c", str(src), "-o", str(obj)])
    if jobs:
        if verbose:
            print(f"[build] compiling {len(jobs)} HIP source(s) for gfx950", flush=True)
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as pool:
            list(pool.map(_run, jobs))
    if force or jobs or _stale(LIB_PATH, objs):
        _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB_PATH), *map(str, objs)])
        if verbose:
            print(f"[build] linked {LIB_PATH}", flush=True)
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv)
