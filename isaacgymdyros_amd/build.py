"""Builds libdyroswalk_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libdyroswalk_hip.so")
# (source, its own flags).  The octet kernels are built with the iterative-ilp machine scheduler: the substep is a chain of short
# dependent regions at two waves per SIMD, and a scheduler that lengthens the distance between an LDS load and its first use
# pays directly (0.1477 -> 0.1444 ms at 16384 envs, 0.1188 -> 0.1176 at 4096, no scratch in the flat kernels; max-ilp,
# iterative-minreg and the default max-occupancy scheduler lose; A/Bs of round 3).  The small kernels keep
# the compiler's default.
# Round 6 (a sweep of backend switches on the final sources, same-box A/Bs: profiles/r06_flag_sweep.txt): no SDWA peephole (the sub-dword forms
# are 8-byte encodings: same instruction count, fewer instruction-fetch waits: 0.1391 -> 0.1366 ms at 16384 envs) and no loop strength
# reduction (its induction variables cost the chain loops registers: 0.1369 -> 0.1356 ms, the height-field kernels' scratch 156 -> 116 / 136 B);
# together flat 0.1394 -> 0.1360 ms, height field 0.1617 -> 0.1582 ms, 4096 envs (hex) 0.0977 -> 0.0967 ms.
KERNEL_UNIT = ["-mllvm", "-amdgpu-sched-strategy=iterative-ilp", "-mllvm", "-amdgpu-sdwa-peephole=false", "-mllvm", "-disable-lsr"]
SOURCES = [("dw_hip.hip", []),
           ("dw_oct_kernels.hip", KERNEL_UNIT),
           ("dw_hex_kernels.hip", KERNEL_UNIT), ("dw_amp.hip", []), ("dw_ppo.hip", ["-munsafe-fp-atomics"])]
HEADERS = ["dw_wave.h", "dw_devmodel.h", "dw_physics.h", "dw_task.h", "dw_params.h", "dw_quad_wave.h", "dw_quad_model.h",
           "dw_limb.h", "dw_bufg.h", "dw_oct.h", "dw_oct_kernels.h", "dw_oct_post.h", "dw_handle.h", "dw_amp.h", "dw_amp_step.h"]
# -fno-slp-vectorize: the SLP vectoriser packs adjacent scalar f32 math into v_pk_*_f32 pairs: in the octet step kernel 1 920
# packed instructions replace 4 079 scalar ones, but 775 v_mov are added to form the pairs and the two-waves-per-SIMD build (256
# registers) goes from 0 to 612 B of scratch: 0.149 -> 0.199 ms (round 3), re-measured in round 4 with -slp-threshold 2..16 (DESIGN.md
# section 7)
# -O2 rather than -O3: 1 % faster with this scheduler (less aggressive unrolling, same zero scratch)
FLAGS = ["--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-fno-strict-aliasing", "-fno-slp-vectorize"]


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found; the MI355X kernels cannot be built")


def stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in [s for s, _ in SOURCES] + HEADERS] + [os.path.join(os.path.dirname(PKG), "include", h) for h in ("dyros_walk.h", "dyros_ppo.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if force or stale():
        objs = []
        os.makedirs(os.path.join(PKG, "_obj"), exist_ok=True)
        for src, extra in SOURCES:
            obj = os.path.join(PKG, "_obj", os.path.splitext(src)[0] + ".o")
            cmd = [hipcc()] + FLAGS + extra + ["-c", "-o", obj, os.path.join(CSRC, src)]
            if verbose:
                cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
            subprocess.check_call(cmd, cwd=CSRC)
            objs.append(obj)
        subprocess.check_call([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
