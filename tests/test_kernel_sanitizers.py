"""GPU AddressSanitizer is not available on the pool, so the kernel bodies are sanitised on the CPU: the host emulations
(tests/emul) are built with -fsanitize=address,undefined and replay both golden fixtures plus the simulate / reset_idx /
in-kernel-RNG / self-collision paths -- for both forms of the octet step (two waves per SIMD; the register-resident form of the
one-wave build) and for the hex instantiation of the same source (16 lanes per env), one fiber per lane -- and the fused TocabiAMPLower kernels.  Any out-of-bounds LDS or buffer index in the shared kernel source aborts the worker."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _lib(name):
    try:
        p = subprocess.check_output(["gcc", "-print-file-name=" + name], text=True).strip()
        return p if os.path.isabs(p) and os.path.exists(p) else None
    except Exception:
        return None


@pytest.mark.parametrize("wave_build", [2, 1, 3], ids=["two_waves", "keep", "hex"])
def test_kernel_body_under_asan_ubsan(wave_build):
    asan, ubsan = _lib("libasan.so"), _lib("libubsan.so")
    if not asan or not ubsan:
        pytest.skip("libasan/libubsan not found")
    subprocess.check_call(["make", "-C", os.path.join(HERE, "emul"), "-s", "_build/libdw_emul_%s_asan.so" % ("hex" if wave_build == 3 else "oct")])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1",
               LD_PRELOAD=asan + ":" + ubsan, OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, os.path.join(HERE, "_asan_worker.py"), str(wave_build)], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "replayed 40 steps" in out.stdout and "simulate / reset_idx / step(noise=None) ok" in out.stdout
    assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr
    if wave_build == 2:          # (the emulation also carries the fused TocabiAMPLower kernels, csrc/dw_amp_step.h)
        assert "fused amp step / reset ok" in out.stdout
