"""Loads the host emulation of the kernel bodies (tests/emul: the kernel source under g++, one fiber per lane) behind the same driver as the oracle."""
import ctypes as C
import os
import subprocess

from isaacgymdyros_amd import abi
from oracle.oracle import OracleSim

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "emul")
_cache = {}


def load(asan=False, hex=False):
    # the octet kernels (also carries the fused TocabiAMPLower kernels), or the same source as the 16-lanes-per-env instantiation
    name = ("libdw_emul_hex" if hex else "libdw_emul_oct") + ("_asan.so" if asan else ".so")
    if name not in _cache:
        subprocess.check_call(["make", "-C", HERE, "-s", "_build/" + name])
        lib = C.CDLL(os.path.join(HERE, "_build", name))
        _cache[name] = (lib, abi.declare(lib, "dwe_"))
    return _cache[name]


class EmulSim(OracleSim):
    def __init__(self, num_envs, task_const=None, **cfg_over):
        # cfg_over debug_wave_build = 1: the register-resident (KEEP) form of the step, 0 / 2: the two-waves form, 3: the hex
        # instantiation (16 lanes per env; its own library, the same source compiled with OCT_LPE = 16)
        super().__init__(num_envs, task_const=task_const, lib_api=load(hex=cfg_over.get("debug_wave_build", 0) == 3), **cfg_over)


class EmulBackend:
    def __init__(self, N, task_const, **cfg):
        self.sim = EmulSim(N, task_const=task_const, **cfg)

    def load_buffers(self, bufs):
        for k, v in bufs.items():
            self.sim.buf[k][...] = v

    def write_state(self, root, dof, cf):
        self.sim.buf["root_states"][...] = root
        self.sim.buf["dof_state"][...] = dof
        self.sim.buf["contact_forces"][...] = cf

    def step(self, a, nz, t):
        self.sim.step(a, nz, t)

    def read_buffers(self):
        return self.sim.buf
