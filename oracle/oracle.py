"""Python driver of the CPU oracle (oracle/dw_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.  It builds the
C restatement with the Makefile next to it and drives it through the same ctypes prototypes
(isaacgymdyros_amd/abi.py) as the HIP library, with numpy arrays standing where device tensors stand.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from isaacgymdyros_amd import abi
from isaacgymdyros_amd.model import load_model

HERE = os.path.dirname(os.path.abspath(__file__))
BUILD = os.path.join(HERE, "_build")


def build(force: bool = False) -> None:
    so = os.path.join(BUILD, "libdw_oracle.so")
    if force or not os.path.exists(so):
        subprocess.check_call(["make", "-C", HERE, "-s"] + (["-B"] if force else []))


_libs = {}


def load(double: bool = False):
    key = "64" if double else "32"
    if key not in _libs:
        build()
        path = os.path.join(BUILD, "libdw_oracle64.so" if double else "libdw_oracle.so")
        lib = C.CDLL(path)
        api = abi.declare(lib, "dwo_")
        lib.dwo_forward_dynamics.restype = C.c_int
        lib.dwo_forward_dynamics.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        _libs[key] = (lib, api)
    return _libs[key]


def default_config(num_envs: int, **over) -> abi.DwConfig:
    _, api = load()
    cfg = abi.DwConfig()
    api["default_config"](C.byref(cfg))
    cfg.num_envs = num_envs
    for k, v in over.items():
        cur = getattr(cfg, k)
        if hasattr(cur, "__len__"):
            for i, x in enumerate(v):
                cur[i] = x
        else:
            setattr(cfg, k, v)
    return cfg


def terrain_config(terrain, max_episode_length_s: float = 32.0) -> dict:
    """DwConfig fields of a isaacgymdyros_amd.terrain.Terrain (height field)."""
    c = terrain.cfg
    return dict(terrain=1, custom_origins=1, terrain_rows=int(terrain.tot_rows), terrain_cols=int(terrain.tot_cols),
                terrain_hscale=float(c.horizontal_scale), terrain_vscale=float(c.vertical_scale),
                terrain_border=float(c.border_size), terrain_curriculum=int(bool(c.curriculum)),
                terrain_num_levels=int(c.num_rows), terrain_num_types=int(c.num_cols),
                terrain_env_length=float(terrain.env_length), max_episode_length_s=float(max_episode_length_s))


def alloc_buffers(num_envs: int):
    bufs = {}
    for name, (shape, dt) in abi.BUFFER_SPECS.items():
        if shape is None:
            bufs[name] = np.zeros((abi.GLOBAL_WORDS[name],), dtype=dt)
        else:
            bufs[name] = np.zeros((num_envs,) + tuple(shape), dtype=dt)
    return bufs


class OracleSim:
    """Owns numpy buffers in the DwBuffers layout and a dwo_ handle."""

    def __init__(self, num_envs: int, task_const=None, double: bool = False, cfg: abi.DwConfig = None,
                 lib_api=None, terrain=None, **cfg_over):
        if terrain is not None:
            cfg_over = dict(terrain_config(terrain), **cfg_over)
        self.lib, self.api = lib_api if lib_api is not None else load(double)
        self.model = load_model()
        self.cfg = cfg if cfg is not None else default_config(num_envs, **cfg_over)
        self.cfg.num_envs = num_envs
        self.N = num_envs
        self.cmodel = self.model.to_c()
        self._task_keep = None
        tptr = None
        if task_const is not None:
            tc = abi.DwTaskConst()
            keep = {}
            for k in ("kp", "kv", "action_high", "initial_dof_pos", "mocap", "obs_mean", "obs_var",
                      "dof_armature_nominal", "dof_damping_nominal"):
                arr = np.ascontiguousarray(task_const[k], dtype=np.float32).ravel()
                keep[k] = arr
                setattr(tc, k, arr.ctypes.data_as(C.POINTER(C.c_float)))
            self._task_keep = (tc, keep)
            tptr = C.byref(tc)
        h = C.c_void_p()
        rc = self.api["create"](C.byref(self.cfg), C.byref(self.cmodel), tptr, C.byref(h))
        if rc != 0:
            raise RuntimeError(self.api["last_error"]().decode())
        self.h = h
        self.buf = alloc_buffers(num_envs)
        self.buf["mass_scale"][:] = 1.0
        self.buf["friction_scale"][:] = 1.0
        self.buf["dof_damping"][:] = 0.1
        from isaacgymdyros_amd.model import ARMATURE
        self.buf["dof_armature"][:] = np.asarray(ARMATURE, dtype=np.float32)
        self.buf["root_states"][:, 6] = 1.0
        self.buf["root_states"][:, 2] = self.cfg.initial_height
        self.buf["total_mass"][:] = np.float32(self.model.nominal_total_mass)
        if terrain is not None:
            self.buf["height_samples"] = np.ascontiguousarray(terrain.heightsamples, dtype=np.int16)
            self.buf["terrain_origins"] = np.ascontiguousarray(terrain.env_origins, dtype=np.float32)
        self.bind()

    def bind(self):
        b = abi.DwBuffers()
        for name in abi.BUFFER_NAMES:
            setattr(b, name, self.buf[name].ctypes.data)
        rc = self.api["bind"](self.h, C.byref(b))
        if rc != 0:
            raise RuntimeError(self.api["last_error"]().decode())

    def _chk(self, rc):
        if rc != 0:
            raise RuntimeError(self.api["last_error"]().decode())

    def simulate(self, tau, push=None):
        tau = np.ascontiguousarray(tau, dtype=np.float32)
        assert tau.shape == (self.N, 33)
        pp = None
        if push is not None:
            push = np.ascontiguousarray(push, dtype=np.float32)
            pp = push.ctypes.data
        self._chk(self.api["simulate"](self.h, tau.ctypes.data, pp, None))

    def step(self, actions, noise=None, step_index=0):
        actions = np.ascontiguousarray(actions, dtype=np.float32)
        assert actions.shape == (self.N, 13)
        nz = None
        if noise is not None:
            noise = np.ascontiguousarray(noise, dtype=np.float32)
            assert noise.shape == (self.N, abi.K["DW_NOISE_WORDS"])
            nz = noise.ctypes.data
        self._chk(self.api["step"](self.h, actions.ctypes.data, nz, step_index, None))

    def terrain_log(self):
        """[N, 15 + terrain types]: the reward columns followed by the curriculum's logging columns (dw_terrain_log; libraries that carry it)."""
        out = np.zeros((self.N, abi.K["DW_NUM_REW"] + int(self.cfg.terrain_num_types)), dtype=np.float32)
        self._chk(self.api["terrain_log"](self.h, out.ctypes.data, None))
        return out

    def reset_idx(self, env_ids, noise=None, step_index=0):
        ids = np.ascontiguousarray(env_ids, dtype=np.int32)
        nz = None
        if noise is not None:
            noise = np.ascontiguousarray(noise, dtype=np.float32)
            nz = noise.ctypes.data
        self._chk(self.api["reset_idx"](self.h, ids.ctypes.data, len(ids), nz, step_index, None))

    def forward_dynamics(self, env, tau):
        tau = np.ascontiguousarray(tau, dtype=np.float32)
        qdd = np.zeros(33)
        a0 = np.zeros(6)
        self._chk(self.lib.dwo_forward_dynamics(self.h, env, tau.ctypes.data, qdd.ctypes.data, a0.ctypes.data))
        return qdd, a0

    def __del__(self):
        try:
            self.api["destroy"](self.h)
        except Exception:
            pass
