"""Worker of tests/test_kernel_sanitizers.py: replays the goldens through the ASan/UBSan build of the kernel-body emulation."""
import sys, os, ctypes as C
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
import numpy as np
import emul_backend, replay as R
from isaacgymdyros_amd import abi
from isaacgymdyros_amd.task_constants import load_task_constants
WB = int(sys.argv[1]) if len(sys.argv) > 1 else 2          # DwConfig.debug_wave_build: 2 = two-waves form, 1 = register-resident form
stem = 'libdw_emul_hex' if WB == 3 else 'libdw_emul_oct'
lib = C.CDLL(os.path.join(HERE, 'emul', '_build', stem + '_asan.so'))
emul_backend._cache[stem + '.so'] = (lib, abi.declare(lib, 'dwe_'))
tc = load_task_constants()
for name, kw in (('task_logic_frozen.npz', dict(debug_freeze_physics=1)), ('whole_step_oracle.npz', {})):
    g = R.load(name)
    be = emul_backend.EmulBackend(int(g['N']), tc, debug_wave_build=WB, randomize_dof_on_reset=0, torch_gpu_div=0, **kw)
    n = 0
    for t, ref, got in R.replay(g, be):
        n += 1
        if n >= 40: break
    print(name, 'replayed', n, 'steps under ASan/UBSan')
# reset_idx + simulate paths
sim = emul_backend.EmulSim(8, task_const=tc, debug_wave_build=WB)
sim.simulate(np.zeros((8,33),np.float32), np.ones((8,2),np.float32))
sim.reset_idx(np.array([0,3,7],np.int32))
sim.step(np.zeros((8,13),np.float32), None, 0)
# self-collision resolution paths (legs crossed, arms pressed into the torso) and a ragged last wave (17 envs)
from test_oracle_physics import _arms_in
sim = emul_backend.EmulSim(17, task_const=tc, debug_wave_build=WB)
q = _arms_in(17)
q[::2, 1] = -0.2
q[::2, 7] = 0.2
sim.buf['root_states'][:, 2] = 3.0
sim.buf['dof_state'][:, :, 0] = q
sim.simulate(np.zeros((17,33),np.float32))
assert (np.linalg.norm(sim.buf['contact_forces'], axis=2) > 1).any(axis=1).sum() >= 8
print('simulate / reset_idx / step(noise=None) ok')
# the fused TocabiAMPLower step and reset (csrc/dw_amp_step.h; exported by the octet emulation): rings and the shifting layout,
# device draws and the caller's, a ragged env count, episodes short enough that every env resets
if WB == 2:
    from amp_emul import AmpEmul
    for ring in (True, False):
        env = AmpEmul(emul_backend.EmulSim(5, self_collision=0), 5, hist_ring=ring, episode_length=6.0, pd_control=not ring)
        rng = np.random.default_rng(0)
        for t in range(14):
            env.reset_done()
            env.step((rng.random((5, 12), dtype=np.float32) * 2 - 1) * 1.2)
        assert np.isfinite(env.a['obs_buf']).all()
    print('fused amp step / reset ok')
