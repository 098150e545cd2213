"""Worker of tests/test_kernel_sanitizers.py: replays the goldens through the ASan/UBSan build of the kernel-body emulation."""
import sys, os, ctypes as C
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
import numpy as np
import emul_backend, replay as R
from isaacgymdyros_amd import abi
from isaacgymdyros_amd.task_constants import load_task_constants
lib = C.CDLL(os.path.join(HERE, 'emul', '_build', 'libdw_emul_asan.so'))
emul_backend._cache['libdw_emul.so'] = (lib, abi.declare(lib, 'dwe_'))
tc = load_task_constants()
for name, kw in (('task_logic_frozen.npz', dict(debug_freeze_physics=1)), ('whole_step_oracle.npz', {})):
    g = R.load(name)
    be = emul_backend.EmulBackend(int(g['N']), tc, randomize_dof_on_reset=0, torch_gpu_div=0, **kw)
    n = 0
    for t, ref, got in R.replay(g, be):
        n += 1
        if n >= 40: break
    print(name, 'replayed', n, 'steps under ASan/UBSan')
# reset_idx + simulate paths
sim = emul_backend.EmulSim(8, task_const=tc)
sim.simulate(np.zeros((8,33),np.float32), np.ones((8,2),np.float32))
sim.reset_idx(np.array([0,3,7],np.int32))
sim.step(np.zeros((8,13),np.float32), None, 0)
print('simulate / reset_idx / step(noise=None) ok')
