"""Per-phase latency of the octet step kernel for wave 0 (profiling build: tools/phase_stamps.sh builds the library with
-DDQ_STAMPS into isaacgymdyros_amd/_ab/ and runs this).  Prints cycles between the stamps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from isaacgymdyros_amd import _lib
_lib.LIB_PATH = os.environ.get("DW_LIB", _lib.LIB_PATH)
from isaacgymdyros_amd.config import default_cfg
from isaacgymdyros_amd.dyros_dynamic_walk import DyrosDynamicWalk
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
cfg = default_cfg(N, "cuda:0")
if os.environ.get("DW_TERRAIN"):          # the height-field variant (10 x 20 curriculum map)
    from isaacgymdyros_amd.config import with_terrain
    cfg = with_terrain(cfg, mesh_type="trimesh", curriculum=True)
cfg["sim"]["mi355"]["pipeline"] = int(os.environ.get("DW_PIPE", "0"))          # 3 octet (the -DDQ_STAMPS build of tools/phase_stamps.sh stamps the octet unit)
env = DyrosDynamicWalk(cfg, "cuda:0", 0, True)
g = torch.Generator(device="cuda").manual_seed(42)
acts = [torch.rand(N, 13, generator=g, device="cuda") * 2 - 1 for _ in range(8)]
names = {0: "substep entry", 1: "base kin -> FK", 2: "FK", 3: "self-collision", 4: "inward", 5: "base solve", 6: "outward", 7: "corners, free twist",
         8: "W rows", 9: "A_kk, start", 10: "PGS", 11: "impulse up-sweep", 12: "final pass", 13: "base integrate"}
acc = None
K = 20
for i in range(30 + K):
    env.step(acts[i % 8])
    if i >= 30:
        torch.cuda.synchronize()
        st = env._buf["gate_acc"][200:256].cpu().numpy().astype("int64")
        acc = st.copy() if acc is None else acc
        d = {}
        tot = st[40] - st[0]
        if i == 30 + K - 1:
            print("kernel total (wave 0): %d cycles; before that, pre_physics_step up to the actuator pass: %d" % (tot, st[0] - st[54]))
            for sub in (0, 1):
                base = 1 + 16 * sub
                print("substep %d: prologue %d" % (sub, st[base + 0] - (st[0] if sub == 0 else st[1 + 14])))
                for n in range(1, 14):
                    print("   %-22s %7d" % (names[n], st[base + n] - st[base + n - 1]))
                print("   %-22s %7d" % ("encoder epilogue", st[base + 14] - st[base + 13]))
            print("self-collision of substep 1: proxies %d, detection %d, resolution %d (any hit in the wave: %d)" % (st[51] - st[1 + 16 + 2], st[52] - st[51], st[1 + 16 + 3] - st[52], st[53]))
            print("kernel epilogue %d" % (st[40] - st[1 + 16 + 14]))
            if st[34] > 0:
                print("encoder epilogue of substep 0: loads issued %d, noise %d, arithmetic + stores %d" % (st[34] - st[1 + 13], st[35] - st[34], st[1 + 14] - st[35]))
            if st[32] > 0:
                print("inward pass of substep 0 (octet kernels): map %d, recursion + hand-over %d" % (st[32], st[33]))
            if st[41] > st[40]:
                print("post_physics_step %d" % (st[41] - st[40]))
                pn = {42: "stage", 43: "Q1 + guard", 44: "Q2 reward", 45: "Q3", 46: "reset", 47: "Q4 obs", 48: "Q5 obs_buf", 49: "Q6", 50: "write back"}
                prev = st[40]
                for n in range(42, 51):
                    if st[n] > 0:
                        print("   %-22s %7d" % (pn[n], st[n] - prev)); prev = st[n]
