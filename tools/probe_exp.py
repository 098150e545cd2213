"""Attribution of the non-bit-identical reward terms (tests/test_hip_gpu.py::test_obs_reward_vs_torch_twin_on_gpu): is it exp?
Compares torch-ROCm's exp() with the candidates a HIP kernel can call, on the arguments the reward terms produce."""
import ctypes, os, subprocess, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
so = "/tmp/libprobe_exp.so"
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(here, "probe_exp.hip")])
lib = ctypes.CDLL(so)
n = 1 << 20
g = torch.Generator(device="cuda").manual_seed(0)
for lo, hi in ((-1.0, 0.0), (-10.0, 0.0), (-60.0, 0.0)):
    x = torch.rand(n, generator=g, device="cuda") * (hi - lo) + lo
    outs = [torch.empty_like(x) for _ in range(4)]
    lib.probe_exp(ctypes.c_void_p(x.data_ptr()), *[ctypes.c_void_p(o.data_ptr()) for o in outs], n, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    ref = torch.exp(x)
    names = ["expf (OCML)", "__expf", "exp2f(x*log2e)", "(float)exp(double)"]
    print("x in [%g, %g]:" % (lo, hi), ", ".join("%s %.4f %% identical to torch.exp" % (nm, 100.0 * float((o == ref).float().mean())) for nm, o in zip(names, outs)))
