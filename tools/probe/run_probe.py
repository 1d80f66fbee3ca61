import ctypes, os, sys, subprocess, torch, numpy as np
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "libprobe.so")
if not os.path.exists(so):
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(here, "mfma_probe.hip"), "-o", so])
lib = ctypes.CDLL(so)
dev = torch.device("cuda:0")
print(torch.cuda.get_device_name(0), torch.cuda.get_device_properties(0).multi_processor_count)
K = 6
g = torch.Generator().manual_seed(0)
A = torch.randn(32, K, generator=g); B = torch.randn(K, 32, generator=g); M = torch.randn(32, 32, generator=g)
Ad, Bd, Md = A.to(dev), B.to(dev), M.to(dev)
raw = torch.zeros(64, 16, device=dev); D = torch.zeros(32, 32, device=dev); Z = torch.zeros(32, 32, device=dev)
s = torch.cuda.current_stream().cuda_stream
p = lambda t: ctypes.c_void_p(t.data_ptr())
rc = lib.probe_run(p(Ad), p(Bd), K, p(Md), p(raw), p(D), p(Z), ctypes.c_void_p(s))
torch.cuda.synchronize()
ref = (A.double() @ B.double())
print("rc", rc, "D err", (D.cpu().double() - ref).abs().max().item(), "Z err", (Z.cpu().double() - M.double() @ ref).abs().max().item())
import ctypes.util
print([l for l in open('/proc/self/maps').read().split('\n') if 'amdhip64' in l][:2])
