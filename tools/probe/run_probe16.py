import ctypes, os, subprocess, torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "libprobe16.so")
if not os.path.exists(so):
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(here, "mfma16_probe.hip"), "-o", so])
lib = ctypes.CDLL(so)
dev = torch.device("cuda:0")
K = 6
g = torch.Generator().manual_seed(0)
A = torch.randn(32, K, generator=g); B = torch.randn(K, 32, generator=g); W = torch.randn(32, 16, generator=g)
Ad, Bd, Wd = A.to(dev), B.to(dev), W.to(dev)
Out = torch.zeros(16, 32, device=dev); clk = torch.zeros(2, dtype=torch.int64, device=dev)
p = lambda t: ctypes.c_void_p(t.data_ptr())
rc = lib.probe16_run(p(Ad), p(Bd), K, p(Wd), p(Out), p(clk), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
ref = W.double().T @ (A.double() @ B.double())
print("rc", rc, "dA16 err", (Out.cpu().double() - ref).abs().max().item(), "ref max", ref.abs().max().item())
print("clocks per v_mfma_f32_16x16x1_4b_f32:", clk[0].item() / 256, " per v_mfma_f32_32x32x2_f32:", clk[1].item() / 256)
