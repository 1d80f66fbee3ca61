import ctypes, os, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "libdma.so"))
dev = "cuda:0"
ld, nrows = 4096, 32
src = torch.arange(64 * ld, dtype=torch.float32, device=dev).view(64, ld)
out = torch.zeros(nrows * 128, device=dev)
p = lambda t: ctypes.c_void_p(t.data_ptr())
rc = lib.probe_dma_run(p(src), ld, nrows, p(out), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
exp = src[:nrows, :128].contiguous().view(-1)
print("rc", rc, "match", bool((out == exp).all()), (out != exp).sum().item())
if not (out == exp).all():
    print(out[:8], exp[:8], out[128:136], exp[128:136])
