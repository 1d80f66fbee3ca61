"""Reference points for the fp32 GEMM kernels: what the vendor library (rocBLAS / hipBLASLt through torch.matmul, fp32) reaches on
the shapes of the discriminator layers and of the blend-shape product at 4096 poses.  Prints TFLOP/s and the fraction of 157.3."""
import torch, time
torch.backends.cuda.matmul.allow_tf32 = False
def run(M, N, K, ta=False, n=50):
    A = torch.randn(K, M, device='cuda').t() if ta else torch.randn(M, K, device='cuda')
    B = torch.randn(K, N, device='cuda')
    for _ in range(5): C = A @ B
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): C = A @ B
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    tf = 2.0 * M * N * K / ms / 1e9
    print(f'M={M} N={N} K={K} transA={ta}: {ms*1e3:.1f} us  {tf:.1f} TFLOP/s  {tf/157.3:.2f} of peak')
for ta in (False, True):
    run(1024, 4096, 1024, ta)
    run(1024, 4096, 768, ta)
    run(768, 4096, 1024, ta)
    run(20736, 4096, 224, ta)
    run(224, 4096, 20736, ta)
