"""How fast is the library GEMM (torch.mm -> hipBLASLt / rocBLAS, fp32) on the weight-gradient shapes of a B = 64 training step?
dW[o][i] = sum_e dY[e][o] X[e][i]: a transposed tall-skinny GEMM with K = rows (edges)."""
import torch, time
dev = torch.device("cuda:0")
torch.backends.cuda.matmul.allow_tf32 = False
shapes = [("gcl W3", 300288, 688, 208), ("gcl W2", 300288, 208, 208), ("gcl W1c", 300288, 208, 688),
          ("equi dp0", 97152, 592, 688), ("equi dp2+rbf", 97152, 624, 688)]
for name, E, no, ni in shapes:
    dY = torch.randn(E, no, device=dev); X = torch.randn(E, ni, device=dev)
    for _ in range(2): (dY.t() @ X)
    torch.cuda.synchronize(); t = time.perf_counter()
    n = 5
    for _ in range(n): out = dY.t() @ X
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
    print(f"{name}: E={E} {no}x{ni}: {dt * 1e3:.3f} ms  {2 * E * no * ni / dt / 1e12:.1f} TFLOP/s")
