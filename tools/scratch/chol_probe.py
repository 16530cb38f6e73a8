import torch, time
for F in (2048, 4096, 8192):
    A = torch.randn(F, 64, dtype=torch.float64, device="cuda")
    iC = A @ A.T / 64 + torch.eye(F, dtype=torch.float64, device="cuda")
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        L = torch.linalg.cholesky(iC)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        C = torch.cholesky_inverse(L)
        torch.cuda.synchronize(); t2 = time.perf_counter()
    print(F, "potrf %.1f ms, potri %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3), float((C @ iC - torch.eye(F, dtype=torch.float64, device="cuda")).abs().max()))
