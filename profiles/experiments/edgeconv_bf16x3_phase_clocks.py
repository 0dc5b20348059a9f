"""Phase clocks of edgeconv_bf16x3 (timing build: edgeconv_bf16x3_phase_clock_kernel.hip compiled in place of the product kernel into
scratch/lib_ecprobe.so; lane 0 of every wave of workgroups 0 and 517 accumulates s_memtime differences per phase and writes them
over x1 rows 0 / 1 -- results of that build are garbage by design).  Record: profiles/rounds4-5/r5n_edgeconv_bf16x3_phase_clocks.txt."""
import sys, torch
sys.path.insert(0, '.')
import vcrnet_amd
from vcrnet_amd import native
native.LIB_PATH = 'scratch/lib_ecprobe.so'
g = torch.Generator().manual_seed(0)
for B, N, k in ((32, 1024, 20), (32, 1024, 40), (64, 4096, 20), (64, 4096, 40)):
    M = B * N
    pq = torch.randn(M, 256, generator=g).cuda(); w2 = (torch.randn(128, 128, generator=g) / 11).cuda(); b2 = torch.randn(128, generator=g).cuda()
    idx = torch.randint(0, N, (M, k), generator=g, dtype=torch.int32).cuda()
    for _ in range(3):
        x1, x2 = native.edgeconv(pq, idx, N, w2, b2, bf16x3=True)
    torch.cuda.synchronize()
    for row in (0, 1):
        for w in range(4):
            v = x1[row, 32 * w: 32 * w + 7].tolist()
            tiles = v[6] * 5
            print(f"B={B} N={N} k={k} block {'0' if row == 0 else '517'} wave {w}: groups {int(v[6])}, per tile cycles: pre-mfma(gather issue) {v[0]/tiles:7.0f}  mfma loop {v[1]/tiles:7.0f}  fold {v[2]/tiles:7.0f}  commit {v[3]/tiles:7.0f}  barrier {v[4]/tiles:7.0f}   total {v[5]/tiles:7.0f}")
