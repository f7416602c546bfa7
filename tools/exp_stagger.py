"""Does the relative placement of the four field arrays in HBM matter?  The fused kernel streams Htau and Ht (same plane,
same row at the same time) and writes the new field and dHdtau (same plane, same row): with every array a multiple of
1 GiB from the next, simultaneous accesses fall on the same position of whatever channel interleave the memory system uses.
Carves the arrays out of one allocation at staggered offsets and times the fused launch.  usage: exp_stagger.py [n]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fpr_amd
F = fpr_amd.load(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
N = n * n * n
dx = 10.0 / n
coef = (dx * dx / 8.1, 5.0, 1 / dx, 1 / dx, 1 / dx, 1 / dx, 1 / dx, 1 / dx)
pool = torch.zeros(5 * N + (1 << 24), dtype=torch.float64, device="cuda")
def carve(offsets):
    out = []
    for i, o in enumerate(offsets):
        t = pool[i * N + o: i * N + o + N].view(n, n, n).permute(2, 1, 0)   # Julia (nx, ny, nz) strides (1, nx, nx*ny)
        out.append(t)
    return out
sq = F.fzeros(2)
for name, offs in (("all arrays a multiple of 8 B x n^3 apart", (0, 0, 0, 0, 0)),
                   ("+4 KiB steps", (0, 512, 1024, 1536, 2048)),
                   ("+64 KiB steps", (0, 8192, 16384, 24576, 32768)),
                   ("+1 MiB steps", (0, 131072, 262144, 393216, 524288)),
                   ("+1 plane + 32 KiB steps", (0, n * n + 4096, 2 * n * n + 8192, 3 * n * n + 12288, 4 * n * n + 16384)),
                   ("all arrays a multiple of 8 B x n^3 apart (again)", (0, 0, 0, 0, 0))):
    Ht, A, O, C, R = carve(offs)
    pool.zero_()
    F.part1.init_local_gaussian((5., 5., 5.), dx, dx, dx, Ht)
    A.copy_(Ht); C.copy_(Ht)
    ts = []
    for rep in range(3):
        for _ in range(5):
            F.part1.diffusion_3D_step_τ2(Ht, A, O, C, R, *coef, 0.2, sq); A, C = C, A
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30):
            F.part1.diffusion_3D_step_τ2(Ht, A, O, C, R, *coef, 0.2, sq); A, C = C, A
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 30)
    print("%-55s fused pair %.1f / %.1f / %.1f us" % (name, *(t * 1e6 for t in ts)))
