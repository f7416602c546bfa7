"""Per-workgroup start / end stamps of the ticketed core launch (EXPERIMENTS 10.1-12).  Needs the temporary hook Diff3Args2::stamps
(option diff3_dbg_stamps) that existed while the split forms were compared; kept for the record of how the placement was read."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, fpr_amd, numpy as np
F = fpr_amd.load(0)
n = 512
dx = 10.0 / n
coef = (dx * dx / 8.1, 5.0, 1 / dx, 1 / dx, 1 / dx, 1 / dx, 1 / dx, 1 / dx)
Ht = F.fzeros(n, n, n); F.part1.init_local_gaussian((5., 5., 5.), dx, dx, dx, Ht)
A, O, C, R, sq = Ht.clone(), F.fzeros(n, n, n), Ht.clone(), F.fzeros(n, n, n), F.fzeros(2)
k = int(sys.argv[1]) if len(sys.argv) > 1 else 16
gz = F.grid.GlobalGrid(n, n, n, dims=(1, 1, 1), periods=(0, 0, 1), transport="rccl", use_dist=False)
F.ctx().set_option("diff3_comm_units", k)
st = torch.zeros(4 * 256, dtype=torch.int64, device=Ht.device)
for rep in range(4):
    gz.step2(Ht, A, O, C, R, *coef, 0.2, sq, join=False); A, C = C, A
gz.join(); torch.cuda.synchronize()
F.ctx().set_option("diff3_dbg_stamps", st.data_ptr())
for rep in range(3):
    st.zero_(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    gz.step2(Ht, A, O, C, R, *coef, 0.2, sq, join=False); A, C = C, A
    gz.join(); torch.cuda.synchronize()
    s = st.cpu().numpy().reshape(256, 4)
    t_0 = s[:, 0][s[:, 0] > 0].min()
    start = (s[:, 0] - t_0) / 100.0; end = (s[:, 1] - t_0) / 100.0
    work = s[:, 2] >= 0
    print("k=%d pair %.0f us: working WGs %d; start of working WGs: max %.1f us; end max %.1f; surplus starts: %s" % (
        k, (time.perf_counter() - t0) * 1e6, work.sum(), start[work].max(), end[work].max(), np.sort(start[~work])[:20].round(0)))
    late = np.where(work & (start > 50))[0]
    print("   late working WGs (blockIdx, xcc, unit, start):", [(int(b), int(s[b, 3]) >> 8, int(s[b, 2]), float(start[b])) for b in late[:12]])
    xcc = (s[:, 3] >> 8) & 7
    print("   WGs per xcc:", np.bincount(xcc.astype(int), minlength=8), " working per xcc:", np.bincount(xcc[work].astype(int), minlength=8), " blockIdx%8==xcc for", int(((np.arange(256) % 8) == xcc).sum()))
    dur = end[work] - start[work]
    print("   comm units found:", F.ctx().get_option("comm_units_found") if hasattr(F.ctx(), "get_option") else "?", " distinct keys:", len(set(s[:, 3].tolist())))
    print("   duration of working WGs: min %.0f median %.0f max %.0f" % (dur.min(), np.median(dur), dur.max()))
