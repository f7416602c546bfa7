// placement.hip -- which of a host's candidate allocations the streaming kernels should get.
//
// Separately allocated arrays carry a label that comes with their physical pages: two arrays streamed at equal offsets get in each
// other's way when the labels agree (a copy between two 1 GiB arrays runs at ~4950 / ~5150 / ~5400 GB/s by label distance; the fused
// diffusion launch takes 0.76 ms on four arrays that all differ and 0.85-0.91 ms on arrays of one class).  The label cannot be computed
// from a pointer, so it is measured: fpr_placement_rank times a copy between every pair of the caller's candidates, ranks the
// assignments of candidates to array positions by their slowest streamed-together pair, and -- given a trial callback -- lets the
// caller's own kernel decide among the best few and a short local search.  The candidates stay the caller's (role of the
// reference's prealloc_dict, multigrid.jl:25-38,49-51: the host owns its arrays); nothing is allocated or freed here.
// One implementation for every host language: the Python mirror (placement.py) and julia/FPRHip.jl allocate, call, free.
#include <algorithm>
#include <vector>

#include "fpr_internal.hpp"

__global__ __launch_bounds__(256) void k_place_copy(double* __restrict__ dst, const double* __restrict__ src, size_t n)
{
    // the access pattern of fpr_copy (16-byte lanes where both arrays are aligned): what the pair times were calibrated on
    const size_t n2 = n / 2;
    const size_t stride = (size_t)gridDim.x * 256;
    if ((((uintptr_t)dst | (uintptr_t)src) & 15) == 0) {
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += stride)
            reinterpret_cast<double2*>(dst)[i] = reinterpret_cast<const double2*>(src)[i];
        if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) dst[n - 1] = src[n - 1];
    } else {
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = src[i];
    }
}

namespace {
struct Ranked {
    double worst, sum;
    std::vector<int> sub;   // candidate per role
};
inline bool better(double w1, double s1, double w2, double s2) { return w1 < w2 || (w1 == w2 && s1 < s2); }

// the `keep` best assignments (distinct candidate SETS) of k candidates to nroles positions: depth-first with the partial cost
// (pairs whose two roles are both assigned) as the bound
struct Search {
    int k, nroles, keep;
    const std::vector<double>* sym;
    std::vector<std::pair<int, int>> rp;   // pairs in role indices, sorted by their later role
    std::vector<Ranked> best;
    std::vector<int> cur;
    std::vector<char> used;
    long nodes = 0;
    double t(int a, int b) const { return (*sym)[(size_t)a * k + b]; }
    void offer(double w, double s)
    {
        std::vector<int> key = cur;
        std::sort(key.begin(), key.end());
        for (auto& b : best) {
            std::vector<int> kb = b.sub;
            std::sort(kb.begin(), kb.end());
            if (kb == key) {                       // same set: keep the better labelling
                if (better(w, s, b.worst, b.sum)) { b.worst = w; b.sum = s; b.sub = cur; }
                std::sort(best.begin(), best.end(), [](const Ranked& x, const Ranked& y) { return better(x.worst, x.sum, y.worst, y.sum); });
                return;
            }
        }
        best.push_back({w, s, cur});
        std::sort(best.begin(), best.end(), [](const Ranked& x, const Ranked& y) { return better(x.worst, x.sum, y.worst, y.sum); });
        if ((int)best.size() > keep) best.pop_back();
    }
    void go(int depth, double w, double s)
    {
        if (++nodes > 20000000L) return;           // (a pool of 40 candidates for 6 roles would be 2.8e9 leaves: the bound prunes, this caps)
        if ((int)best.size() == keep && w > best.back().worst) return;    // (the partial worst pair only grows)
        if (depth == nroles) { offer(w, s); return; }
        for (int c = 0; c < k; ++c) {
            if (used[c]) continue;
            double w2 = w, s2 = s;
            cur[depth] = c;
            for (auto& p : rp)
                if (p.second == depth) { const double v = t(cur[p.first], c); w2 = v > w2 ? v : w2; s2 += v; }
            used[c] = 1;
            go(depth + 1, w2, s2);
            used[c] = 0;
        }
    }
};
}  // namespace

extern "C" int fpr_placement_rank(fpr_ctx* ctx, double* const* cand, int k, size_t n, int count, const int* pairs, int npairs,
                                  fpr_place_trial_fn trial, void* user, int* chosen_out, double* report)
{
    if (!ctx) return FPR_ERR_INVALID;
    FPR_REQUIRE(ctx, cand && chosen_out && report, "null pointer");
    FPR_REQUIRE(ctx, count >= 1 && k >= count && k <= 4096, "need at least `count` candidates");
    FPR_REQUIRE(ctx, npairs >= 0 && (npairs == 0 || pairs), "pairs");
    for (int i = 0; i < FPR_PLACE_REPORT_LEN; ++i) report[i] = 0.0;
    for (int i = 0; i < k; ++i) FPR_REQUIRE(ctx, cand[i], "null candidate");
    // positions streamed together (default: every pair of positions)
    std::vector<std::pair<int, int>> pp;
    if (npairs == 0)
        for (int i = 0; i < count; ++i)
            for (int j = i + 1; j < count; ++j) pp.push_back({i, j});
    for (int q = 0; q < npairs; ++q) {
        const int i = pairs[2 * q], j = pairs[2 * q + 1];
        FPR_REQUIRE(ctx, i >= 0 && j >= 0 && i < count && j < count && i != j, "pair index out of range");
        pp.push_back({i, j});
    }
    std::vector<int> roles;
    for (auto& p : pp) { roles.push_back(p.first); roles.push_back(p.second); }
    std::sort(roles.begin(), roles.end());
    roles.erase(std::unique(roles.begin(), roles.end()), roles.end());
    const int nroles = (int)roles.size();
    auto role_of = [&](int pos) { return (int)(std::lower_bound(roles.begin(), roles.end(), pos) - roles.begin()); };

    // ---- copy time of every ordered pair (events on the compute stream) ----
    hipStream_t s = ctx->stream[0];
    struct Events {          // (destroyed on every return path)
        hipEvent_t a = nullptr, b = nullptr;
        ~Events() { if (a) hipEventDestroy(a); if (b) hipEventDestroy(b); }
    } ev;
    FPR_HIP(ctx, hipEventCreate(&ev.a));
    FPR_HIP(ctx, hipEventCreate(&ev.b));
    hipEvent_t e0 = ev.a, e1 = ev.b;
    const int reps = 2;
    const size_t nb = (n + 1) / 2;
    // 2048 workgroups walking through the arrays together (fpr_copy's grid): the pattern the class thresholds were measured with, and
    // the one that resembles a march -- a grid of one 16-byte access per thread copies at 6.2-6.4 TB/s whatever the classes
    const unsigned grid = (unsigned)std::max<size_t>(1, std::min<size_t>((nb + 255) / 256, 2048));
    std::vector<double> t((size_t)k * k, 0.0), sym((size_t)k * k, 0.0);
    const double bytes2 = 2.0 * 8.0 * (double)n;
    if (nroles >= 2) {
        for (int i = 0; i < k; ++i)
            for (int j = 0; j < k; ++j) {
                if (i == j) continue;
                k_place_copy<<<grid, 256, 0, s>>>(cand[j], cand[i], n);      // warm-up (clocks, first touch)
                FPR_HIP(ctx, hipEventRecord(e0, s));
                for (int r = 0; r < reps; ++r) k_place_copy<<<grid, 256, 0, s>>>(cand[j], cand[i], n);
                FPR_HIP(ctx, hipEventRecord(e1, s));
                FPR_HIP(ctx, hipEventSynchronize(e1));
                float ms = 0.f;
                FPR_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
                t[(size_t)i * k + j] = (double)ms / reps;
            }
        FPR_CHECK_LAUNCH(ctx);
        for (int i = 0; i < k; ++i)
            for (int j = 0; j < k; ++j) sym[(size_t)i * k + j] = 0.5 * (t[(size_t)i * k + j] + t[(size_t)j * k + i]);
    }
    auto gbs = [&](double ms) { return ms > 0 ? bytes2 / (ms * 1e-3) / 1e9 : 0.0; };
    std::vector<double> flat;
    for (int i = 0; i < k; ++i)
        for (int j = i + 1; j < k; ++j) flat.push_back(sym[(size_t)i * k + j]);
    std::sort(flat.begin(), flat.end());
    if (!flat.empty()) {
        report[FPR_PLACE_POOL_FASTEST_GBS] = gbs(flat.front());
        report[FPR_PLACE_POOL_MEDIAN_GBS] = gbs(flat[flat.size() / 2]);
        report[FPR_PLACE_POOL_SLOWEST_GBS] = gbs(flat.back());
    }

    // ---- rank the assignments by their slowest streamed-together pair ----
    const int ntrials = 4;          // assignments handed to the caller's kernel besides the candidates as given
    Search S;
    S.k = k; S.nroles = nroles; S.keep = trial ? ntrials : 1; S.sym = &sym;
    for (auto& p : pp) {
        int a = role_of(p.first), b = role_of(p.second);
        if (a > b) std::swap(a, b);
        S.rp.push_back({a, b});
    }
    S.cur.assign(nroles, 0);
    S.used.assign(k, 0);
    if (nroles >= 2) S.go(0, 0.0, 0.0);
    if (S.best.empty()) {            // nothing streamed together: the candidates as given
        Ranked r{0.0, 0.0, {}};
        for (int q = 0; q < nroles; ++q) r.sub.push_back(q);
        S.best.push_back(r);
    }
    // full assignment (every position) from a choice for the roles: the other positions take unused candidates in order
    auto build = [&](const std::vector<int>& sub, int* out) {
        std::vector<char> used(k, 0);
        for (int c : sub) used[c] = 1;
        for (int pos = 0; pos < count; ++pos) out[pos] = -1;
        for (int q = 0; q < nroles; ++q) out[roles[q]] = sub[q];
        int nxt = 0;
        for (int pos = 0; pos < count; ++pos)
            if (out[pos] < 0) {
                while (used[nxt]) ++nxt;
                out[pos] = nxt;
                used[nxt] = 1;
            }
    };
    std::vector<int> best_sub = S.best[0].sub;
    double best_ms = 0.0, first_ms = 0.0, worst_ms = 0.0, least_ms = 0.0, ident_ms = 0.0;
    int tried = 0;
    std::vector<int> full(count);
    auto run_trial = [&](const std::vector<int>& sub) -> double {
        build(sub, full.data());
        const double ms = trial(user, full.data(), count);
        if (ms > 0) {
            if (tried == 0) { first_ms = worst_ms = least_ms = ms; }
            worst_ms = ms > worst_ms ? ms : worst_ms;
            least_ms = ms < least_ms ? ms : least_ms;
            ++tried;
        }
        return ms;
    };
    if (trial) {
        // the copy times only rank the candidates roughly; the caller's kernel decides.  First the candidates AS GIVEN (positions
        // 0 .. count-1 = what a host that simply allocates would use: the result is never worse than that), then the best few
        // assignments, then a local search: one position at a time swapped for a candidate not in use, kept when faster.
        const double gain = 5e-3;       // a swap of the local search is kept when it is 0.5 % faster
        {
            std::vector<int> ident;
            for (int q = 0; q < nroles; ++q) ident.push_back(roles[q]);
            ident_ms = run_trial(ident);
            if (ident_ms > 0) { best_sub = ident; best_ms = ident_ms; }
        }
        for (auto& r : S.best) {
            const double ms = run_trial(r.sub);
            if (ms > 0 && (best_ms <= 0 || ms < best_ms)) { best_sub = r.sub; best_ms = ms; }
        }
        if (best_ms > 0) {
            long budget = 6L * k;
            bool improved = true;
            while (improved && budget > 0) {
                improved = false;
                for (int q = 0; q < nroles && budget > 0; ++q)
                    for (int c = 0; c < k && budget > 0; ++c) {
                        if (std::find(best_sub.begin(), best_sub.end(), c) != best_sub.end()) continue;
                        std::vector<int> sub = best_sub;
                        sub[q] = c;
                        --budget;
                        const double ms = run_trial(sub);
                        if (ms > 0 && ms < (1.0 - gain) * best_ms) { best_sub = sub; best_ms = ms; improved = true; }
                    }
            }
        }
    }
    build(best_sub, chosen_out);
    double cw = 0.0, cs = 0.0;
    for (auto& p : pp) {
        const double v = sym[(size_t)chosen_out[p.first] * k + chosen_out[p.second]];
        cw = v > cw ? v : cw;
        cs += v;
    }
    report[FPR_PLACE_CHOSEN_SLOWEST_GBS] = gbs(cw);
    report[FPR_PLACE_CHOSEN_MEAN_GBS] = pp.empty() ? 0.0 : gbs(cs / (double)pp.size());
    report[FPR_PLACE_TRIALS] = tried;
    report[FPR_PLACE_TRIAL_BEST_MS] = best_ms;
    report[FPR_PLACE_TRIAL_FIRST_MS] = first_ms;
    report[FPR_PLACE_TRIAL_WORST_MS] = worst_ms;
    report[FPR_PLACE_TRIAL_IDENTITY_MS] = ident_ms;
    const double spread = (tried >= 2 && least_ms > 0) ? worst_ms / least_ms - 1.0 : 0.0;
    report[FPR_PLACE_TRIAL_SPREAD] = spread;
    // "more candidates would help": every pair of the pool copies below the rate at which pools with a second class start
    // (measured 5076-5160 GB/s where one exists, < 4950 where not: 1 GiB arrays), or the caller's kernel sees less than 2.5 % between any
    // two assignments (arrays that fit in the Infinity Cache copy at cache speed whatever their pages: only the trial can rank them)
    const double below = 5050.0;
    const double min_spread = 25e-3;
    const bool uniform_copy = !flat.empty() && report[FPR_PLACE_POOL_FASTEST_GBS] < below;
    const bool uniform_trial = trial && tried >= 4 && spread < min_spread;
    report[FPR_PLACE_WANT_MORE] = uniform_copy ? 1.0 : (uniform_trial ? 2.0 : 0.0);
    report[FPR_PLACE_SEARCH_NODES] = S.nodes > 20000000L ? -(double)S.nodes : (double)S.nodes;   // negative: the search hit its cap (the best found so far was kept)
    return FPR_OK;
}
