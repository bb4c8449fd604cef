// Host-side check of libdfx's argument validation and planner under AddressSanitizer + UBSan (no GPU needed: every
// call here must return before it launches anything).  Built and run by tests/test_cpu_oracle_and_host.py.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/dfx.h"

// layers of v_part per z-chunk on a fast plan (include/dfx.h): row parts for r = 16, 32, 64 and 1 for r = 2, 4, 8 on
// planes with 128-byte aligned rows (Y % 32 == 0); else the strip tile: the most 4-row strips a block row meets
static int layers_per_zchunk(int r, int Xs, long long Y) {
    if ((r == 2 || r == 4 || r == 8) && Y % 32 == 0) return 1;
    if ((r == 16 || r == 32 || r == 64) && Y % 32 == 0) return r / 4;
    if (r == 2 || r == 4 || r == 8 || r == 16 || r == 32 || r == 64) return r;   // flat map
    (void)Xs;
    return (r >= 3 && r <= 128) ? 2 * r : 1;   // flat map for any r
}

int main() {
    int bad = 0;
    unsigned long long seed = 12345;
    auto rnd = [&](int lo, int hi) {
        seed = seed * 6364136223846793005ULL + 1442695040888963407ULL;
        return lo + (int)((seed >> 33) % (unsigned)(hi - lo + 1));
    };
    for (int it = 0; it < 20000; ++it) {
        const int r = (int[]){1, 2, 3, 4, 4, 4, 8, 5, 16, 32}[rnd(0, 9)];
        dfx_geom g = {rnd(1, 600), rnd(1, 40), rnd(1, 40), (int64_t)r * rnd(1, 160), (int64_t)r * rnd(1, 160), r, 0};
        dfx_plan pl;
        const int tb = (int[]){0, 1, 64, 256, 304, 2048}[rnd(0, 5)];
        const int zc = (int[]){0, 0, 4, 8}[rnd(0, 3)];
        const int steps = (int[]){0, 0, 1, 4, 64, 1000}[rnd(0, 5)];
        if (dfx_plan_make_ex(&g, tb, zc, steps, &pl) != 0) { std::printf("plan failed: %s\n", dfx_last_error()); ++bad; continue; }
        const long long M = g.nT * g.nL, S = g.nZ * g.X * g.Y, Ps = (g.X / r) * (g.Y / r);
        if (pl.fast) {
            if ((long long)pl.zs * pl.n_zchunks < g.nZ || (long long)pl.steps * pl.n_chunks < M || pl.zs > pl.zc ||
                pl.n_chunks % pl.cpb || pl.u_part_elems != (long long)(pl.n_chunks / pl.cpb) * S ||
                pl.v_part_elems != (long long)pl.n_zchunks * layers_per_zchunk(r, (int)(g.X / r), g.Y) * M * Ps)
                { std::printf("inconsistent fast plan\n"); ++bad; }
        } else if (pl.n_zchunks != 1 || pl.cpb != 1 || pl.n_chunks < 1 || pl.n_chunks > M || (long long)pl.steps * pl.n_chunks < M ||
                   pl.n_blocks % pl.n_chunks || pl.u_part_elems != (long long)pl.n_chunks * S) { std::printf("inconsistent generic plan\n"); ++bad; }
        if (pl.part_elems < (long long)(pl.n_blocks * DFX_MAX_SLICES + DFX_EW_MAXBLOCKS) * DFX_NPART) { std::printf("part too small\n"); ++bad; }
        // every entry point must reject NULL buffers / bad ranges before touching the device
        float *nf = nullptr; double *nd = nullptr;
        if (dfx_sweep(DFX_SWEEP_XP, &g, &pl, nf, nf, nf, nd, nf, nf, nd, nullptr) == 0) ++bad;
        if (dfx_sweep_chunks(7, &g, &pl, nf, nf, nf, nd, nf, nf, nd, 0, 1, 0, 1, nullptr) == 0) ++bad;
        if (dfx_sweep_factored(DFX_SWEEP_PROJECT, &g, &pl, nf, nf, nf, nf, nd, nf, nf, nd, 0, 1, 0, 1, nullptr) == 0) ++bad;
        if (dfx_post_sweep(&g, &pl, nf, nf, nf, nf, nd, nd, 1, DFX_POST_U, 0, 0, 0, nullptr) == 0) ++bad;
        if (dfx_post_sweep(&g, &pl, nf, nf, nf, nf, (double *)&g, (double *)&g, 99, 0, 0, 0, 0, nullptr) == 0) ++bad;
        if (dfx_resid_cg_step(&g, nf, nf, nf, nf, 1, 1, 1e-10, 1, nd, nd, nd, nullptr) == 0) ++bad;
        if (dfx_resid_update(&g, nf, nf, nf, nf, 1, 1, 1, nd, 0, nd, nd, nd, nullptr) == 0) ++bad;
        if (dfx_resid_update_noalloc(&g, nf, nf, nf, nf, 1, 1, 1, nd, 0, nd, nullptr) == 0) ++bad;
        if (dfx_bcast(&g, &pl, nf, nf, nf, nf, nf, 1, 1, 1, nd, nd, nd, nullptr) == 0) ++bad;
        if (dfx_adam_step(&g, &pl, nf, nf, nf, nf, nf, nf, 1e-3, 0.9, 0.999, 1e-8, 1, 0, nd, nf, nf, nd, nullptr) == 0) ++bad;
        if (dfx_adam_step_factored(&g, &pl, nf, nf, nf, nf, nf, nf, nf, 1e-3, 0.9, 0.999, 1e-8, 1, 0, nd, nf, nf, nd, nullptr) == 0) ++bad;
        if (dfx_adam_apply_project(&g, &pl, nf, nf, nf, nf, 1e-3, 0.9, 0.999, 1e-8, 1, 0, nd, nf, nf, nd, nullptr) == 0) ++bad;
        if (dfx_reduce_u(&g, &pl, nf, nf, nd, nd, nullptr) == 0) ++bad;
        if (dfx_reduce_v(&g, &pl, nf, nf, 0, 1, nullptr) == 0) ++bad;
        if (dfx_baseline_init(&g, nf, nf, nf, 1, nd, nd, nd, nullptr) == 0) ++bad;
    }
    dfx_geom g0 = {0, 1, 1, 4, 4, 4, 0}, g1 = {2, 2, 2, 6, 8, 4, 0}, g2 = {1, 1, 1, 1 << 20, 1 << 20, 4, 0};
    dfx_plan pl;
    if (dfx_plan_make(&g0, 0, &pl) == 0 || dfx_plan_make(&g1, 0, &pl) == 0 || dfx_plan_make(&g2, 0, &pl) == 0 ||
        dfx_plan_make(nullptr, 0, &pl) == 0 || dfx_plan_make(&g1, 0, nullptr) == 0) ++bad;
    float *nf = nullptr; double *nd = nullptr;
    if (dfx_adam_apply(nf, nf, nf, nf, 16, 1e-3, 0.9, 0.999, 1e-8, 1, 0, 0, nd, nd, nullptr) == 0) ++bad;
    if (dfx_adam_prep(nf, nf, nf, nf, nf, 4, nf, nf, nf, 0, nf, nf, 4, 1, 1, 0.1, nd, nd, nullptr) == 0) ++bad;
    if (dfx_resize_stack(nd, nd, nd, nd, 1, 8, 8, 4, 4, 0, nullptr) == 0) ++bad;
    if (dfx_sqnorm(nf, 4, nd, 0, nd, nullptr) == 0 || dfx_sqdiff(nf, nf, nf, 0, 4, 1, nd, 0, nd, nullptr) == 0 ||
        dfx_sum(nf, 4, nd, nd, nullptr) == 0 || dfx_axpy(nf, nf, 4, 1, nd, nd, nd, nullptr) == 0 ||
        dfx_scale(nf, 4, nf, 0, 1, nd, nullptr) == 0 || dfx_lerp(nf, nf, 4, 0.1, nullptr) == 0 ||
        dfx_diff_scale(nf, nf, nf, nf, 0, 4, 1, nullptr) == 0 || dfx_sum_axis(nf, nf, 1, 1, 1, nullptr) == 0 ||
        dfx_expand_axis(nf, nf, 1, 1, 1, 1, nullptr) == 0 || dfx_block_sum(nf, nf, 1, 4, 4, 4, nullptr) == 0 ||
        dfx_block_expand(nf, nf, 1, 4, 4, 4, 1, nullptr) == 0 || dfx_recip(nd, nd, 1, nullptr) == 0 ||
        dfx_cg_start(nd, nullptr) == 0 || dfx_cg_alpha(nd, 1, 1, 1, nullptr) == 0 || dfx_cg_beta(nd, 1, 0, nullptr) == 0) ++bad;
    std::printf(bad ? "host_sanitize: %d PROBLEMS\n" : "host_sanitize: OK\n", bad);
    return bad ? 1 : 0;
}
