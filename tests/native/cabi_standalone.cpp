// The C ABI used with nothing but the HIP runtime: no Python, no torch.  Builds a small volume, runs
// dfx_sweep(PROJECT) + dfx_reduce_u/v and one factored XP sweep, and checks them against host loops.
// Built by __graft_entry__.build() into tests/native/cabi_standalone; run by a -m gpu test.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/dfx.h"

#define HIP_OK(e)                                                                      \
    do {                                                                               \
        hipError_t e_ = (e);                                                           \
        if (e_ != hipSuccess) {                                                        \
            std::printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); \
            return 10;                                                                 \
        }                                                                              \
    } while (0)
#define DFX_OK(e)                                                                 \
    do {                                                                          \
        int rc_ = (e);                                                            \
        if (rc_ != 0) {                                                           \
            std::printf("dfx error %d (%s) at line %d\n", rc_, dfx_last_error(), __LINE__); \
            return 11;                                                            \
        }                                                                         \
    } while (0)

template <class T>
static T *dev(size_t n) {
    void *p = nullptr;
    if (hipMalloc(&p, n * sizeof(T)) != hipSuccess) std::abort();
    return static_cast<T *>(p);
}

int main() {
    const int nT = 3, nL = 2, nZ = 5, X = 8, Y = 12, r = 4, Xs = X / r, Ys = Y / r, M = nT * nL;
    const long long S = (long long)nZ * X * Y, N = M * S, Ps = Xs * Ys;
    dfx_geom g = {nT, nL, nZ, X, Y, r, 0};
    dfx_plan pl;
    DFX_OK(dfx_plan_make(&g, 0, &pl));
    if (!pl.fast) return 12;

    std::vector<float> x(N), p(N), ra(S), rb(M * Ps);
    for (long long i = 0; i < N; ++i) x[i] = 0.25f * (float)(i % 7) - 0.5f, p[i] = 0.125f * (float)(i % 5) - 0.25f;
    for (long long i = 0; i < S; ++i) ra[i] = 0.0625f * (float)(i % 9);
    for (long long i = 0; i < M * Ps; ++i) rb[i] = -0.03125f * (float)(i % 11);

    float *dx = dev<float>(N), *dp = dev<float>(N), *dra = dev<float>(S), *drb = dev<float>(M * Ps);
    float *du = dev<float>(S), *dv = dev<float>(M * Ps);
    float *u_part = dev<float>(pl.u_part_elems), *v_part = dev<float>(pl.v_part_elems);
    double *part = dev<double>(pl.part_elems), *scal = dev<double>(DFX_NSCAL);
    HIP_OK(hipMemcpy(dx, x.data(), N * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dp, p.data(), N * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dra, ra.data(), S * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(drb, rb.data(), M * Ps * 4, hipMemcpyHostToDevice));
    std::vector<double> hs(DFX_NSCAL, 0.0);
    const double alpha = 0.375, beta = -0.5;
    hs[DFX_S_ALPHA] = alpha, hs[DFX_S_BETA] = beta;
    HIP_OK(hipMemcpy(scal, hs.data(), DFX_NSCAL * 8, hipMemcpyHostToDevice));

    auto project = [&](const std::vector<float> &a, std::vector<double> &u, std::vector<double> &v) {
        u.assign(S, 0.0), v.assign(M * Ps, 0.0);
        for (int m = 0; m < M; ++m)
            for (int z = 0; z < nZ; ++z)
                for (int xi = 0; xi < X; ++xi)
                    for (int yi = 0; yi < Y; ++yi) {
                        const double val = a[(long long)m * S + ((long long)z * X + xi) * Y + yi];
                        u[((long long)z * X + xi) * Y + yi] += val;
                        v[(long long)m * Ps + (xi / r) * Ys + yi / r] += val;
                    }
    };
    const long long v_layers = pl.v_part_elems / (M * Ps);   // z-chunks (x row parts for r = 16, 32, 64)
    auto fetch_v = [&](std::vector<float> &out) -> int {   // v = sum of the layers of v_part
        const float *src = v_part;
        if (v_layers > 1) {
            DFX_OK(dfx_reduce_v(&g, &pl, v_part, dv, 0, M, nullptr));
            src = dv;
        }
        out.resize(M * Ps);
        HIP_OK(hipMemcpy(out.data(), src, M * Ps * 4, hipMemcpyDeviceToHost));
        return 0;
    };
    auto check = [&](const char *what, const std::vector<float> &got, const std::vector<double> &want) {
        double num = 0, den = 0;
        for (size_t i = 0; i < want.size(); ++i) num += (got[i] - want[i]) * (got[i] - want[i]), den += want[i] * want[i];
        const double e = std::sqrt(num / (den > 0 ? den : 1));
        std::printf("%-28s rel-L2 %.2e\n", what, e);
        return e <= 1e-6;
    };

    // 1. projection of x
    DFX_OK(dfx_sweep(DFX_SWEEP_PROJECT, &g, &pl, dx, nullptr, nullptr, scal, u_part, v_part, part, nullptr));
    DFX_OK(dfx_reduce_u(&g, &pl, u_part, du, scal, part, nullptr));
    std::vector<float> hu(S), hv;
    HIP_OK(hipMemcpy(hu.data(), du, S * 4, hipMemcpyDeviceToHost));
    if (int rc = fetch_v(hv)) return rc;
    std::vector<double> ru, rv;
    project(x, ru, rv);
    bool ok = check("u(x)", hu, ru) && check("v(x)", hv, rv);
    HIP_OK(hipMemcpy(hs.data(), scal, DFX_NSCAL * 8, hipMemcpyDeviceToHost));
    double xx = 0, uu = 0;
    for (float t : x) xx += (double)t * t;
    for (double t : ru) uu += t * t;
    ok = ok && std::fabs(hs[DFX_S_XX] - xx) <= 1e-9 * xx && std::fabs(hs[DFX_S_UU] - uu) <= 1e-6 * uu;

    // 2. factored XP sweep: x += alpha p ; p = (ra + rb) + beta p ; u(p), v(p)
    DFX_OK(dfx_sweep_factored(DFX_SWEEP_XP, &g, &pl, dx, dra, drb, dp, scal, u_part, v_part, part, 0, pl.cpb, 0, 1,
                              nullptr));
    DFX_OK(dfx_reduce_u(&g, &pl, u_part, du, scal, part, nullptr));
    std::vector<float> gx(N), gp(N), wantp(N);
    std::vector<double> wx(N), wp(N);
    HIP_OK(hipMemcpy(gx.data(), dx, N * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(gp.data(), dp, N * 4, hipMemcpyDeviceToHost));
    for (int m = 0; m < M; ++m)
        for (long long j = 0; j < S; ++j) {
            const long long i = (long long)m * S + j;
            const int xi = (int)((j % ((long long)X * Y)) / Y), yi = (int)(j % Y);
            const double res = (double)ra[j] + (double)rb[(long long)m * Ps + (xi / r) * Ys + yi / r];
            wx[i] = x[i] + alpha * p[i];
            wp[i] = res + beta * p[i];
            wantp[i] = (float)wp[i];
        }
    ok = ok && check("x + alpha p", gx, wx) && check("(ra + rb) + beta p", gp, wp);
    project(wantp, ru, rv);
    HIP_OK(hipMemcpy(hu.data(), du, S * 4, hipMemcpyDeviceToHost));
    if (int rc = fetch_v(hv)) return rc;
    ok = ok && check("u(p)", hu, ru) && check("v(p)", hv, rv);

    // 3. one whole CG iteration as three calls: residual update + beta, sweep left unfolded, post-sweep
    //    with the next alpha - checked against the host on the scalars that close the iteration
    {
        const double c1 = 0.5 / M, c2 = 0.5 / (nZ * r * r), eps = 1e-10;
        double *zab = dev<double>(Ps);
        std::vector<double> h0(DFX_NSCAL);
        HIP_OK(hipMemcpy(h0.data(), scal, DFX_NSCAL * 8, hipMemcpyDeviceToHost));
        h0[DFX_S_RSOLD] = 3.25, h0[DFX_S_ALPHA] = 0.125;
        HIP_OK(hipMemcpy(scal, h0.data(), DFX_NSCAL * 8, hipMemcpyHostToDevice));
        // u, v on the device are u(p), v(p) of step 2 (hu, hv hold them); ra, rb as uploaded
        const float *v_dev = v_layers > 1 ? dv : v_part;
        DFX_OK(dfx_resid_cg_step(&g, dra, drb, du, v_dev, c1, c2, eps, 1, zab, scal, part, nullptr));
        std::vector<float> gra(S), grb(M * Ps);
        HIP_OK(hipMemcpy(gra.data(), dra, S * 4, hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(grb.data(), drb, M * Ps * 4, hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(hs.data(), scal, DFX_NSCAL * 8, hipMemcpyDeviceToHost));
        std::vector<double> wra(S), wrb(M * Ps);
        for (long long j = 0; j < S; ++j) wra[j] = (double)ra[j] - 0.125 * c1 * hu[j];
        for (long long j = 0; j < M * Ps; ++j) wrb[j] = (double)rb[j] - 0.125 * c2 * hv[j];
        ok = ok && check("ra - alpha c1 u", gra, wra) && check("rb - alpha c2 v", grb, wrb);
        double rr = 0;      // ||bcast(ra) + bcast(rb)||^2 over the 5-D volume, from the stored fp32 factors
        for (int m = 0; m < M; ++m)
            for (long long j = 0; j < S; ++j) {
                const int xi = (int)((j % ((long long)X * Y)) / Y), yi = (int)(j % Y);
                const double t = (double)gra[j] + (double)grb[(long long)m * Ps + (xi / r) * Ys + yi / r];
                rr += t * t;
            }
        const double beta_w = (double)((float)rr / ((float)3.25 + (float)eps));
        std::printf("%-28s %.9e vs %.9e ; beta %.7e vs %.7e\n", "RR (small-space identity)", hs[DFX_S_RR], rr, hs[DFX_S_BETA], beta_w);
        ok = ok && std::fabs(hs[DFX_S_RR] - rr) <= 1e-9 * rr && std::fabs(hs[DFX_S_BETA] - beta_w) <= 1e-6 * std::fabs(beta_w) &&
             hs[DFX_S_RSOLD] == hs[DFX_S_RR] && hs[DFX_S_ALPHA_LAST] == 0.125 && hs[DFX_S_ITER] == 1.0;
        // the sweep with its scalar rows left to dfx_post_sweep
        HIP_OK(hipMemcpy(gx.data(), dx, N * 4, hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(gp.data(), dp, N * 4, hipMemcpyDeviceToHost));
        DFX_OK(dfx_sweep_factored(DFX_SWEEP_XP, &g, &pl, dx, dra, drb, dp, scal, u_part, v_part, part, 0, pl.cpb, 0,
                                  DFX_LAST_NOFOLD, nullptr));
        DFX_OK(dfx_post_sweep(&g, &pl, u_part, du, v_part, v_layers > 1 ? dv : nullptr, scal, part, 1,
                              DFX_POST_U | DFX_POST_V | DFX_POST_FOLD_SWEEP | DFX_POST_ALPHA, c1, c2, eps, nullptr));
        const double al = 0.125, be = hs[DFX_S_BETA];
        double pp = 0, xxn = 0, xp = 0;
        for (int m = 0; m < M; ++m)
            for (long long j = 0; j < S; ++j) {
                const long long i = (long long)m * S + j;
                const int xi = (int)((j % ((long long)X * Y)) / Y), yi = (int)(j % Y);
                const double res = (double)(gra[j] + grb[(long long)m * Ps + (xi / r) * Ys + yi / r]);
                const double xn = (double)gx[i] + (double)(float)al * gp[i], pn = res + (double)(float)be * gp[i];
                wantp[i] = (float)pn;
                pp += pn * pn, xxn += xn * xn, xp += xn * pn;
            }
        project(wantp, ru, rv);
        HIP_OK(hipMemcpy(hu.data(), du, S * 4, hipMemcpyDeviceToHost));
        hv.resize(M * Ps);
        HIP_OK(hipMemcpy(hv.data(), v_layers > 1 ? dv : v_part, M * Ps * 4, hipMemcpyDeviceToHost));
        ok = ok && check("u(p) via dfx_post_sweep", hu, ru) && check("v(p) via dfx_post_sweep", hv, rv);
        HIP_OK(hipMemcpy(hs.data(), scal, DFX_NSCAL * 8, hipMemcpyDeviceToHost));
        double uu2 = 0, vv2 = 0;
        for (float t : hu) uu2 += (double)t * t;
        for (float t : hv) vv2 += (double)t * t;
        const double alpha_w = (double)((float)hs[DFX_S_RSOLD] / ((float)(c1 * uu2 + c2 * vv2) + (float)eps));
        std::printf("%-28s PP %.6e/%.6e XX %.6e/%.6e XP %.6e/%.6e alpha %.7e/%.7e\n", "post-sweep scalars", hs[DFX_S_PP], pp,
                    hs[DFX_S_XX], xxn, hs[DFX_S_XP], xp, hs[DFX_S_ALPHA], alpha_w);
        ok = ok && std::fabs(hs[DFX_S_PP] - pp) <= 1e-5 * pp && std::fabs(hs[DFX_S_XX] - xxn) <= 1e-5 * xxn &&
             std::fabs(hs[DFX_S_XP] - xp) <= 1e-5 * std::fabs(xp) + 1e-7 * std::sqrt(pp * xxn) &&
             std::fabs(hs[DFX_S_UU] - uu2) <= 1e-9 * uu2 && std::fabs(hs[DFX_S_VV] - vv2) <= 1e-9 * vv2 &&
             std::fabs(hs[DFX_S_ALPHA] - alpha_w) <= 1e-6 * std::fabs(alpha_w);
        // 4. the entry point that takes no workspace gives what the one with a workspace gives
        std::vector<float> a1(S), b1(M * Ps), a2(S), b2(M * Ps);
        DFX_OK(dfx_resid_update(&g, dra, drb, du, v_layers > 1 ? dv : v_part, 0.3f, -0.7f, 1.5f, nullptr, 0, zab, scal,
                                part, nullptr));
        HIP_OK(hipMemcpy(a1.data(), dra, S * 4, hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(b1.data(), drb, M * Ps * 4, hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(hs.data(), scal, DFX_NSCAL * 8, hipMemcpyDeviceToHost));
        const double rr1 = hs[DFX_S_RR];
        DFX_OK(dfx_resid_update_noalloc(&g, dra, drb, du, v_layers > 1 ? dv : v_part, 0.3f, -0.7f, 1.5f, nullptr, 0,
                                        scal, nullptr));
        HIP_OK(hipMemcpy(a2.data(), dra, S * 4, hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(b2.data(), drb, M * Ps * 4, hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(hs.data(), scal, DFX_NSCAL * 8, hipMemcpyDeviceToHost));
        const bool same = a1 == a2 && b1 == b2 && hs[DFX_S_RR] == rr1;
        std::printf("%-28s %s\n", "dfx_resid_update_noalloc", same ? "bit-identical" : "DIFFERS");
        ok = ok && same;
    }
    HIP_OK(hipDeviceSynchronize());
    std::printf(ok ? "cabi_standalone: OK\n" : "cabi_standalone: MISMATCH\n");
    return ok ? 0 : 1;
}
