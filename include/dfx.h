/* dfx.h - C ABI of libdfx.so: hand-written gfx950 (MI355X) kernels for the FusionCG /
 * FusionAdam / baseline() hot path of FluoLab/datafusion.
 *
 * The reference has no FFI for this path (it is eager torch); each entry point
 * below names the reference lines (paths relative to the reference checkout) whose work it
 * replaces.  All tensors are fp32, layout (t, lambda, z, x, y) with y contiguous, exactly as
 * datafusion/fusion.py:162-168.  With z-sharding, nZ in dfx_geom is the LOCAL plane count.
 *
 * Conventions
 *   - every function returns 0 on success, a negative DFX_E* code for a rejected argument,
 *     or a positive hipError_t; dfx_last_error() gives the message (thread local).
 *   - no entry point allocates, frees or synchronises (the dfx_arena_* set-up functions at the end of this
 *     file excepted: they are the allocator); the caller owns every buffer (sizes come from dfx_plan_make)
 *     and passes the hipStream_t to launch on.
 *   - pointers are device pointers; big vectors must be 16-byte aligned.
 *   - reductions are deterministic: per-workgroup double partials in the caller's `part`
 *     workspace, folded in a fixed order by a one-workgroup kernel launched by the same entry
 *     point into a slot of the device-resident state `scal`; no float atomics anywhere.
 *   - `scal` (double[DFX_NSCAL]) lives on the device for the whole solve: alpha, beta and the
 *     norms never visit the host unless the caller copies them.
 */
#ifndef DFX_H
#define DFX_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DFX_VERSION 130

enum { DFX_EINVAL = -1, DFX_EALIGN = -2, DFX_EPLAN = -3 };

/* Extents of the (local shard of the) 5-D volume; X = r*Xs, Y = r*Ys (fusion.py:158-169). */
typedef struct dfx_geom {
    int64_t nT, nL, nZ, X, Y;
    int32_t r;        /* spatial_increase, fusion.py:169 */
    int32_t pad_;
} dfx_geom;

/* Launch plan + workspace sizes for one geometry.  fast=1: float4 wave-tile kernels
 * (Y % 4 == 0 and 2 <= r <= 128: row maps for r in {2, 4, 8}, a row-part tile for {16, 32, 64}, flat maps otherwise;
 * small-space rows B / v_part 8-byte aligned when r = 2); fast=0: shape-generic kernels (any r, any Y).
 * v_part holds LAYERS of nT*nL*Xs*Ys floats that dfx_post_sweep / dfx_reduce_v add up: one per z-chunk, times a
 * factor of the lane map (r / 4 for r in {16, 32, 64}; r for a power-of-two r on planes with Y % 32 != 0; 2 r for
 * other r); layers = v_part_elems / (nT*nL*Xs*Ys).  With ONE layer v_part is v itself. */
typedef struct dfx_plan {
    int32_t fast;
    int32_t zc;          /* z planes a workgroup keeps in registers */
    int32_t n_zchunks;   /* ceil(nZ / zs) */
    int32_t steps;       /* (t,lambda) slabs per u_part chunk (flush interval of the fp32 u sums) */
    int32_t n_chunks;    /* ceil(nT*nL / steps) */
    int32_t cpb;         /* consecutive chunks one workgroup walks through */
    int32_t n_blocks;    /* workgroups per sweep / bcast launch */
    int32_t zs;          /* planes per z-chunk: ceil(nZ / n_zchunks) <= zc (chunks are balanced) */
    int64_t u_part_elems;    /* floats: (n_chunks / cpb) * nZ*X*Y: one row per workgroup of the slab axis */
    int64_t v_part_elems;    /* floats: layers * nT*nL*Xs*Ys, layers = n_zchunks (x r / 4 for r in {16, 32, 64}) */
    int64_t part_elems;      /* doubles: (n_blocks * DFX_MAX_SLICES + DFX_EW_MAXBLOCKS) * DFX_NPART */
} dfx_plan;

/* Slots of the device-resident state `scal` (double[DFX_NSCAL]). */
enum {
    DFX_S_RSOLD = 0, DFX_S_RSNEW = 1, DFX_S_PAP = 2, DFX_S_ALPHA = 3, DFX_S_BETA = 4,
    DFX_S_TOL = 5,
    DFX_S_UU = 6,   /* ||u(p)||^2 over the local z planes (dfx_reduce_u)            */
    DFX_S_PP = 7,   /* ||p||^2 (dfx_sweep)                                            */
    DFX_S_XX = 8,   /* ||x||^2 at the time p was formed (dfx_sweep)                   */
    DFX_S_XP = 9,   /* <x, p> (dfx_sweep)                                             */
    DFX_S_VV = 10,  /* ||v(p)||^2 after the cross-shard sum (dfx_sqnorm)              */
    DFX_S_RR = 11,  /* ||out||^2 of the last dfx_bcast                                */
    DFX_S_XXNEW = 12, /* ||y||^2 of the last dfx_axpy                                 */
    DFX_S_NONFINITE = 13, /* sticky: >0 once rsnew was not finite (fusion.py:448)     */
    DFX_S_ITER = 14,
    DFX_S_SUM = 15, /* plain sum (dfx_sum, dfx_baseline_init)                         */
    DFX_S_SCALE = 16, /* dfx_recip output by convention                               */
    DFX_S_LOSS1 = 17, DFX_S_LOSS2 = 18, /* dfx_sqdiff outputs by convention           */
    DFX_S_ADAM_DD = 19, /* ||x_new - x_old||^2 of the last dfx_adam_step              */
    DFX_S_ADAM_XX = 20, /* ||x_new||^2 of the last dfx_adam_step; the Adam entry points also set
                           TOL = sqrt(ADAM_DD / ADAM_XX) (fusion.py:244-252 as fusion.py:348 uses it) */
    DFX_S_RA2 = 21, DFX_S_RB2 = 22, DFX_S_RAB = 23, /* sum ra^2, sum rb^2, sum rb*zab of the last dfx_resid_update */
    DFX_S_ALPHA_LAST = 24, /* the alpha of the iteration dfx_cg_beta closed (ALPHA is renewed right after the sweep) */
    DFX_NSCAL = 32
};
#define DFX_NPART 4          /* doubles per workgroup row of `part` */
#define DFX_EW_MAXBLOCKS 2048 /* grid cap of the element-wise kernels */
#define DFX_MAX_SLICES 16     /* most launches one sweep may be cut into (dfx_sweep_chunks) */
#define DFX_NOALLOC_MAX_PS 65536 /* largest (X/r)*(Y/r) dfx_resid_update_noalloc accepts (its static scratch) */

/* `last` argument of dfx_sweep_chunks / dfx_sweep_factored */
enum { DFX_LAST_NO = 0, DFX_LAST_FOLD = 1, /* final slice: fold PP, XX, XP with a one-workgroup launch */
       DFX_LAST_NOFOLD = 2 /* final slice: leave the partial rows to dfx_post_sweep(DFX_POST_FOLD_SWEEP) */ };
/* dfx_post_sweep flags */
enum { DFX_POST_U = 1,           /* u = sum of the u_part rows, UU                                            */
       DFX_POST_V = 2,           /* v = sum of the v_part layers (if more than one), VV                       */
       DFX_POST_V_SUMMED = 4,    /* with DFX_POST_V: `v_part` is the finished v (e.g. after the all-reduce)    */
       DFX_POST_FOLD_SWEEP = 8,  /* PP, XX, XP from the rows of a sweep issued with DFX_LAST_NOFOLD            */
       DFX_POST_FOLD_ADAM = 16,  /* ADAM_DD, ADAM_XX from the rows of dfx_adam_step*(DFX_ADAM_NOFOLD)          */
       DFX_POST_ALPHA = 32 };    /* then PAP, ALPHA of the next iteration (dfx_cg_alpha)                        */
/* `non_neg` argument of dfx_adam_step*: bit flags */
enum { DFX_ADAM_NONNEG = 1, DFX_ADAM_NOFOLD = 2 };

/* sweep modes */
enum {
    DFX_SWEEP_PROJECT = 0, /* read x: u(x), v(x), XX = ||x||^2                                  */
    DFX_SWEEP_XP = 1,      /* x += alpha p ; p = r + beta p ; u(p), v(p), PP, XX, XP             */
    DFX_SWEEP_P = 2,       /* p = r + beta p ; u(p), v(p), PP          (x untouched)             */
    DFX_SWEEP_PINIT = 3    /* p = r ; u(p), v(p), PP, XX, XP    (reads x, r; writes p)           */
};

int dfx_version(void);
const char *dfx_last_error(void);

/* (No counterpart in the reference, which is eager ATen: launch geometry and workspace sizes are this library's.)
 * Choose the launch plan and workspace sizes.  target_blocks <= 0 -> default (one workgroup per CU of the
 * current device; 256 when no device is visible). */
int dfx_plan_make(const dfx_geom *g, int32_t target_blocks, dfx_plan *out);
/* Same with overrides for tuning: zc in {0 = auto, 4, 8}, steps (0 = auto). */
int dfx_plan_make_ex(const dfx_geom *g, int32_t target_blocks, int32_t zc, int32_t steps, dfx_plan *out);

/* One pass over the 5-D vectors (replaces, fused: T,S,D,R forward of fusion.py:44-51,89-98 as
 * called from the closure A fusion.py:423-426; the axpy `x + p*alpha` fusion.py:445; the
 * direction update fusion.py:455; the dot products / norms of fusion.py:250-251,444).
 * alpha, beta are read from scal[DFX_S_ALPHA], scal[DFX_S_BETA] on the device.
 * Outputs: u_part [n_chunks/cpb][nZ][X][Y], v_part [layers][nT*nL][Xs][Ys], scal[PP, XX, XP]. */
int dfx_sweep(int32_t mode, const dfx_geom *g, const dfx_plan *pl, float *x, const float *r, float *p,
              double *scal, float *u_part, float *v_part, double *part, void *stream);

/* The same sweep (same reference lines as dfx_sweep: fusion.py:44-51,89-98,423-426,445,455) cut along the slab axis, for overlapping the cross-shard sum of v with the rest of the sweep:
 * launch `slice` (0, 1, ...) handles chunks [c_lo, c_hi) of the plan's cpb, i.e. the contiguous slabs
 * [c_lo*steps*gz, min(c_hi*steps*gz, nT*nL)) with gz = n_chunks / cpb; when it has run, those rows of v_part
 * are final.  Slices must be issued in order with increasing `slice`; `last` != 0 on the final one folds the
 * scalar partials of all slices.  Fast plans only (a generic plan has cpb = 1). */
int dfx_sweep_chunks(int32_t mode, const dfx_geom *g, const dfx_plan *pl, float *x, const float *r, float *p,
                     double *scal, float *u_part, float *v_part, double *part, int32_t c_lo, int32_t c_hi,
                     int32_t slice, int32_t last, void *stream);

/* Factored residual.  b - A(x0) and every A(p) of fusion.py:423-430 are a (z,x,y) array broadcast over (t,l) plus a
 * (t,l,x/r,y/r) array broadcast over z and the r x r block, so the CG residual of fusion.py:435,446 always has the
 * form r[t,l,z,x,y] = ra[z,x,y] + rb[t,l,x/r,y/r].  Kept as those two small arrays it is never stored or read as a
 * 5-D vector: `r - Ap*alpha` (fusion.py:446) becomes the small-space update below and the sweep forms r on the fly.
 *
 * dfx_resid_update:  ra = (accumulate ? ra : 0) + coef*sa*A[z,x,y] ;  rb = (accumulate ? rb : 0) + coef*sb*B[t,l,xb,yb]
 *   coef = coef_host * (coef_dev ? *coef_dev : 1);  zab = workspace of Xs*Ys doubles (sum_z sum_block ra);
 *   scal[DFX_S_RR] = ||r||^2 = nT*nL*sum ra^2 + 2*sum rb*zab + nZ*r^2*sum rb^2   (torch.dot(r, r), fusion.py:437,447)
 *   with nZ = g->nZ, the local planes: partials of z-shards add up when rb is replicated.
 * dfx_sweep_factored: dfx_sweep_chunks (modes XP, P, PINIT) with r given by its factors: x, p read + written
 *   (4N words for XP instead of 5N + the 2N of dfx_bcast). */
int dfx_resid_update(const dfx_geom *g, float *ra, float *rb, const float *A, const float *B, float sa, float sb,
                     float coef_host, const double *coef_dev, int32_t accumulate, double *zab, double *scal,
                     double *part, void *stream);
int dfx_sweep_factored(int32_t mode, const dfx_geom *g, const dfx_plan *pl, float *x, const float *ra,
                       const float *rb, float *p, double *scal, float *u_part, float *v_part, double *part,
                       int32_t c_lo, int32_t c_hi, int32_t slice, int32_t last, void *stream);

/* dfx_resid_update without the `zab` and `part` workspace arguments.  It has NO counterpart in the reference: it exists
 * for callers that cannot allocate device memory (code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc) and therefore cannot bring a workspace.  Same kernels, same results bit for bit; the scratch
 * (zab + the per-workgroup partials, 576 KiB) is a static __device__ array inside the library, one copy per GPU,
 * held until the process ends.  Limits: (X/r)*(Y/r) <= DFX_NOALLOC_MAX_PS, else DFX_EINVAL; the device of `stream`
 * must be the current device; every call on one GPU shares that GPU's scratch, so calls that overlap on one GPU
 * (two streams) race - serialise them.  Calls no hipMalloc / hipFree, and does not synchronise.  The FIRST call on
 * a device resolves the scratch's address (hipGetSymbolAddress: not a stream operation): make one call outside any
 * capture before capturing it into a HIP graph. */
int dfx_resid_update_noalloc(const dfx_geom *g, float *ra, float *rb, const float *A, const float *B, float sa,
                             float sb, float coef_host, const double *coef_dev, int32_t accumulate, double *scal,
                             void *stream);

/* One CG iteration's small-space half in three launches (fusion.py:446-448,455-456 and the tolerance of
 * fusion.py:244-252): ra -= ALPHA*c1*u ; rb -= ALPHA*c2*v (ALPHA read from scal) ; then, in the one-workgroup launch
 * that folds RR = ||r||^2 by the identity above, dfx_cg_beta's bookkeeping (RSNEW, NONFINITE, TOL, BETA, RSOLD, ITER).
 * One-GPU path: a z-sharded solve all-reduces RR between dfx_resid_update and dfx_cg_beta. */
int dfx_resid_cg_step(const dfx_geom *g, float *ra, float *rb, const float *u, const float *v, double c1, double c2,
                      double eps, int32_t deferred, double *zab, double *scal, double *part, void *stream);

/* Everything that follows a sweep, in TWO launches (flags above) instead of seven: one many-workgroup pass for the
 * last stage of the T, S sums (u, UU: dfx_reduce_u), of the D sum (v: dfx_reduce_v) and ||v||^2 (dfx_sqnorm) of
 * fusion.py:44-51,89-98,444; one one-workgroup launch that folds those partials, the sweep's (or fused Adam step's)
 * per-workgroup scalar partials, and computes alpha (fusion.py:444, dfx_cg_alpha) - same arithmetic, fixed order.
 * `part` is the buffer the preceding sweep wrote its rows into; n_slices = the number of launches that sweep was cut
 * into.  v may be NULL with one layer in v_part (v_part IS v). */
int dfx_post_sweep(const dfx_geom *g, const dfx_plan *pl, const float *u_part, float *u, const float *v_part,
                   float *v, double *scal, double *part, int32_t n_slices, int32_t flags, double c1, double c2,
                   double eps, void *stream);

/* out = (in ? in : 0) + coef * mask * (sa * A[z,x,y] + sb * B[t,l,x/r,y/r]),
 * coef = coef_host * (coef_dev ? *coef_dev : 1); mask (nZ,X,Y) optional (NULL = ones).
 * Broadcast-on-read adjoints: replaces T.T, S.T, D.T (fusion.py:36-42), R.T (fusion.py:78-87),
 * the weighted sum of fusion.py:423-430 and `r - Ap*alpha` (fusion.py:446); with in=NULL it
 * writes b - A(x0) (fusion.py:428-435) or the Adam gradient (autograd of fusion.py:226-232,
 * masked as fusion.py:274-278).  scal[DFX_S_RR] = sum(out^2)  (fusion.py:437,447). */
int dfx_bcast(const dfx_geom *g, const dfx_plan *pl, float *out, const float *in, const float *A,
              const float *B, const float *mask, float sa, float sb, float coef_host,
              const double *coef_dev, double *scal, double *part, void *stream);

/* One fused Adam iteration on x (replaces optimizer.step() of fusion.py:324-346 with
 * torch.optim.Adam(lr, betas, eps, amsgrad=False, weight_decay=0), the materialised gradient of
 * loss.backward() fusion.py:338, the gradient mask fusion.py:274-278, the clamp fusion.py:344-346, the
 * clone + norms of fusion.py:329,348 and the next iteration's T,S,D,R forward fusion.py:226-232):
 *   g = mask * (A[z,x,y] + B[t,l,x/r,y/r]);  m1, m2, x updated in place (6N words);
 *   u_part, v_part <- projections of the NEW x;  scal[ADAM_DD] = ||dx||^2, scal[ADAM_XX] = ||x||^2.
 * `step` is the 1-based Adam step count (bias corrections are computed on the host in double).
 * `non_neg`: DFX_ADAM_NONNEG (the clamp) | DFX_ADAM_NOFOLD (leave ADAM_DD, ADAM_XX to dfx_post_sweep). */
int dfx_adam_step(const dfx_geom *g, const dfx_plan *pl, float *x, float *m1, float *m2, const float *A,
                  const float *B, const float *mask, double lr, double beta1, double beta2, double eps, int32_t step,
                  int32_t non_neg, double *scal, float *u_part, float *v_part, double *part, void *stream);

/* The same step (optimizer.step() of fusion.py:340 with the Adam of fusion.py:325) with the FIRST moment kept in
 * factored form.  exp_avg is linear in the gradients and every gradient
 * is mask * (A[z,x,y] + B[t,l,x/r,y/r]), so m1 = mask * (m1a[z,x,y] + m1b[t,l,x/r,y/r]) exactly and the 5-D m1 array
 * never exists: the caller advances the factors with dfx_lerp(m1a, A, 1-beta1), dfx_lerp(m1b, B, 1-beta1) (torch's
 * exp_avg.lerp_) BEFORE this call; x and m2 are read and written (4N words instead of 6N).  No counterpart for m2:
 * g^2 has cross terms A*B that do not factor. */
int dfx_adam_step_factored(const dfx_geom *g, const dfx_plan *pl, float *x, const float *m1a, const float *m1b,
                           float *m2, const float *A, const float *B, const float *mask, double lr, double beta1,
                           double beta2, double eps, int32_t step, int32_t non_neg, double *scal, float *u_part,
                           float *v_part, double *part, void *stream);
/* The same two steps (optimizer.step() of fusion.py:340-346) cut along the slab axis like dfx_sweep_chunks, so that a
 * z-sharded FusionAdam can all-reduce each finished band of v_part (the RD partials of the NEXT iteration's loss,
 * fusion.py:226-232) while the rest of the volume is still being stepped: launch `slice` handles chunks [c_lo, c_hi)
 * of the plan's cpb; slices in order; pass either m1 (explicit first moment) or m1a and m1b (factored), the other(s)
 * NULL.  The slice with c_hi == cpb folds ADAM_DD, ADAM_XX over the rows of all slices unless DFX_ADAM_NOFOLD leaves
 * them to dfx_post_sweep(DFX_POST_FOLD_ADAM, n_slices).  Fast plans only. */
int dfx_adam_step_chunks(const dfx_geom *g, const dfx_plan *pl, float *x, float *m1, const float *m1a, const float *m1b,
                         float *m2, const float *A, const float *B, const float *mask, double lr, double beta1,
                         double beta2, double eps, int32_t step, int32_t non_neg, double *scal, float *u_part,
                         float *v_part, double *part, int32_t c_lo, int32_t c_hi, int32_t slice, void *stream);
/* torch.optim.Adam's step (amsgrad=False, weight_decay=0) on an EXPLICIT gradient: the optimizer.step() of
 * fusion.py:342 when loss.backward() (fusion.py:338) has materialised x.grad - the autograd formulation, e.g. with
 * extra differentiable terms added to the loss.  x, grad, m1 (exp_avg), m2 (exp_avg_sq): n floats each, 16-byte
 * aligned; one launch, 7n words.  flags: DFX_ADAM_NONNEG fuses the clamp of fusion.py:344-346.
 * scal[ADAM_DD] = ||x_new - x_old||^2, scal[ADAM_XX] = ||x_new||^2 (the tolerance of fusion.py:244-252 without the
 * clone of fusion.py:329).  n_blocks <= 0: one workgroup per CU.  part: DFX_EW_MAXBLOCKS * DFX_NPART doubles. */
int dfx_adam_apply(float *x, const float *grad, float *m1, float *m2, int64_t n, double lr, double beta1,
                   double beta2, double eps, int32_t step, int32_t flags, int32_t n_blocks, double *scal, double *part,
                   void *stream);
/* The same optimizer.step() (fusion.py:342, on the explicit x.grad of fusion.py:338) for a parameter that IS the 5-D
 * volume of geometry g, fused with the projections the NEXT iteration's loss() starts with (T, S, D, R of fusion.py:
 * 226-232 on the stepped x): the step runs in the sweep's lane-mapped structure (x, grad, m1, m2 read, x, m1, m2
 * written: 7N words) and leaves u_part, v_part and the ADAM_DD / ADAM_XX partials exactly as dfx_adam_step(...,
 * DFX_ADAM_NOFOLD) does - finish with dfx_post_sweep(DFX_POST_U | DFX_POST_V | DFX_POST_FOLD_ADAM).  Saves the N-word
 * projection pass of the following forward.  Arithmetic: dfx_adam_apply's (IEEE square root and division; eps = 0
 * allowed).  flags: DFX_ADAM_NONNEG.  Fast (lane-mapped) plans only, else DFX_EPLAN. */
int dfx_adam_apply_project(const dfx_geom *g, const dfx_plan *pl, float *x, const float *grad, float *m1, float *m2,
                           double lr, double beta1, double beta2, double eps, int32_t step, int32_t flags, double *scal,
                           float *u_part, float *v_part, double *part, void *stream);

/* The small-space half of one fused Adam iteration in ONE launch (fusion.py:332-338: loss(), and what
 * loss.backward() does on the cmos- and spc-sized arrays):
 *   A[i] = -2 w1 cmask[i] (cmos[i] - u[i]), i < n_u ;  B[i] = -2 w2 smask[i % smask_n] (spc[i] - v[i]), i < n_v
 *   (masks optional, fusion.py:216-223) ; m1a, m1b (both or neither) advanced by lerp(., A / B, lerp_weight) ;
 *   scal[LOSS1] = w1 sum cmask (cmos - u)^2, scal[LOSS2] = w2 sum smask (spc - v)^2  (fusion.py:226-232).
 * Same arithmetic as dfx_diff_scale + dfx_lerp + dfx_sqdiff, which it replaces (8 launches -> 2). */
int dfx_adam_prep(const float *u, const float *cmos, const float *cmask, float *A, float *m1a, int64_t n_u,
                  const float *v, const float *spc, const float *smask, int64_t smask_n, float *B, float *m1b,
                  int64_t n_v, double w1, double w2, double lerp_weight, double *scal, double *part, void *stream);
/* m[i] += (float)weight * (g[i] - m[i]), i < n  (torch.lerp_ as torch.optim.Adam applies it to exp_avg inside
 * optimizer.step(), fusion.py:340). */
int dfx_lerp(float *m, const float *g, int64_t n, double weight, void *stream);

/* u[i] = sum over the n_chunks/cpb rows of u_part[row][i] (i < nZ*X*Y); scal[DFX_S_UU] = sum u^2: the last stage of
 * the T and S sums of fusion.py:44-51 and the ST part of torch.dot(p, Ap), fusion.py:444. */
int dfx_reduce_u(const dfx_geom *g, const dfx_plan *pl, const float *u_part, float *u, double *scal,
                 double *part, void *stream);
/* v[m] = sum_zc v_part[zc][m] for the slabs m_lo <= m < m_hi (rows of Xs*Ys floats): the last stage of the D sum of
 * fusion.py:44-51 feeding R, fusion.py:89-98. */
int dfx_reduce_v(const dfx_geom *g, const dfx_plan *pl, const float *v_part, float *v, int64_t m_lo, int64_t m_hi,
                 void *stream);
/* scal[slot] = sum_i a[i]^2  (||v(p)||^2: the RD part of torch.dot(p, Ap), fusion.py:444). */
int dfx_sqnorm(const float *a, int64_t n, double *scal, int32_t slot, double *part, void *stream);
/* scal[slot] = w * sum_i mask[i % mask_n] * (a[i] - b[i])^2, mask optional (loss terms,
 * fusion.py:216-232; squared_l2 fusion.py:12-18). */
int dfx_sqdiff(const float *a, const float *b, const float *mask, int64_t mask_n, int64_t n, double w,
               double *scal, int32_t slot, double *part, void *stream);

/* alpha (fusion.py:444): PAP = c1*UU + c2*VV ; ALPHA = RSOLD / (PAP + eps). */
int dfx_cg_alpha(double *scal, double c1, double c2, double eps, void *stream);
/* After r was updated (RR valid): RSNEW = RR; NONFINITE += !isfinite(RR) (fusion.py:448);
 * TOL = |alpha| ||p|| / ||x_new|| (fusion.py:244-252; ||x_new||^2 = XX + 2 alpha XP + alpha^2 PP if
 * deferred, XXNEW otherwise); BETA = RSNEW/(RSOLD+eps) (fusion.py:455); RSOLD = RSNEW (fusion.py:456);
 * ITER += 1. */
int dfx_cg_beta(double *scal, double eps, int32_t deferred, void *stream);
/* First residual: RSOLD = RR (fusion.py:437). */
int dfx_cg_start(double *scal, void *stream);

/* y[i] += coef_host*(*coef_dev) * x[i]; scal[DFX_S_XXNEW] = sum(y^2)  (fusion.py:445 stand-alone). */
int dfx_axpy(float *y, const float *x, int64_t n, float coef_host, const double *coef_dev, double *scal,
             double *part, void *stream);
/* y[i] = (mask && mask[i % mask_n] == 0) ? 0 : y[i], then *= coef_host*(coef_dev ? *coef_dev : 1)
 * (fusion.py:197-199,204-209).  mask has period mask_n (= nZ*X*Y) over the flat index. */
int dfx_scale(float *y, int64_t n, const float *mask, int64_t mask_n, float coef_host,
              const double *coef_dev, void *stream);
/* scal[DFX_S_SUM] = sum_i y[i]  (fusion.py:209 tensor.sum()). */
int dfx_sum(const float *y, int64_t n, double *scal, double *part, void *stream);
/* out[i] = coef * (a[i] - b[i]) * (mask ? mask[i % mask_n] : 1), i < n: the small-space differences cmos_n - u(x) and
 * spc_n - v(x) of b - A(x0) (fusion.py:428-435) and of the loss gradient (autograd of fusion.py:216-232, with the
 * masks of fusion.py:218-223). */
int dfx_diff_scale(float *out, const float *a, const float *b, const float *mask, int64_t mask_n, int64_t n,
                   float coef, void *stream);

/* out[a][b] = sum_{k<R} x[(a*R + k)*B + b]  (x viewed as (A, R, B)): the marginals callers take right after a
 * solve - x.sum(axis=0): A=1, R=nT; x.sum(axis=1): A=nT, R=nL (analysis/cells.ipynb cell 27bf9830e81df430) -
 * computed on the device so the 5-D volume need not cross PCIe (fusion.py:489-493 returns it whole). */
int dfx_sum_axis(const float *x, float *out, int64_t A, int64_t R, int64_t B, void *stream);
/* Stand-alone forms of the reference's operator objects (the solvers use the fused kernels above instead):
 * dfx_sum_axis is SumOperator.__call__ (fusion.py:44-51);
 * dfx_expand_axis is SumOperator.T (fusion.py:36-42): out[(a*R + k)*B + b] = scale * y[a*B + b], scale = 1/size;
 * dfx_block_sum is SumPoolOperator.__call__ (fusion.py:89-98): r x r block sums of `rows` planes of X x Y;
 * dfx_block_expand is SumPoolOperator.T (fusion.py:78-87): out[row][x][y] = scale * in[row][x/r][y/r], scale = 1/r^2. */
int dfx_expand_axis(const float *y, float *out, int64_t A, int64_t R, int64_t B, float scale, void *stream);
int dfx_block_sum(const float *in, float *out, int64_t rows, int64_t X, int64_t Y, int32_t r, void *stream);
int dfx_block_expand(const float *in, float *out, int64_t rows, int64_t X, int64_t Y, int32_t r, float scale,
                     void *stream);
/* dst[0] = num / src[0]  (device scalars; total_energy / sum, fusion.py:209). */
int dfx_recip(double *dst, const double *src, double num, void *stream);

/* baseline.py:38-50 in one pass: x[t,l,z,x,y] = coef * up(spc)[t,l,x,y] * zw[z,x,y], up = bilinear
 * (align_corners = False) upsampling by r, zw = cmos / sum_z cmos (times the cmos mask when
 * mask_noise, fusion.py:197-198).  coef = coef_host * (coef_dev ? *coef_dev : 1).  If x == NULL
 * nothing is written (sum only).  scal[DFX_S_SUM] = sum(x). */
int dfx_baseline_init(const dfx_geom *g, const float *spc, const float *zw, float *x, float coef_host,
                      const double *coef_dev, double *scal, double *part, void *stream);

/* Input stage, utils.py:411-427 (load_data): `skimage.transform.resize(cmos_xyz, (S1, S2, nZ))` with its defaults
 * (order 1, mode "reflect" - numpy.pad's name, which skimage hands to scipy.ndimage as "mirror": whole-sample
 * symmetric, the edge sample not repeated -, anti-aliasing Gaussian sigma = (factor - 1) / 2 truncated at 4 sigma on the
 * axes that shrink), restated per plane of a float64 (planes, X, Y) stack on the device: out (planes, S1, S2), or
 * (planes, S2, S1) when transpose_out (load_data's final np.transpose(., (2, 1, 0)) swaps x and y).
 * tmp_a, tmp_b: workspaces of planes*X*Y doubles each.  The z axis is not resized (factor 1).
 * scikit-image is not available to check against: PARITY UNPINNED; the tests compare with the scipy.ndimage calls
 * skimage's resize is built from (gaussian_filter + zoom(order=1, grid_mode=True), both with mode="mirror"). */
int dfx_resize_stack(const double *in, double *tmp_a, double *tmp_b, double *out, int64_t planes, int64_t X,
                     int64_t Y, int64_t S1, int64_t S2, int32_t transpose_out, void *stream);

/* ---- Placement of the 5-D vectors in device memory (set-up time; the ONLY entry points that allocate and
 * synchronise).  No counterpart in the reference, whose vectors are whatever torch's allocator returns
 * (fusion.py:196, 436, 445): on MI355X the 288 GiB fall into three classes of 96 GiB, and two vectors swept in
 * lock-step stream at 6.4-6.5 TB/s when they lie in different classes at equal offsets, 5.4-5.5 TB/s when they share one
 * (DESIGN.md "Placement of large vectors").  An arena owns physical chunks (hipMemCreate) of the CURRENT device; the
 * caller measures each chunk's class (datafusion_amd/placement.py times the real sweep on pairs of chunks) and maps
 * chunks, in an order of its choice, behind contiguous virtual ranges (hipMemAddressReserve + hipMemMap) that the
 * other entry points then use like any device pointer.  A chunk is mapped in at most one view at a time. */
typedef struct dfx_arena dfx_arena;
/* n_chunks physical chunks of chunk_bytes each (rounded up to the device's recommended granularity). */
int dfx_arena_create(int64_t chunk_bytes, int64_t n_chunks, dfx_arena **out);
/* n_more further chunks (indices continue after the existing ones); all or nothing. */
int dfx_arena_grow(dfx_arena *a, int64_t n_more);
int dfx_arena_info(const dfx_arena *a, int64_t *chunk_bytes, int64_t *n_chunks, int32_t *device);
/* A new view: chunks[0..n) mapped in that order at one contiguous, read-write virtual range -> *out_ptr. */
int dfx_arena_map(dfx_arena *a, const int32_t *chunks, int64_t n, void **out_ptr);
/* Remove the view whose base address is ptr (synchronises the device first); its chunks can be mapped again. */
int dfx_arena_unmap(dfx_arena *a, void *ptr);
/* Give unmapped chunks back to the driver (e.g. those the chosen layout does not use). */
int dfx_arena_release_chunks(dfx_arena *a, const int32_t *chunks, int64_t n);
/* Unmap every view, release every chunk, free the arena. */
int dfx_arena_destroy(dfx_arena *a);
/* Message of the last failed dfx_arena_* call on this thread. */
const char *dfx_arena_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
