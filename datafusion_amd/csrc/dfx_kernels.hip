// libdfx: gfx950 (MI355X, CDNA4) kernels for the FusionCG / FusionAdam / baseline() hot path.
// C ABI in include/dfx.h.  Everything here is HBM-bandwidth-bound fp32 streaming + reductions:
// no MFMA.  The 5-D vectors x, p (and r when it is kept explicit; layout (t, l, z, x, y), y contiguous) are swept
// by workgroups of four 64-lane waves; each wave owns a "wave tile" of consecutive r x r downsampling blocks
// (LaneMap<R>; r = 4: 4 rows x 64 columns = sixteen 4x4 blocks), each lane a float4, for `zc` z-planes held in
// registers, walking interleaved (t,l) slabs:
//   * u = sum_{t,l} (fusion.py:44-51, T and S) accumulates in zc float4 registers per lane and is flushed every
//     `steps` slabs into the workgroup's row of u_part;
//   * v = sum_z sum_{r x r} (fusion.py:44-51 D, fusion.py:89-98 R) is a lane-local sum over z and the float4, then
//     cross-lane adds over the lanes holding the other rows of the block (r = 4: l^16, l^32);
//   * the CG residual is never a 5-D array by default: r = ra[z,x,y] + rb[t,l,x/r,y/r] (k_resid_*).
// No LDS tile and no barrier in the streaming loop; LDS is only used for the per-workgroup scalar partials.
// The same kernels run over four more lane maps (lane_map_of): a row-part tile for r = 16, 32, 64 (LaneMap<16>), and -
// for planes whose rows are not 128-byte aligned, and for every other r up to 128 - lanes flat over the contiguous plane
// (LaneMap<1>, LaneMap<3>), whose partial block sums go to layers of v_part.  Y % 4 != 0 takes the generic kernels
// (k_gen_sweep4 / k_gen_sweep2 + k_gen_proj_v2).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>

#include "../../include/dfx.h"

typedef float f4 __attribute__((ext_vector_type(4)));

// Build-time tuning knobs (tools/tune_kernels.py builds variants; the defaults are the measured best).
#ifndef DFX_NT
#define DFX_NT 1          // 1: non-temporal loads/stores for the once-streamed 5-D vectors
#endif
#ifndef DFX_MINWAVES
#define DFX_MINWAVES 1    // __launch_bounds__ 2nd argument (waves per SIMD) of the fast kernels
#endif
#ifndef DFX_WAVES
#define DFX_WAVES 4       // waves per workgroup of the fast kernels (each wave owns one wave tile)
#endif
#ifndef DFX_APPLY_UNROLL
#define DFX_APPLY_UNROLL 1   // float4 groups per array a lane of k_adam_apply loads before it stores (C2: 1: 5.35 ms, 2: 5.67, 4: 5.62)
#endif
#ifndef DFX_ADAM_ZB
#define DFX_ADAM_ZB 2     // same for k_adam_fast
#endif
#ifndef DFX_BCAST_INFLIGHT
#define DFX_BCAST_INFLIGHT 8  // float4 loads a lane of k_bcast_fast keeps in flight (slabs per trip = this / zc)
#endif
#ifndef DFX_ADAM_FASTMATH
#define DFX_ADAM_FASTMATH 1   // 1: v_sqrt_f32 / v_rcp_f32 (1 ulp each) instead of the IEEE sqrt and division sequences
#endif
#ifndef DFX_ZB
#define DFX_ZB 2          // z planes whose loads are issued together, then updated and stored (finer read/write
                          // interleaving measured faster: 3.48 / 3.55 / 3.72 ms at 2 / 4 / 8 on the C2 sweep)
#endif

namespace {

thread_local char g_err[256] = "";

int fail(int code, const char *msg) {
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}
int hipfail(hipError_t e, const char *where) {
    snprintf(g_err, sizeof(g_err), "%s: %s", where, hipGetErrorString(e));
    return (int)e;
}
#define DFX_LAUNCH_CHECK(where)                         \
    do {                                                \
        hipError_t e_ = hipGetLastError();              \
        if (e_ != hipSuccess) return hipfail(e_, where); \
    } while (0)

struct Geo {
    long long S;   // elements per (t,l) slab = nZ*X*Y
    long long P;   // elements per plane = X*Y
    long long Ps;  // Xs*Ys
    long long N;   // M*S
    int M;         // nT*nL
    int nZ, X, Y, Xs, Ys, r;
    int WT;        // wave tiles per plane (fast path): ceil(nblk / BW), BW = 64, 16, 4 blocks per wave for r = 2, 4, 8;
                   // row-part tile: 4-row strips x column groups; flat maps: ceil(X Y / 256)
    int gTPR;      // row-part tile: column groups per strip = ceil(Y / 64)
    int vq;        // layers of v_part per z-chunk, all dense (every layer holds every block; the reducers add them like
                   // z-chunks): 1 for the r = 2, 4, 8 row maps and the generic kernels; r / 4 for the row-part tile (the
                   // waves that share a block write a layer each); r for the flat map (a layer per row of a block); 2 r
                   // for the flat map of other r (that, times two for runs a wave boundary cuts).
    int lm;        // lane map of the fast kernels: 2, 4, 8 (row maps), 16 (row-part tile), 1 (flat map, power-of-two r), 3 (flat map, other r); -1: none
    int nblk;      // r x r blocks per plane = Xs*Ys
    int zs;        // planes per z-chunk of the plan in use (<= the kernel's ZC); set by the entry points
};

// Row-part tile (LaneMap<16>): r = 16, 32, 64 with the wave shape of r = 4 (4 rows x 64 columns, 256 B row segments);
// a block's r rows are spread over r / 4 waves, which leave their partial block sums in layers of v_part.
bool rowpart_tile(int r) {
    static const bool off = getenv("DFX_NO_ROWPART_TILE") != nullptr;   // measurement knob: 16 -> general tile, 32 / 64 -> generic
    return !off && (r == 16 || r == 32 || r == 64);
}

int make_geo(const dfx_geom *g, Geo *o) {
    if (!g) return fail(DFX_EINVAL, "geom is NULL");
    if (g->nT <= 0 || g->nL <= 0 || g->nZ <= 0 || g->X <= 0 || g->Y <= 0 || g->r <= 0)
        return fail(DFX_EINVAL, "geom extents must be positive");
    if (g->X % g->r || g->Y % g->r) return fail(DFX_EINVAL, "X and Y must be multiples of r");
    long long M = g->nT * g->nL;
    if (M > 0x7fffffffLL || g->nZ > 0x7fffffffLL || g->X > 0x7fffffffLL || g->Y > 0x7fffffffLL)
        return fail(DFX_EINVAL, "extent too large");
    o->M = (int)M;
    o->nZ = (int)g->nZ;
    o->X = (int)g->X;
    o->Y = (int)g->Y;
    o->r = g->r;
    o->Xs = o->X / o->r;
    o->Ys = o->Y / o->r;
    o->P = (long long)o->X * o->Y;
    o->Ps = (long long)o->Xs * o->Ys;
    o->S = o->P * o->nZ;
    o->N = o->S * M;
    if (o->S > 0x7fffffffLL) return fail(DFX_EINVAL, "one (t,l) slab must stay below 2^31 elements");
    o->nblk = o->Xs * o->Ys;
    o->gTPR = 0;
    o->vq = 1;
    // the r = 2, 4, 8 and row-part maps stream 256 B row segments with non-temporal accesses, which need plane rows that
    // start on 128-byte lines (Y % 32 == 0): r = 4 on 120 x 120 planes ran at 0.34 of peak (every partial line fetched
    // twice).  On other widths a power-of-two r takes the flat map (lanes over the contiguous plane, one KiB per
    // wave-instruction), and so does every other r (3 .. 128).
    static const bool nt_any = getenv("DFX_NT_ANY_WIDTH") != nullptr, no_flat = getenv("DFX_NO_FLAT_MAP") != nullptr;
    static const bool no_general = getenv("DFX_NO_GENERAL_TILE") != nullptr;   // measurement knob: other r -> generic kernels
    static const bool flat_any = getenv("DFX_FLAT_ANY_WIDTH") != nullptr;   // measurement knob: flat map on aligned rows too
    const bool aligned_rows = ((o->Y % 32) == 0 && !flat_any) || nt_any;
    const bool pow2 = o->r == 2 || o->r == 4 || o->r == 8 || o->r == 16 || o->r == 32 || o->r == 64;
    o->lm = -1;
    o->WT = 0;
    if ((o->Y % 4) != 0) {
        // generic kernels
    } else if ((o->r == 2 || o->r == 4 || o->r == 8) && aligned_rows) {
        const int bw = o->r == 2 ? 64 : (o->r == 8 ? 4 : 16);   // LaneMap<R>::BW
        o->WT = (o->nblk + bw - 1) / bw;
        o->lm = o->r;
    } else if (rowpart_tile(o->r) && aligned_rows) {
        o->vq = o->r / 4;
        o->gTPR = (o->Y + 63) / 64;
        const long long wt = (long long)o->Xs * o->gTPR * o->vq;
        if (wt > 0x7fffffffLL) return fail(DFX_EINVAL, "plane too large");
        o->WT = (int)wt;
        o->lm = 16;
    } else if (pow2 && !no_flat && o->P / 4 <= 0x3fffffffLL) {
        o->vq = o->r;                      // one layer of v_part per row of a block
        o->WT = (int)((o->P / 4 + 63) / 64);
        o->lm = 1;
    } else if (o->r >= 3 && o->r <= 128 && !no_general && o->P / 4 <= 0x3fffffffLL) {
        o->vq = 2 * o->r;                  // a layer per row of a block, times two: the part a wave boundary cuts off
        o->WT = (int)((o->P / 4 + 63) / 64);
        o->lm = 3;
    }
    return 0;
}

// Y a multiple of 4 (every plane row 16-byte aligned) and 2 <= r <= 128: one of the lane maps make_geo chose
bool fast_ok(const Geo &q) { return q.lm >= 0; }
// which lane map a fast plan uses: the r = 2, 4, 8 row maps, the row-part tile (16), the flat maps (1, 3)
int lane_map_of(const Geo &q) { return q.lm; }

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// Per-workgroup partials (a, b, c) -> row[0..2] (row[3] = 0).  Fixed order: deterministic.
__device__ __forceinline__ void block_store3(double a, double b, double c, double *row) {
    __shared__ double sh[16][3];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    a = wave_sum(a);
    b = wave_sum(b);
    c = wave_sum(c);
    if (lane == 0) {
        sh[wave][0] = a;
        sh[wave][1] = b;
        sh[wave][2] = c;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double s0 = 0, s1 = 0, s2 = 0;
        for (int w = 0; w < nw; ++w) {
            s0 += sh[w][0];
            s1 += sh[w][1];
            s2 += sh[w][2];
        }
        row[0] = s0;
        row[1] = s1;
        row[2] = s2;
        row[3] = 0.0;
    }
}

// out[k] = sum over rows [0, n_rows) of column k (k < NC <= 3) of a DFX_NPART-wide table, by one workgroup of 256
// threads in ONE pass: strided per-thread sums, wave shuffles, the four wave results added in wave order.  Fixed
// order: deterministic.  Every thread returns the sums.
template <int NC>
__device__ __forceinline__ void fold_cols(const double *parts, int n_rows, double (&out)[NC]) {
    __shared__ double sh[4][NC];
    double acc[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) acc[k] = 0.0;
    for (int b = threadIdx.x; b < n_rows; b += 256) {
#pragma unroll
        for (int k = 0; k < NC; ++k) acc[k] += parts[(long long)b * DFX_NPART + k];
    }
#pragma unroll
    for (int k = 0; k < NC; ++k) acc[k] = wave_sum(acc[k]);
    __syncthreads();                                   // sh may still be read from a previous call
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < NC; ++k) sh[threadIdx.x >> 6][k] = acc[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NC; ++k) out[k] = ((sh[0][k] + sh[1][k]) + sh[2][k]) + sh[3][k];
}

__device__ __forceinline__ f4 ld4(const float *p) { return *reinterpret_cast<const f4 *>(p); }
__device__ __forceinline__ void st4(float *p, f4 v) { *reinterpret_cast<f4 *>(p) = v; }
__device__ __forceinline__ f4 ld4s(const float *p) {  // streamed once
#if DFX_NT
    return __builtin_nontemporal_load(reinterpret_cast<const f4 *>(p));
#else
    return ld4(p);
#endif
}
__device__ __forceinline__ void st4s(float *p, f4 v) {
#if DFX_NT
    __builtin_nontemporal_store(v, reinterpret_cast<f4 *>(p));
#else
    st4(p, v);
#endif
}
// Every lane map streams the 5-D vectors non-temporally (ld4s / st4s).  That needs wave accesses that start and end on
// 128-byte lines - the row maps on planes with Y % 32 == 0, the flat maps anywhere: a tile with row segments of 48 - 336 B
// (this round's first general-r tile) ran 1.5 - 2.7x slower non-temporally than through the L2, and r = 4's row map on
// 120-wide planes at 0.34 of peak.
__device__ __forceinline__ float hsum(f4 v) { return (v.x + v.y) + (v.z + v.w); }
__device__ __forceinline__ float dot4(f4 a, f4 b) { return (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w); }

// ------------------------------------------------------------------------------------------
// Lane mapping of the fast kernels, r in {2, 4, 8}.  A lane owns one float4 of a plane; a wave covers R rows of
// 64/R float4 each, i.e. BW consecutive r x r blocks (row-major over (x/r, y/r)):
//   r = 2: 2 rows x 32 float4, a float4 spans TWO blocks        -> 64 blocks, partner row in lane ^ 32
//   r = 4: 4 rows x 16 float4, a float4 is one block row        -> 16 blocks, rows in lanes l, l^16, l^32, l^48
//   r = 8: 8 rows x  8 float4, two float4 make one block row    ->  4 blocks, halves in l^1, rows in l^8, ^16, ^32
// All lanes of one block are active together, so the shuffles below never read an idle lane.
// ------------------------------------------------------------------------------------------
template <int R>
struct LaneMap {
    static constexpr int UPR = 64 / R;                               // float4 units per wave row
    static constexpr int BW = (R == 2) ? 64 : (R == 4 ? 16 : 4);     // blocks per wave
    int blk;          // first (r = 2) or only block this lane contributes to
    bool act;
    long long off2;   // element offset of the lane's float4 inside a plane
    __device__ __forceinline__ LaneMap(const Geo &q, int wt, int lane) {
        const int dx = lane / UPR, j = lane % UPR;
        blk = wt * BW + (R == 2 ? 2 * j : (R == 4 ? j : (j >> 1)));
        act = blk < q.nblk;
        const int xb = blk / q.Ys, yb = blk - xb * q.Ys;
        off2 = (long long)(xb * R + dx) * q.Y + yb * R + (R == 8 ? (j & 1) * 4 : 0);
    }
    // s4 = the lane's float4 summed over z; writes the r x r block sums to vrow[block]
    __device__ __forceinline__ void vstore(f4 s4, float *vrow, int lane) const {
        if (R == 2) {
            float s0 = s4.x + s4.y, s1 = s4.z + s4.w;
            s0 += __shfl_xor(s0, 32);
            s1 += __shfl_xor(s1, 32);
            if (lane < 32) *reinterpret_cast<float2 *>(vrow + blk) = make_float2(s0, s1);
        } else if (R == 4) {
            float s = hsum(s4);
            s += __shfl_xor(s, 16);
            s += __shfl_xor(s, 32);
            if (lane < 16) vrow[blk] = s;
        } else {
            float s = hsum(s4);
            s += __shfl_xor(s, 1);
            s += __shfl_xor(s, 8);
            s += __shfl_xor(s, 16);
            s += __shfl_xor(s, 32);
            if (lane < 8 && !(lane & 1)) vrow[blk] = s;
        }
    }
    // the small-space value(s) under the lane's float4, spread over its four elements
    __device__ __forceinline__ f4 bload(const float *brow) const {
        if (R == 2) {
            const float2 b = *reinterpret_cast<const float2 *>(brow + blk);
            return (f4){b.x, b.x, b.y, b.y};
        }
        const float b = brow[blk];
        return (f4){b, b, b, b};
    }
};

// Row-part tile (R = 16: r = 16, 32, 64).  The wave has the shape of r = 4 - 4 rows x 16 float4, 256 B per row segment -
// and covers a 4-row strip of 64 / r blocks; wave tiles are numbered like r = 4's (column groups, then strips).  A wave adds its 4 rows x r columns per block with shuffles and
// stores the partial in ITS layer of v_part (layer = z-chunk * r/4 + row part); the layers are added where the
// z-chunks' are (dfx_post_sweep / dfx_reduce_v).  The general tile's 16 x 16 wave (64 B row segments, cached accesses)
// ran the C2 volume with r = 16 at 0.68 of peak.
template <>
struct LaneMap<16> {
    int blk;
    bool act, lead;
    long long off2;
    int part;         // which 4 rows of the block's r
    int r;
    __device__ __forceinline__ LaneMap(const Geo &q, int wt, int lane) {
        r = q.r;
        // wave tiles in the order of r = 4: column groups fastest, then the 4-row strips of the plane - a workgroup's four
        // waves touch one contiguous run of rows (2.69 ms at C2 size with r = 16 against 2.80 with the strips of one block first)
        const int strip = wt / q.gTPR, cg = wt - strip * q.gTPR;
        const int xb = strip / q.vq;
        part = strip - xb * q.vq;
        const int dx = lane >> 4, j = lane & 15;
        const int col = cg * 64 + 4 * j;
        act = wt < q.WT && col < q.Y;
        off2 = (long long)(xb * r + part * 4 + dx) * q.Y + col;
        blk = xb * q.Ys + col / r;
        lead = act && dx == 0 && (col % r) == 0;
    }
    __device__ __forceinline__ void vstore(f4 s4, float *vrow, int lane) const {
        float s = hsum(s4);
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        if (r >= 32) s += __shfl_xor(s, 4);
        if (r >= 64) s += __shfl_xor(s, 8);
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        if (lead) vrow[blk] = s;
    }
    __device__ __forceinline__ f4 bload(const float *brow) const {
        const float b = brow[blk];
        return (f4){b, b, b, b};
    }
};
// Flat map (R = 1: r = 2, 4, 8, 16, 32, 64 on planes whose rows are not 128-byte aligned).  Lanes run over the plane as
// it lies in memory - float4 f = 64 wt + lane - so every wave-instruction moves one contiguous KiB whatever Y is, and the
// accesses can be non-temporal.  The rows of a block are then in different waves; nothing is exchanged: a float4 (r = 4),
// r / 4 neighbouring lanes (r >= 8: r / 4 divides Y / 4, so the group never straddles a row or a wave) or each half of a
// float4 (r = 2) is one ROW of a block, and row rho of every block goes to layer rho of v_part (Geo::vq = r dense layers,
// added by the reducers like the z-chunks').
template <>
struct LaneMap<1> {
    int blk, layer, r;
    bool act, lead;
    long long off2;
    __device__ __forceinline__ LaneMap(const Geo &q, int wt, int lane) {
        r = q.r;
        const int Y4 = q.Y >> 2;
        const long long f = (long long)wt * 64 + lane;
        act = wt < q.WT && f < (q.P >> 2);
        const int fi = act ? (int)f : 0;
        const int row = fi / Y4, j = fi - row * Y4;
        off2 = 4LL * fi;
        const int xb = row / r;
        layer = row - xb * r;
        blk = xb * q.Ys + (4 * j) / r;
        lead = act && (r <= 4 || (4 * j) % r == 0);
    }
    __device__ __forceinline__ void vstore(f4 s4, float *vrow, int lane) const {
        if (r == 2) {
            *reinterpret_cast<float2 *>(vrow + blk) = make_float2(s4.x + s4.y, s4.z + s4.w);
            return;
        }
        float s = hsum(s4);
        if (r >= 8) s += __shfl_xor(s, 1);
        if (r >= 16) s += __shfl_xor(s, 2);
        if (r >= 32) s += __shfl_xor(s, 4);
        if (r >= 64) s += __shfl_xor(s, 8);
        if (lead) vrow[blk] = s;
    }
    __device__ __forceinline__ f4 bload(const float *brow) const {
        if (r == 2) {
            const float2 b = *reinterpret_cast<const float2 *>(brow + blk);
            return (f4){b.x, b.x, b.y, b.y};
        }
        const float b = brow[blk];
        return (f4){b, b, b, b};
    }
};
// Flat map for any other r (R = 3: 3 <= r <= 128).  Lanes over the plane as it lies in memory, like LaneMap<1>; here a
// float4 can straddle two blocks and a block's columns need not start at a lane, a row or a wave.  Per row, the lanes'
// values form a sequence a_0 b_0 a_1 b_1 ... (a: the float4's elements in its first block, b: those in the next) in
// which every block row is a run; the run's first element - its "head" - adds the a's of the lanes behind it (wave-
// private LDS: one ds_write_b32 per lane, the heads read <= r / 4 + 1 values) and stores the sum in layer 2 rho of
// v_part, rho = the row inside the block.  A run that a wave boundary cuts in two leaves its second part in layer
// 2 rho + 1 (written by lane 0 of the next wave); every other head writes a zero there, so that all 2 r layers are
// dense and the reducers add them like z-chunks.  Nothing else is exchanged between lanes.
template <>
struct LaneMap<3> {
    bool act;
    long long off2;
    int bi, cut;          // block under the float4's first element; how many of the four elements lie in it (>= 4: all)
    int a_n, b_n;         // lanes whose a this lane adds as head of its first / second block's run (0: not a head); the
                          // a-run includes the lane itself, the b-run starts with its own b and adds the a of b_n - 1 lanes behind
    int a_sub;            // 1: the a-run continues a block row cut by the wave boundary (layer 2 rho + 1)
    bool a_zero, b_zero;  // the run ends inside this wave: write the zero of layer 2 rho + 1
    int layer;            // 2 rho
    long long lstride;    // floats between two layers
    __device__ __forceinline__ LaneMap(const Geo &q, int wt, int lane) {
        const int r = q.r, Y4 = q.Y >> 2;
        const long long f = (long long)wt * 64 + lane, F = q.P >> 2;
        act = wt < q.WT && f < F;
        const int fi = act ? (int)f : 0;
        const int row = fi / Y4, j = fi - row * Y4, col = 4 * j;
        off2 = 4LL * fi;
        const int xb = row / r, ca = col / r;
        layer = 2 * (row - xb * r);
        lstride = (long long)q.M * q.Ps;
        bi = xb * q.Ys + ca;
        cut = r - (col - ca * r);
        // lanes of this wave behind this one that are live and in the same row
        const long long wave_end = min((long long)wt * 64 + 64, F);
        const int behind = (int)min((long long)(Y4 - 1 - j), wave_end - 1 - f);
        // first block: head when the block starts at this float4, at the start of the row or at the start of the wave
        const bool a_head = act && (lane == 0 || j == 0 || col - ca * r == 0);
        const int a_last = (ca * r + r - 1) >> 2;                  // float4 (within the row) holding the block's last column
        a_n = a_head ? min(a_last - j, behind) + 1 : 0;
        a_sub = (a_head && lane == 0 && j != 0 && col - ca * r != 0) ? 1 : 0;
        a_zero = a_head && !a_sub && (a_last - j <= behind);
        // second block: always a head when the float4 reaches into it
        const bool b_head = act && cut < 4;
        const int b_last = ((ca + 1) * r + r - 1) >> 2;
        b_n = b_head ? min(b_last - j, behind) + 1 : 0;
        b_zero = b_head && (b_last - j <= behind);
    }
    __device__ __forceinline__ void vstore(f4 s4, float *vrow, int lane) const {
        __shared__ float runs[DFX_WAVES][64];
        float *t = runs[threadIdx.x >> 6];
        const float e0 = s4.x, e1 = s4.y, e2 = s4.z, e3 = s4.w;
        const float a = cut >= 4 ? (e0 + e1) + (e2 + e3) : (cut == 3 ? (e0 + e1) + e2 : (cut == 2 ? e0 + e1 : e0));
        const float b = cut >= 4 ? 0.f : (cut == 3 ? e3 : (cut == 2 ? e2 + e3 : (e1 + e2) + e3));
        t[lane] = a;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (a_n) {
            float sum = a;
            for (int k = 1; k < a_n; ++k) sum += t[lane + k];
            vrow[(long long)a_sub * lstride + bi] = sum;
            if (a_zero) vrow[lstride + bi] = 0.f;
        }
        if (b_n) {
            float sum = b;
            for (int k = 1; k < b_n; ++k) sum += t[lane + k];
            vrow[bi + 1] = sum;
            if (b_zero) vrow[lstride + bi + 1] = 0.f;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the next slab's store stays behind these reads
        __builtin_amdgcn_wave_barrier();
    }
    __device__ __forceinline__ f4 bload(const float *brow) const {
        const float b0 = brow[bi];
        const float b1 = cut < 4 ? brow[bi + 1] : b0;
        return (f4){b0, cut > 1 ? b0 : b1, cut > 2 ? b0 : b1, cut > 3 ? b0 : b1};
    }
};
// the layer of v_part a wave writes: its z-chunk, times the row parts of the row-part tile
template <int R>
__device__ __forceinline__ long long v_layer(const LaneMap<R> &L, const Geo &q) {
    if constexpr (R == 16) return (long long)blockIdx.y * q.vq + L.part;
    else if constexpr (R == 1 || R == 3) return (long long)blockIdx.y * q.vq + L.layer;
    else return (long long)blockIdx.y;
}

// ------------------------------------------------------------------------------------------
// Fast sweep: x += alpha p ; p = r + beta p ; u(p), v(p), ||p||^2, ||x||^2, <x,p>   (MODE XP)
// ------------------------------------------------------------------------------------------
struct SweepArgs {
    float *x;
    const float *r;    // explicit residual, or NULL when it is given in factored form:
    const float *ra;   //   r[t,l,z,x,y] = ra[z,x,y] + rb[t,l,x/r,y/r]   (FACT kernels; never read from a 5-D array)
    const float *rb;
    float *p;
    const double *scal;
    float *u_part;
    float *v_part;
    double *part;
    Geo q;
    int steps;     // slabs per u_part chunk (flush interval of the u accumulators)
    int n_chunks;  // ceil(M / steps)
    int cpb;       // chunks one workgroup walks through in a full sweep
    int c_lo, c_hi;  // chunk range of THIS launch (a full sweep is [0, cpb))
    int row0;      // first row of `part` this launch writes
};

// ZN = planes of this z-chunk that exist (ZC for every chunk but possibly the last): a compile-time count, so a
// ragged chunk runs the same straight-line code as a full one over fewer planes
template <int MODE, int ZC, int R, bool FACT, int ZN>
__device__ __forceinline__ void sweep_steps(const SweepArgs &a, int m0, int m1, int ms, long long off,
                                            const LaneMap<R> &L, float fa, float fb, const f4 (&rav)[ZC],
                                            f4 (&uacc)[ZC], double &dpp, double &dxx, double &dxp) {
    float *vbase = a.v_part + v_layer(L, a.q) * a.q.M * a.q.Ps;
    const int lane = threadIdx.x & 63;
    // the ZN live planes go in groups of DFX_ZB (loads issued together, then updated and stored); the LAST group takes
    // the remainder too (ZN = 5 -> 2 + 3, 7 -> 2 + 2 + 3), so that no group runs with fewer loads in flight than the
    // others (5 as 2 + 2 + 1 cost three group latencies for five planes: 87 % of the per-plane rate of 4 or 8)
    constexpr int ZB = DFX_ZB;
    constexpr int NG = (ZN / ZB) > 0 ? ZN / ZB : 1;
    constexpr int GS = (2 * ZB - 1) < ZN ? (2 * ZB - 1) : ZN;     // largest group
    for (int m = m0; m < m1; m += ms) {
        const long long base = (long long)m * a.q.S + off;
        f4 s4 = {0.f, 0.f, 0.f, 0.f};
        float pp = 0.f, xx = 0.f, xp = 0.f;
        f4 rb4 = {0.f, 0.f, 0.f, 0.f};
        if (FACT) rb4 = L.bload(a.rb + (long long)m * a.q.Ps);
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            const int g = gi * ZB;
            const int cnt = (gi == NG - 1) ? ZN - g : ZB;
            f4 xv[GS], rv[GS], pv[GS];
#pragma unroll
            for (int j = 0; j < GS; ++j) {
                if (j < cnt) {
                    const long long i = base + (long long)(g + j) * a.q.P;
                    if (MODE == DFX_SWEEP_PROJECT) {
                        xv[j] = ld4s(a.x + i);
                    } else if (MODE == DFX_SWEEP_XP) {
                        xv[j] = ld4s(a.x + i);
                        rv[j] = FACT ? rav[g + j] + rb4 : ld4s(a.r + i);
                        pv[j] = ld4s(a.p + i);
                    } else if (MODE == DFX_SWEEP_P) {
                        rv[j] = FACT ? rav[g + j] + rb4 : ld4s(a.r + i);
                        pv[j] = ld4s(a.p + i);
                    } else {  // PINIT
                        xv[j] = ld4s(a.x + i);
                        rv[j] = FACT ? rav[g + j] + rb4 : ld4s(a.r + i);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < GS; ++j) {
                if (j < cnt) {
                    const long long i = base + (long long)(g + j) * a.q.P;
                    f4 pn;
                    if (MODE == DFX_SWEEP_PROJECT) {
                        pn = xv[j];
                        xx += dot4(pn, pn);
                    } else if (MODE == DFX_SWEEP_XP) {
                        const f4 xn = xv[j] + fa * pv[j];
                        pn = rv[j] + fb * pv[j];
                        st4s(a.x + i, xn);
                        st4s(a.p + i, pn);
                        xx += dot4(xn, xn);
                        xp += dot4(xn, pn);
                        pp += dot4(pn, pn);
                    } else if (MODE == DFX_SWEEP_P) {
                        pn = rv[j] + fb * pv[j];
                        st4s(a.p + i, pn);
                        pp += dot4(pn, pn);
                    } else {
                        pn = rv[j];
                        st4s(a.p + i, pn);
                        xx += dot4(xv[j], xv[j]);
                        xp += dot4(xv[j], pn);
                        pp += dot4(pn, pn);
                    }
                    uacc[g + j] += pn;
                    s4 += pn;
                }
            }
        }
        L.vstore(s4, vbase + (long long)m * a.q.Ps, lane);
        dpp += (double)pp;
        dxx += (double)xx;
        dxp += (double)xp;
    }
}

template <int MODE, int ZC, int R, bool FACT>
__global__ __launch_bounds__(64 * DFX_WAVES, DFX_MINWAVES) void k_sweep_fast(const SweepArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wt = blockIdx.x * DFX_WAVES + wave;
    const int z0 = blockIdx.y * a.q.zs;   // z-chunks are balanced: zs = ceil(nZ / n_zchunks) <= ZC planes each
    const int zcnt = min(min(a.q.zs, ZC), a.q.nZ - z0);
    const int gz = gridDim.z, bz = blockIdx.z;
    double dpp = 0, dxx = 0, dxp = 0;
    // a wave owns LaneMap<R>::BW consecutive r x r blocks of the plane; only the last wave can have idle lanes
    const LaneMap<R> L(a.q, wt, lane);
    if (L.act) {
        const long long off = (long long)z0 * a.q.P + L.off2;
        const float fa = (MODE == DFX_SWEEP_XP) ? (float)a.scal[DFX_S_ALPHA] : 0.f;
        const float fb = (MODE == DFX_SWEEP_XP || MODE == DFX_SWEEP_P) ? (float)a.scal[DFX_S_BETA] : 0.f;
        f4 rav[ZC];   // this lane's piece of the (z, x, y) factor of r: stays in registers for the whole launch
#pragma unroll
        for (int k = 0; k < ZC; ++k) {
            rav[k] = (f4){0.f, 0.f, 0.f, 0.f};
            if (FACT && k < zcnt) rav[k] = ld4(a.ra + off + (long long)k * a.q.P);
        }
        for (int c = a.c_lo; c < a.c_hi; ++c) {
            // the u accumulators are flushed every `steps` slabs: bounds the length of an fp32 running sum
            // concurrently running workgroups take ADJACENT slabs (m = i*gz + bz): the chip streams one
            // contiguous moving window of each vector; chunk c of every workgroup covers slabs
            // [c*steps*gz, (c+1)*steps*gz), so a range of chunks is a contiguous range of slabs
            const int m0 = c * a.steps * gz + bz, ms = gz;
            const int m1 = min((c + 1) * a.steps * gz, a.q.M);
            f4 uacc[ZC];
#pragma unroll
            for (int k = 0; k < ZC; ++k) uacc[k] = (f4){0.f, 0.f, 0.f, 0.f};
#define DFX_ZN_CASE(n)                                                                                     \
    case n:                                                                                                \
        if constexpr (n <= ZC)                                                                             \
            sweep_steps<MODE, ZC, R, FACT, n>(a, m0, m1, ms, off, L, fa, fb, rav, uacc, dpp, dxx, dxp);   \
        break;
            switch (zcnt) {
                DFX_ZN_CASE(1) DFX_ZN_CASE(2) DFX_ZN_CASE(3) DFX_ZN_CASE(4)
                DFX_ZN_CASE(5) DFX_ZN_CASE(6) DFX_ZN_CASE(7) DFX_ZN_CASE(8)
                default: break;
            }
#undef DFX_ZN_CASE
            // one u_part row per workgroup of the slab axis: chunk 0 writes it, later chunks add to it (same lane,
            // same order in every run: deterministic; a sliced sweep continues the rows of the slices before it)
            float *up = a.u_part + (long long)bz * a.q.S + off;
#pragma unroll
            for (int k = 0; k < ZC; ++k) {
                if (k < zcnt) {
                    f4 t = uacc[k];
                    if (c > 0) t += ld4(up + (long long)k * a.q.P);
                    st4(up + (long long)k * a.q.P, t);
                }
            }
        }
    }
    const long long bid = ((long long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    block_store3(dpp, dxx, dxp, a.part + (a.row0 + bid) * DFX_NPART);
}

// ------------------------------------------------------------------------------------------
// Fast broadcast-axpy: out = in + coef*mask*(sa*A[z,x,y] + sb*B[m,x/r,y/r]); sum(out^2)
// ------------------------------------------------------------------------------------------
struct BcastArgs {
    float *out;
    const float *in;
    const float *A;
    const float *B;
    const float *mask;
    const double *coef_dev;
    double *part;
    float sa, sb, coef_host;
    Geo q;
    int steps, n_chunks, cpb;
};

template <int ZC, int R, bool HAS_IN, bool HAS_MASK, int ZN>
__device__ __forceinline__ void bcast_steps(const BcastArgs &a, int m0, int m1, int ms, long long off,
                                            const LaneMap<R> &L, float cb, const f4 (&av)[ZC], const f4 (&mk)[ZC],
                                            double &drr) {
    // MU slabs per trip: one wave per SIMD is the fastest residency for these kernels, so the loads in
    // flight per lane (MU * ZC float4) are what hides HBM latency
    constexpr int MU = (DFX_BCAST_INFLIGHT / ZN) > 0 ? DFX_BCAST_INFLIGHT / ZN : 1;
    for (int m = m0; m < m1; m += ms * MU) {
        f4 b[MU];
        f4 iv[MU][ZC];
#pragma unroll
        for (int u = 0; u < MU; ++u) {
            const int mm = m + u * ms;
            if (mm < m1) {
                b[u] = cb * L.bload(a.B + (long long)mm * a.q.Ps);
                if (HAS_IN) {
                    const long long base = (long long)mm * a.q.S + off;
#pragma unroll
                    for (int k = 0; k < ZN; ++k) iv[u][k] = ld4s(a.in + base + (long long)k * a.q.P);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < MU; ++u) {
            const int mm = m + u * ms;
            if (mm < m1) {
                const long long base = (long long)mm * a.q.S + off;
                float rr = 0.f;
#pragma unroll
                for (int k = 0; k < ZN; ++k) {
                    {
                        f4 o = av[k] + (HAS_MASK ? mk[k] * b[u] : b[u]);
                        if (HAS_IN) o += iv[u][k];
                        st4s(a.out + base + (long long)k * a.q.P, o);
                        rr += dot4(o, o);
                    }
                }
                drr += (double)rr;
            }
        }
    }
}

template <int ZC, int R, bool HAS_IN, bool HAS_MASK>
__global__ __launch_bounds__(64 * DFX_WAVES, DFX_MINWAVES) void k_bcast_fast(const BcastArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wt = blockIdx.x * DFX_WAVES + wave;
    const int z0 = blockIdx.y * a.q.zs;   // z-chunks are balanced: zs = ceil(nZ / n_zchunks) <= ZC planes each
    const int zcnt = min(min(a.q.zs, ZC), a.q.nZ - z0);
    const int m0 = blockIdx.z, ms = gridDim.z, m1 = a.q.M;
    double drr = 0;
    const LaneMap<R> L(a.q, wt, lane);
    if (L.act) {
        const long long off = (long long)z0 * a.q.P + L.off2;
        const float coef = a.coef_host * (a.coef_dev ? (float)(*a.coef_dev) : 1.f);
        const float ca = coef * a.sa, cb = coef * a.sb;
        f4 av[ZC], mk[ZC];
#pragma unroll
        for (int k = 0; k < ZC; ++k) {
            av[k] = (f4){0.f, 0.f, 0.f, 0.f};
            mk[k] = (f4){1.f, 1.f, 1.f, 1.f};
            if (k < zcnt) {
                av[k] = ca * ld4(a.A + off + (long long)k * a.q.P);
                if (HAS_MASK) {
                    mk[k] = ld4(a.mask + off + (long long)k * a.q.P);
                    av[k] *= mk[k];
                }
            }
        }
#define DFX_ZN_CASE(n)                                                                               \
    case n:                                                                                          \
        if constexpr (n <= ZC)                                                                       \
            bcast_steps<ZC, R, HAS_IN, HAS_MASK, n>(a, m0, m1, ms, off, L, cb, av, mk, drr);        \
        break;
        switch (zcnt) {
            DFX_ZN_CASE(1) DFX_ZN_CASE(2) DFX_ZN_CASE(3) DFX_ZN_CASE(4)
            DFX_ZN_CASE(5) DFX_ZN_CASE(6) DFX_ZN_CASE(7) DFX_ZN_CASE(8)
            default: break;
        }
#undef DFX_ZN_CASE
    }
    const long long bid = ((long long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    block_store3(drr, 0.0, 0.0, a.part + bid * DFX_NPART);
}

// ------------------------------------------------------------------------------------------
// Fused Adam step (fusion.py:324-346 with torch.optim.Adam(amsgrad=False, weight_decay=0)):
//   g = mask * (A[z,x,y] + B[m,x/r,y/r])        gradient, broadcast on read: never stored in HBM
//   m1 += (1-b1)(g - m1) ; m2 = b2 m2 + (1-b2) g^2
//   x -= (lr/bc1) * m1 / (sqrt(m2)/sqrt(bc2) + eps) ; optional clamp(min=0)
// plus, in the same pass, what the next iteration needs: u(x), v(x), ||dx||^2, ||x||^2.
// x, m1, m2 read + written: 6N words per iteration.
//
// Factored first moment (M1F).  m1 is linear in the gradients, and every gradient is mask * (A (+) B): so
//   m1 = mask * (m1a[z,x,y] + m1b[m,x/r,y/r]),  m1a <- lerp(m1a, A, 1-b1),  m1b <- lerp(m1b, B, 1-b1)   (dfx_lerp)
// exactly, and the 5-D m1 array need not exist: x and m2 read + written, 4N words per iteration.  (m2 is an average of
// g^2 = A^2 + 2AB + B^2; the cross terms do not factor.)
// ------------------------------------------------------------------------------------------
struct AdamArgs {
    float *x, *m1, *m2;
    const float *m1a, *m1b;   // M1F: the first moment AFTER this step's lerp, as its two small-space factors
    const float *A, *B, *mask;
    float *u_part, *v_part;
    double *part;
    float b1, b2, omb1, omb2, eps, step_size, inv_sqrt_bc2;   // omb = 1 - beta, rounded from double like torch
    int non_neg;
    Geo q;
    int steps, cpb;
    int c_lo, c_hi;   // chunk range of THIS launch (a full step is [0, cpb)), as in SweepArgs
    int row0;         // first row of `part` this launch writes
    const float *grad;   // GRAD kernels (dfx_adam_apply_project): the explicit gradient; A, B, mask, m1a, m1b unused
};

// FAST: the hardware square root and reciprocal (1 ulp each; v_sqrt_f32 flushes denormal inputs, harmless while
// eps > 0, which the fused entry points require) - the 5-D kernels of the fused FusionAdam path, where the IEEE
// sequences made the pass VALU-bound.  !FAST: torch's arithmetic bit for bit (IEEE sqrt and division): k_adam_apply
// (DfxAdam, which claims torch.optim.Adam's update and is bound by its seven memory streams either way) and the
// generic kernel.
template <bool M1F, bool FAST>
__device__ __forceinline__ float adam_elem(float x, float g, float &m1, float &m2, const AdamArgs &a) {
    if (!FAST) {
        // DfxAdam's arithmetic, spelled out so that it does not depend on which kernel it is inlined into (the
        // compiler's choice of fused multiply-adds differed between k_adam_apply and the lane-mapped kernel by one ulp):
        // the roundings of torch's own sequence - lerp_ as fma(w, g - m1, m1); mul_(beta2) rounded, then addcmul_ as
        // fma(value * g, g, .); sqrt, scale, + eps; addcdiv_ as fma(-step_size, m1 / denom, x)
#pragma clang fp contract(off)
        if (!M1F) m1 = __builtin_fmaf(a.omb1, g - m1, m1);
        const float m2b = a.b2 * m2;
        m2 = __builtin_fmaf(a.omb2 * g, g, m2b);
        const float denom = __fsqrt_rn(m2) * a.inv_sqrt_bc2 + a.eps;
        float xn = __builtin_fmaf(-a.step_size, __fdiv_rn(m1, denom), x);
        if (a.non_neg) xn = fmaxf(xn, 0.f);               // fusion.py:344-346
        return xn;
    }
    if (!M1F) m1 = m1 + a.omb1 * (g - m1);                // exp_avg.lerp_(grad, 1 - beta1); M1F: m1 is already new
    m2 = a.b2 * m2 + a.omb2 * (g * g);                    // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    float xn;
    if (FAST) {
        // the step changes by < 3e-7 of itself, far below one ulp of x for any step that does not swamp x; exp_avg_sq
        // (the state carried on) is exact either way
        const float denom = __builtin_amdgcn_sqrtf(m2) * a.inv_sqrt_bc2 + a.eps;
        xn = x - a.step_size * (m1 * __builtin_amdgcn_rcpf(denom));
    } else {
        const float denom = __fsqrt_rn(m2) * a.inv_sqrt_bc2 + a.eps;
        xn = x - a.step_size * __fdiv_rn(m1, denom);      // param.addcdiv_(exp_avg, denom, value=-step_size)
    }
    if (a.non_neg) xn = fmaxf(xn, 0.f);                   // fusion.py:344-346
    return xn;
}

template <bool M1F, bool FAST = (DFX_ADAM_FASTMATH != 0)>
__device__ __forceinline__ f4 adam_elem4(f4 x, f4 g, f4 &m1, f4 &m2, const AdamArgs &a) {
    float x0 = x.x, x1 = x.y, x2 = x.z, x3 = x.w;
    float p0 = m1.x, p1 = m1.y, p2 = m1.z, p3 = m1.w;
    float q0 = m2.x, q1 = m2.y, q2 = m2.z, q3 = m2.w;
    x0 = adam_elem<M1F, FAST>(x0, g.x, p0, q0, a);
    x1 = adam_elem<M1F, FAST>(x1, g.y, p1, q1, a);
    x2 = adam_elem<M1F, FAST>(x2, g.z, p2, q2, a);
    x3 = adam_elem<M1F, FAST>(x3, g.w, p3, q3, a);
    m1 = (f4){p0, p1, p2, p3};
    m2 = (f4){q0, q1, q2, q3};
    return (f4){x0, x1, x2, x3};
}

// The slabs m0, m0 + gz, ... < m1e of one u_part chunk, ZN live planes: the sweep's structure - DFX_ADAM_ZB planes
// loaded, updated and stored at a time.  (Measured at C2, factored first moment, with the hardware sqrt / rcp: 2 planes
// 2.70 ms, 3: 2.78, 4: 2.80, 8: 2.87, 1: 3.80; whole slabs in two register sets in ping-pong, the round-1 loop, 2.87;
// with the IEEE sqrt / division sequences 2 planes take 2.85: both changes are needed.)
template <int ZC, int R, bool HAS_MASK, bool M1F, int ZN, bool GRAD = false>
__device__ __forceinline__ void adam_chunk(const AdamArgs &a, const LaneMap<R> &L, int m0, int m1e, int gz,
                                              long long off, const f4 (&av)[ZC], const f4 (&mk)[ZC],
                                              const f4 (&m1av)[ZC], f4 (&uacc)[ZC], float *vbase, int lane,
                                              double &ddd, double &dxx) {
    constexpr int NG = (ZN / DFX_ADAM_ZB) > 0 ? ZN / DFX_ADAM_ZB : 1;      // groups as in sweep_steps: the last one
    constexpr int GS = (2 * DFX_ADAM_ZB - 1) < ZN ? (2 * DFX_ADAM_ZB - 1) : ZN;   // takes the remainder
    for (int m = m0; m < m1e; m += gz) {
        const long long base = (long long)m * a.q.S + off;
        f4 b4 = {0.f, 0.f, 0.f, 0.f};
        if (!GRAD) b4 = L.bload(a.B + (long long)m * a.q.Ps);
        f4 mb4 = {0.f, 0.f, 0.f, 0.f};
        if (M1F) mb4 = L.bload(a.m1b + (long long)m * a.q.Ps);
        f4 s4 = {0.f, 0.f, 0.f, 0.f};
        float dd = 0.f, xx = 0.f;
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            const int g = gi * DFX_ADAM_ZB;
            const int cnt = (gi == NG - 1) ? ZN - g : DFX_ADAM_ZB;
            f4 xv[GS], e1[GS], e2[GS], gv[GRAD ? GS : 1];
#pragma unroll
            for (int j = 0; j < GS; ++j) {
                if (j < cnt) {
                    const long long i = base + (long long)(g + j) * a.q.P;
                    xv[j] = ld4s(a.x + i);
                    if (GRAD) gv[j] = ld4s(a.grad + i);
                    if (!M1F) e1[j] = ld4s(a.m1 + i);
                    e2[j] = ld4s(a.m2 + i);
                }
            }
#pragma unroll
            for (int j = 0; j < GS; ++j) {
                if (j < cnt) {
                    const int k = g + j;
                    const long long i = base + (long long)k * a.q.P;
                    f4 gr = GRAD ? gv[GRAD ? j : 0] : av[k] + b4;
                    if (HAS_MASK) gr *= mk[k];
                    if (M1F) {
                        e1[j] = m1av[k] + mb4;
                        if (HAS_MASK) e1[j] *= mk[k];
                    }
                    // GRAD: DfxAdam's step - torch's arithmetic bit for bit (IEEE sqrt and division), like k_adam_apply
                    const f4 xn = GRAD ? adam_elem4<false, false>(xv[j], gr, e1[j], e2[j], a)
                                       : adam_elem4<M1F>(xv[j], gr, e1[j], e2[j], a);
                    st4s(a.x + i, xn);
                    if (!M1F) st4s(a.m1 + i, e1[j]);
                    st4s(a.m2 + i, e2[j]);
                    const f4 d = xn - xv[j];
                    dd += dot4(d, d);
                    xx += dot4(xn, xn);
                    uacc[k] += xn;
                    s4 += xn;
                }
            }
        }
        L.vstore(s4, vbase + (long long)m * a.q.Ps, lane);
        ddd += (double)dd;
        dxx += (double)xx;
    }
}

template <int ZC, int R, bool HAS_MASK, bool M1F, bool GRAD = false>
__global__ __launch_bounds__(64 * DFX_WAVES, DFX_MINWAVES) void k_adam_fast(const AdamArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wt = blockIdx.x * DFX_WAVES + wave;
    const int z0 = blockIdx.y * a.q.zs;   // z-chunks are balanced: zs = ceil(nZ / n_zchunks) <= ZC planes each
    const int zcnt = min(min(a.q.zs, ZC), a.q.nZ - z0);
    const int gz = gridDim.z, bz = blockIdx.z;
    double ddd = 0, dxx = 0;
    const LaneMap<R> L(a.q, wt, lane);
    if (L.act) {
        const long long off = (long long)z0 * a.q.P + L.off2;
        float *vbase = a.v_part + v_layer(L, a.q) * a.q.M * a.q.Ps;
        f4 av[ZC], mk[ZC], m1av[ZC];
#pragma unroll
        for (int k = 0; k < ZC; ++k) {
            av[k] = (f4){0.f, 0.f, 0.f, 0.f};
            mk[k] = (f4){1.f, 1.f, 1.f, 1.f};
            m1av[k] = (f4){0.f, 0.f, 0.f, 0.f};
            if (k < zcnt) {
                if (!GRAD) av[k] = ld4(a.A + off + (long long)k * a.q.P);
                if (HAS_MASK) mk[k] = ld4(a.mask + off + (long long)k * a.q.P);
                if (M1F) m1av[k] = ld4(a.m1a + off + (long long)k * a.q.P);
            }
        }
        for (int c = a.c_lo; c < a.c_hi; ++c) {
            const int m0 = c * a.steps * gz + bz;
            const int m1e = min((c + 1) * a.steps * gz, a.q.M);
            f4 uacc[ZC];
#pragma unroll
            for (int k = 0; k < ZC; ++k) uacc[k] = (f4){0.f, 0.f, 0.f, 0.f};
#define DFX_ZN_CASE(n)                                                                                          \
    case n:                                                                                                     \
        if constexpr (n <= ZC)                                                                                  \
            adam_chunk<ZC, R, HAS_MASK, M1F, n, GRAD>(a, L, m0, m1e, gz, off, av, mk, m1av, uacc, vbase, lane, ddd, dxx); \
        break;
            switch (zcnt) {
                DFX_ZN_CASE(1) DFX_ZN_CASE(2) DFX_ZN_CASE(3) DFX_ZN_CASE(4)
                DFX_ZN_CASE(5) DFX_ZN_CASE(6) DFX_ZN_CASE(7) DFX_ZN_CASE(8)
                default: break;
            }
#undef DFX_ZN_CASE
            float *up = a.u_part + (long long)bz * a.q.S + off;
#pragma unroll
            for (int k = 0; k < ZC; ++k) {
                if (k < zcnt) {
                    f4 t = uacc[k];
                    if (c > 0) t += ld4(up + (long long)k * a.q.P);
                    st4(up + (long long)k * a.q.P, t);
                }
            }
        }
    }
    const long long bid = ((long long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    block_store3(ddd, dxx, 0.0, a.part + (a.row0 + bid) * DFX_NPART);
}

// torch.optim.Adam's step on an EXPLICIT gradient (the autograd formulation, fusion.py:338-342: loss.backward() has
// left x.grad, optimizer.step() consumes it): x, grad, exp_avg, exp_avg_sq read, x, exp_avg, exp_avg_sq written - 7N
// words in one launch (torch's own fused implementation takes 52 launches per step on a 4 GiB tensor, 7.6 ms at C2).
// Workgroup b takes the float4 groups b, b + gridDim, ...: the chip streams one contiguous moving window per array.
// Measured at C2 (tools/tune_adam_apply.py): 5.35 ms = 5.6 TB/s with one workgroup per CU and one group per array
// loaded before the stores; 2 or 4 groups 5.6 - 5.7 ms, 512 - 2048 workgroups 5.8 - 6.5 ms, runs of 4 - 64 consecutive
// 4 KiB pieces per workgroup 5.47 - 5.49 ms: seven streams top out here (six: 6.0 TB/s, four: 6.4).
template <int UNROLL>
__global__ __launch_bounds__(256) void k_adam_apply(float *x, const float *g, float *m1, float *m2, long long n,
                                                    AdamArgs a) {
    double ddd = 0, dxx = 0;
    const long long n4 = n >> 2;
    const long long stride = (long long)gridDim.x * 256;
    for (long long i0 = (long long)blockIdx.x * 256 + threadIdx.x; i0 < n4; i0 += stride * UNROLL) {
        f4 xv[UNROLL], gv[UNROLL], e1[UNROLL], e2[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const long long i = i0 + u * stride;
            if (i < n4) {
                xv[u] = ld4s(x + 4 * i);
                gv[u] = ld4s(g + 4 * i);
                e1[u] = ld4s(m1 + 4 * i);
                e2[u] = ld4s(m2 + 4 * i);
            }
        }
        float dd = 0.f, xx = 0.f;
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const long long i = i0 + u * stride;
            if (i < n4) {
                const f4 xn = adam_elem4<false, false>(xv[u], gv[u], e1[u], e2[u], a);
                st4s(x + 4 * i, xn);
                st4s(m1 + 4 * i, e1[u]);
                st4s(m2 + 4 * i, e2[u]);
                const f4 d = xn - xv[u];
                dd += dot4(d, d);
                xx += dot4(xn, xn);
            }
        }
        ddd += (double)dd;
        dxx += (double)xx;
    }
    for (long long i = 4 * n4 + (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {   // tail (n % 4)
        float e1 = m1[i], e2 = m2[i];
        const float xo = x[i], xn = adam_elem<false, false>(xo, g[i], e1, e2, a);
        x[i] = xn, m1[i] = e1, m2[i] = e2;
        ddd += (double)((xn - xo) * (xn - xo));
        dxx += (double)(xn * xn);
    }
    block_store3(ddd, dxx, 0.0, a.part + (long long)blockIdx.x * DFX_NPART);
}

__global__ __launch_bounds__(256) void k_gen_adam(AdamArgs a) {
    double ddd = 0, dxx = 0;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < a.q.N; i += stride) {
        const long long m = i / a.q.S;
        const int j = (int)(i - m * a.q.S);
        const int xy = (int)(j % a.q.P);
        const int xi = xy / a.q.Y, yi = xy - xi * a.q.Y;
        const long long jb = m * a.q.Ps + (long long)(xi / a.q.r) * a.q.Ys + yi / a.q.r;
        float g = a.A[j] + a.B[jb];
        if (a.mask) g *= a.mask[j];
        const float xo = a.x[i];
        float e1, e2 = a.m2[i], xn;
        if (a.m1) {
            e1 = a.m1[i];
            xn = adam_elem<false, false>(xo, g, e1, e2, a);
            a.m1[i] = e1;
        } else {
            e1 = a.m1a[j] + a.m1b[jb];
            if (a.mask) e1 *= a.mask[j];
            xn = adam_elem<true, false>(xo, g, e1, e2, a);
        }
        a.x[i] = xn;
        a.m2[i] = e2;
        ddd += (double)((xn - xo) * (xn - xo));
        dxx += (double)(xn * xn);
    }
    block_store3(ddd, dxx, 0.0, a.part + (long long)blockIdx.x * DFX_NPART);
}

// ------------------------------------------------------------------------------------------
// Generic (any r, X, Y) kernels: element-wise pass + two projection passes.
// ------------------------------------------------------------------------------------------
// residual element i of the generic kernels: explicit array, or the factored form ra[z,x,y] + rb[t,l,x/r,y/r]
__device__ __forceinline__ float gen_resid(const float *r, const float *ra, const float *rb, const Geo &q,
                                           long long i) {
    if (r) return r[i];
    const long long m = i / q.S;
    const int j = (int)(i - m * q.S);
    const int xy = j % (int)q.P;
    const int xi = xy / q.Y, yi = xy - xi * q.Y;
    return ra[j] + rb[m * q.Ps + (long long)(xi / q.r) * q.Ys + yi / q.r];
}

// The generic sweep (any r, any Y): the fast path's loop structure with scalar lanes.  A thread owns DFX_GEN_K positions
// j of the slab (consecutive threads, consecutive j: 256 B per wave-instruction) and walks the slabs m = bz, bz + gz, ...;
// what depends on j only - the block under it, its ra - is worked out once per tile instead of per element (the first
// form, k_gen_sweep, spent its time in 64-bit divisions: 3.5 TB/s), and u accumulates in registers (fp32 for 64 slabs,
// then into a double), so the separate projection pass for u (one more read of the volume) is gone.  v keeps its own
// pass (k_gen_proj_v).  Grid: (tile slots, gz); u_part has gz rows.
#define DFX_GEN_K 4
template <int MODE>
__global__ __launch_bounds__(256) void k_gen_sweep2(float *x, const float *r, const float *ra, const float *rb, float *p,
                                                    const double *scal, Geo q, int n_tiles, float *u_part,
                                                    double *part) {
    const int gz = gridDim.y, bz = blockIdx.y;
    const float fa = (MODE == DFX_SWEEP_XP) ? (float)scal[DFX_S_ALPHA] : 0.f;
    const float fb = (MODE == DFX_SWEEP_XP || MODE == DFX_SWEEP_P) ? (float)scal[DFX_S_BETA] : 0.f;
    double dpp = 0, dxx = 0, dxp = 0;
    const int S = (int)q.S, P = (int)q.P;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        int j[DFX_GEN_K], blk[DFX_GEN_K];
        bool ok[DFX_GEN_K];
        float rav[DFX_GEN_K], uf[DFX_GEN_K];
        double ud[DFX_GEN_K];
#pragma unroll
        for (int k = 0; k < DFX_GEN_K; ++k) {
            const long long jj = ((long long)tile * DFX_GEN_K + k) * 256 + threadIdx.x;
            ok[k] = jj < S;
            j[k] = ok[k] ? (int)jj : 0;
            const int xy = j[k] % P, xi = xy / q.Y, yi = xy - xi * q.Y;
            blk[k] = (xi / q.r) * q.Ys + yi / q.r;
            rav[k] = (ra && ok[k]) ? ra[j[k]] : 0.f;
            uf[k] = 0.f;
            ud[k] = 0.0;
        }
        int run = 0;
        for (int m = bz; m < q.M; m += gz) {
            const long long base = (long long)m * S;
            const float *rbm = rb ? rb + (long long)m * q.Ps : nullptr;
            float xv[DFX_GEN_K], pv[DFX_GEN_K], rv[DFX_GEN_K];
#pragma unroll
            for (int k = 0; k < DFX_GEN_K; ++k) {
                xv[k] = pv[k] = rv[k] = 0.f;
                if (ok[k]) {
                    const long long i = base + j[k];
                    if (MODE != DFX_SWEEP_P) xv[k] = x[i];
                    if (MODE == DFX_SWEEP_XP || MODE == DFX_SWEEP_P) pv[k] = p[i];
                    if (MODE != DFX_SWEEP_PROJECT) rv[k] = r ? r[i] : rav[k] + rbm[blk[k]];
                }
            }
            float pp = 0.f, xx = 0.f, xp = 0.f;
#pragma unroll
            for (int k = 0; k < DFX_GEN_K; ++k) {
                if (ok[k]) {
                    const long long i = base + j[k];
                    float pn;
                    if (MODE == DFX_SWEEP_PROJECT) {
                        pn = xv[k];
                        xx += pn * pn;
                    } else if (MODE == DFX_SWEEP_XP) {
                        const float xn = xv[k] + fa * pv[k];
                        pn = rv[k] + fb * pv[k];
                        x[i] = xn;
                        p[i] = pn;
                        xx += xn * xn;
                        xp += xn * pn;
                        pp += pn * pn;
                    } else if (MODE == DFX_SWEEP_P) {
                        pn = rv[k] + fb * pv[k];
                        p[i] = pn;
                        pp += pn * pn;
                    } else {
                        pn = rv[k];
                        p[i] = pn;
                        xx += xv[k] * xv[k];
                        xp += xv[k] * pn;
                        pp += pn * pn;
                    }
                    uf[k] += pn;
                }
            }
            dpp += (double)pp;
            dxx += (double)xx;
            dxp += (double)xp;
            if (++run == 64) {   // bounds the length of an fp32 running sum, like the fast path's flush interval
                run = 0;
#pragma unroll
                for (int k = 0; k < DFX_GEN_K; ++k) ud[k] += (double)uf[k], uf[k] = 0.f;
            }
        }
#pragma unroll
        for (int k = 0; k < DFX_GEN_K; ++k)
            if (ok[k]) u_part[(long long)bz * S + j[k]] = (float)(ud[k] + (double)uf[k]);
    }
    block_store3(dpp, dxx, dxp, part + ((long long)bz * gridDim.x + blockIdx.x) * DFX_NPART);
}

// The same with float4 lanes, for slabs whose length is a multiple of 4 (then every float4 of the FLAT volume is 16-byte
// aligned whatever Y is): a thread owns DFX_GEN_K4 float4 groups of the slab, each wave-instruction moves 1 KiB of
// contiguous memory and the accesses are non-temporal like the fast path's.  The four elements of a group may lie in
// different rows and blocks (Y % 4 != 0), so each has its own block index for the rb gather.
#ifndef DFX_GEN_K4
#define DFX_GEN_K4 1      // measured at C2 size (126 x 126 planes, r = 3): 1: 3.20 ms, 2: 3.55, 4: 4.9
#endif
template <int MODE>
__global__ __launch_bounds__(256) void k_gen_sweep4(float *x, const float *r, const float *ra, const float *rb, float *p,
                                                    const double *scal, Geo q, int n_tiles, float *u_part,
                                                    double *part) {
    const int gz = gridDim.y, bz = blockIdx.y;
    const float fa = (MODE == DFX_SWEEP_XP) ? (float)scal[DFX_S_ALPHA] : 0.f;
    const float fb = (MODE == DFX_SWEEP_XP || MODE == DFX_SWEEP_P) ? (float)scal[DFX_S_BETA] : 0.f;
    double dpp = 0, dxx = 0, dxp = 0;
    const int S = (int)q.S, P = (int)q.P;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        int j[DFX_GEN_K4], blk[DFX_GEN_K4][4];
        bool ok[DFX_GEN_K4];
        f4 rav[DFX_GEN_K4], uf[DFX_GEN_K4];
#pragma unroll
        for (int k = 0; k < DFX_GEN_K4; ++k) {
            const long long jj = 4 * (((long long)tile * DFX_GEN_K4 + k) * 256 + threadIdx.x);
            ok[k] = jj < S;
            j[k] = ok[k] ? (int)jj : 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int xy = (j[k] + e) % P, xi = xy / q.Y, yi = xy - xi * q.Y;
                blk[k][e] = (xi / q.r) * q.Ys + yi / q.r;
            }
            rav[k] = (ra && ok[k]) ? ld4(ra + j[k]) : (f4){0.f, 0.f, 0.f, 0.f};
            uf[k] = (f4){0.f, 0.f, 0.f, 0.f};
        }
        int run = 0;
        bool first = true;
        for (int m = bz; m < q.M; m += gz) {
            const long long base = (long long)m * S;
            const float *rbm = rb ? rb + (long long)m * q.Ps : nullptr;
            f4 xv[DFX_GEN_K4], pv[DFX_GEN_K4], rv[DFX_GEN_K4];
#pragma unroll
            for (int k = 0; k < DFX_GEN_K4; ++k) {
                xv[k] = pv[k] = rv[k] = (f4){0.f, 0.f, 0.f, 0.f};
                if (ok[k]) {
                    const long long i = base + j[k];
                    if (MODE != DFX_SWEEP_P) xv[k] = ld4s(x + i);
                    if (MODE == DFX_SWEEP_XP || MODE == DFX_SWEEP_P) pv[k] = ld4s(p + i);
                    if (MODE != DFX_SWEEP_PROJECT)
                        rv[k] = r ? ld4s(r + i)
                                  : rav[k] + (f4){rbm[blk[k][0]], rbm[blk[k][1]], rbm[blk[k][2]], rbm[blk[k][3]]};
                }
            }
            float pp = 0.f, xx = 0.f, xp = 0.f;
#pragma unroll
            for (int k = 0; k < DFX_GEN_K4; ++k) {
                if (ok[k]) {
                    const long long i = base + j[k];
                    f4 pn;
                    if (MODE == DFX_SWEEP_PROJECT) {
                        pn = xv[k];
                        xx += dot4(pn, pn);
                    } else if (MODE == DFX_SWEEP_XP) {
                        const f4 xn = xv[k] + fa * pv[k];
                        pn = rv[k] + fb * pv[k];
                        st4s(x + i, xn);
                        st4s(p + i, pn);
                        xx += dot4(xn, xn);
                        xp += dot4(xn, pn);
                        pp += dot4(pn, pn);
                    } else if (MODE == DFX_SWEEP_P) {
                        pn = rv[k] + fb * pv[k];
                        st4s(p + i, pn);
                        pp += dot4(pn, pn);
                    } else {
                        pn = rv[k];
                        st4s(p + i, pn);
                        xx += dot4(xv[k], xv[k]);
                        xp += dot4(xv[k], pn);
                        pp += dot4(pn, pn);
                    }
                    uf[k] += pn;
                }
            }
            dpp += (double)pp;
            dxx += (double)xx;
            dxp += (double)xp;
            if (++run == 64) {   // flush: bounds the length of an fp32 running sum (the row is this workgroup row's own)
                run = 0;
#pragma unroll
                for (int k = 0; k < DFX_GEN_K4; ++k) {
                    if (ok[k]) {
                        float *up = u_part + (long long)bz * S + j[k];
                        st4(up, first ? uf[k] : uf[k] + ld4(up));
                    }
                    uf[k] = (f4){0.f, 0.f, 0.f, 0.f};
                }
                first = false;
            }
        }
#pragma unroll
        for (int k = 0; k < DFX_GEN_K4; ++k) {
            if (ok[k]) {
                float *up = u_part + (long long)bz * S + j[k];
                st4(up, first ? uf[k] : uf[k] + ld4(up));
            }
        }
    }
    block_store3(dpp, dxx, dxp, part + ((long long)bz * gridDim.x + blockIdx.x) * DFX_NPART);
}

__global__ __launch_bounds__(256) void k_gen_proj_u(const float *a, float *u, Geo q) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < q.S; j += stride) {
        float s = 0.f;
        for (int m = 0; m < q.M; ++m) s += a[(long long)m * q.S + j];
        u[j] = s;
    }
}

__global__ __launch_bounds__(256) void k_gen_proj_v(const float *a, float *v, Geo q) {
    const long long total = (long long)q.M * q.Ps;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < total; j += stride) {
        const int m = (int)(j / q.Ps);
        const int rem = (int)(j - (long long)m * q.Ps);
        const int xb = rem / q.Ys, yb = rem - xb * q.Ys;
        float s = 0.f;
        for (int z = 0; z < q.nZ; ++z) {
            const float *pl = a + (long long)m * q.S + (long long)z * q.P;
            for (int dx = 0; dx < q.r; ++dx)
                for (int dy = 0; dy < q.r; ++dy) s += pl[(long long)(xb * q.r + dx) * q.Y + yb * q.r + dy];
        }
        v[j] = s;
    }
}

// v of the generic path, second form: one wave per (slab, block row, column chunk); lanes walk the columns of the chunk
// (coalesced row segments of up to 1 KiB) and add the nZ * r rows that feed the block row; the r columns of each block
// are then added through a wave-private LDS row (LDS operations of one wave execute in order; wavefront-scope fences
// for the compiler).  The first form gave every thread a whole block to gather (4 TB/s).
__global__ __launch_bounds__(256) void k_gen_proj_v2(const float *a, float *v, Geo q) {
    __shared__ float cs_all[4][256];
    float *cs = cs_all[threadIdx.x >> 6];
    const int lane = threadIdx.x & 63;
    const int C = q.r * (256 / q.r);                       // chunk width: whole blocks, at most 256 columns
    const int n_ch = (q.Y + C - 1) / C;
    const long long items = (long long)q.M * q.Xs * n_ch;
    const long long wid = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (long long)gridDim.x * 4;
    for (long long it = wid; it < items; it += nw) {
        const int ch = (int)(it % n_ch);
        const long long t = it / n_ch;
        const int xb = (int)(t % q.Xs);
        const long long m = t / q.Xs;
        const int y0 = ch * C, cw = min(C, q.Y - y0);
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        const float *row0 = a + m * q.S + (long long)(xb * q.r) * q.Y + y0;
        const int n_rows = q.nZ * q.r;          // the rows that feed this block row, four at a time (loads in flight)
        for (int rr0 = 0; rr0 < n_rows; rr0 += 4) {
            float t[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int rr = rr0 + u, z = rr / q.r, dx = rr - z * q.r;
                const float *row = row0 + (long long)z * q.P + (long long)dx * q.Y;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int c = lane + 64 * k;
                    t[u][k] = (rr < n_rows && c < cw) ? row[c] : 0.f;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[k] += t[u][k];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = lane + 64 * k;
            if (c < cw) cs[c] = acc[k];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int nb = cw / q.r;
        for (int b = lane; b < nb; b += 64) {
            float sum = 0.f;
            for (int d = 0; d < q.r; ++d) sum += cs[b * q.r + d];
            v[m * q.Ps + (long long)xb * q.Ys + y0 / q.r + b] = sum;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

__global__ __launch_bounds__(256) void k_gen_bcast(BcastArgs a) {
    const float coef = a.coef_host * (a.coef_dev ? (float)(*a.coef_dev) : 1.f);
    const float ca = coef * a.sa, cb = coef * a.sb;
    double drr = 0;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < a.q.N; i += stride) {
        const long long m = i / a.q.S;
        const int j = (int)(i - m * a.q.S);  // z*P + x*Y + y
        const int xy = (int)(j % a.q.P);
        const int xi = xy / a.q.Y, yi = xy - xi * a.q.Y;
        float o = ca * a.A[j] + cb * a.B[m * a.q.Ps + (long long)(xi / a.q.r) * a.q.Ys + yi / a.q.r];
        if (a.mask) o *= a.mask[j];
        if (a.in) o += a.in[i];
        a.out[i] = o;
        drr += (double)(o * o);
    }
    block_store3(drr, 0.0, 0.0, a.part + (long long)blockIdx.x * DFX_NPART);
}

// The generic broadcast-axpy and fused Adam step with float4 lanes over the flat volume (slab length a multiple of 4),
// in the structure of k_gen_sweep4: thread-owned slab positions, everything that depends on the position only (A, mask,
// the blocks under the four elements, m1a) worked out once, the slabs walked with 1 KiB per wave-instruction.  The
// element-wise forms above (index arithmetic per element) stay for slabs whose length is not a multiple of 4.
__global__ __launch_bounds__(256) void k_gen_bcast4(BcastArgs a, int n_tiles) {
    const int gz = gridDim.y, bz = blockIdx.y;
    const float coef = a.coef_host * (a.coef_dev ? (float)(*a.coef_dev) : 1.f);
    const float ca = coef * a.sa, cb = coef * a.sb;
    const int S = (int)a.q.S, P = (int)a.q.P;
    double drr = 0;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const long long jj = 4 * ((long long)tile * 256 + threadIdx.x);
        if (jj >= S) continue;
        const int j = (int)jj;
        int blk[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int xy = (j + e) % P, xi = xy / a.q.Y, yi = xy - xi * a.q.Y;
            blk[e] = (xi / a.q.r) * a.q.Ys + yi / a.q.r;
        }
        const f4 av = ca * ld4(a.A + j);
        const f4 mk = a.mask ? ld4(a.mask + j) : (f4){1.f, 1.f, 1.f, 1.f};
        for (int m = bz; m < a.q.M; m += gz) {
            const long long i = (long long)m * S + j;
            const float *bm = a.B + (long long)m * a.q.Ps;
            f4 o = (av + cb * (f4){bm[blk[0]], bm[blk[1]], bm[blk[2]], bm[blk[3]]}) * mk;
            if (a.in) o += ld4s(a.in + i);
            st4s(a.out + i, o);
            drr += (double)dot4(o, o);
        }
    }
    block_store3(drr, 0.0, 0.0, a.part + ((long long)bz * gridDim.x + blockIdx.x) * DFX_NPART);
}

template <bool M1F>
__global__ __launch_bounds__(256) void k_gen_adam4(AdamArgs a, int n_tiles) {
    const int gz = gridDim.y, bz = blockIdx.y;
    const int S = (int)a.q.S, P = (int)a.q.P;
    double ddd = 0, dxx = 0;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const long long jj = 4 * ((long long)tile * 256 + threadIdx.x);
        if (jj >= S) continue;
        const int j = (int)jj;
        int blk[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int xy = (j + e) % P, xi = xy / a.q.Y, yi = xy - xi * a.q.Y;
            blk[e] = (xi / a.q.r) * a.q.Ys + yi / a.q.r;
        }
        const f4 av = ld4(a.A + j);
        const f4 mk = a.mask ? ld4(a.mask + j) : (f4){1.f, 1.f, 1.f, 1.f};
        const f4 m1av = M1F ? ld4(a.m1a + j) : (f4){0.f, 0.f, 0.f, 0.f};
        f4 uf = {0.f, 0.f, 0.f, 0.f};
        float *up = a.u_part + (long long)bz * S + j;
        int run = 0;
        bool first = true;
        for (int m = bz; m < a.q.M; m += gz) {
            const long long i = (long long)m * S + j;
            const float *bm = a.B + (long long)m * a.q.Ps;
            const f4 xo = ld4s(a.x + i);
            f4 e2 = ld4s(a.m2 + i), e1;
            if (M1F) {
                const float *mbm = a.m1b + (long long)m * a.q.Ps;
                e1 = (m1av + (f4){mbm[blk[0]], mbm[blk[1]], mbm[blk[2]], mbm[blk[3]]}) * mk;
            } else {
                e1 = ld4s(a.m1 + i);
            }
            const f4 gr = (av + (f4){bm[blk[0]], bm[blk[1]], bm[blk[2]], bm[blk[3]]}) * mk;
            const f4 xn = adam_elem4<M1F, false>(xo, gr, e1, e2, a);      // the generic kernels' arithmetic: IEEE
            st4s(a.x + i, xn);
            if (!M1F) st4s(a.m1 + i, e1);
            st4s(a.m2 + i, e2);
            const f4 d = xn - xo;
            ddd += (double)dot4(d, d);
            dxx += (double)dot4(xn, xn);
            uf += xn;
            if (++run == 64) {
                run = 0;
                st4(up, first ? uf : uf + ld4(up));
                uf = (f4){0.f, 0.f, 0.f, 0.f};
                first = false;
            }
        }
        st4(up, first ? uf : uf + ld4(up));
    }
    block_store3(ddd, dxx, 0.0, a.part + ((long long)bz * gridDim.x + blockIdx.x) * DFX_NPART);
}

// ------------------------------------------------------------------------------------------
// Small-space reductions and scalar bookkeeping
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_reduce_u(const float *u_part, float *u, long long len, int n_chunks,
                                                  double *part) {
    double acc = 0;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += stride) {
        float s = 0.f;
        for (int c = 0; c < n_chunks; ++c) s += u_part[(long long)c * len + i];
        u[i] = s;
        acc += (double)s * (double)s;
    }
    block_store3(acc, 0.0, 0.0, part + (long long)blockIdx.x * DFX_NPART);
}

// v = the sum of the layers of v_part (z-chunks x Geo::vq), slabs [lo, hi)
__global__ __launch_bounds__(256) void k_reduce_v(const float *v_part, float *v, long long lo, long long hi,
                                                  long long len, int n_layers) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = lo + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < hi; i += stride) {
        float s = 0.f;
        for (int c = 0; c < n_layers; ++c) s += v_part[(long long)c * len + i];
        v[i] = s;
    }
}

// mode 0: sum a^2 ; 1: sum mask*(a-b)^2 ; 2: sum a
template <int KIND>
__global__ __launch_bounds__(256) void k_reduce1(const float *a, const float *b, const float *mask,
                                                 long long mask_n, long long n, double *part) {
    double acc = 0;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (KIND == 0) {
            acc += (double)a[i] * (double)a[i];
        } else if (KIND == 1) {
            const double d = (double)(a[i] - b[i]);
            const double w = mask ? (double)mask[i % mask_n] : 1.0;
            acc += w * d * d;
        } else {
            acc += (double)a[i];
        }
    }
    block_store3(acc, 0.0, 0.0, part + (long long)blockIdx.x * DFX_NPART);
}

// dst[k] = scale * sum_rows parts[row*DFX_NPART + k], k < nvals <= 3; one workgroup; fixed order.
// tol_dst != NULL (the Adam steps): *tol_dst = sqrt(r[0] / r[1]) = ||x - prev_x|| / ||x||, fusion.py:244-252,348.
__global__ __launch_bounds__(256) void k_fold(const double *parts, int n_rows, int nvals, double *dst,
                                              double scale, double *tol_dst) {
    double r[3];
    fold_cols<3>(parts, n_rows, r);     // rows always hold three values (block_store3): unused columns cost nothing
    if (threadIdx.x == 0) {
        for (int k = 0; k < nvals; ++k) dst[k] = scale * r[k];
        if (tol_dst) *tol_dst = sqrt(r[0] / r[1]);
    }
}

__device__ __forceinline__ void cg_alpha_body(double *s, double c1, double c2, double eps) {
    // float arithmetic where the reference has 0-dim fp32 tensors (fusion.py:444)
    const float pap = (float)(c1 * s[DFX_S_UU] + c2 * s[DFX_S_VV]);
    s[DFX_S_PAP] = (double)pap;
    const float alpha = (float)s[DFX_S_RSOLD] / (pap + (float)eps);
    s[DFX_S_ALPHA] = (double)alpha;
}

__device__ __forceinline__ void cg_beta_body(double *s, double eps, int deferred) {
    const double rsnew = s[DFX_S_RR];
    s[DFX_S_RSNEW] = rsnew;
    if (!isfinite(rsnew)) s[DFX_S_NONFINITE] += 1.0;
    const double al = s[DFX_S_ALPHA];
    const double pp = s[DFX_S_PP];
    const double xx_new = deferred ? (s[DFX_S_XX] + 2.0 * al * s[DFX_S_XP] + al * al * pp) : s[DFX_S_XXNEW];
    s[DFX_S_TOL] = fabs(al) * sqrt(pp) / sqrt(xx_new);
    const float beta = (float)rsnew / ((float)s[DFX_S_RSOLD] + (float)eps);
    s[DFX_S_BETA] = (double)beta;
    s[DFX_S_RSOLD] = rsnew;
    s[DFX_S_ALPHA_LAST] = al;      // the alpha of the iteration just closed (ALPHA itself is renewed after the sweep)
    s[DFX_S_ITER] += 1.0;
}

// Everything that follows a sweep (or a fused Adam step), in TWO launches:
//  k_post_sweep, many workgroups:
//   workgroups [0, nbu):        u = sum_rows u_part ; partial sum u^2                      (DFX_POST_U)
//   workgroups [nbu, nbu+nbv):  v = sum_zchunks v_part (if n_zchunks > 1) ; partial sum v^2  (DFX_POST_V)
//  k_post_fold, one workgroup (fixed order: deterministic):
//   UU, VV ; the sweep's / Adam step's own partial rows -> PP, XX, XP or ADAM_DD, ADAM_XX (DFX_POST_FOLD_*) ;
//   alpha of the next iteration (DFX_POST_ALPHA, fusion.py:444).
// (Folding inside k_post_sweep by "the last workgroup to finish" was measured and dropped: the device-scope release
// every workgroup needs before taking its ticket writes back the L2 the launch has just filled with u and v, and the
// tickets serialise at ~12 ns each: 53 us for the launch at C2, against 10 + 4 us for these two; a dependent kernel
// boundary costs 1.5 - 1.9 us on this GPU.)
struct PostArgs {
    const float *u_part;
    float *u;
    const float *v_part;
    float *v;          // NULL: v aliases v_part (one z-chunk), only ||v||^2 is taken
    double *scal;
    const double *fold_rows;   // the preceding launch's partial rows
    double *part;              // this launch's partial rows
    long long S, vlen;
    int n_urows, n_zchunks, nbu, nbv, n_fold_rows, flags;   // n_zchunks: layers of v_part (z-chunks x Geo::vq)
    double c1, c2, eps;
};

__global__ __launch_bounds__(256) void k_post_sweep(const PostArgs a) {
    double uu = 0, vv = 0;
    const int b = blockIdx.x;
    if (b < a.nbu) {
        if (a.flags & DFX_POST_U) {
            const long long stride = (long long)a.nbu * 256;
            const long long t0 = (long long)b * 256 + threadIdx.x;
            const bool vec = (a.S % 4 == 0) && ((((uintptr_t)a.u_part | (uintptr_t)a.u) & 15) == 0);
            if (vec) {
                for (long long i = t0; i < a.S / 4; i += stride) {
                    f4 sacc = ld4(a.u_part + 4 * i);
                    for (int c = 1; c < a.n_urows; ++c) sacc += ld4(a.u_part + (long long)c * a.S + 4 * i);
                    st4(a.u + 4 * i, sacc);
                    uu += ((double)sacc.x * sacc.x + (double)sacc.y * sacc.y) + ((double)sacc.z * sacc.z + (double)sacc.w * sacc.w);
                }
            } else {
                for (long long i = t0; i < a.S; i += stride) {
                    float sacc = 0.f;
                    for (int c = 0; c < a.n_urows; ++c) sacc += a.u_part[(long long)c * a.S + i];
                    a.u[i] = sacc;
                    uu += (double)sacc * (double)sacc;
                }
            }
        }
    } else if (a.flags & DFX_POST_V) {
        const long long stride = (long long)a.nbv * 256;
        const long long t0 = (long long)(b - a.nbu) * 256 + threadIdx.x;
        const bool vec = (a.vlen % 4 == 0) && ((((uintptr_t)a.v_part | (uintptr_t)a.v) & 15) == 0);
        if (vec) {
            for (long long i = t0; i < a.vlen / 4; i += stride) {
                f4 sacc = ld4(a.v_part + 4 * i);
                if (a.v) {
                    for (int c = 1; c < a.n_zchunks; ++c) sacc += ld4(a.v_part + (long long)c * a.vlen + 4 * i);
                    st4(a.v + 4 * i, sacc);
                }
                vv += ((double)sacc.x * sacc.x + (double)sacc.y * sacc.y) + ((double)sacc.z * sacc.z + (double)sacc.w * sacc.w);
            }
        } else {
            for (long long i = t0; i < a.vlen; i += stride) {
                float sacc = a.v_part[i];
                if (a.v) {
                    for (int c = 1; c < a.n_zchunks; ++c) sacc += a.v_part[(long long)c * a.vlen + i];
                    a.v[i] = sacc;
                }
                vv += (double)sacc * (double)sacc;
            }
        }
    }
    block_store3(uu, vv, 0.0, a.part + (long long)b * DFX_NPART);
}

__global__ __launch_bounds__(256) void k_post_fold(const PostArgs a) {
    double *s = a.scal;
    double own[2], rows[3];
    fold_cols<2>(a.part, a.nbu + a.nbv, own);
    if (a.flags & (DFX_POST_FOLD_SWEEP | DFX_POST_FOLD_ADAM)) fold_cols<3>(a.fold_rows, a.n_fold_rows, rows);
    if (threadIdx.x != 0) return;
    if (a.flags & DFX_POST_U) s[DFX_S_UU] = own[0];
    if (a.flags & DFX_POST_V) s[DFX_S_VV] = own[1];
    if (a.flags & DFX_POST_FOLD_SWEEP) {
        s[DFX_S_PP] = rows[0];       // PP, XX, XP
        s[DFX_S_XX] = rows[1];
        s[DFX_S_XP] = rows[2];
    }
    if (a.flags & DFX_POST_FOLD_ADAM) {
        s[DFX_S_ADAM_DD] = rows[0];
        s[DFX_S_ADAM_XX] = rows[1];
        s[DFX_S_TOL] = sqrt(rows[0] / rows[1]);   // fusion.py:244-252 as FusionAdam uses it (fusion.py:348); one shard's view
    }
    if (a.flags & DFX_POST_ALPHA) cg_alpha_body(s, a.c1, a.c2, a.eps);
}

// Everything a fused Adam iteration does in the small space BEFORE its 5-D pass, in one launch (fusion.py:332-338:
// loss(), and the part of loss.backward() that lives on the cmos- and spc-sized arrays):
//   A = -2 w1 mask1 (cmos - u) ; B = -2 w2 mask2 (spc - v)          d loss / d u, d loss / d v  (k_diff_scale)
//   m1a <- lerp(m1a, A, 1-b1) ; m1b <- lerp(m1b, B, 1-b1)           exp_avg's two factors       (k_lerp)
//   LOSS1 = w1 sum mask1 (cmos - u)^2 ; LOSS2 = w2 sum mask2 (spc - v)^2                        (k_reduce1<1>)
struct PrepArgs {
    const float *u, *cmos, *cmask;
    float *A, *m1a;
    long long n_u;
    const float *v, *spc, *smask;
    long long smask_n;
    float *B, *m1b;
    long long n_v;
    float ca, cb, omb1;
    double w1, w2;
    double *part, *scal;
    int nbu, nbv;
};

__global__ __launch_bounds__(256) void k_adam_prep(const PrepArgs a) {
    double l1 = 0, l2 = 0;
    const int b = blockIdx.x;
    if (b < a.nbu) {
        const long long stride = (long long)a.nbu * 256;
        for (long long i = (long long)b * 256 + threadIdx.x; i < a.n_u; i += stride) {
            const float d = a.cmos[i] - a.u[i];
            float g = a.ca * d;
            double w = 1.0;
            if (a.cmask) {
                const float mk = a.cmask[i];
                g *= mk;
                w = (double)mk;
            }
            l1 += w * (double)d * (double)d;
            a.A[i] = g;
            if (a.m1a) {
                const float m = a.m1a[i];
                a.m1a[i] = m + a.omb1 * (g - m);
            }
        }
    } else {
        const long long stride = (long long)a.nbv * 256;
        for (long long i = (long long)(b - a.nbu) * 256 + threadIdx.x; i < a.n_v; i += stride) {
            const float d = a.spc[i] - a.v[i];
            float g = a.cb * d;
            double w = 1.0;
            if (a.smask) {
                const float mk = a.smask[i % a.smask_n];   // the (x/r, y/r) mask repeats over (t, l)
                g *= mk;
                w = (double)mk;
            }
            l2 += w * (double)d * (double)d;
            a.B[i] = g;
            if (a.m1b) {
                const float m = a.m1b[i];
                a.m1b[i] = m + a.omb1 * (g - m);
            }
        }
    }
    block_store3(l1, l2, 0.0, a.part + (long long)b * DFX_NPART);
}

__global__ __launch_bounds__(256) void k_prep_fold(const double *part, int n_rows, double w1, double w2, double *scal) {
    double r[2];
    fold_cols<2>(part, n_rows, r);
    if (threadIdx.x == 0) {
        scal[DFX_S_LOSS1] = w1 * r[0];
        scal[DFX_S_LOSS2] = w2 * r[1];
    }
}

// ------------------------------------------------------------------------------------------
// Factored residual.  r0 = b - A x0 and every A p are sums of a (z,x,y) array broadcast over (t,l) and a
// (t,l,x/r,y/r) array broadcast over z and the r x r block (fusion.py:423-430), so the CG residual always has the
// form r = ra[z,x,y] + rb[t,l,x/r,y/r]: it is kept as those two small arrays and never stored as a 5-D vector.
//   ||r||^2 = M * sum ra^2 + 2 * sum_{t,l,xb,yb} rb * zab[xb,yb] + nZ r^2 * sum rb^2,   zab = sum_z sum_block ra
// ------------------------------------------------------------------------------------------
// ra update + sum ra^2 + zab in one pass.  Workgroup = 16 blocks x 16 z-lanes: thread (b, zz) walks planes
// zz, zz+16, ... of block b (r rows of r contiguous floats: neighbouring threads touch neighbouring segments).
__global__ __launch_bounds__(256) void k_resid_az(float *ra, const float *A, double *zab, Geo q, float sa,
                                                  float coef_host, const double *coef_dev, int accumulate,
                                                  double *part) {
    __shared__ double sh[16][17];
    const float c = coef_host * (coef_dev ? (float)(*coef_dev) : 1.f) * sa;
    const int b = threadIdx.x & 15, zz = threadIdx.x >> 4;
    const int groups = (int)((q.Ps + 15) / 16);
    double acc = 0;
    for (int gi = blockIdx.x; gi < groups; gi += gridDim.x) {
        const int j = gi * 16 + b;
        double bs = 0;   // fp64: ||r||^2 is a difference of the three sums when ra and rb partly cancel
        if (j < q.Ps) {
            const int xb = j / q.Ys, yb = j - xb * q.Ys;
            for (int z = zz; z < q.nZ; z += 16) {
                const long long o = (long long)z * q.P + (long long)(xb * q.r) * q.Y + yb * q.r;
                for (int dx = 0; dx < q.r; ++dx)
                    for (int dy = 0; dy < q.r; ++dy) {
                        const long long i = o + (long long)dx * q.Y + dy;
                        const float v = (accumulate ? ra[i] : 0.f) + c * A[i];
                        ra[i] = v;
                        acc += (double)v * (double)v;
                        bs += (double)v;
                    }
            }
        }
        sh[zz][b] = bs;
        __syncthreads();
        if (threadIdx.x < 16 && j < q.Ps) {
            double s = 0;
            for (int k = 0; k < 16; ++k) s += sh[k][threadIdx.x];
            zab[j] = s;
        }
        __syncthreads();
    }
    block_store3(acc, 0.0, 0.0, part + (long long)blockIdx.x * DFX_NPART);
}

// The same for large blocks (r >= 16: r^2 >= 256 elements per block and plane): a WORKGROUP per block, its threads
// over the block's nZ * r * r elements (row segments of r contiguous floats).  Thread-per-block above walks r^2 elements
// per plane serially and leaves one workgroup per 16 blocks: 0.58 ms at t256 x l16 x z16 x 128^2 with r = 32 (16 blocks).
__global__ __launch_bounds__(256) void k_resid_az_wide(float *ra, const float *A, double *zab, Geo q, float sa,
                                                       float coef_host, const double *coef_dev, int accumulate,
                                                       double *part) {
    __shared__ double shw[4];
    const float c = coef_host * (coef_dev ? (float)(*coef_dev) : 1.f) * sa;
    const int r = q.r, rr = r * r;
    const int per = q.nZ * rr;   // <= S < 2^31
    double acc = 0;
    for (int j = blockIdx.x; j < (int)q.Ps; j += gridDim.x) {
        const int xb = j / q.Ys, yb = j - xb * q.Ys;
        const long long o0 = (long long)(xb * r) * q.Y + yb * r;
        double bs = 0;
        for (int e = threadIdx.x; e < per; e += 256) {
            const int z = e / rr, w = e - z * rr, dx = w / r, dy = w - dx * r;
            const long long i = (long long)z * q.P + o0 + (long long)dx * q.Y + dy;
            const float v = (accumulate ? ra[i] : 0.f) + c * A[i];
            ra[i] = v;
            acc += (double)v * (double)v;
            bs += (double)v;
        }
        bs = wave_sum(bs);
        if ((threadIdx.x & 63) == 0) shw[threadIdx.x >> 6] = bs;
        __syncthreads();
        if (threadIdx.x == 0) zab[j] = (shw[0] + shw[1]) + (shw[2] + shw[3]);
        __syncthreads();
    }
    block_store3(acc, 0.0, 0.0, part + (long long)blockIdx.x * DFX_NPART);
}

// rb update + sum rb^2 + sum rb*zab: a workgroup takes whole rows (t,l) of rb, so the zab index needs no modulo
__global__ __launch_bounds__(256) void k_resid_b(float *rb, const float *B, const double *zab, int M, int Ps,
                                                 float sb, float coef_host, const double *coef_dev, int accumulate,
                                                 double *part) {
    const float c = coef_host * (coef_dev ? (float)(*coef_dev) : 1.f) * sb;
    double a2 = 0, ab = 0;
    const bool vec = (Ps % 4 == 0) && ((((uintptr_t)rb | (uintptr_t)B) & 15) == 0);
    for (int m = blockIdx.x; m < M; m += gridDim.x) {
        float *row = rb + (long long)m * Ps;
        const float *brow = B + (long long)m * Ps;
        if (vec) {
            for (int j = 4 * threadIdx.x; j < Ps; j += 4 * 256) {
                f4 v = c * ld4(brow + j);
                if (accumulate) v += ld4(row + j);
                st4(row + j, v);
                a2 += ((double)v.x * v.x + (double)v.y * v.y) + ((double)v.z * v.z + (double)v.w * v.w);
                ab += ((double)v.x * zab[j] + (double)v.y * zab[j + 1]) + ((double)v.z * zab[j + 2] + (double)v.w * zab[j + 3]);
            }
        } else {
            for (int j = threadIdx.x; j < Ps; j += 256) {
                const float v = (accumulate ? row[j] : 0.f) + c * brow[j];
                row[j] = v;
                a2 += (double)v * (double)v;
                ab += (double)v * zab[j];
            }
        }
    }
    block_store3(0.0, a2, ab, part + (long long)blockIdx.x * DFX_NPART);
}

// One workgroup: RR = M * sum ra^2 + 2 * sum rb zab + nZ r^2 * sum rb^2 from the rows the two kernels above wrote
// (torch.dot(r, r), fusion.py:437,447); with beta_mode >= 0 it also closes the iteration's scalar bookkeeping
// (k_cg_beta; beta_mode = the deferred flag).
__global__ __launch_bounds__(256) void k_resid_rr(const double *parts, int n_rows, double m, double zr2, double *s,
                                                  int beta_mode, double eps) {
    double r[3];
    fold_cols<3>(parts, n_rows, r);
    if (threadIdx.x == 0) {
        s[DFX_S_RA2] = r[0];
        s[DFX_S_RB2] = r[1];
        s[DFX_S_RAB] = r[2];
        s[DFX_S_RR] = m * r[0] + 2.0 * r[2] + zr2 * r[1];
        if (beta_mode >= 0) cg_beta_body(s, eps, beta_mode);
    }
}

__global__ void k_cg_start(double *s) {
    s[DFX_S_RSOLD] = s[DFX_S_RR];
    s[DFX_S_RSNEW] = s[DFX_S_RR];
    s[DFX_S_ALPHA] = 0.0;
    s[DFX_S_ALPHA_LAST] = 0.0;
    s[DFX_S_BETA] = 0.0;
    s[DFX_S_TOL] = 0.0;
    s[DFX_S_ITER] = 0.0;
    s[DFX_S_NONFINITE] = isfinite(s[DFX_S_RR]) ? 0.0 : 1.0;
}

__global__ void k_cg_alpha(double *s, double c1, double c2, double eps) { cg_alpha_body(s, c1, c2, eps); }

__global__ void k_cg_beta(double *s, double eps, int deferred) { cg_beta_body(s, eps, deferred); }

__global__ void k_recip(double *dst, const double *src, double num) { dst[0] = num / src[0]; }

// ------------------------------------------------------------------------------------------
// Element-wise helpers
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_axpy(float *y, const float *x, long long n, float coef_host,
                                              const double *coef_dev, double *part) {
    const float c = coef_host * (coef_dev ? (float)(*coef_dev) : 1.f);
    double acc = 0;
    const long long n4 = ((((uintptr_t)y | (uintptr_t)x) & 15) == 0) ? (n >> 2) : 0;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long t0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (long long i = t0; i < n4; i += stride) {
        const f4 v = ld4s(y + 4 * i) + c * ld4s(x + 4 * i);
        st4s(y + 4 * i, v);
        acc += (double)dot4(v, v);
    }
    for (long long i = 4 * n4 + t0; i < n; i += stride) {
        const float v = y[i] + c * x[i];
        y[i] = v;
        acc += (double)(v * v);
    }
    block_store3(acc, 0.0, 0.0, part + (long long)blockIdx.x * DFX_NPART);
}

__global__ __launch_bounds__(256) void k_scale(float *y, long long n, const float *mask, long long mask_n,
                                               float coef_host, const double *coef_dev) {
    const float c = coef_host * (coef_dev ? (float)(*coef_dev) : 1.f);
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float v = y[i];
        // fusion.py:198 assigns 0 (not multiplies): NaN/Inf under the mask must become 0 too
        if (mask && mask[i % mask_n] == 0.f) v = 0.f;
        y[i] = v * c;
    }
}

// out = coef * (a - b) * mask: the small-space residuals cmos_n - u(x), spc_n - v(x) behind r0 (fusion.py:428-435),
// the loss gradient (autograd of fusion.py:216-232) and its masked variants
__global__ __launch_bounds__(256) void k_diff_scale(float *out, const float *a, const float *b, const float *mask,
                                                    long long mask_n, long long n, float coef) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float v = coef * (a[i] - b[i]);
        if (mask) v *= mask[i % mask_n];   // a (x/r, y/r) mask repeats over (t, l)
        out[i] = v;
    }
}

// m += w * (g - m): torch's exp_avg.lerp_(grad, w) on the small-space factors of the first Adam moment
__global__ __launch_bounds__(256) void k_lerp(float *m, const float *g, long long n, float w) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float v = m[i];
        m[i] = v + w * (g[i] - v);
    }
}

// out[a][b] = sum_{k < R} x[(a*R + k)*B + b]: the marginals a caller takes right after a solve
// (x.sum(axis=0) -> A=1,R=nT ; x.sum(axis=1) -> A=nT,R=nL ; cells.ipynb cell 27bf9830e81df430), one read of x.
__global__ __launch_bounds__(256) void k_sum_axis(const float *x, float *out, long long A, long long R, long long B) {
    const long long B4 = ((B & 3) == 0 && ((((uintptr_t)x | (uintptr_t)out) & 15) == 0)) ? (B >> 2) : 0;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long t0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (B4) {
        for (long long j = t0; j < A * B4; j += stride) {
            const long long a = j / B4, b = (j - a * B4) * 4;
            const float *src = x + a * R * B + b;
            f4 acc = {0.f, 0.f, 0.f, 0.f};
            for (long long k = 0; k < R; ++k) acc += ld4s(src + k * B);
            st4(out + a * B + b, acc);
        }
    } else {
        for (long long j = t0; j < A * B; j += stride) {
            const long long a = j / B, b = j - a * B;
            const float *src = x + a * R * B + b;
            float acc = 0.f;
            for (long long k = 0; k < R; ++k) acc += src[k * B];
            out[a * B + b] = acc;
        }
    }
}

// Stand-alone operator kernels behind the SumOperator / SumPoolOperator API mirrors (fusion.py:21-98).  The solvers
// never launch these (their work is fused into k_sweep_fast / k_bcast_fast); they exist so that T, S, D, R and
// their normalised transposes are callable on device tensors exactly like the reference's objects.
// out[(a*R + k)*B + b] = scale * y[a*B + b]   (repeat_interleave(R, dim) / size, fusion.py:42)
__global__ __launch_bounds__(256) void k_expand_axis(const float *y, float *out, long long A, long long R, long long B,
                                                     float scale) {
    const long long n = A * R * B, stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const long long b = i % B, a = i / (R * B);
        out[i] = scale * y[a * B + b];
    }
}

// out[row][xb][yb] = sum_{r x r} in[row][xb*r + dx][yb*r + dy]   (depthwise ones conv, stride r, fusion.py:95-97)
__global__ __launch_bounds__(256) void k_block_sum(const float *in, float *out, long long rows, int X, int Y, int r) {
    const int Xs = X / r, Ys = Y / r;
    const long long n = rows * Xs * Ys, stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int yb = (int)(i % Ys), xb = (int)((i / Ys) % Xs);
        const long long row = i / ((long long)Xs * Ys);
        const float *src = in + row * X * Y + (long long)xb * r * Y + (long long)yb * r;
        float s = 0.f;
        for (int dx = 0; dx < r; ++dx)
            for (int dy = 0; dy < r; ++dy) s += src[(long long)dx * Y + dy];
        out[i] = s;
    }
}

// out[row][x][y] = scale * in[row][x/r][y/r]   (conv_transpose2d with the ones kernel / r^2, fusion.py:84-87)
__global__ __launch_bounds__(256) void k_block_expand(const float *in, float *out, long long rows, int X, int Y, int r,
                                                      float scale) {
    const int Xs = X / r, Ys = Y / r;
    const long long n = rows * X * Y, stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int y = (int)(i % Y), x = (int)((i / Y) % X);
        const long long row = i / ((long long)X * Y);
        out[i] = scale * in[row * Xs * Ys + (long long)(x / r) * Ys + y / r];
    }
}

// ------------------------------------------------------------------------------------------
// Input stage (utils.py:411-427 load_data): skimage.transform.resize(cmos, (S1, S2, nZ)) with its defaults, restated
// per (x, y) plane of a float64 stack: Gaussian pre-filter (scipy.ndimage.gaussian_filter: separable, truncate 4,
// mode "mirror" = whole-sample symmetric, which is what skimage translates its default mode "reflect" to) on the axes
// that shrink, then order-1 interpolation at the centres of the output grid
// (scipy.ndimage.zoom(order=1, mode="mirror", grid_mode=True)).  PARITY UNPINNED against scikit-image
// itself (absent from the image); checked against the scipy calls it is built from.
// ------------------------------------------------------------------------------------------
#define DFX_RESIZE_MAXR 64
struct GaussArgs {
    const double *in;
    double *out;
    long long planes;
    int X, Y, axis, radius;      // axis 0: along x (stride Y), 1: along y (stride 1)
    double w[DFX_RESIZE_MAXR + 1];   // w[k], k = 0..radius (symmetric kernel, normalised)
};

// skimage's resize hands its default mode "reflect" (numpy.pad's name) to scipy.ndimage as "mirror"
// (skimage/_shared/utils.py: _to_ndimage_mode): whole-sample symmetric, the edge sample is NOT repeated.
__device__ __forceinline__ int reflect_idx(int i, int n) {   // d c b | a b c d | c b a
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
    return i;
}

__global__ __launch_bounds__(256) void k_gauss_axis(const GaussArgs a) {
    const long long P = (long long)a.X * a.Y, n = a.planes * P;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const long long pl = i / P;
        const int xy = (int)(i - pl * P), x = xy / a.Y, y = xy - x * a.Y;
        const double *base = a.in + pl * P;
        const int len = a.axis == 0 ? a.X : a.Y, pos = a.axis == 0 ? x : y;
        // scipy's correlate1d for a symmetric kernel: centre tap, then the pairs from the outermost inwards
        double t = base[xy] * a.w[0];
        for (int k = a.radius; k >= 1; --k) {
            const int lo = reflect_idx(pos - k, len), hi = reflect_idx(pos + k, len);
            const double vlo = a.axis == 0 ? base[(long long)lo * a.Y + y] : base[(long long)x * a.Y + lo];
            const double vhi = a.axis == 0 ? base[(long long)hi * a.Y + y] : base[(long long)x * a.Y + hi];
            t += (vlo + vhi) * a.w[k];
        }
        a.out[i] = t;
    }
}

// out[pl][i][j] (or out[pl][j][i] when transpose) = bilinear sample of in[pl] at ((i + 0.5) X / S1 - 0.5, (j + 0.5) Y / S2 - 0.5)
__global__ __launch_bounds__(256) void k_zoom_linear(const double *in, double *out, long long planes, int X, int Y,
                                                     int S1, int S2, int transpose) {
    const long long Po = (long long)S1 * S2, n = planes * Po;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const double fx = (double)X / (double)S1, fy = (double)Y / (double)S2;
    for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < n; o += stride) {
        const long long pl = o / Po;
        const int ij = (int)(o - pl * Po), i = ij / S2, j = ij - i * S2;
        const double cx = ((double)i + 0.5) * fx - 0.5, cy = ((double)j + 0.5) * fy - 0.5;
        const double flx = floor(cx), fly = floor(cy);
        const double wx = cx - flx, wy = cy - fly;
        const int x0 = reflect_idx((int)flx, X), x1 = reflect_idx((int)flx + 1, X);
        const int y0 = reflect_idx((int)fly, Y), y1 = reflect_idx((int)fly + 1, Y);
        const double *b = in + pl * (long long)X * Y;
        const double v = (1.0 - wx) * ((1.0 - wy) * b[(long long)x0 * Y + y0] + wy * b[(long long)x0 * Y + y1]) +
                         wx * ((1.0 - wy) * b[(long long)x1 * Y + y0] + wy * b[(long long)x1 * Y + y1]);
        out[transpose ? pl * Po + (long long)j * S1 + i : o] = v;
    }
}

struct BaseArgs {
    const float *spc;
    const float *zw;
    float *x;
    const double *coef_dev;
    double *part;
    float coef_host;
    Geo q;
    int vec;  // 4 if Y % 4 == 0 else 1
};

// bilinear source index/weight, align_corners=False, scale 1/r (ATen area_pixel_compute_source_index)
__device__ __forceinline__ void src_idx(int i, float inv_r, int n_in, int &i0, int &i1, float &w1) {
    float s = inv_r * ((float)i + 0.5f) - 0.5f;
    s = s < 0.f ? 0.f : s;
    i0 = (int)s;
    if (i0 > n_in - 1) i0 = n_in - 1;
    i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
    w1 = s - (float)i0;
}

__global__ __launch_bounds__(256) void k_baseline(BaseArgs a) {
    const Geo &q = a.q;
    const float coef = a.coef_host * (a.coef_dev ? (float)(*a.coef_dev) : 1.f);
    const float inv_r = 1.0f / (float)q.r;
    const int yq_n = q.Y / a.vec;
    const long long items = (long long)q.M * q.X * yq_n;
    const long long stride = (long long)gridDim.x * blockDim.x;
    double acc = 0;
    for (long long it = (long long)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += stride) {
        const int yq = (int)(it % yq_n);
        const long long t2 = it / yq_n;
        const int xi = (int)(t2 % q.X);
        const long long m = t2 / q.X;
        int x0, x1;
        float wx;
        src_idx(xi, inv_r, q.Xs, x0, x1, wx);
        const float *s0 = a.spc + m * q.Ps + (long long)x0 * q.Ys;
        const float *s1 = a.spc + m * q.Ps + (long long)x1 * q.Ys;
        float up[4];
        for (int e = 0; e < a.vec; ++e) {
            int y0, y1;
            float wy;
            src_idx(yq * a.vec + e, inv_r, q.Ys, y0, y1, wy);
            const float top = (1.f - wy) * s0[y0] + wy * s0[y1];
            const float bot = (1.f - wy) * s1[y0] + wy * s1[y1];
            up[e] = (1.f - wx) * top + wx * bot;
        }
        const long long sp = (long long)xi * q.Y + (long long)yq * a.vec;
        float fsum = 0.f;
        for (int z = 0; z < q.nZ; ++z) {
            const long long zo = (long long)z * q.P + sp;
            if (a.vec == 4) {
                const f4 w = ld4(a.zw + zo);
                const f4 v = {coef * (up[0] * w.x), coef * (up[1] * w.y), coef * (up[2] * w.z),
                              coef * (up[3] * w.w)};
                if (a.x) st4s(a.x + m * q.S + zo, v);
                fsum += hsum(v);
            } else {
                const float v = coef * (up[0] * a.zw[zo]);
                if (a.x) a.x[m * q.S + zo] = v;
                fsum += v;
            }
        }
        acc += (double)fsum;
    }
    block_store3(acc, 0.0, 0.0, a.part + (long long)blockIdx.x * DFX_NPART);
}

int ew_blocks(long long n, int per_thread) {
    long long b = (n + 256LL * per_thread - 1) / (256LL * per_thread);
    if (b < 1) b = 1;
    if (b > DFX_EW_MAXBLOCKS) b = DFX_EW_MAXBLOCKS;
    return (int)b;
}

// workgroups along the slab positions of the generic sweep: one per tile of 256 * DFX_GEN_K positions, at most 2048
// (a workgroup then walks several tiles)
// k_gen_proj_v2: one wave per (slab, block row, column chunk) item, four waves per workgroup
int gen_projv_blocks(const Geo &q) {
    const int C = q.r <= 256 ? q.r * (256 / q.r) : q.r;
    const long long items = (long long)q.M * q.Xs * ((q.Y + C - 1) / C);
    long long b = (items + 3) / 4;
    if (b < 1) b = 1;
    if (b > 8192) b = 8192;
    return (int)b;
}
bool gen_vec4(const Geo &q) { return (q.S % 4) == 0; }   // float4 lanes over the flat volume (k_gen_sweep4)
int gen_tiles(const Geo &q) {
    const long long per = gen_vec4(q) ? 4LL * 256 * DFX_GEN_K4 : 256LL * DFX_GEN_K;
    return (int)((q.S + per - 1) / per);
}
int gen_tile_slots(const Geo &q) {
    const int t = gen_tiles(q);
    return t < 2048 ? t : 2048;
}

// v of the generic path: the coalesced form for every block size a wave's LDS row can hold, the gather form beyond
void launch_gen_proj_v(const float *src, float *v_part, const Geo &q, hipStream_t st) {
    if (q.r <= 256)
        hipLaunchKernelGGL(k_gen_proj_v2, dim3(gen_projv_blocks(q)), dim3(256), 0, st, src, v_part, q);
    else
        hipLaunchKernelGGL(k_gen_proj_v, dim3(ew_blocks((long long)q.M * q.Ps, 1)), dim3(256), 0, st, src, v_part, q);
}

int check_plan(const Geo &q, const dfx_plan *pl) {
    if (!pl) return fail(DFX_EPLAN, "plan is NULL");
    if (pl->steps <= 0 || pl->n_chunks <= 0 || pl->n_blocks <= 0 || pl->zc <= 0 || pl->n_zchunks <= 0)
        return fail(DFX_EPLAN, "plan not initialised (call dfx_plan_make)");
    if (pl->fast) {
        if (!fast_ok(q)) return fail(DFX_EPLAN, "fast plan on a geometry that needs the generic path");
        if (pl->zc != 4 && pl->zc != 8) return fail(DFX_EPLAN, "fast plan: zc must be 4 or 8");
        if (pl->cpb <= 0 || pl->n_chunks % pl->cpb) return fail(DFX_EPLAN, "plan: n_chunks must be a multiple of cpb");
        if (pl->zs <= 0 || pl->zs > pl->zc) return fail(DFX_EPLAN, "fast plan: need 0 < zs <= zc");
        if ((long long)pl->zs * pl->n_zchunks < q.nZ || (long long)pl->steps * pl->n_chunks < q.M)
            return fail(DFX_EPLAN, "plan does not cover the geometry");
        if (pl->n_zchunks > 65535) return fail(DFX_EPLAN, "plan grid too large");
        const long long gz = (pl->n_chunks + pl->cpb - 1) / pl->cpb;
        if (gz > 65535) return fail(DFX_EPLAN, "plan grid too large");
        const long long nb = (long long)((q.WT + DFX_WAVES - 1) / DFX_WAVES) * pl->n_zchunks * gz;
        if (nb != pl->n_blocks) return fail(DFX_EPLAN, "plan n_blocks inconsistent");
    } else {
        if (pl->n_zchunks != 1 || pl->cpb != 1) return fail(DFX_EPLAN, "generic plan must have n_zchunks = cpb = 1");
        if (pl->n_chunks > q.M || pl->n_chunks > 4096 || (long long)pl->steps * pl->n_chunks < q.M)
            return fail(DFX_EPLAN, "generic plan does not cover the geometry");
        if ((long long)gen_tile_slots(q) * pl->n_chunks != pl->n_blocks) return fail(DFX_EPLAN, "plan n_blocks inconsistent");
    }
    return 0;
}

bool aligned16(const void *p) { return (((uintptr_t)p) & 15) == 0; }
bool aligned8(const void *p) { return (((uintptr_t)p) & 7) == 0; }   // small-space rows: float2 access when r = 2

// One workgroup per CU for the lane-mapped streaming kernels.  They stream fastest at ONE 4-wave workgroup per CU (see
// dfx_plan_make_ex); a plane of 512 x 512 and more has more (tile group, z-chunk) pairs than the chip has CUs, and the
// dispatcher would then run up to four workgroups per CU side by side (four concurrent windows per vector).  A dynamic
// LDS request of more than half a CU's 160 KiB (never touched by the kernels) makes the hardware admit one workgroup
// per CU, the rest following as CUs drain: config 4 whole on one GPU 90.1 -> 88.4 ms, t64 x l16 x z16 x 512^2
// 11.29 -> 10.94 ms, t16 x l4 x z16 x 512^2 0.793 -> 0.731 ms; launches of at most one workgroup per CU are unchanged.
// DFX_SOLO_LDS=<bytes> overrides (0: off).
static int solo_lds_bytes() {
    static const int v = [] {
        const char *e = getenv("DFX_SOLO_LDS");
        int want = e ? atoi(e) : 96 * 1024;
        if (want < 0) want = 0;
        return want;
    }();
    return v;
}
#define DFX_LAUNCH_SOLO(KFN, grid, block, st, ...)                                                                   \
    do {                                                                                                             \
        auto kfn_ = KFN;                                                                                             \
        static unsigned long long ok_ = 0, bad_ = 0;   /* per kernel instance: devices (bit = id) with the attribute set */ \
        int dyn_ = solo_lds_bytes();                                                                                 \
        if (dyn_ > 0) {                                                                                              \
            int dev_ = 0;                                                                                            \
            (void)hipGetDevice(&dev_);                                                                               \
            const unsigned long long bit_ = 1ull << (dev_ & 63);                                                     \
            if (!((ok_ | bad_) & bit_)) {                                                                            \
                int cap_ = 0;                                                                                        \
                if (hipDeviceGetAttribute(&cap_, hipDeviceAttributeMaxSharedMemoryPerBlock, dev_) == hipSuccess &&   \
                    cap_ >= dyn_ + 16 * 1024 &&                                                                      \
                    hipFuncSetAttribute((const void *)kfn_, hipFuncAttributeMaxDynamicSharedMemorySize, dyn_) ==     \
                        hipSuccess)                                                                                  \
                    ok_ |= bit_;                                                                                     \
                else                                                                                                 \
                    bad_ |= bit_, (void)hipGetLastError();                                                           \
            }                                                                                                        \
            if (bad_ & bit_) dyn_ = 0;                                                                               \
        }                                                                                                            \
        hipLaunchKernelGGL(kfn_, grid, block, dyn_, st, __VA_ARGS__);                                                \
    } while (0)

// the fast kernels exist for zc in {4, 8} x lane map in {2, 4, 8, 16 (row-part tile), 1, 3 (flat maps)}; check_plan / fast_ok admit nothing else
#define DFX_DISPATCH_R(ZCV, lm, ...)                                       \
    do {                                                                   \
        constexpr int ZC_ = ZCV;                                           \
        if ((lm) == 2) { constexpr int R_ = 2; __VA_ARGS__; }              \
        else if ((lm) == 4) { constexpr int R_ = 4; __VA_ARGS__; }         \
        else if ((lm) == 8) { constexpr int R_ = 8; __VA_ARGS__; }         \
        else if ((lm) == 16) { constexpr int R_ = 16; __VA_ARGS__; }       \
        else if ((lm) == 1) { constexpr int R_ = 1; __VA_ARGS__; }         \
        else { constexpr int R_ = 3; __VA_ARGS__; }                        \
    } while (0)
#define DFX_DISPATCH_ZC_R(zc, r, ...)                                      \
    do {                                                                   \
        if ((zc) == 4) DFX_DISPATCH_R(4, r, __VA_ARGS__);                  \
        else DFX_DISPATCH_R(8, r, __VA_ARGS__);                            \
    } while (0)

template <int MODE>
int launch_sweep_fast(const SweepArgs &a, const dfx_plan *pl, hipStream_t st) {
    dim3 grid((a.q.WT + DFX_WAVES - 1) / DFX_WAVES, pl->n_zchunks, (pl->n_chunks + pl->cpb - 1) / pl->cpb), block(64 * DFX_WAVES);
    if (MODE != DFX_SWEEP_PROJECT && a.ra)
        DFX_DISPATCH_ZC_R(pl->zc, lane_map_of(a.q), DFX_LAUNCH_SOLO((k_sweep_fast<MODE, ZC_, R_, MODE != DFX_SWEEP_PROJECT>), grid, block, st, a));
    else
        DFX_DISPATCH_ZC_R(pl->zc, lane_map_of(a.q), DFX_LAUNCH_SOLO((k_sweep_fast<MODE, ZC_, R_, false>), grid, block, st, a));
    DFX_LAUNCH_CHECK("dfx_sweep(fast)");
    return 0;
}

template <bool HAS_IN, bool HAS_MASK>
int launch_bcast_fast(const BcastArgs &a, const dfx_plan *pl, hipStream_t st) {
    dim3 grid((a.q.WT + DFX_WAVES - 1) / DFX_WAVES, pl->n_zchunks, (pl->n_chunks + pl->cpb - 1) / pl->cpb), block(64 * DFX_WAVES);
    DFX_DISPATCH_ZC_R(pl->zc, lane_map_of(a.q), DFX_LAUNCH_SOLO((k_bcast_fast<ZC_, R_, HAS_IN, HAS_MASK>), grid, block, st, a));
    DFX_LAUNCH_CHECK("dfx_bcast(fast)");
    return 0;
}

// ------------------------------------------------------------------------------------------
// Scratch of dfx_resid_update_noalloc: the zab workspace (DFX_NOALLOC_MAX_PS doubles) followed by the rows of
// per-workgroup partials (DFX_EW_MAXBLOCKS * DFX_NPART doubles), 576 KiB, held by every GPU the library runs on until
// the process ends.  In host code the array's NAME is a host-side stand-in, not a device address: host code only
// ever uses the per-device pointer hipGetSymbolAddress returned (noalloc_scratch()), never the name.
// ------------------------------------------------------------------------------------------
static __device__ __align__(256) double g_noalloc_scratch[DFX_NOALLOC_MAX_PS + DFX_EW_MAXBLOCKS * DFX_NPART];

constexpr int kMaxDevices = 64;
std::atomic<double *> g_noalloc_ptr[kMaxDevices];   // device address of g_noalloc_scratch on each GPU (each has its own)

// The current device's copy of the scratch array; `stream` (if not the NULL stream) must belong to that device.
int noalloc_scratch(hipStream_t st, double **out) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return hipfail(e, "dfx_resid_update_noalloc: hipGetDevice");
    if (dev < 0 || dev >= kMaxDevices) return fail(DFX_EINVAL, "dfx_resid_update_noalloc: device ordinal out of range");
    if (st) {
        int sdev = -1;
        e = hipStreamGetDevice(st, &sdev);
        if (e != hipSuccess) return hipfail(e, "dfx_resid_update_noalloc: hipStreamGetDevice");
        if (sdev != dev)
            return fail(DFX_EINVAL, "dfx_resid_update_noalloc: the stream's device is not the current device");
    }
    double *p = g_noalloc_ptr[dev].load(std::memory_order_acquire);
    if (!p) {   // once per device, with that device current
        void *sym = nullptr;
        e = hipGetSymbolAddress(&sym, HIP_SYMBOL(g_noalloc_scratch));
        if (e != hipSuccess) return hipfail(e, "dfx_resid_update_noalloc: hipGetSymbolAddress");
        if (!sym) return fail(DFX_EINVAL, "dfx_resid_update_noalloc: hipGetSymbolAddress returned NULL");
        p = static_cast<double *>(sym);
        g_noalloc_ptr[dev].store(p, std::memory_order_release);
    }
    *out = p;
    return 0;
}

int fold(const double *parts, int n_rows, int nvals, double *dst, double scale, hipStream_t st,
         double *tol_dst = nullptr) {
    hipLaunchKernelGGL(k_fold, dim3(1), dim3(256), 0, st, parts, n_rows, nvals, dst, scale, tol_dst);
    DFX_LAUNCH_CHECK("fold");
    return 0;
}

}  // namespace

// ==========================================================================================
// C ABI
// ==========================================================================================
extern "C" {

int dfx_version(void) { return DFX_VERSION; }
const char *dfx_last_error(void) { return g_err; }

int dfx_plan_make(const dfx_geom *g, int32_t target_blocks, dfx_plan *out) {
    return dfx_plan_make_ex(g, target_blocks, 0, 0, out);
}

int dfx_plan_make_ex(const dfx_geom *g, int32_t target_blocks, int32_t zc_req, int32_t steps_req, dfx_plan *out) {
    Geo q;
    if (int rc = make_geo(g, &q)) return rc;
    if (!out) return fail(DFX_EINVAL, "plan out is NULL");
    memset(out, 0, sizeof(*out));
    if (fast_ok(q)) {
        out->fast = 1;
        int zc;
        // measured on MI355X at t256 x l16 x z16 x 128^2 (tools/tune_kernels.py): zc = 8 beats 16 (which
        // needs 256 VGPRs + AGPRs, and spills in the Adam kernel) by 5 % and 4 by 3 %: 16 is not built
        // z-chunks are balanced: n = ceil(nZ / ZC) chunks of zs = ceil(nZ / n) planes each, so every workgroup streams
        // (almost) the same bytes; the inner loops are compiled for every live-plane count 1..ZC, so a chunk of
        // zs < ZC planes costs only its unused registers.  Pick the ZC that pads nZ least, with a small preference
        // for full register tiles; 8 on a tie.  nZ = 16 -> 8 + 8; 20 -> 5 x 4; 5 -> 5; 10 -> 5 + 5; 21 -> 7 + 7 + 7.
        int zs;
        {
            double best = 1e30;
            zc = 8, zs = 8;
            for (int cand = 8; cand >= 4; cand -= 4) {
                const int n = (q.nZ + cand - 1) / cand, per = (q.nZ + n - 1) / n;
                const double cost = (double)n * per / q.nZ * (1.0 + 0.05 * (cand - per) / cand);
                if (cost < best - 1e-9) best = cost, zc = cand, zs = per;
            }
        }
        if (zc_req) {
            if (zc_req != 4 && zc_req != 8) return fail(DFX_EINVAL, "zc must be 0 (auto), 4 or 8");
            zc = zc_req;
            const int n = (q.nZ + zc - 1) / zc;
            zs = (q.nZ + n - 1) / n;
        }
        // Grid sizing.  Measured on MI355X (tools/tune_kernels.py): these kernels stream fastest with ONE
        // 4-wave workgroup per CU walking a long run of slabs (C2: 6.0 TB/s at 256 workgroups vs 5.5 at 2048:
        // fewer concurrent DRAM streams), so the default target is the CU count.  `steps` only sets how often
        // the fp32 u accumulators are flushed to u_part (accuracy + 2N/steps words of extra traffic).
        long long steps = steps_req > 0 ? steps_req : 64;
        if (steps > q.M) steps = q.M;
        long long n_chunks = (q.M + steps - 1) / steps;
        if (target_blocks <= 0) {
            int dev = 0, cus = 0;
            if (hipGetDevice(&dev) != hipSuccess ||
                hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
                cus = 256;
            (void)hipGetLastError();
            target_blocks = cus;
        }
        const long long tile_groups = (q.WT + DFX_WAVES - 1) / DFX_WAVES;
        // workgroups along the slab axis for `nzc` z-chunks: fill whole "rounds" of one workgroup per CU (80 tile groups
        // on 256 CUs: gz = 3 or 6 -> 240 / 480 workgroups, not 4 -> 320, which leaves a quarter-full second round)
        auto pick_gz = [&](long long nzc, double *score_out) {
            const long long groups = tile_groups * nzc;
            long long gz = 1;
            double best = -1.0;
            const long long gz_hi = (4LL * target_blocks + groups - 1) / groups;
            for (long long c = 1; c <= gz_hi && c <= n_chunks && c <= 65535; ++c) {
                const long long nb = groups * c, rounds = (nb + target_blocks - 1) / target_blocks;
                // a second or third round of workgroups costs more than its fill suggests: 512 workgroups ran 5.7 %
                // slower than 256 at C2 (more concurrent DRAM streams)
                const double score = (double)nb / (double)(rounds * target_blocks) - 0.06 * (double)(rounds - 1);
                if (score > best + 1e-9) best = score, gz = c;
            }
            if (score_out) *score_out = best;
            return gz;
        };
        double fill = 0.0;
        long long gz = pick_gz((q.nZ + zs - 1) / zs, &fill);
        // Planes of at most 16 tile groups (128 x 128 and smaller) on the row maps stream faster in z-chunks of TWO planes
        // whenever those still fill the chip in one round: more z-chunks mean fewer slab lanes (gz 8 -> 2 at C2), i.e. a
        // tighter window of concurrently streamed slabs.  Measured (tools/exp_zs.sh, sweep of the factored iteration,
        // 8 -> 2 planes): t256 x l16 x z16 x 128^2 2.696 -> 2.629 ms (r = 8: 2.719 -> 2.661, r = 16: 2.701 -> 2.630),
        // t64 x l16 x z16 x 128^2 0.717 -> 0.671, z8 1.362 -> 1.311, t1024 x l16 x z16 x 64^2 3.323 -> 3.052, z32 x 128^2
        // 5.348 -> 5.362; the fused Adam step 2.703 -> 2.644.  NOT for larger planes (256^2 x z16 at t512 x l32: 43.2 ->
        // 45.3 ms with 4 planes), r = 2 (2.727 -> 2.743) or the flat maps (120^2: 3.005 -> 3.865 ms).
        if (!zc_req && (q.lm == 4 || q.lm == 8 || q.lm == 16) && tile_groups <= 16 && zs > 2 && q.nZ % 2 == 0 && fill >= 0.99) {
            double f2 = 0.0;
            const long long gz2 = pick_gz(q.nZ / 2, &f2);
            if (f2 >= 0.99 && tile_groups * (q.nZ / 2) * gz2 <= target_blocks) zc = 4, zs = 2, gz = gz2, fill = f2;
        }
        // Filling idle CUs with another number of z-chunks does not pay: t64 x l16 x z20 x 128^2 as 16 tile groups x 4
        // chunks of 5 planes = 256 workgroups ran 0.939 ms against 0.876 ms as 5 chunks of 4 planes = 240 workgroups
        // (DFX_PLAN_ZS=5, a measurement knob: planes per z-chunk).
        static const int zs_env = [] { const char *e = getenv("DFX_PLAN_ZS"); return e ? atoi(e) : 0; }();
        if (!zc_req && zs_env >= 1 && zs_env <= 8) {
            zs = zs_env < q.nZ ? zs_env : q.nZ;
            zc = zs <= 4 ? 4 : 8;
            gz = pick_gz((q.nZ + zs - 1) / zs, &fill);
        }
        out->zc = zc;
        out->zs = zs;
        out->n_zchunks = (q.nZ + zs - 1) / zs;
        const long long groups = tile_groups * out->n_zchunks;
        long long cpb = (n_chunks + gz - 1) / gz;
        gz = (n_chunks + cpb - 1) / cpb;
        n_chunks = cpb * gz;   // with interleaving the last chunks may cover fewer (or no) slabs
        out->steps = (int)steps;
        out->n_chunks = (int)n_chunks;
        out->cpb = (int)cpb;
        const long long nb = groups * gz;
        if (nb > 0x7fffffffLL) return fail(DFX_EINVAL, "too many workgroups");
        out->n_blocks = (int)nb;
    } else {
        // generic kernels (k_gen_sweep2): grid = (tile slots, gz) with a tile = 256 * DFX_GEN_K slab positions; gz
        // workgroups along the slab axis (one u_part row each) so that about two workgroups per CU are in flight
        out->fast = 0;
        out->zc = 1;
        out->zs = 1;
        out->n_zchunks = 1;
        int dev = 0, cus = 0;
        if (target_blocks <= 0) {
            if (hipGetDevice(&dev) != hipSuccess ||
                hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
                cus = 256;
            (void)hipGetLastError();
            // float4 lanes: two workgroups per CU; scalar lanes move a quarter of the bytes per instruction and want
            // many more waves in flight (C2-sized volume, 125 x 125 planes: 8.1 ms at 2 per CU, 6.5 ms at 32)
            target_blocks = (gen_vec4(q) ? 2 : 32) * cus;
        }
        const int slots = gen_tile_slots(q);
        long long gz = (target_blocks + slots - 1) / slots;
        if (gz > q.M) gz = q.M;
        if (gz > 4096) gz = 4096;
        if (gz < 1) gz = 1;
        out->steps = (int)((q.M + gz - 1) / gz);
        gz = (q.M + out->steps - 1) / out->steps;      // no workgroup row without a slab
        out->n_chunks = (int)gz;
        out->cpb = 1;
        out->n_blocks = (int)(slots * gz);
    }
    out->u_part_elems = (long long)(out->n_chunks / out->cpb) * q.S;   // one row per workgroup of the slab axis
    out->v_part_elems = (long long)out->n_zchunks * q.vq * q.M * q.Ps;   // layers: z-chunks x row parts (LaneMap<16>)
    // rows of the sweep's slices, then DFX_EW_MAXBLOCKS rows for dfx_post_sweep's own partials (which must not
    // overlap the rows it folds); the element-wise entry points use rows [0, DFX_EW_MAXBLOCKS)
    long long rows = (long long)out->n_blocks * DFX_MAX_SLICES + DFX_EW_MAXBLOCKS;
    out->part_elems = rows * DFX_NPART;
    return 0;
}

int dfx_sweep(int32_t mode, const dfx_geom *g, const dfx_plan *pl, float *x, const float *r, float *p,
              double *scal, float *u_part, float *v_part, double *part, void *stream) {
    if (!pl) return fail(DFX_EPLAN, "plan is NULL");
    return dfx_sweep_chunks(mode, g, pl, x, r, p, scal, u_part, v_part, part, 0, pl->cpb, 0, 1, stream);
}

static int sweep_impl(int32_t mode, const dfx_geom *g, const dfx_plan *pl, float *x, const float *r, const float *ra,
                      const float *rb, float *p, double *scal, float *u_part, float *v_part, double *part,
                      int32_t c_lo, int32_t c_hi, int32_t slice, int32_t last, void *stream) {
    Geo q;
    if (int rc = make_geo(g, &q)) return rc;
    if (int rc = check_plan(q, pl)) return rc;
    q.zs = pl->fast ? pl->zs : 1;
    if (c_lo < 0 || c_hi > pl->cpb || c_lo >= c_hi || slice < 0 || slice >= DFX_MAX_SLICES)
        return fail(DFX_EINVAL, "dfx_sweep_chunks: need 0 <= c_lo < c_hi <= cpb and 0 <= slice < DFX_MAX_SLICES");
    if (!pl->fast && (c_lo != 0 || c_hi != pl->cpb || slice != 0 || !last))
        return fail(DFX_EINVAL, "dfx_sweep_chunks: the generic path sweeps whole volumes only");
    if (mode < DFX_SWEEP_PROJECT || mode > DFX_SWEEP_PINIT) return fail(DFX_EINVAL, "dfx_sweep: bad mode");
    if (!x || !scal || !u_part || !v_part || !part) return fail(DFX_EINVAL, "dfx_sweep: NULL buffer");
    if (mode != DFX_SWEEP_PROJECT && (!p || !(r || (ra && rb))))
        return fail(DFX_EINVAL, "dfx_sweep: p and r (or its factors ra, rb) required");
    if (r) ra = rb = nullptr;
    hipStream_t st = (hipStream_t)stream;
    int n_rows;
    if (pl->fast) {
        if (!aligned16(x) || !aligned16(r) || !aligned16(p) || !aligned16(u_part))
            return fail(DFX_EALIGN, "dfx_sweep: x, r, p, u_part must be 16-byte aligned");
        if (!aligned8(v_part) || !aligned16(ra) || !aligned8(rb))
            return fail(DFX_EALIGN, "dfx_sweep: v_part, rb must be 8-byte, ra 16-byte aligned");
        SweepArgs a{x, r, ra, rb, p, scal, u_part, v_part, part, q, pl->steps, pl->n_chunks, pl->cpb, c_lo, c_hi,
                    slice * pl->n_blocks};
        int rc;
        switch (mode) {
            case DFX_SWEEP_PROJECT: rc = launch_sweep_fast<DFX_SWEEP_PROJECT>(a, pl, st); break;
            case DFX_SWEEP_XP: rc = launch_sweep_fast<DFX_SWEEP_XP>(a, pl, st); break;
            case DFX_SWEEP_P: rc = launch_sweep_fast<DFX_SWEEP_P>(a, pl, st); break;
            default: rc = launch_sweep_fast<DFX_SWEEP_PINIT>(a, pl, st); break;
        }
        if (rc) return rc;
        n_rows = (slice + 1) * pl->n_blocks;
    } else {
        const int nb = pl->n_blocks;
        const dim3 grid(gen_tile_slots(q), pl->n_chunks);
        const int nt = gen_tiles(q);
        const bool v4 = gen_vec4(q) && aligned16(x) && aligned16(p) && aligned16(r) && aligned16(ra) && aligned16(u_part);
        if (gen_vec4(q) && !v4) return fail(DFX_EALIGN, "dfx_sweep: x, r, p, ra, u_part must be 16-byte aligned");
#define DFX_GEN_LAUNCH(M_)                                                                                                 \
    do {                                                                                                                  \
        if (v4) hipLaunchKernelGGL(k_gen_sweep4<M_>, grid, dim3(256), 0, st, x, r, ra, rb, p, scal, q, nt, u_part, part);  \
        else hipLaunchKernelGGL(k_gen_sweep2<M_>, grid, dim3(256), 0, st, x, r, ra, rb, p, scal, q, nt, u_part, part);    \
    } while (0)
        switch (mode) {
            case DFX_SWEEP_PROJECT: DFX_GEN_LAUNCH(DFX_SWEEP_PROJECT); break;
            case DFX_SWEEP_XP: DFX_GEN_LAUNCH(DFX_SWEEP_XP); break;
            case DFX_SWEEP_P: DFX_GEN_LAUNCH(DFX_SWEEP_P); break;
            default: DFX_GEN_LAUNCH(DFX_SWEEP_PINIT); break;
        }
#undef DFX_GEN_LAUNCH
        DFX_LAUNCH_CHECK("dfx_sweep(generic)");
        const float *src = (mode == DFX_SWEEP_PROJECT) ? x : p;
        launch_gen_proj_v(src, v_part, q, st);
        DFX_LAUNCH_CHECK("dfx_sweep(generic v)");
        n_rows = nb;
    }
    // (pp, xx, xp) -> scal[PP], scal[XX], scal[XP] (consecutive slots), over the rows of every slice so far
    // (last == DFX_LAST_NOFOLD: the rows are left for dfx_post_sweep, which folds them in its own launch)
    if (!last || last == DFX_LAST_NOFOLD) return 0;
    return fold(part, n_rows, 3, scal + DFX_S_PP, 1.0, st);
}

int dfx_sweep_chunks(int32_t mode, const dfx_geom *g, const dfx_plan *pl, float *x, const float *r, float *p,
                     double *scal, float *u_part, float *v_part, double *part, int32_t c_lo, int32_t c_hi,
                     int32_t slice, int32_t last, void *stream) {
    if (mode != DFX_SWEEP_PROJECT && !r) return fail(DFX_EINVAL, "dfx_sweep: r and p required");
    return sweep_impl(mode, g, pl, x, r, nullptr, nullptr, p, scal, u_part, v_part, part, c_lo, c_hi, slice, last,
                      stream);
}

int dfx_sweep_factored(int32_t mode, const dfx_geom *g, const dfx_plan *pl, float *x, const float *ra,
                       const float *rb, float *p, double *scal, float *u_part, float *v_part, double *part,
                       int32_t c_lo, int32_t c_hi, int32_t slice, int32_t last, void *stream) {
    if (mode == DFX_SWEEP_PROJECT || !ra || !rb)
        return fail(DFX_EINVAL, "dfx_sweep_factored: modes XP, P, PINIT with both factors ra, rb");
    return sweep_impl(mode, g, pl, x, nullptr, ra, rb, p, scal, u_part, v_part, part, c_lo, c_hi, slice, last,
                      stream);
}

static int resid_update_impl(const Geo &q, float *ra, float *rb, const float *A, const float *B, float sa, float sb,
                             float coef_host, const double *coef_dev, int32_t accumulate, double *zab, double *scal,
                             double *part, int beta_mode, double eps, hipStream_t st) {
    const bool wide = q.r >= 16;   // a workgroup per block instead of a thread per block
    const int groups = wide ? (int)q.Ps : (int)((q.Ps + 15) / 16);
    const int nba = groups < DFX_EW_MAXBLOCKS / 2 ? groups : DFX_EW_MAXBLOCKS / 2;
    const int nbb = q.M < DFX_EW_MAXBLOCKS / 2 ? q.M : DFX_EW_MAXBLOCKS / 2;
    if (wide)
        hipLaunchKernelGGL(k_resid_az_wide, dim3(nba), dim3(256), 0, st, ra, A, zab, q, sa, coef_host, coef_dev,
                           accumulate, part);
    else
        hipLaunchKernelGGL(k_resid_az, dim3(nba), dim3(256), 0, st, ra, A, zab, q, sa, coef_host, coef_dev, accumulate,
                           part);
    DFX_LAUNCH_CHECK("dfx_resid_update(a)");
    hipLaunchKernelGGL(k_resid_b, dim3(nbb), dim3(256), 0, st, rb, B, (const double *)zab, q.M, (int)q.Ps, sb,
                       coef_host, coef_dev, accumulate, part + (long long)nba * DFX_NPART);
    DFX_LAUNCH_CHECK("dfx_resid_update(b)");
    hipLaunchKernelGGL(k_resid_rr, dim3(1), dim3(256), 0, st, (const double *)part, nba + nbb, (double)q.M,
                       (double)q.nZ * q.r * q.r, scal, beta_mode, eps);
    DFX_LAUNCH_CHECK("dfx_resid_update(rr)");
    return 0;
}

int dfx_resid_update(const dfx_geom *g, float *ra, float *rb, const float *A, const float *B, float sa, float sb,
                     float coef_host, const double *coef_dev, int32_t accumulate, double *zab, double *scal,
                     double *part, void *stream) {
    Geo q;
    if (int rc = make_geo(g, &q)) return rc;
    if (!ra || !rb || !A || !B || !zab || !scal || !part) return fail(DFX_EINVAL, "dfx_resid_update: NULL buffer");
    if (q.Ps > 0x7fffffffLL) return fail(DFX_EINVAL, "dfx_resid_update: plane too large");
    return resid_update_impl(q, ra, rb, A, B, sa, sb, coef_host, coef_dev, accumulate, zab, scal, part, -1, 0.0,
                             (hipStream_t)stream);
}

int dfx_resid_cg_step(const dfx_geom *g, float *ra, float *rb, const float *u, const float *v, double c1, double c2,
                      double eps, int32_t deferred, double *zab, double *scal, double *part, void *stream) {
    Geo q;
    if (int rc = make_geo(g, &q)) return rc;
    if (!ra || !rb || !u || !v || !zab || !scal || !part) return fail(DFX_EINVAL, "dfx_resid_cg_step: NULL buffer");
    if (q.Ps > 0x7fffffffLL) return fail(DFX_EINVAL, "dfx_resid_cg_step: plane too large");
    return resid_update_impl(q, ra, rb, u, v, (float)c1, (float)c2, -1.0f, scal + DFX_S_ALPHA, 1, zab, scal, part,
                             deferred ? 1 : 0, eps, (hipStream_t)stream);
}

int dfx_resid_update_noalloc(const dfx_geom *g, float *ra, float *rb, const float *A, const float *B, float sa,
                             float sb, float coef_host, const double *coef_dev, int32_t accumulate, double *scal,
                             void *stream) {
    Geo q;
    if (int rc = make_geo(g, &q)) return rc;
    if (q.Ps > DFX_NOALLOC_MAX_PS)
        return fail(DFX_EINVAL, "dfx_resid_update_noalloc: (X/r)*(Y/r) exceeds DFX_NOALLOC_MAX_PS (65536); use "
                                "dfx_resid_update with a caller-provided workspace");
    if (!ra || !rb || !A || !B || !scal) return fail(DFX_EINVAL, "dfx_resid_update_noalloc: NULL buffer");
    hipStream_t st = (hipStream_t)stream;
    double *scratch = nullptr;   // this GPU's copy of g_noalloc_scratch (never the array's name)
    if (int rc = noalloc_scratch(st, &scratch)) return rc;
    // every element the kernels read is written earlier in this same call (zab[0, Ps) by k_resid_az, each partial row
    // by its workgroup), so nothing needs clearing between calls; calls overlapping on one GPU would race on it
    return resid_update_impl(q, ra, rb, A, B, sa, sb, coef_host, coef_dev, accumulate, scratch, scal,
                             scratch + DFX_NOALLOC_MAX_PS, -1, 0.0, st);
}

int dfx_bcast(const dfx_geom *g, const dfx_plan *pl, float *out, const float *in, const float *A,
              const float *B, const float *mask, float sa, float sb, float coef_host,
              const double *coef_dev, double *scal, double *part, void *stream) {
    Geo q;
    if (int rc = make_geo(g, &q)) return rc;
    if (int rc = check_plan(q, pl)) return rc;
    q.zs = pl->fast ? pl->zs : 1;
    if (!out || !A || !B || !scal || !part) return fail(DFX_EINVAL, "dfx_bcast: NULL buffer");
    hipStream_t st = (hipStream_t)stream;
    BcastArgs a{out, in, A, B, mask, coef_dev, part, sa, sb, coef_host, q, pl->steps, pl->n_chunks, pl->cpb};
    int n_rows;
    if (pl->fast) {
        if (!aligned16(out) || !aligned16(in) || !aligned16(A) || !aligned16(mask))
            return fail(DFX_EALIGN, "dfx_bcast: out, in, A, mask must be 16-byte aligned");
        if (!aligned8(B)) return fail(DFX_EALIGN, "dfx_bcast: B must be 8-byte aligned");
        int rc;
        if (in)
            rc = mask ? launch_bcast_fast<true, true>(a, pl, st) : launch_bcast_fast<true, false>(a, pl, st);
        else
            rc = mask ? launch_bcast_fast<false, true>(a, pl, st) : launch_bcast_fast<false, false>(a, pl, st);
        if (rc) return rc;
        n_rows = pl->n_blocks;
    } else {
        n_rows = pl->n_blocks;
        if (gen_vec4(q) && aligned16(out) && aligned16(in) && aligned16(A) && aligned16(mask))
            hipLaunchKernelGGL(k_gen_bcast4, dim3(gen_tile_slots(q), pl->n_chunks), dim3(256), 0, st, a, gen_tiles(q));
        else
            hipLaunchKernelGGL(k_gen_bcast, dim3(n_rows), dim3(256), 0, st, a);
        DFX_LAUNCH_CHECK("dfx_bcast(generic)");
    }
    return fold(part, n_rows, 1, scal + DFX_S_RR, 1.0, st);
}

static int adam_impl(const dfx_geom *g, const dfx_plan *pl, float *x, float *m1, const float *m1a, const float *m1b,
                     float *m2, const float *A, const float *B, const float *mask, double lr, double beta1,
                     double beta2, double eps, int32_t step, int32_t non_neg, double *scal, float *u_part,
                     float *v_part, double *part, int32_t c_lo, int32_t c_hi, int32_t slice, void *stream) {
    Geo q;
    if (int rc = make_geo(g, &q)) return rc;
    if (int rc = check_plan(q, pl)) return rc;
    q.zs = pl->fast ? pl->zs : 1;
    if (c_hi < 0) c_hi = pl->cpb;       // the whole-volume entry points
    if (c_lo < 0 || c_hi > pl->cpb || c_lo >= c_hi || slice < 0 || slice >= DFX_MAX_SLICES)
        return fail(DFX_EINVAL, "dfx_adam_step_chunks: need 0 <= c_lo < c_hi <= cpb and 0 <= slice < DFX_MAX_SLICES");
    if (!pl->fast && (c_lo != 0 || c_hi != pl->cpb || slice != 0))
        return fail(DFX_EINVAL, "dfx_adam_step_chunks: the generic path steps whole volumes only");
    const bool last_slice = (c_hi == pl->cpb);
    const bool m1f = (m1 == nullptr);
    if (!x || !m2 || !A || !B || !scal || !u_part || !v_part || !part || (m1f && (!m1a || !m1b)))
        return fail(DFX_EINVAL, "dfx_adam_step: NULL buffer");
    if (step < 1 || !(beta1 >= 0 && beta1 < 1) || !(beta2 >= 0 && beta2 < 1))
        return fail(DFX_EINVAL, "dfx_adam_step: step >= 1 and betas in [0, 1) required");
    if (!(eps > 0))   // the fast kernels' hardware sqrt flushes denormal exp_avg_sq: only eps keeps the quotient finite
        return fail(DFX_EINVAL, "dfx_adam_step: eps > 0 required");
    hipStream_t st = (hipStream_t)stream;
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    AdamArgs a{x, m1, m2, m1a, m1b, A, B, mask, u_part, v_part, part, (float)beta1, (float)beta2,
               (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps,
               (float)(lr / bc1), (float)(1.0 / sqrt(bc2)), (non_neg & DFX_ADAM_NONNEG) ? 1 : 0, q, pl->steps, pl->cpb,
               c_lo, c_hi, slice * pl->n_blocks};
    int n_rows;
    if (pl->fast) {
        if (!aligned16(x) || !aligned16(m1) || !aligned16(m2) || !aligned16(A) || !aligned16(mask) ||
            !aligned16(u_part) || !aligned16(m1a))
            return fail(DFX_EALIGN, "dfx_adam_step: x, m1 (m1a), m2, A, mask, u_part must be 16-byte aligned");
        if (!aligned8(B) || !aligned8(v_part) || !aligned8(m1b))
            return fail(DFX_EALIGN, "dfx_adam_step: B, m1b, v_part must be 8-byte aligned");
        dim3 grid((q.WT + DFX_WAVES - 1) / DFX_WAVES, pl->n_zchunks, (pl->n_chunks + pl->cpb - 1) / pl->cpb),
            block(64 * DFX_WAVES);
        if (mask && m1f)
            DFX_DISPATCH_ZC_R(pl->zc, lane_map_of(q), DFX_LAUNCH_SOLO((k_adam_fast<ZC_, R_, true, true>), grid, block, st, a));
        else if (mask)
            DFX_DISPATCH_ZC_R(pl->zc, lane_map_of(q), DFX_LAUNCH_SOLO((k_adam_fast<ZC_, R_, true, false>), grid, block, st, a));
        else if (m1f)
            DFX_DISPATCH_ZC_R(pl->zc, lane_map_of(q), DFX_LAUNCH_SOLO((k_adam_fast<ZC_, R_, false, true>), grid, block, st, a));
        else
            DFX_DISPATCH_ZC_R(pl->zc, lane_map_of(q), DFX_LAUNCH_SOLO((k_adam_fast<ZC_, R_, false, false>), grid, block, st, a));
        DFX_LAUNCH_CHECK("dfx_adam_step(fast)");
        n_rows = (slice + 1) * pl->n_blocks;
    } else {
        n_rows = pl->n_blocks;
        if (gen_vec4(q) && aligned16(x) && aligned16(m1) && aligned16(m2) && aligned16(A) && aligned16(mask) &&
            aligned16(m1a) && aligned16(u_part)) {
            const dim3 grid(gen_tile_slots(q), pl->n_chunks);
            if (m1f) hipLaunchKernelGGL(k_gen_adam4<true>, grid, dim3(256), 0, st, a, gen_tiles(q));
            else hipLaunchKernelGGL(k_gen_adam4<false>, grid, dim3(256), 0, st, a, gen_tiles(q));
            DFX_LAUNCH_CHECK("dfx_adam_step(generic)");
        } else {
            hipLaunchKernelGGL(k_gen_adam, dim3(n_rows), dim3(256), 0, st, a);
            DFX_LAUNCH_CHECK("dfx_adam_step(generic)");
            hipLaunchKernelGGL(k_gen_proj_u, dim3(ew_blocks(q.S, 1)), dim3(256), 0, st, (const float *)x, u_part, q);
            DFX_LAUNCH_CHECK("dfx_adam_step(generic u)");
            if (pl->n_chunks > 1) {   // the generic plan has one u_part row per slab-axis workgroup of the sweep; this pass fills row 0
                hipError_t e = hipMemsetAsync(u_part + q.S, 0, sizeof(float) * (size_t)(pl->n_chunks - 1) * (size_t)q.S, st);
                if (e != hipSuccess) return hipfail(e, "dfx_adam_step(generic u rows)");
            }
        }
        launch_gen_proj_v((const float *)x, v_part, q, st);
        DFX_LAUNCH_CHECK("dfx_adam_step(generic v)");
    }
    // the rows are left for dfx_post_sweep(DFX_POST_FOLD_ADAM), or this is not the final slice
    if ((non_neg & DFX_ADAM_NOFOLD) || !last_slice) return 0;
    return fold(part, n_rows, 2, scal + DFX_S_ADAM_DD, 1.0, st, scal + DFX_S_TOL);   // (DD, XX) -> consecutive slots, TOL
}

int dfx_adam_step(const dfx_geom *g, const dfx_plan *pl, float *x, float *m1, float *m2, const float *A,
                  const float *B, const float *mask, double lr, double beta1, double beta2, double eps, int32_t step,
                  int32_t non_neg, double *scal, float *u_part, float *v_part, double *part, void *stream) {
    if (!m1) return fail(DFX_EINVAL, "dfx_adam_step: NULL buffer");
    return adam_impl(g, pl, x, m1, nullptr, nullptr, m2, A, B, mask, lr, beta1, beta2, eps, step, non_neg, scal, u_part,
                     v_part, part, 0, -1, 0, stream);
}

int dfx_adam_step_chunks(const dfx_geom *g, const dfx_plan *pl, float *x, float *m1, const float *m1a, const float *m1b,
                         float *m2, const float *A, const float *B, const float *mask, double lr, double beta1,
                         double beta2, double eps, int32_t step, int32_t non_neg, double *scal, float *u_part,
                         float *v_part, double *part, int32_t c_lo, int32_t c_hi, int32_t slice, void *stream) {
    if ((m1 != nullptr) == (m1a != nullptr || m1b != nullptr) || ((m1a == nullptr) != (m1b == nullptr)))
        return fail(DFX_EINVAL, "dfx_adam_step_chunks: pass either m1 or both m1a and m1b");
    if (c_hi < 0) return fail(DFX_EINVAL, "dfx_adam_step_chunks: need 0 <= c_lo < c_hi <= cpb");
    return adam_impl(g, pl, x, m1, m1a, m1b, m2, A, B, mask, lr, beta1, beta2, eps, step, non_neg, scal, u_part, v_part,
                     part, c_lo, c_hi, slice, stream);
}

int dfx_adam_step_factored(const dfx_geom *g, const dfx_plan *pl, float *x, const float *m1a, const float *m1b,
                           float *m2, const float *A, const float *B, const float *mask, double lr, double beta1,
                           double beta2, double eps, int32_t step, int32_t non_neg, double *scal, float *u_part,
                           float *v_part, double *part, void *stream) {
    if (!m1a || !m1b) return fail(DFX_EINVAL, "dfx_adam_step_factored: NULL buffer");
    return adam_impl(g, pl, x, nullptr, m1a, m1b, m2, A, B, mask, lr, beta1, beta2, eps, step, non_neg, scal, u_part,
                     v_part, part, 0, -1, 0, stream);
}

int dfx_adam_apply(float *x, const float *grad, float *m1, float *m2, int64_t n, double lr, double beta1,
                   double beta2, double eps, int32_t step, int32_t flags, int32_t n_blocks, double *scal, double *part,
                   void *stream) {
    if (!x || !grad || !m1 || !m2 || !scal || !part || n <= 0) return fail(DFX_EINVAL, "dfx_adam_apply: bad argument");
    if (step < 1 || !(beta1 >= 0 && beta1 < 1) || !(beta2 >= 0 && beta2 < 1))
        return fail(DFX_EINVAL, "dfx_adam_apply: step >= 1 and betas in [0, 1) required");
    if (!aligned16(x) || !aligned16(grad) || !aligned16(m1) || !aligned16(m2))
        return fail(DFX_EALIGN, "dfx_adam_apply: x, grad, m1, m2 must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    AdamArgs a{};
    a.part = part;
    a.b1 = (float)beta1, a.b2 = (float)beta2, a.omb1 = (float)(1.0 - beta1), a.omb2 = (float)(1.0 - beta2);
    a.eps = (float)eps, a.step_size = (float)(lr / bc1), a.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    a.non_neg = (flags & DFX_ADAM_NONNEG) ? 1 : 0;
    int nb = n_blocks;
    if (nb <= 0) {      // one workgroup of four waves per CU, like the sweep
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            cus = 256;
        (void)hipGetLastError();
        nb = cus;
    }
    const long long need = ((n >> 2) + 255) / 256;
    if (nb > need) nb = (int)(need < 1 ? 1 : need);
    if (nb > DFX_EW_MAXBLOCKS) nb = DFX_EW_MAXBLOCKS;
    hipLaunchKernelGGL(k_adam_apply<DFX_APPLY_UNROLL>, dim3(nb), dim3(256), 0, st, x, grad, m1, m2, (long long)n, a);
    DFX_LAUNCH_CHECK("dfx_adam_apply");
    return fold(part, nb, 2, scal + DFX_S_ADAM_DD, 1.0, st, scal + DFX_S_TOL);
}

int dfx_adam_apply_project(const dfx_geom *g, const dfx_plan *pl, float *x, const float *grad, float *m1, float *m2,
                           double lr, double beta1, double beta2, double eps, int32_t step, int32_t flags, double *scal,
                           float *u_part, float *v_part, double *part, void *stream) {
    Geo q;
    if (int rc = make_geo(g, &q)) return rc;
    if (int rc = check_plan(q, pl)) return rc;
    if (!pl->fast) return fail(DFX_EPLAN, "dfx_adam_apply_project: lane-mapped plans only (use dfx_adam_apply + dfx_sweep)");
    q.zs = pl->zs;
    if (!x || !grad || !m1 || !m2 || !scal || !u_part || !v_part || !part)
        return fail(DFX_EINVAL, "dfx_adam_apply_project: NULL buffer");
    if (step < 1 || !(beta1 >= 0 && beta1 < 1) || !(beta2 >= 0 && beta2 < 1) || !(eps >= 0))
        return fail(DFX_EINVAL, "dfx_adam_apply_project: step >= 1, betas in [0, 1) and eps >= 0 required");
    if (!aligned16(x) || !aligned16(grad) || !aligned16(m1) || !aligned16(m2) || !aligned16(u_part) || !aligned8(v_part))
        return fail(DFX_EALIGN, "dfx_adam_apply_project: x, grad, m1, m2, u_part must be 16-byte, v_part 8-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    AdamArgs a{x, m1, m2, nullptr, nullptr, nullptr, nullptr, nullptr, u_part, v_part, part, (float)beta1, (float)beta2,
               (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps, (float)(lr / bc1), (float)(1.0 / sqrt(bc2)),
               (flags & DFX_ADAM_NONNEG) ? 1 : 0, q, pl->steps, pl->cpb, 0, pl->cpb, 0, grad};
    dim3 grid((q.WT + DFX_WAVES - 1) / DFX_WAVES, pl->n_zchunks, (pl->n_chunks + pl->cpb - 1) / pl->cpb), block(64 * DFX_WAVES);
    DFX_DISPATCH_ZC_R(pl->zc, lane_map_of(q), DFX_LAUNCH_SOLO((k_adam_fast<ZC_, R_, false, false, true>), grid, block, st, a));
    DFX_LAUNCH_CHECK("dfx_adam_apply_project");
    return 0;   // u, v, ADAM_DD, ADAM_XX, TOL: dfx_post_sweep(DFX_POST_U | DFX_POST_V | DFX_POST_FOLD_ADAM)
}

int dfx_adam_prep(const float *u, const float *cmos, const float *cmask, float *A, float *m1a, int64_t n_u,
                  const float *v, const float *spc, const float *smask, int64_t smask_n, float *B, float *m1b,
                  int64_t n_v, double w1, double w2, double lerp_weight, double *scal, double *part, void *stream) {
    if (!u || !cmos || !A || !v || !spc || !B || !scal || !part || n_u <= 0 || n_v <= 0 ||
        (smask && (smask_n <= 0 || n_v % smask_n)) || ((m1a == nullptr) != (m1b == nullptr)))
        return fail(DFX_EINVAL, "dfx_adam_prep: bad argument");
    PrepArgs a{u, cmos, cmask, A, m1a, (long long)n_u, v, spc, smask, (long long)(smask ? smask_n : 1), B, m1b,
               (long long)n_v, (float)(-2.0 * w1), (float)(-2.0 * w2), (float)lerp_weight, w1, w2, part, scal,
               ew_blocks(n_u, 4) > 512 ? 512 : ew_blocks(n_u, 4), ew_blocks(n_v, 4) > 1024 ? 1024 : ew_blocks(n_v, 4)};
    hipLaunchKernelGGL(k_adam_prep, dim3(a.nbu + a.nbv), dim3(256), 0, (hipStream_t)stream, a);
    DFX_LAUNCH_CHECK("dfx_adam_prep");
    hipLaunchKernelGGL(k_prep_fold, dim3(1), dim3(256), 0, (hipStream_t)stream, (const double *)part, a.nbu + a.nbv, w1,
                       w2, scal);
    DFX_LAUNCH_CHECK("dfx_adam_prep(fold)");
    return 0;
}

int dfx_lerp(float *m, const float *g, int64_t n, double weight, void *stream) {
    if (!m || !g || n <= 0) return fail(DFX_EINVAL, "dfx_lerp: bad argument");
    hipLaunchKernelGGL(k_lerp, dim3(ew_blocks(n, 4)), dim3(256), 0, (hipStream_t)stream, m, g, (long long)n,
                       (float)weight);
    DFX_LAUNCH_CHECK("dfx_lerp");
    return 0;
}

int dfx_post_sweep(const dfx_geom *g, const dfx_plan *pl, const float *u_part, float *u, const float *v_part,
                   float *v, double *scal, double *part, int32_t n_slices, int32_t flags, double c1, double c2,
                   double eps, void *stream) {
    Geo q;
    if (int rc = make_geo(g, &q)) return rc;
    if (int rc = check_plan(q, pl)) return rc;
    if (!scal || !part) return fail(DFX_EINVAL, "dfx_post_sweep: NULL buffer");
    if ((flags & DFX_POST_U) && (!u_part || !u)) return fail(DFX_EINVAL, "dfx_post_sweep: DFX_POST_U needs u_part, u");
    if ((flags & DFX_POST_V) && !v_part) return fail(DFX_EINVAL, "dfx_post_sweep: DFX_POST_V needs v_part");
    const int v_layers = pl->n_zchunks * q.vq;
    if ((flags & DFX_POST_V) && v_layers > 1 && !v && !(flags & DFX_POST_V_SUMMED))
        return fail(DFX_EINVAL, "dfx_post_sweep: v_part has more than one layer: v needed to hold the sum");
    if ((flags & DFX_POST_FOLD_SWEEP) && (flags & DFX_POST_FOLD_ADAM))
        return fail(DFX_EINVAL, "dfx_post_sweep: fold either the sweep's or the Adam step's rows");
    if (n_slices < 1 || n_slices > DFX_MAX_SLICES) return fail(DFX_EINVAL, "dfx_post_sweep: bad n_slices");
    PostArgs a;
    a.u_part = u_part;
    a.u = u;
    // DFX_POST_V_SUMMED: `v_part` already holds the summed (and, when sharded, all-reduced) v: only ||v||^2 is taken
    const bool summed = (flags & DFX_POST_V_SUMMED) || v_layers == 1;
    a.v_part = v_part;
    a.v = summed ? nullptr : v;
    a.scal = scal;
    a.fold_rows = part;
    a.part = part + (long long)pl->n_blocks * DFX_MAX_SLICES * DFX_NPART;   // behind the sweep's rows (dfx_plan.part_elems)
    a.S = q.S;
    a.vlen = (long long)q.M * q.Ps;
    a.n_urows = pl->n_chunks / pl->cpb;
    a.n_zchunks = v_layers;
    a.nbu = 0;
    a.nbv = 0;
    if (flags & DFX_POST_U) {
        long long nb = (q.S / 4 + 255) / 256;
        a.nbu = (int)(nb < 1 ? 1 : (nb > 512 ? 512 : nb));
    }
    if (flags & DFX_POST_V) {
        long long nb = (a.vlen / 4 + 1023) / 1024;
        a.nbv = (int)(nb < 1 ? 1 : (nb > 1024 ? 1024 : nb));
    }
    a.n_fold_rows = (pl->fast ? n_slices : 1) * pl->n_blocks;
    a.flags = flags;
    a.c1 = c1;
    a.c2 = c2;
    a.eps = eps;
    if (flags & (DFX_POST_U | DFX_POST_V)) {
        hipLaunchKernelGGL(k_post_sweep, dim3(a.nbu + a.nbv), dim3(256), 0, (hipStream_t)stream, a);
        DFX_LAUNCH_CHECK("dfx_post_sweep");
    }
    hipLaunchKernelGGL(k_post_fold, dim3(1), dim3(256), 0, (hipStream_t)stream, a);
    DFX_LAUNCH_CHECK("dfx_post_sweep(fold)");
    return 0;
}

int dfx_reduce_u(const dfx_geom *g, const dfx_plan *pl, const float *u_part, float *u, double *scal,
                 double *part, void *stream) {
    Geo q;
    if (int rc = make_geo(g, &q)) return rc;
    if (int rc = check_plan(q, pl)) return rc;
    q.zs = pl->fast ? pl->zs : 1;
    if (!u_part || !u || !scal || !part) return fail(DFX_EINVAL, "dfx_reduce_u: NULL buffer");
    hipStream_t st = (hipStream_t)stream;
    const int nb = ew_blocks(q.S, 1);
    hipLaunchKernelGGL(k_reduce_u, dim3(nb), dim3(256), 0, st, u_part, u, q.S, pl->n_chunks / pl->cpb, part);
    DFX_LAUNCH_CHECK("dfx_reduce_u");
    return fold(part, nb, 1, scal + DFX_S_UU, 1.0, st);
}

int dfx_reduce_v(const dfx_geom *g, const dfx_plan *pl, const float *v_part, float *v, int64_t m_lo, int64_t m_hi,
                 void *stream) {
    Geo q;
    if (int rc = make_geo(g, &q)) return rc;
    if (int rc = check_plan(q, pl)) return rc;
    q.zs = pl->fast ? pl->zs : 1;
    if (!v_part || !v) return fail(DFX_EINVAL, "dfx_reduce_v: NULL buffer");
    if (m_lo < 0 || m_hi > q.M || m_lo >= m_hi) return fail(DFX_EINVAL, "dfx_reduce_v: bad slab range");
    hipLaunchKernelGGL(k_reduce_v, dim3(ew_blocks((m_hi - m_lo) * q.Ps, 1)), dim3(256), 0, (hipStream_t)stream, v_part,
                       v, (long long)m_lo * q.Ps, (long long)m_hi * q.Ps, (long long)q.M * q.Ps, pl->n_zchunks * q.vq);
    DFX_LAUNCH_CHECK("dfx_reduce_v");
    return 0;
}

int dfx_sqnorm(const float *a, int64_t n, double *scal, int32_t slot, double *part, void *stream) {
    if (!a || !scal || !part || n <= 0 || slot < 0 || slot >= DFX_NSCAL) return fail(DFX_EINVAL, "dfx_sqnorm: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const int nb = ew_blocks(n, 4);
    hipLaunchKernelGGL(k_reduce1<0>, dim3(nb), dim3(256), 0, st, a, (const float *)nullptr, (const float *)nullptr,
                       (long long)1, (long long)n, part);
    DFX_LAUNCH_CHECK("dfx_sqnorm");
    return fold(part, nb, 1, scal + slot, 1.0, st);
}

int dfx_sqdiff(const float *a, const float *b, const float *mask, int64_t mask_n, int64_t n, double w,
               double *scal, int32_t slot, double *part, void *stream) {
    if (!a || !b || !scal || !part || n <= 0 || slot < 0 || slot >= DFX_NSCAL || (mask && mask_n <= 0))
        return fail(DFX_EINVAL, "dfx_sqdiff: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const int nb = ew_blocks(n, 4);
    hipLaunchKernelGGL(k_reduce1<1>, dim3(nb), dim3(256), 0, st, a, b, mask, (long long)(mask ? mask_n : 1),
                       (long long)n, part);
    DFX_LAUNCH_CHECK("dfx_sqdiff");
    return fold(part, nb, 1, scal + slot, w, st);
}

int dfx_sum(const float *y, int64_t n, double *scal, double *part, void *stream) {
    if (!y || !scal || !part || n <= 0) return fail(DFX_EINVAL, "dfx_sum: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const int nb = ew_blocks(n, 4);
    hipLaunchKernelGGL(k_reduce1<2>, dim3(nb), dim3(256), 0, st, y, (const float *)nullptr, (const float *)nullptr,
                       (long long)1, (long long)n, part);
    DFX_LAUNCH_CHECK("dfx_sum");
    return fold(part, nb, 1, scal + DFX_S_SUM, 1.0, st);
}

int dfx_cg_start(double *scal, void *stream) {
    if (!scal) return fail(DFX_EINVAL, "dfx_cg_start: scal is NULL");
    hipLaunchKernelGGL(k_cg_start, dim3(1), dim3(1), 0, (hipStream_t)stream, scal);
    DFX_LAUNCH_CHECK("dfx_cg_start");
    return 0;
}

int dfx_cg_alpha(double *scal, double c1, double c2, double eps, void *stream) {
    if (!scal) return fail(DFX_EINVAL, "dfx_cg_alpha: scal is NULL");
    hipLaunchKernelGGL(k_cg_alpha, dim3(1), dim3(1), 0, (hipStream_t)stream, scal, c1, c2, eps);
    DFX_LAUNCH_CHECK("dfx_cg_alpha");
    return 0;
}

int dfx_cg_beta(double *scal, double eps, int32_t deferred, void *stream) {
    if (!scal) return fail(DFX_EINVAL, "dfx_cg_beta: scal is NULL");
    hipLaunchKernelGGL(k_cg_beta, dim3(1), dim3(1), 0, (hipStream_t)stream, scal, eps, (int)deferred);
    DFX_LAUNCH_CHECK("dfx_cg_beta");
    return 0;
}

int dfx_axpy(float *y, const float *x, int64_t n, float coef_host, const double *coef_dev, double *scal,
             double *part, void *stream) {
    if (!y || !x || !scal || !part || n <= 0) return fail(DFX_EINVAL, "dfx_axpy: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const int nb = ew_blocks(n, 16);
    hipLaunchKernelGGL(k_axpy, dim3(nb), dim3(256), 0, st, y, x, (long long)n, coef_host, coef_dev, part);
    DFX_LAUNCH_CHECK("dfx_axpy");
    return fold(part, nb, 1, scal + DFX_S_XXNEW, 1.0, st);
}

int dfx_scale(float *y, int64_t n, const float *mask, int64_t mask_n, float coef_host,
              const double *coef_dev, void *stream) {
    if (!y || n <= 0 || (mask && mask_n <= 0)) return fail(DFX_EINVAL, "dfx_scale: bad argument");
    hipLaunchKernelGGL(k_scale, dim3(ew_blocks(n, 4)), dim3(256), 0, (hipStream_t)stream, y, (long long)n, mask,
                       (long long)(mask ? mask_n : 1), coef_host, coef_dev);
    DFX_LAUNCH_CHECK("dfx_scale");
    return 0;
}

int dfx_diff_scale(float *out, const float *a, const float *b, const float *mask, int64_t mask_n, int64_t n,
                   float coef, void *stream) {
    if (!out || !a || !b || n <= 0 || (mask && (mask_n <= 0 || n % mask_n)))
        return fail(DFX_EINVAL, "dfx_diff_scale: bad argument");
    hipLaunchKernelGGL(k_diff_scale, dim3(ew_blocks(n, 4)), dim3(256), 0, (hipStream_t)stream, out, a, b, mask,
                       (long long)(mask ? mask_n : 1), (long long)n, coef);
    DFX_LAUNCH_CHECK("dfx_diff_scale");
    return 0;
}

int dfx_sum_axis(const float *x, float *out, int64_t A, int64_t R, int64_t B, void *stream) {
    if (!x || !out || A <= 0 || R <= 0 || B <= 0) return fail(DFX_EINVAL, "dfx_sum_axis: bad argument");
    const long long items = (long long)A * ((B & 3) == 0 ? B / 4 : B);
    long long nb = (items + 255) / 256;
    if (nb > 65536) nb = 65536;
    hipLaunchKernelGGL(k_sum_axis, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, x, out, (long long)A,
                       (long long)R, (long long)B);
    DFX_LAUNCH_CHECK("dfx_sum_axis");
    return 0;
}

int dfx_expand_axis(const float *y, float *out, int64_t A, int64_t R, int64_t B, float scale, void *stream) {
    if (!y || !out || A <= 0 || R <= 0 || B <= 0) return fail(DFX_EINVAL, "dfx_expand_axis: bad argument");
    hipLaunchKernelGGL(k_expand_axis, dim3(ew_blocks(A * R * B, 4)), dim3(256), 0, (hipStream_t)stream, y, out,
                       (long long)A, (long long)R, (long long)B, scale);
    DFX_LAUNCH_CHECK("dfx_expand_axis");
    return 0;
}

int dfx_block_sum(const float *in, float *out, int64_t rows, int64_t X, int64_t Y, int32_t r, void *stream) {
    if (!in || !out || rows <= 0 || X <= 0 || Y <= 0 || r <= 0 || X % r || Y % r || X > 0x7fffffff || Y > 0x7fffffff)
        return fail(DFX_EINVAL, "dfx_block_sum: bad argument");
    hipLaunchKernelGGL(k_block_sum, dim3(ew_blocks(rows * (X / r) * (Y / r), 1)), dim3(256), 0, (hipStream_t)stream, in,
                       out, (long long)rows, (int)X, (int)Y, (int)r);
    DFX_LAUNCH_CHECK("dfx_block_sum");
    return 0;
}

int dfx_block_expand(const float *in, float *out, int64_t rows, int64_t X, int64_t Y, int32_t r, float scale,
                     void *stream) {
    if (!in || !out || rows <= 0 || X <= 0 || Y <= 0 || r <= 0 || X % r || Y % r || X > 0x7fffffff || Y > 0x7fffffff)
        return fail(DFX_EINVAL, "dfx_block_expand: bad argument");
    hipLaunchKernelGGL(k_block_expand, dim3(ew_blocks(rows * X * Y, 4)), dim3(256), 0, (hipStream_t)stream, in, out,
                       (long long)rows, (int)X, (int)Y, (int)r, scale);
    DFX_LAUNCH_CHECK("dfx_block_expand");
    return 0;
}

int dfx_recip(double *dst, const double *src, double num, void *stream) {
    if (!dst || !src) return fail(DFX_EINVAL, "dfx_recip: NULL");
    hipLaunchKernelGGL(k_recip, dim3(1), dim3(1), 0, (hipStream_t)stream, dst, src, num);
    DFX_LAUNCH_CHECK("dfx_recip");
    return 0;
}

static int gauss_pass(const double *in, double *out, int64_t planes, int X, int Y, int axis, double factor,
                      hipStream_t st) {
    GaussArgs a;
    a.in = in;
    a.out = out;
    a.planes = planes;
    a.X = X;
    a.Y = Y;
    a.axis = axis;
    const double sigma = (factor - 1.0) / 2.0;                 // skimage: anti_aliasing_sigma = (factor - 1) / 2
    const int radius = (int)(4.0 * sigma + 0.5);               // scipy: truncate = 4.0
    if (radius > DFX_RESIZE_MAXR) return fail(DFX_EINVAL, "dfx_resize_stack: shrink factor too large (radius > 64)");
    a.radius = radius;
    double sum = 0.0;                                          // scipy _gaussian_kernel1d: exp(-0.5 x^2 / sigma^2), normalised
    for (int k = -radius; k <= radius; ++k) sum += exp(-0.5 / (sigma * sigma) * (double)k * (double)k);
    for (int k = 0; k <= radius; ++k) a.w[k] = exp(-0.5 / (sigma * sigma) * (double)k * (double)k) / sum;
    hipLaunchKernelGGL(k_gauss_axis, dim3(ew_blocks(planes * X * Y, 1)), dim3(256), 0, st, a);
    DFX_LAUNCH_CHECK("dfx_resize_stack(gauss)");
    return 0;
}

int dfx_resize_stack(const double *in, double *tmp_a, double *tmp_b, double *out, int64_t planes, int64_t X,
                     int64_t Y, int64_t S1, int64_t S2, int32_t transpose_out, void *stream) {
    if (!in || !tmp_a || !tmp_b || !out || planes <= 0 || X <= 0 || Y <= 0 || S1 <= 0 || S2 <= 0 ||
        X > 0x7fff || Y > 0x7fff || S1 > 0x7fff || S2 > 0x7fff)
        return fail(DFX_EINVAL, "dfx_resize_stack: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const double fx = (double)X / (double)S1, fy = (double)Y / (double)S2;
    const double *cur = in;
    if (fx > 1.0) {            // anti-aliasing only on the axes that shrink; sigma = 0 elsewhere (no pass)
        if (int rc = gauss_pass(cur, tmp_a, planes, (int)X, (int)Y, 0, fx, st)) return rc;
        cur = tmp_a;
    }
    if (fy > 1.0) {
        if (int rc = gauss_pass(cur, tmp_b, planes, (int)X, (int)Y, 1, fy, st)) return rc;
        cur = tmp_b;
    }
    hipLaunchKernelGGL(k_zoom_linear, dim3(ew_blocks(planes * S1 * S2, 1)), dim3(256), 0, st, cur, out,
                       (long long)planes, (int)X, (int)Y, (int)S1, (int)S2, (int)transpose_out);
    DFX_LAUNCH_CHECK("dfx_resize_stack(zoom)");
    return 0;
}

int dfx_baseline_init(const dfx_geom *g, const float *spc, const float *zw, float *x, float coef_host,
                      const double *coef_dev, double *scal, double *part, void *stream) {
    Geo q;
    if (int rc = make_geo(g, &q)) return rc;
    if (!spc || !zw || !scal || !part) return fail(DFX_EINVAL, "dfx_baseline_init: NULL buffer");
    const int vec = (q.Y % 4 == 0 && aligned16(zw) && aligned16(x)) ? 4 : 1;
    BaseArgs a{spc, zw, x, coef_dev, part, coef_host, q, vec};
    hipStream_t st = (hipStream_t)stream;
    const int nb = ew_blocks((long long)q.M * q.X * (q.Y / vec), 1);
    hipLaunchKernelGGL(k_baseline, dim3(nb), dim3(256), 0, st, a);
    DFX_LAUNCH_CHECK("dfx_baseline_init");
    return fold(part, nb, 1, scal + DFX_S_SUM, 1.0, st);
}

}  // extern "C"
