// Which workgroup should stream which 4 KiB chunk?  256 workgroups (one per CU; workgroup w runs on XCD w % 8) copy a
// large buffer, x and p style (two arrays read and written), in "rounds" of 256 chunks of G bytes; in round i workgroup w
// takes chunk i * 256 + map(w).  The maps: identity; shift by s (XCD x -> class x + s); spread (the workgroups of one XCD
// take chunks of all 8 residues); block (an XCD takes 32 neighbouring chunks).  Prints GB/s per map and granularity.
//   hipcc -O3 --offload-arch=gfx950 -o variants/xcd_affinity_probe tools/xcd_affinity_probe.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_copy2(float *x, float *p, const int *map, long long rounds, int g4 /* unused */) {
    // per round a workgroup takes TWO 4 KiB chunks of each array, 256 chunks apart (same residue mod 8): four loads in flight
    const int w = blockIdx.x, t = threadIdx.x;
    const long long c0 = map[w];
    for (long long i = 0; i + 1 < rounds; i += 2) {
        long long o[2];
        f4 a[2], b[2];
        for (int u = 0; u < 2; ++u) {
            o[u] = (((i + u) * gridDim.x + c0) * 256 + t) * 4;
            a[u] = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(x + o[u]));
            b[u] = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(p + o[u]));
        }
        for (int u = 0; u < 2; ++u) {
            __builtin_nontemporal_store(a[u] + 0.5f * b[u], reinterpret_cast<f4 *>(x + o[u]));
            __builtin_nontemporal_store(b[u] * 0.7f + a[u], reinterpret_cast<f4 *>(p + o[u]));
        }
    }
}

int main() {
    const int W = 256;
    const long long bytes = 4LL << 30;   // per array
    float *x, *p;
    int *dmap;
    CK(hipMalloc(&x, bytes));
    CK(hipMalloc(&p, bytes));
    CK(hipMalloc(&dmap, W * sizeof(int)));
    CK(hipMemset(x, 0, bytes));
    CK(hipMemset(p, 0, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int dyn = 96 * 1024;
    CK(hipFuncSetAttribute((const void *)k_copy2, hipFuncAttributeMaxDynamicSharedMemorySize, dyn));
    for (int g4 = 1; g4 <= 1; g4 *= 2) {   // chunk = 256 threads * 16 B * g4 = 4 KiB * g4 ... wait 256*16 = 4 KiB at g4 = 1
        const long long chunk = 256LL * 16 * g4;
        const long long rounds = bytes / (chunk * W);
        struct M { const char *name; std::vector<int> m; };
        std::vector<M> maps;
        { M m{"identity", std::vector<int>(W)}; for (int w = 0; w < W; ++w) m.m[w] = w; maps.push_back(m); }
        for (int s : {1, 2, 4}) { M m{s == 1 ? "shift1" : (s == 2 ? "shift2" : "shift4"), std::vector<int>(W)}; for (int w = 0; w < W; ++w) m.m[w] = (w + s) % W; maps.push_back(m); }
        { M m{"spread", std::vector<int>(W)}; for (int w = 0; w < W; ++w) m.m[w] = (w / 8) * 8 + (w % 8 + w / 8) % 8; maps.push_back(m); }
        { M m{"block32", std::vector<int>(W)}; for (int w = 0; w < W; ++w) m.m[w] = (w % 8) * 32 + w / 8; maps.push_back(m); }
        { M m{"pairs", std::vector<int>(W)}; for (int w = 0; w < W; ++w) m.m[w] = (w / 16) * 16 + (w % 8) * 2 + (w / 8) % 2; maps.push_back(m); }
        for (auto &m : maps) {
            CK(hipMemcpy(dmap, m.m.data(), W * sizeof(int), hipMemcpyHostToDevice));
            float best = 1e30f;
            for (int rep = 0; rep < 4; ++rep) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(k_copy2, dim3(W), dim3(256), dyn, 0, x, p, dmap, rounds, g4);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep && ms < best) best = ms;
            }
            printf("chunk %6lld B  map %-9s  %.3f ms  %.0f GB/s\n", chunk, m.name, best, 4.0 * bytes / best / 1e6);
            fflush(stdout);
        }
    }
    return 0;
}
