// tools/microbench.hip — platform probes for the paged-decode design (run under rocprofv3 --kernel-trace --stats).
// How long does a bare 16 MiB HBM read take as a function of launch geometry?  What is the floor of a launch?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void k_empty(int * p) { if (p == (int *) 1) *p = 0; }

// each thread reads NL x 16 B, all issued up front; bytes = grid * BLOCK * NL * 16
template <int NL, int BLOCK>
__global__ void __launch_bounds__(BLOCK) k_read(const uint4 * __restrict__ src, uint32_t * __restrict__ dst) {
    const size_t base = (size_t) blockIdx.x * BLOCK * NL + threadIdx.x;
    uint4 v[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) v[i] = src[base + (size_t) i * BLOCK];
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < NL; ++i) acc ^= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
    if (acc == 0x12345678u) dst[blockIdx.x] = acc;
}

template <int NL, int BLOCK>
__global__ void __launch_bounds__(BLOCK) k_read_nt(const uint4 * __restrict__ src, uint32_t * __restrict__ dst) {
    const size_t base = (size_t) blockIdx.x * BLOCK * NL + threadIdx.x;
    uint4 v[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const uint4 * p = src + base + (size_t) i * BLOCK;
        v[i].x = __builtin_nontemporal_load(&p->x); v[i].y = __builtin_nontemporal_load(&p->y);
        v[i].z = __builtin_nontemporal_load(&p->z); v[i].w = __builtin_nontemporal_load(&p->w);
    }
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < NL; ++i) acc ^= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
    if (acc == 0x12345678u) dst[blockIdx.x] = acc;
}

template <int NL, int BLOCK, bool NT = false>
void run_read(const char * name, const std::vector<uint4 *> & bufs, uint32_t * dst, size_t bytes, int iters) {
    const int grid = (int) (bytes / ((size_t) BLOCK * NL * 16));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 4; ++i) if (NT) hipLaunchKernelGGL((k_read_nt<NL, BLOCK>), dim3(grid), dim3(BLOCK), 0, 0, bufs[i % bufs.size()], dst); else hipLaunchKernelGGL((k_read<NL, BLOCK>), dim3(grid), dim3(BLOCK), 0, 0, bufs[i % bufs.size()], dst);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) { if (NT) hipLaunchKernelGGL((k_read_nt<NL, BLOCK>), dim3(grid), dim3(BLOCK), 0, 0, bufs[i % bufs.size()], dst); else hipLaunchKernelGGL((k_read<NL, BLOCK>), dim3(grid), dim3(BLOCK), 0, 0, bufs[i % bufs.size()], dst); }
    CK(hipEventRecord(b));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("%-28s grid %5d block %4d NL %2d : %.2f us/launch back-to-back (%.0f GB/s incl. gaps)\n", name, grid, BLOCK, NL,
           ms * 1e3 / iters, bytes / (ms * 1e-3 / iters) / 1e9);
}

int main(int argc, char ** argv) {
    const size_t bytes = (argc > 1 ? atol(argv[1]) : 16) << 20;
    const int nbuf = 40, iters = 400;   // 40 x 16 MiB = 640 MiB > 2x Infinity Cache
    std::vector<uint4 *> bufs(nbuf);
    for (auto & p : bufs) { CK(hipMalloc(&p, bytes)); CK(hipMemset(p, 1, bytes)); }
    uint32_t * dst; CK(hipMalloc(&dst, 1 << 20));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, 0, (int *) dst);
    CK(hipEventRecord(b)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("empty kernel: %.2f us/launch back-to-back\n", ms * 1e3 / iters);
    run_read<16, 256>("read 256wg x 16", bufs, dst, bytes, iters);
    run_read<8, 256>("read 512wg x 8", bufs, dst, bytes, iters);
    run_read<4, 256>("read 1024wg x 4", bufs, dst, bytes, iters);
    run_read<2, 256>("read 2048wg x 2", bufs, dst, bytes, iters);
    run_read<1, 256>("read 4096wg x 1", bufs, dst, bytes, iters);
    run_read<8, 512>("read 256wg(512t) x 8", bufs, dst, bytes, iters);
    run_read<4, 1024>("read 256wg(1024t) x 4", bufs, dst, bytes, iters);
    run_read<4, 512>("read 512wg(512t) x 4", bufs, dst, bytes, iters);
    run_read<16, 64>("read 1024wg(64t) x 16", bufs, dst, bytes, iters);
    run_read<8, 64>("read 2048wg(64t) x 8", bufs, dst, bytes, iters);
    run_read<8, 512, true>("nt read 256wg(512t) x 8", bufs, dst, bytes, iters);
    run_read<2, 256, true>("nt read 2048wg x 2", bufs, dst, bytes, iters);
    run_read<16, 256, true>("nt read 256wg x 16", bufs, dst, bytes, iters);
    return 0;
}
