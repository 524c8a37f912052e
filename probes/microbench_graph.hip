// probes/microbench_graph.hip — platform floor for the single-sequence paged decode launch (16 MiB of KV per launch).
// Every variant is captured as 32 dependent launches in ONE hipGraph (what bench.py replays), cycling through 40 distinct
// 16 MiB buffers (640 MiB > 2x Infinity Cache), and timed with events around back-to-back replays: us per launch includes
// the in-graph dispatch gap, exactly like bench.py's kernel_us.mean.  (The r01 microbench launched eagerly: its 3.5 us
// "empty kernel" was the host's launch rate, not the device's.)
//   hipcc -O3 --offload-arch=gfx950 -o probes/bin/microbench_graph probes/microbench_graph.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <functional>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

struct Args { const uint4 * src; const int * table; uint32_t * dst; int n_wg; };

__global__ void k_empty(const Args a) { if (a.n_wg == -1) a.dst[0] = 0; }

// NL x 16 B per thread, all issued up front, tile = blockIdx (direct) or table[blockIdx] (one dependent scalar load, like the block table)
template <int NL, int BLOCK, bool TABLE>
__global__ void __launch_bounds__(BLOCK) k_read(const Args a) {
    const int tile = TABLE ? a.table[blockIdx.x] : blockIdx.x;
    const uint4 * p = a.src + (size_t) tile * BLOCK * NL + threadIdx.x;
    uint4 v[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) v[i] = p[(size_t) i * BLOCK];
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < NL; ++i) acc ^= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
    if (acc == 0x12345678u) a.dst[blockIdx.x] = acc;
}
// the same with a compute tail of `WORK` dependent-ish VALU ops per loaded dword and an LDS exchange + barrier + 2 KB store per
// workgroup: a stand-in for QK^T / softmax / PV / merge / partial store
template <int NL, int BLOCK, int WORK>
__global__ void __launch_bounds__(BLOCK) k_read_work(const Args a) {
    __shared__ float red[BLOCK];
    const int tile = a.table[blockIdx.x];
    const uint4 * p = a.src + (size_t) tile * BLOCK * NL + threadIdx.x;
    uint4 v[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) v[i] = p[(size_t) i * BLOCK];
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const float f[4] = {__uint_as_float(v[i].x), __uint_as_float(v[i].y), __uint_as_float(v[i].z), __uint_as_float(v[i].w)};
#pragma unroll
        for (int w = 0; w < WORK; ++w)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = fmaf(f[j], acc[(j + w) & 3] + 1.0f, acc[j]);
    }
    red[threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
    __syncthreads();
    a.dst[(size_t) blockIdx.x * BLOCK + threadIdx.x] = __float_as_uint(red[threadIdx.x ^ 64]);
}

// bare read with wall-clock (100 MHz) stamps per wave: entry, all loads issued, all data landed
template <int NL, int BLOCK>
__global__ void __launch_bounds__(BLOCK) k_read_stamped(const Args a, unsigned long long * st) {
    unsigned long long t0, t1, t2;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    const int tile = a.table[blockIdx.x];
    const uint4 * p = a.src + (size_t) tile * BLOCK * NL + threadIdx.x;
    uint4 v[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) v[i] = p[(size_t) i * BLOCK];
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2) :: "memory");
    __builtin_amdgcn_sched_barrier(0);
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < NL; ++i) acc ^= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
    if (acc == 0x12345678u) a.dst[blockIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) {
        unsigned long long * d = st + ((size_t) blockIdx.x * (BLOCK / 64) + threadIdx.x / 64) * 3;
        d[0] = t0; d[1] = t1; d[2] = t2;
    }
}

static std::vector<uint4 *> g_bufs;
static int * g_table; static uint32_t * g_dst;

template <typename F>
static double time_graph(F launch, int n_nodes, int reps) {
    hipStream_t st; CK(hipStreamCreate(&st));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < n_nodes; ++i) launch(st, i);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a, st));
    for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(b, st));
    CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); CK(hipStreamDestroy(st));
    return ms * 1e3 / ((double) reps * n_nodes);
}

template <int NL, int BLOCK, bool TABLE>
static void run_read(const char * name, size_t bytes) {
    const int grid = (int) (bytes / ((size_t) BLOCK * NL * 16));
    const double us = time_graph([&](hipStream_t st, int i) {
        Args a{g_bufs[i % g_bufs.size()], g_table, g_dst, grid};
        hipLaunchKernelGGL((k_read<NL, BLOCK, TABLE>), dim3(grid), dim3(BLOCK), 0, st, a);
    }, 32, 40);
    printf("%-44s grid %5d block %4d NL %2d : %6.2f us/launch  %7.0f GB/s\n", name, grid, BLOCK, NL, us, bytes / us / 1e3);
}
template <int NL, int BLOCK, int WORK>
static void run_work(const char * name, size_t bytes) {
    const int grid = (int) (bytes / ((size_t) BLOCK * NL * 16));
    const double us = time_graph([&](hipStream_t st, int i) {
        Args a{g_bufs[i % g_bufs.size()], g_table, g_dst, grid};
        hipLaunchKernelGGL((k_read_work<NL, BLOCK, WORK>), dim3(grid), dim3(BLOCK), 0, st, a);
    }, 32, 40);
    printf("%-44s grid %5d block %4d NL %2d : %6.2f us/launch  %7.0f GB/s\n", name, grid, BLOCK, NL, us, bytes / us / 1e3);
}

int main(int argc, char ** argv) {
    const size_t bytes = (argc > 1 ? atol(argv[1]) : 16) << 20;
    const int nbuf = (int) ((640ull << 20) / bytes) + 1;
    g_bufs.resize(nbuf);
    for (auto & p : g_bufs) { CK(hipMalloc(&p, bytes)); CK(hipMemset(p, 1, bytes)); }
    CK(hipMalloc(&g_dst, 64 << 20));
    std::vector<int> tab(1 << 16);
    for (size_t i = 0; i < tab.size(); ++i) tab[i] = (int) i;    // identity, but the kernel cannot know
    CK(hipMalloc(&g_table, tab.size() * sizeof(int)));
    CK(hipMemcpy(g_table, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice));
    CK(hipDeviceSynchronize());
    printf("bytes per launch: %zu MiB, %d buffers cycled, 32 launches per graph\n", bytes >> 20, nbuf);
    for (int wg : {1, 256, 1024}) {
        const double us = time_graph([&](hipStream_t st, int) { Args a{nullptr, nullptr, g_dst, 0}; hipLaunchKernelGGL(k_empty, dim3(wg), dim3(512), 0, st, a); }, 32, 40);
        printf("empty kernel, %4d x 512 threads                                          : %6.2f us/launch\n", wg, us);
    }
    run_read<8, 512, false>("read direct   (WG = decode's: 512 thr x 8)", bytes);
    run_read<8, 512, true >("read via table", bytes);
    run_read<8, 256, false>("read direct   256 thr x 8", bytes);
    run_read<8, 256, true >("read via table 256 thr x 8", bytes);
    run_read<4, 256, true >("read via table 256 thr x 4", bytes);
    run_read<8, 128, true >("read via table 128 thr x 8", bytes);
    run_read<4, 512, true >("read via table 512 thr x 4", bytes);
    run_read<16, 256, true>("read via table 256 thr x 16", bytes);
    run_read<16, 512, true>("read via table 512 thr x 16", bytes);
    run_read<2, 256, true >("read via table 256 thr x 2", bytes);
    run_work<8, 512, 1>("read + light tail (1 fma/dword) 512 thr x 8", bytes);
    run_work<8, 512, 4>("read + tail (4 fma/dword)       512 thr x 8", bytes);
    run_work<8, 512, 8>("read + tail (8 fma/dword)       512 thr x 8", bytes);
    run_work<8, 256, 4>("read + tail (4 fma/dword)       256 thr x 8", bytes);
    run_work<4, 256, 4>("read + tail (4 fma/dword)       256 thr x 4", bytes);
    run_work<4, 128, 4>("read + tail (4 fma/dword)       128 thr x 4", bytes);
    {   // wall-clock picture of the bare read (decode's geometry) INSIDE a graph of 32 back-to-back launches: when do waves enter,
        // finish issuing, have their data; and how far apart are consecutive launches?
        constexpr int NL = 8, BLOCK = 512, NG = 32;
        const int grid = (int) (bytes / ((size_t) BLOCK * NL * 16)), nw = grid * (BLOCK / 64);
        unsigned long long * st; CK(hipMalloc(&st, (size_t) NG * nw * 3 * 8));
        std::vector<unsigned long long> h((size_t) NG * nw * 3);
        hipStream_t s2; CK(hipStreamCreate(&s2));
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s2, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < NG; ++i) {
            Args a{g_bufs[i % g_bufs.size()], g_table, g_dst, grid};
            hipLaunchKernelGGL((k_read_stamped<NL, BLOCK>), dim3(grid), dim3(BLOCK), 0, s2, a, st + (size_t) i * nw * 3);
        }
        CK(hipStreamEndCapture(s2, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        std::vector<double> e_last, i_med, i_last, d_med, d_last, d_first, cad, gap;
        for (int r = 0; r < 12; ++r) {
            for (int k = 0; k < 3; ++k) CK(hipGraphLaunch(ge, s2));
            CK(hipStreamSynchronize(s2));
            CK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
            if (r < 2) continue;
            unsigned long long prev_first = 0, prev_last = 0;
            for (int l = 0; l < NG; ++l) {
                const unsigned long long * hl = h.data() + (size_t) l * nw * 3;
                unsigned long long t0 = ~0ull, t9 = 0;
                for (int w = 0; w < nw; ++w) { t0 = std::min(t0, hl[(size_t) w * 3]); t9 = std::max(t9, hl[(size_t) w * 3 + 2]); }
                std::vector<double> e(nw), i(nw), d(nw);
                for (int w = 0; w < nw; ++w) { e[w] = (hl[(size_t) w * 3] - t0) * 10.0; i[w] = (hl[(size_t) w * 3 + 1] - t0) * 10.0; d[w] = (hl[(size_t) w * 3 + 2] - t0) * 10.0; }
                std::sort(e.begin(), e.end()); std::sort(i.begin(), i.end()); std::sort(d.begin(), d.end());
                e_last.push_back(e.back()); i_med.push_back(i[nw / 2]); i_last.push_back(i.back()); d_first.push_back(d[0]); d_med.push_back(d[nw / 2]); d_last.push_back(d.back());
                if (l > 0) { cad.push_back((t0 - prev_first) * 10.0); gap.push_back(((double) t0 - (double) prev_last) * 10.0); }
                prev_first = t0; prev_last = t9;
            }
        }
        auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
        printf("bare read 256 x 512 thr x 8 in a graph, wall ns after the launch's first wave entry (medians): last wave entry %.0f | loads issued: median wave %.0f, last %.0f | "
               "data landed: first wave %.0f, median %.0f, last %.0f | cadence %.0f, gap (next first entry - last data) %.0f\n",
               med(e_last), med(i_med), med(i_last), med(d_first), med(d_med), med(d_last), med(cad), med(gap));
    }
    return 0;
}
