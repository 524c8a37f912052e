// probes/gemv_read_pattern.hip — how much of the decode-step products' 4.3-5.3 TB/s ceiling is the ADDRESS PATTERN of GGUF-order K-quant rows?
//   Bare reads, no dots, no column prologue: a wave takes row pairs in bands exactly as gemv_kernel does (workgroup b, wave wv
// -> row pairs b + grid * (wv + NWV k)), a lane owns 64-element groups, two units (2 items per lane each) in flight.  Variants per shape:
//   gguf   : q4_K rows as a GGUF holds them: per item three 16-byte loads at a 144-byte block stride (header re-read by the 4 lanes of a block)
//            q6_K: four 16-byte loads (two of them misaligned) + a 2-byte load at a 210-byte stride
//   planes : the plane layout of csrc/pkv_wlayout.h: the same loads, lane-consecutive items = consecutive 16 bytes of a plane
//   planes_nt : ... non-temporal
//   linear : the same bytes of a row pair as one contiguous run, 16 bytes per lane (the upper bound for this wave / band structure)
//   linear_nt
// 32 dependent launches in one hipGraph, weights cycled over NBUF distinct buffers (> 2 x the 256 MiB Infinity Cache), events around replays.
//   hipcc -O3 --offload-arch=gfx950 -o probes/bin/gemv_read_pattern probes/gemv_read_pattern.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

typedef unsigned gv_u4 __attribute__((ext_vector_type(4)));
template <bool NT> __device__ __forceinline__ uint4 ld16(const unsigned char * p) {
    if constexpr (NT) { const gv_u4 v = __builtin_nontemporal_load(reinterpret_cast<const gv_u4 *>(p)); return make_uint4(v.x, v.y, v.z, v.w); }
    else return *reinterpret_cast<const uint4 *>(p);
}

struct Args { const unsigned char * w; uint32_t * dst; int n_rp; int gpr; int row_bytes; };

enum { GGUF = 0, PLANES = 1, LINEAR = 2 };

// TYPE 12 = q4_K, 14 = q6_K.  Unit = 128 items of a row pair (2 per lane).
template <int TYPE, int PAT, bool NT, int NWV>
__global__ void __launch_bounds__(64 * NWV) k_stream(const Args a) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int grid = gridDim.x, stride = grid * NWV;
    const int gpr = a.gpr;
    uint32_t acc = 0;
    constexpr int NL = TYPE == 12 ? 3 : 4;
    for (int rp = blockIdx.x + grid * wv; rp < a.n_rp; rp += stride) {
        const unsigned char * r0 = a.w + (size_t) (2 * rp) * a.row_bytes;
        if constexpr (PAT == LINEAR) {
            const int n16 = 2 * a.row_bytes / 16;                         // 16-byte pieces of the row pair
            for (int i0 = 0; i0 < n16; i0 += 64 * 2 * NL) {                // the same number of loads in flight per lane as a unit pair of the item patterns
                uint4 v[2 * NL];
#pragma unroll
                for (int j = 0; j < 2 * NL; ++j) { const int i = i0 + 64 * j + lane; v[j] = ld16<NT>(r0 + 16 * (size_t) (i < n16 ? i : n16 - 1)); }
#pragma unroll
                for (int j = 0; j < 2 * NL; ++j) acc ^= v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
            }
        } else {
            for (int f0 = 0; f0 < 2 * gpr; f0 += 128) {
                uint4 v[2][NL]; uint32_t dh[2] = {0, 0};
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    int f = f0 + 64 * u + lane; if (f >= 2 * gpr) f = 2 * gpr - 1;
                    const int rs = f >= gpr, g = f - rs * gpr, p = g & 3;
                    const unsigned char * rw = r0 + (size_t) rs * a.row_bytes;
                    if constexpr (PAT == GGUF) {
                        if constexpr (TYPE == 12) {
                            const unsigned char * bp = rw + (size_t) (g >> 2) * 144;
                            v[u][0] = ld16<NT>(bp); v[u][1] = ld16<NT>(bp + 16 + 32 * p); v[u][2] = ld16<NT>(bp + 32 + 32 * p);
                        } else {
                            const unsigned char * bp = rw + (size_t) (g >> 2) * 210; const int n = p >> 1, is = p & 1;
                            v[u][0] = ld16<NT>(bp + 64 * n + 16 * is); v[u][1] = ld16<NT>(bp + 64 * n + 16 * is + 32); v[u][2] = ld16<NT>(bp + 128 + 32 * n + 16 * is);
                            v[u][3] = ld16<NT>(bp + 192); dh[u] = *reinterpret_cast<const uint16_t *>(bp + 208);
                        }
                    } else {
                        const unsigned char * it = rw + 16 * g, * bl = rw + 16 * (g >> 2);
                        if constexpr (TYPE == 12) {
                            v[u][0] = ld16<NT>(bl + 32 * gpr); v[u][1] = ld16<NT>(it); v[u][2] = ld16<NT>(it + 16 * gpr);
                        } else {
                            v[u][0] = ld16<NT>(it); v[u][1] = ld16<NT>(it + 16 * gpr); v[u][2] = ld16<NT>(it + 32 * gpr); v[u][3] = ld16<NT>(bl + 48 * gpr);
                            dh[u] = *reinterpret_cast<const uint16_t *>(rw + 52 * gpr + 2 * (g >> 2));
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    acc ^= dh[u];
#pragma unroll
                    for (int j = 0; j < NL; ++j) acc ^= v[u][j].x ^ v[u][j].y ^ v[u][j].z ^ v[u][j].w;
                }
            }
        }
    }
    if (acc == 0x12345678u) a.dst[blockIdx.x * 64 * NWV + threadIdx.x] = acc;
}

template <typename F> static double time_graph(F launch, hipStream_t st, int nbuf) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < 32; ++i) launch(i % nbuf);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 10;
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    return ms * 1e3 / (reps * 32);
}

template <int TYPE> static void shape(const char * name, int M, int K, hipStream_t st) {
    const int row_bytes = K / 256 * (TYPE == 12 ? 144 : 210);
    const size_t bytes = (size_t) M * row_bytes;
    const int nbuf = (int) std::max<size_t>(2, (600ull << 20) / bytes + 1);
    std::vector<unsigned char *> bufs(nbuf);
    for (auto & b : bufs) { CK(hipMalloc(&b, bytes + 256)); CK(hipMemset(b, 0x5a, bytes + 256)); }
    uint32_t * dst; CK(hipMalloc(&dst, 256 * 1024 * 4));
    const int n_rp = M / 2;
    auto run = [&](auto kern, int nwv) {
        const int grid = std::min(256, (n_rp + nwv - 1) / nwv);
        return time_graph([&](int i) { Args a{bufs[i], dst, n_rp, K / 64, row_bytes}; hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * nwv), 0, st, a); }, st, nbuf);
    };
    const double t_g = run(k_stream<TYPE, GGUF, false, 16>, 16), t_p = run(k_stream<TYPE, PLANES, false, 16>, 16), t_pn = run(k_stream<TYPE, PLANES, true, 16>, 16),
                 t_l = run(k_stream<TYPE, LINEAR, false, 16>, 16), t_ln = run(k_stream<TYPE, LINEAR, true, 16>, 16),
                 t_p8 = run(k_stream<TYPE, PLANES, true, 8>, 8);
    auto tb = [&](double us) { return bytes / us * 1e-6; };
    printf("%-28s %7.1f MB | gguf %6.2f us %5.2f TB/s | planes %6.2f us %5.2f | planes_nt %6.2f us %5.2f | planes_nt 8 waves %6.2f us %5.2f | linear %6.2f us %5.2f | linear_nt %6.2f us %5.2f\n",
           name, bytes / 1e6, t_g, tb(t_g), t_p, tb(t_p), t_pn, tb(t_pn), t_p8, tb(t_p8), t_l, tb(t_l), t_ln, tb(t_ln));
    for (auto b : bufs) CK(hipFree(b));
    CK(hipFree(dst));
}

int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    printf("bare reads with the decode-step products' wave / band structure (16 waves per CU, one workgroup per CU, two units in flight), us per launch in a graph of 32\n");
    shape<12>("q4_K gate+up 28672 x 4096", 28672, 4096, st);
    shape<12>("q4_K ffn_down 4096 x 14336", 4096, 14336, st);
    shape<14>("q6_K ffn_down 4096 x 14336", 4096, 14336, st);
    shape<14>("q6_K lm_head 128256 x 4096", 128256, 4096, st);
    shape<12>("q4_K wq+wk 5120 x 4096", 5120, 4096, st);
    shape<12>("q4_K wo 4096 x 4096", 4096, 4096, st);
    return 0;
}
