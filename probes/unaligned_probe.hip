// does global_load_dwordx4 / dwordx2 / dword work at 2-byte alignment on this box (block_q6_K is 210 bytes: odd blocks sit 2 bytes off)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
__global__ void k(const unsigned char * src, uint4 * out4, uint2 * out2, unsigned * out1, int shift) {
    const unsigned char * p = src + shift + threadIdx.x * 18;      // 18-byte stride: every 2-byte phase
    out4[threadIdx.x] = *reinterpret_cast<const uint4 *>(p);
    out2[threadIdx.x] = *reinterpret_cast<const uint2 *>(p);
    out1[threadIdx.x] = *reinterpret_cast<const unsigned *>(p);
}
int main() {
    std::vector<unsigned char> h(8192);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned char) (i * 7 + 3);
    unsigned char * d; uint4 * o4; uint2 * o2; unsigned * o1;
    hipMalloc(&d, h.size()); hipMalloc(&o4, 64 * 16); hipMalloc(&o2, 64 * 8); hipMalloc(&o1, 64 * 4);
    hipMemcpy(d, h.data(), h.size(), hipMemcpyHostToDevice);
    int bad = 0;
    for (int shift : {0, 2, 6, 10}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o4, o2, o1, shift);
        if (hipDeviceSynchronize() != hipSuccess) { printf("shift %d: launch failed: %s\n", shift, hipGetErrorString(hipGetLastError())); return 1; }
        uint4 r4[64]; uint2 r2[64]; unsigned r1[64];
        hipMemcpy(r4, o4, sizeof r4, hipMemcpyDeviceToHost); hipMemcpy(r2, o2, sizeof r2, hipMemcpyDeviceToHost); hipMemcpy(r1, o1, sizeof r1, hipMemcpyDeviceToHost);
        for (int t = 0; t < 64; ++t) {
            const unsigned char * p = h.data() + shift + t * 18;
            bad += memcmp(&r4[t], p, 16) != 0; bad += memcmp(&r2[t], p, 8) != 0; bad += memcmp(&r1[t], p, 4) != 0;
        }
        printf("shift %d: mismatches so far %d\n", shift, bad);
    }
    printf(bad ? "UNALIGNED LOADS WRONG\n" : "UNALIGNED LOADS OK\n");
    return bad != 0;
}
