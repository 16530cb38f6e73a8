// Does buffer_load_dwordx4 ... offen lds (LDS-DMA through a buffer descriptor, scalar row offsets) deliver the same bytes as
// global_load_lds_dwordx4 on gfx950?   hipcc --offload-arch=gfx950 -O3 -o /tmp/buf_lds_probe tools/probes/buf_lds_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void *lptr_t;
__global__ void k(const float *P, int ld, int rows, float *out, int use_buf) {
    __shared__ float lds[8 * 256];
    const int lane = threadIdx.x;  // 64 threads
    const uint64_t pb = (uint64_t)P;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(pb >> 32)) << 32) | (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)pb)), 0,
        rows * ld * 4, 0x00020000);
    for (int i = 0; i < 8; ++i) {
        float *dst = lds + i * 256;
        if (use_buf)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lptr_t)dst, 16, 16u * lane, (unsigned)(i * ld * 4), 0, 0);
        else
            __builtin_amdgcn_global_load_lds((__attribute__((address_space(1))) const void *)(P + (int64_t)i * ld + 4 * lane), (lptr_t)dst, 16, 0, 0);
    }
    __syncthreads();
    for (int i = lane; i < 8 * 256; i += 64) out[i] = lds[i];
}
int main() {
    const int ld = 4096, rows = 8;
    std::vector<float> h((size_t)ld * rows);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)i;
    float *P, *o;
    hipMalloc(&P, h.size() * 4); hipMalloc(&o, 8 * 256 * 4);
    hipMemcpy(P, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int ub = 0; ub < 2; ++ub) {
        hipMemset(o, 0, 8 * 256 * 4);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, P, ld, rows, o, ub);
        hipError_t e = hipDeviceSynchronize();
        std::vector<float> r(8 * 256);
        hipMemcpy(r.data(), o, r.size() * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 8; ++i) for (int j = 0; j < 256; ++j) if (r[i * 256 + j] != (float)(i * ld + j)) ++bad;
        printf("use_buf=%d err=%d mismatches=%d first=%g %g %g\n", ub, (int)e, bad, r[0], r[1], r[256]);
    }
    return 0;
}
