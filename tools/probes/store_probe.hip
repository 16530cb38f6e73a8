// What HBM write rate can a plain store kernel reach on this MI355X?  (Context for the write-bound feature kernels.)
//   hipcc --offload-arch=gfx950 -O3 tools/probes/store_probe.hip -o /tmp/store_probe && /tmp/store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));

template <bool NT>
__global__ void __launch_bounds__(256) store_x4(f4 *out, size_t n4, float v) {
    const f4 val = {v, v + 1, v + 2, v + 3};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        if (NT) __builtin_nontemporal_store(val, &out[i]);
        else out[i] = val;
    }
}
// each wave owns whole 1 KiB row segments of a (rows, ld) matrix, like the FastFood kernel's output
__global__ void __launch_bounds__(256) store_rows(f4 *out, size_t rows, size_t ld4, float v) {
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6, nw = (size_t)gridDim.x * 4;
    const f4 val = {v, v, v, v};
    for (size_t r = wave; r < rows; r += nw)
        for (size_t c = lane; c < ld4; c += 64) out[r * ld4 + c] = val;
}
// 4-byte stores, half-wave = 128 contiguous bytes of a row, 16 rows per lane (the MFMA feature kernel's pattern)
__global__ void __launch_bounds__(256) store_dword_tiles(float *out, size_t rows, size_t ld, float v) {
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6, nw = (size_t)gridDim.x * 4;
    const size_t tiles_r = rows / 32, tiles_c = ld / 32;
    for (size_t t = wave; t < tiles_r * tiles_c; t += nw) {
        const size_t r0 = (t / tiles_c) * 32, c0 = (t % tiles_c) * 32;
#pragma unroll
        for (int e = 0; e < 16; ++e) out[(r0 + 4 * h + (e & 3) + 8 * (e >> 2)) * ld + c0 + j] = v;
    }
}

// The FastFood chain kernel's output pattern: a workgroup of WPB waves owns 4 WPB adjacent blocks of 128 columns (cos at
// column c, sin at n + c) for the rows [r0, r1); a wave's store instruction covers its 4 blocks: 16 lanes x 16 B = 256
// contiguous bytes per block and group (PIECE = 0), or 64 lanes x 16 B = 1 KiB contiguous (PIECE = 1).
template <int WPB, int PIECE>
__global__ void __launch_bounds__(64 * WPB) store_fastfood(float *out, int rows, int rp, int n, float v) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l16 = lane & 15, sub = lane >> 4;
    const int jb = (blockIdx.x * WPB + wave) * 4;
    const f4 val = {v, v, v, v};
    const int r0 = blockIdx.y * rp, r1 = min(rows, r0 + rp);
    for (int r = r0; r < r1; ++r) {
        float *orow = out + (size_t)r * 2 * n;
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                float *o = PIECE ? orow + half * n + jb * 128 + g * 256 + lane * 4
                                 : orow + half * n + (jb + sub) * 128 + g * 64 + l16 * 4;
                *reinterpret_cast<f4 *>(o) = val;
            }
    }
}

int main() {
    const size_t bytes = (size_t)32 << 30;
    float *d;
    if (hipMalloc(&d, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    auto run = [&](const char *name, auto launch) {
        launch();
        hipDeviceSynchronize();
        hipEventRecord(a);
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("%-44s %.2f TB/s\n", name, 3.0 * bytes / (ms * 1e-3) / 1e12);
    };
    for (int blocks : {2048, 8192, 65536}) {
        char nm[96];
        snprintf(nm, 96, "dwordx4 grid-stride, %d blocks", blocks);
        run(nm, [&] { hipLaunchKernelGGL(store_x4<false>, dim3(blocks), dim3(256), 0, 0, (f4 *)d, bytes / 16, 1.f); });
        snprintf(nm, 96, "dwordx4 grid-stride nt, %d blocks", blocks);
        run(nm, [&] { hipLaunchKernelGGL(store_x4<true>, dim3(blocks), dim3(256), 0, 0, (f4 *)d, bytes / 16, 1.f); });
    }
    run("1 KiB row segments per wave (ld 16384)", [&] { hipLaunchKernelGGL(store_rows, dim3(8192), dim3(256), 0, 0, (f4 *)d, bytes / 65536, (size_t)4096, 2.f); });
    run("dword tiles, 128 B per half-wave (ld 4096)", [&] { hipLaunchKernelGGL(store_dword_tiles, dim3(8192), dim3(256), 0, 0, d, bytes / 16384, (size_t)4096, 3.f); });
    {
        const int rows = 262144, n = 8192;
        const size_t fb = (size_t)rows * 2 * n * 4;
        auto runf = [&](const char *name, auto launch) {
            launch();
            hipDeviceSynchronize();
            hipEventRecord(a);
            for (int i = 0; i < 3; ++i) launch();
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            printf("%-60s %.2f TB/s\n", name, 3.0 * fb / (ms * 1e-3) / 1e12);
        };
        for (int rp : {342, 128, 32, 8}) {
            char nm[128];
            const unsigned gy = (rows + rp - 1) / rp;
            snprintf(nm, 128, "fastfood pattern 4 waves, 256 B pieces, %d rows/wg", rp);
            runf(nm, [&] { hipLaunchKernelGGL((store_fastfood<4, 0>), dim3(4, gy), dim3(256), 0, 0, d, rows, rp, n, 1.f); });
            snprintf(nm, 128, "fastfood pattern 4 waves, 1 KiB pieces, %d rows/wg", rp);
            runf(nm, [&] { hipLaunchKernelGGL((store_fastfood<4, 1>), dim3(4, gy), dim3(256), 0, 0, d, rows, rp, n, 1.f); });
            snprintf(nm, 128, "fastfood pattern 16 waves (whole rows), 256 B pieces, %d rows/wg", rp);
            runf(nm, [&] { hipLaunchKernelGGL((store_fastfood<16, 0>), dim3(1, gy), dim3(1024), 0, 0, d, rows, rp, n, 1.f); });
            snprintf(nm, 128, "fastfood pattern 16 waves (whole rows), 1 KiB pieces, %d rows/wg", rp);
            runf(nm, [&] { hipLaunchKernelGGL((store_fastfood<16, 1>), dim3(1, gy), dim3(1024), 0, 0, d, rows, rp, n, 1.f); });
        }
    }
    run("hipMemsetAsync", [&] { hipMemsetAsync(d, 0, bytes, 0); });
    return 0;
}
