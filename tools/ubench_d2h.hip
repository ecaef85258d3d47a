// micro-benchmark (tools only): device -> PINNED host memory, the way the combined single-frame launch sets return mvImagePyramid (0.75 MB per
// 640x480 frame, written by the set's own kernels), against the DMA engines; and whether it matters that the CPU has the destination lines in
// its caches (it reads the previous frame's pyramid from the same buffer).  Output committed as profiles/r05_d2h_pinned.txt; DESIGN.md section 7,
// round-5 table row 5.
//   hipcc --offload-arch=gfx950 -O3 -mclflushopt -o tools/_build/ubench_d2h tools/ubench_d2h.hip && tools/_build/ubench_d2h
#include <hip/hip_runtime.h>
#include <immintrin.h>
#include <cstdio>
#include <cstdint>
__global__ void k_copy(const uint4 *src, uint4 *dst, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i]; }
int main()
{
    const size_t maxB = 48u << 20;
    uint8_t *d, *h;
    if (hipMalloc(&d, maxB) != hipSuccess || hipHostMalloc(&h, maxB, hipHostMallocDefault) != hipSuccess) return 1;
    (void)hipMemset(d, 1, maxB);
    hipStream_t s; (void)hipStreamCreate(&s);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    volatile uint64_t sink = 0;
    printf("# copy engines against kernel stores (16 bytes per lane), best of 6\n");
    for (size_t bytes : {(size_t)768 << 10, (size_t)3 << 20, (size_t)6 << 20, (size_t)48 << 20})
        for (int mode = 0; mode < 4; mode++) {
            const int blocks = mode == 1 ? 48 : mode == 2 ? 288 : 1024;
            float best = 1e9;
            for (int rep = 0; rep < 6; rep++) {
                (void)hipEventRecord(e0, s);
                if (mode == 0) (void)hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s);
                else k_copy<<<blocks, mode == 3 ? 256 : 1024, 0, s>>>((const uint4 *)d, (uint4 *)h, bytes / 16);
                (void)hipEventRecord(e1, s); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            printf("%8zu KB  %-44s %8.1f us  %6.1f GB/s\n", bytes >> 10, mode == 0 ? "hipMemcpyAsync D2H" : mode == 1 ? "kernel, 48 workgroups x 1024" : mode == 2 ? "kernel, 288 workgroups x 1024" : "kernel, 1024 workgroups x 256",
                   best * 1e3, bytes / (best * 1e-3) / 1e9);
        }
    printf("# kernel stores (48 x 1024) into a buffer the host has just read / dirtied / flushed, mean of 6\n");
    for (size_t bytes : {(size_t)768 << 10, (size_t)6 << 20})
        for (int mode = 0; mode < 4; mode++) {
            float sum = 0;
            for (int rep = 0; rep < 8; rep++) {
                if (mode >= 1) { uint64_t a = 0; for (size_t i = 0; i < bytes; i += 8) a += *(const uint64_t *)(h + i); sink += a; }
                if (mode == 3) for (size_t i = 0; i < bytes; i += 64) h[i] = (uint8_t)rep;
                if (mode == 2) { for (size_t i = 0; i < bytes; i += 64) _mm_clflushopt(h + i); _mm_sfence(); }
                (void)hipEventRecord(e0, s);
                k_copy<<<48, 1024, 0, s>>>((const uint4 *)d, (uint4 *)h, bytes / 16);
                (void)hipEventRecord(e1, s); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (rep >= 2) sum += ms;
            }
            printf("%8zu KB  %-44s %8.1f us  %6.1f GB/s\n", bytes >> 10,
                   mode == 0 ? "host never touches the buffer" : mode == 1 ? "host read all of it before the copy" : mode == 2 ? "host read it, then clflushopt" : "host read and dirtied it", sum / 6 * 1e3,
                   bytes / (sum / 6 * 1e-3) / 1e9);
        }
    return (int)(sink & 1) * 0;
}
