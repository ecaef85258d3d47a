// tools/ubench_hbm_calib.hip -- calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access patterns the ORB kernels use
// (the MI355X guide calibrates only 16 B/lane streaming reads: FETCH_SIZE reports half of those bytes).
// Every kernel moves a KNOWN number of bytes over a 1 GiB region (4x the 256 MiB Infinity Cache); run it under
//   rocprofv3 --pmc FETCH_SIZE ...   and   rocprofv3 --pmc WRITE_SIZE ...   (separate passes)
// and divide the counter by the known byte count (tools/summarize_calib.py).  Patterns:
//   k_read16      16 B / lane streaming read (the guide's case)
//   k_read8u      8 B / lane, rows of 80 bytes starting at a 16-byte-aligned but not 128-byte-aligned address (k_fast_cells / k_blur window staging)
//   k_read4       4 B / lane streaming read (k_resize source words, k_orient disc rows)
//   k_gather1     byte gathers inside a 37x37 patch at a random position per wave (k_describe)
//   k_write16 / k_write4 / k_write1   streaming writes of 16 / 4 / 1 byte per lane (level images are written as dwords, descriptors as bytes)
// hipcc --offload-arch=gfx950 -O3 -o ubench_hbm_calib tools/ubench_hbm_calib.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_read16(const uint4 *__restrict__ src, size_t n, uint32_t *sink)
{
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { uint4 v = src[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void k_read4(const uint32_t *__restrict__ src, size_t n, uint32_t *sink)
{
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc ^= src[i];
    if (acc == 0x12345678u) *sink = acc;
}
// one wave stages a window of 38 rows x 80 bytes (10 lanes x 8 B per row) out of an image with pitch 704, windows 64 px apart: the detector's pattern
__global__ void k_read8u(const uint8_t *__restrict__ src, size_t nwin, int pitch, int winPerRow, uint32_t *sink)
{
    const int lane = threadIdx.x & 63;
    const size_t wv = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6, nw = ((size_t)gridDim.x * blockDim.x) >> 6;
    uint32_t acc = 0;
    for (size_t w = wv; w < nwin; w += nw) {
        const size_t ty = w / winPerRow, tx = w % winPerRow;
        const uint8_t *base = src + ty * 32 * (size_t)pitch + tx * 64 + 16;     // 16-byte aligned, not line aligned
        for (int i = lane; i < 38 * 10; i += 64) {
            const int r = i / 10, c = i % 10;
            const uint2 v = *(const uint2 *)(base + (size_t)r * pitch + c * 8);
            acc ^= v.x ^ v.y;
        }
    }
    if (acc == 0x12345678u) *sink = acc;
}
// one wave per "keypoint": 512 byte loads inside a 37x37 patch
__global__ void k_gather1(const uint8_t *__restrict__ src, size_t nkp, int pitch, int rows, uint32_t *sink)
{
    const int lane = threadIdx.x & 63;
    const size_t wv = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6, nw = ((size_t)gridDim.x * blockDim.x) >> 6;
    uint32_t acc = 0;
    for (size_t k = wv; k < nkp; k += nw) {
        uint64_t h = k * 0x9E3779B97F4A7C15ull; h ^= h >> 29;
        const size_t y = 20 + (h % (size_t)(rows - 40)), x = 20 + ((h >> 32) % (size_t)(pitch - 40));
        const uint8_t *c = src + y * (size_t)pitch + x;
        for (int s = 0; s < 8; s++) {
            const uint32_t g = (uint32_t)(lane * 8 + s) * 2654435761u;
            const int dx = (int)(g % 37) - 18, dy = (int)((g >> 8) % 37) - 18;
            acc += c[dy * pitch + dx];
        }
    }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void k_write16(uint4 *dst, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = make_uint4((uint32_t)i, 1, 2, 3); }
__global__ void k_write4(uint32_t *dst, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = (uint32_t)i; }
__global__ void k_write1(uint8_t *dst, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = (uint8_t)i; }

int main()
{
    const size_t N = (size_t)1 << 30;
    uint8_t *buf; uint32_t *sink;
    CHECK(hipMalloc(&buf, N)); CHECK(hipMalloc(&sink, 4));
    CHECK(hipMemset(buf, 1, N));
    const int pitch = 704, rows = (int)(N / pitch);
    const int winPerRow = (pitch - 96) / 64;                  // windows 64 px apart, 80 bytes wide
    const size_t nwin = (size_t)(rows / 32 - 2) * winPerRow;
    const size_t nkp = 4 << 20;
    printf("known_bytes k_read16 %zu\nknown_bytes k_read4 %zu\nknown_bytes k_read8u %zu (touched rows x 80 B; the windows of a tile row overlap by 16 B and 6 rows)\n", N, N, nwin * 38 * 80);
    printf("known_bytes k_gather1 %zu (bytes requested; each patch spans 37 rows)\nknown_bytes k_write16 %zu\nknown_bytes k_write4 %zu\nknown_bytes k_write1 %zu\n", nkp * 512, N, N, N / 4);
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k_read16, dim3(4096), dim3(256), 0, 0, (const uint4 *)buf, N / 16, sink);
        hipLaunchKernelGGL(k_read4, dim3(4096), dim3(256), 0, 0, (const uint32_t *)buf, N / 4, sink);
        hipLaunchKernelGGL(k_read8u, dim3(4096), dim3(256), 0, 0, buf, nwin, pitch, winPerRow, sink);
        hipLaunchKernelGGL(k_gather1, dim3(4096), dim3(256), 0, 0, buf, nkp, pitch, rows, sink);
        hipLaunchKernelGGL(k_write16, dim3(4096), dim3(256), 0, 0, (uint4 *)buf, N / 16);
        hipLaunchKernelGGL(k_write4, dim3(4096), dim3(256), 0, 0, (uint32_t *)buf, N / 4);
        hipLaunchKernelGGL(k_write1, dim3(4096), dim3(256), 0, 0, buf, N / 4);
        CHECK(hipDeviceSynchronize());
    }
    printf("done\n");
    return 0;
}
