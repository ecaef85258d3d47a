// micro-benchmark: issue rate of a few integer VALU ops on gfx950 (wave64).  tools only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int OP> __global__ void k(unsigned *out, int iters)
{
    unsigned a = threadIdx.x * 2654435761u, b = a ^ 0x9e3779b9u, c = a + 7, d = b + 13;
    unsigned e = a ^ 0x1234567u, f = b + 0x89abcdefu, g = c * 3u, h = d ^ 0xdeadbeefu;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            if (OP == 0) { a = __popc(a) + b; b = __popc(b) + c; c = __popc(c) + d; d = __popc(d) + e; e = __popc(e) + f; f = __popc(f) + g; g = __popc(g) + h; h = __popc(h) + a; }
            if (OP == 1) { a ^= b; b ^= c; c ^= d; d ^= e; e ^= f; f ^= g; g ^= h; h ^= a + 1; }
            if (OP == 2) { a = min(a, b) + 1; b = min(b, c) + 1; c = min(c, d) + 1; d = min(d, e) + 1; e = min(e, f) + 1; f = min(f, g) + 1; g = min(g, h) + 1; h = min(h, a) + 1; }
            if (OP == 3) { a = __builtin_amdgcn_perm(a, b, 0x07020500u); b = __builtin_amdgcn_perm(b, c, 0x07020500u); c = __builtin_amdgcn_perm(c, d, 0x07020500u); d = __builtin_amdgcn_perm(d, e, 0x07020500u);
                           e = __builtin_amdgcn_perm(e, f, 0x07020500u); f = __builtin_amdgcn_perm(f, g, 0x07020500u); g = __builtin_amdgcn_perm(g, h, 0x07020500u); h = __builtin_amdgcn_perm(h, a, 0x07020500u); }
            if (OP == 4) { a = __builtin_amdgcn_udot4(a, b, c, false); b = __builtin_amdgcn_udot4(b, c, d, false); c = __builtin_amdgcn_udot4(c, d, e, false); d = __builtin_amdgcn_udot4(d, e, f, false);
                           e = __builtin_amdgcn_udot4(e, f, g, false); f = __builtin_amdgcn_udot4(f, g, h, false); g = __builtin_amdgcn_udot4(g, h, a, false); h = __builtin_amdgcn_udot4(h, a, b, false); }
            if (OP == 5) { a = __builtin_amdgcn_sad_u8(a, b, c); b = __builtin_amdgcn_sad_u8(b, c, d); c = __builtin_amdgcn_sad_u8(c, d, e); d = __builtin_amdgcn_sad_u8(d, e, f);
                           e = __builtin_amdgcn_sad_u8(e, f, g); f = __builtin_amdgcn_sad_u8(f, g, h); g = __builtin_amdgcn_sad_u8(g, h, a); h = __builtin_amdgcn_sad_u8(h, a, b); }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + e + f + g + h;
}
template <int OP> void run(const char *name, int opsPerIter)
{
    unsigned *out; hipMalloc(&out, 256 * 8 * 256 * 4 * 4);
    const int iters = 2000, blocks = 256 * 8;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(out, 10);
    hipEventRecord(e0); k<OP><<<blocks, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double waveInstr = (double)blocks * 4 * iters * 16 * opsPerIter;
    printf("%-10s %.3f ms  %.2f cycles per wave-instr per SIMD @2.4GHz (lower bound if clocks are lower)\n", name, ms, ms * 1e-3 * 2.4e9 * 1024 / waveInstr);
    hipFree(out);
}
int main() { run<0>("bcnt+add", 8); run<1>("xor", 8); run<2>("min+add", 16); run<3>("perm", 8); run<4>("dot4", 8); run<5>("sad_u8", 8); return 0; }
