// Layout check of v_mfma_f64_16x16x4_f64 on gfx950 (used by the LBA Cholesky's panel update): prints OK when
//   A: lane l holds A[l % 16][l / 16], B: lane l holds B[l / 16][l % 16], D: lane l, element v holds D[4 * v + l / 16][l % 16].
//   hipcc --offload-arch=gfx950 -O2 tools/ubench_mfma_f64.hip -o /tmp/ubench_mfma_f64 && /tmp/ubench_mfma_f64
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double double4_t __attribute__((ext_vector_type(4)));
__global__ void k(const double *A, const double *B, double *D)
{
    const int l = threadIdx.x;
    double4_t acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[(l % 16) * 4 + l / 16], B[(l / 16) * 16 + l % 16], acc, 0, 0, 0);
    for (int v = 0; v < 4; v++) D[l * 4 + v] = acc[v];
}
int main()
{
    double hA[64], hB[64], hD[256], *dA, *dB, *dD;
    for (int i = 0; i < 64; i++) { hA[i] = (i * 37 % 11) - 5 + 0.25 * (i % 3); hB[i] = (i * 53 % 13) - 6 + 0.5 * (i % 5); }
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; l++)
        for (int v = 0; v < 4; v++) {
            const int i = 4 * v + l / 16, j = l % 16;
            double ref = 0;
            for (int kk = 0; kk < 4; kk++) ref += hA[i * 4 + kk] * hB[kk * 16 + j];
            if (ref != hD[l * 4 + v]) bad++;
        }
    if (!bad) { printf("OK: layout as documented\n"); return 0; }
    printf("MISMATCH in %d of 256 entries; searching the layout of lane 17, element 1 = %g\n", bad, hD[17 * 4 + 1]);
    for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) { double ref = 0; for (int kk = 0; kk < 4; kk++) ref += hA[i * 4 + kk] * hB[kk * 16 + j]; if (ref == hD[17 * 4 + 1]) printf("  candidate D[%d][%d]\n", i, j); }
    return 1;
}
