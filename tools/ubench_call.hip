// micro-benchmark (tools only): the fixed cost of ONE small host call - inputs of 64 KB up, a 5 us kernel, 8 KB of results down - in the forms the
// single-call entry points can take.  Output committed as profiles/r06_call_floor.txt.
//   hipcc --offload-arch=gfx950 -O3 -o tools/_build/ubench_call tools/ubench_call.hip && tools/_build/ubench_call
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
__global__ void k_work(const uint32_t *in, uint32_t *out, int n, int spin)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t v = i < n ? in[i] : 0;
    for (int k = 0; k < spin; k++) v = v * 1664525u + 1013904223u;
    if (i < 2048) out[i] = v;
}
__global__ void k_work_pub(const uint32_t *in, uint32_t *out, int n, int spin, unsigned *counter, unsigned long long *flag, unsigned long long seq)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t v = i < n ? in[i] : 0;
    for (int k = 0; k < spin; k++) v = v * 1664525u + 1013904223u;
    if (i < 2048) out[i] = v;
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0 && atomicAdd(counter, 1u) == gridDim.x - 1) {
        *counter = 0;
        __threadfence_system();
        __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
__global__ void k_work_pub2(const uint32_t *in, uint32_t *out, int n, int spin, unsigned *counter, unsigned long long *flag, unsigned long long seq)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t v = i < n ? in[i] : 0;
    for (int k = 0; k < spin; k++) v = v * 1664525u + 1013904223u;
    if (i < 2048) out[i] = v;
    __syncthreads();      // (every wave's stores are issued)
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (__hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) {
            __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
__global__ void k_publish(const uint32_t *src, uint32_t *dst, int n, unsigned long long *flag, unsigned long long seq)
{
    for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
static double med(std::vector<double> &v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }
int main()
{
    const int N = 16384, NB = N * 4, OUTB = 8192, ITER = 400, SPIN = 600;
    uint32_t *dIn, *dOut, *hPinIn, *hMapIn, *hMapOut, *dMapIn, *dMapOut;
    unsigned *dCounter;
    unsigned long long *hFlag, *dFlag;
    hipMalloc(&dIn, NB); hipMalloc(&dOut, OUTB); hipMalloc(&dCounter, 4); hipMemset(dCounter, 0, 4);
    hipHostMalloc(&hPinIn, NB, hipHostMallocDefault);
    hipHostMalloc(&hMapIn, NB, hipHostMallocMapped); hipHostGetDevicePointer((void **)&dMapIn, hMapIn, 0);
    hipHostMalloc(&hMapOut, OUTB, hipHostMallocMapped); hipHostGetDevicePointer((void **)&dMapOut, hMapOut, 0);
    hipHostMalloc(&hFlag, 64, hipHostMallocMapped); hipHostGetDevicePointer((void **)&dFlag, hFlag, 0);
    *hFlag = 0;
    std::vector<uint32_t> pageIn(N, 7), pageOut(OUTB / 4), pageOut2(OUTB / 4);
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    unsigned long long seq = 0;
    auto spinwait = [&](unsigned long long want) { while (__atomic_load_n(hFlag, __ATOMIC_ACQUIRE) < want) __builtin_ia32_pause(); };
    const char *names[] = {"A  pageable H2D async, kernel, hipStreamSynchronize, 2 x hipMemcpy D2H pageable (round-5 shape)",
                           "B  memcpy to pinned + hipMemcpyAsync H2D, kernel, hipStreamSynchronize, 2 x hipMemcpy D2H",
                           "C  pinned H2D async, kernel, hipMemcpyAsync D2H into pinned, hipStreamSynchronize, memcpy out",
                           "D  pinned H2D async, kernel, k_publish (copy to mapped host + flag), host spins on the flag, memcpy out",
                           "E  pinned H2D async, kernel writes mapped host itself + last-arriver flag, host spins, memcpy out",
                           "F  kernel READS mapped host input (zero copy) and writes mapped host + flag, host spins, memcpy out",
                           "G  as E with three kernels in the chain (two more 5 us kernels)",
                           "H  as A with three kernels in the chain",
                           "I  as F, completion by hipStreamSynchronize instead of the flag (no fences, no counter)",
                           "J  as F, agent-scope fence per workgroup, system-scope release only at the flag",
                           "K  as J with three kernels in the chain",
                           "L  as E (pinned H2D async), agent-scope fences"};
    for (int mode = 0; mode < 12; mode++) {
        std::vector<double> t;
        for (int it = 0; it < ITER + 20; it++) {
            pageIn[0] = (uint32_t)it;
            const auto t0 = std::chrono::steady_clock::now();
            if (mode == 0 || mode == 7) {
                hipMemcpyAsync(dIn, pageIn.data(), NB, hipMemcpyHostToDevice, s);
                k_work<<<N / 256, 256, 0, s>>>(dIn, dOut, N, SPIN);
                if (mode == 7) { k_work<<<N / 256, 256, 0, s>>>(dIn, dOut, N, SPIN); k_work<<<N / 256, 256, 0, s>>>(dIn, dOut, N, SPIN); }
                hipStreamSynchronize(s);
                hipMemcpy(pageOut.data(), dOut, OUTB / 2, hipMemcpyDeviceToHost);
                hipMemcpy(pageOut2.data(), dOut + OUTB / 8, OUTB / 2, hipMemcpyDeviceToHost);
            } else if (mode == 1) {
                memcpy(hPinIn, pageIn.data(), NB);
                hipMemcpyAsync(dIn, hPinIn, NB, hipMemcpyHostToDevice, s);
                k_work<<<N / 256, 256, 0, s>>>(dIn, dOut, N, SPIN);
                hipStreamSynchronize(s);
                hipMemcpy(pageOut.data(), dOut, OUTB / 2, hipMemcpyDeviceToHost);
                hipMemcpy(pageOut2.data(), dOut + OUTB / 8, OUTB / 2, hipMemcpyDeviceToHost);
            } else if (mode == 2) {
                memcpy(hPinIn, pageIn.data(), NB);
                hipMemcpyAsync(dIn, hPinIn, NB, hipMemcpyHostToDevice, s);
                k_work<<<N / 256, 256, 0, s>>>(dIn, dOut, N, SPIN);
                hipMemcpyAsync(hMapOut, dOut, OUTB, hipMemcpyDeviceToHost, s);
                hipStreamSynchronize(s);
                memcpy(pageOut.data(), hMapOut, OUTB);
            } else if (mode == 3) {
                memcpy(hPinIn, pageIn.data(), NB);
                hipMemcpyAsync(dIn, hPinIn, NB, hipMemcpyHostToDevice, s);
                k_work<<<N / 256, 256, 0, s>>>(dIn, dOut, N, SPIN);
                k_publish<<<1, 1024, 0, s>>>(dOut, dMapOut, OUTB / 4, dFlag, ++seq);
                spinwait(seq);
                memcpy(pageOut.data(), hMapOut, OUTB);
            } else if (mode == 4 || mode == 6) {
                memcpy(hPinIn, pageIn.data(), NB);
                hipMemcpyAsync(dIn, hPinIn, NB, hipMemcpyHostToDevice, s);
                if (mode == 6) { k_work<<<N / 256, 256, 0, s>>>(dIn, dOut, N, SPIN); k_work<<<N / 256, 256, 0, s>>>(dIn, dOut, N, SPIN); }
                k_work_pub<<<N / 256, 256, 0, s>>>(dIn, dMapOut, N, SPIN, dCounter, dFlag, ++seq);
                spinwait(seq);
                memcpy(pageOut.data(), hMapOut, OUTB);
            } else if (mode == 5) {
                memcpy(hMapIn, pageIn.data(), NB);
                k_work_pub<<<N / 256, 256, 0, s>>>(dMapIn, dMapOut, N, SPIN, dCounter, dFlag, ++seq);
                spinwait(seq);
                memcpy(pageOut.data(), hMapOut, OUTB);
            }
            else if (mode == 8) {
                memcpy(hMapIn, pageIn.data(), NB);
                k_work<<<N / 256, 256, 0, s>>>(dMapIn, dMapOut, N, SPIN);
                hipStreamSynchronize(s);
                memcpy(pageOut.data(), hMapOut, OUTB);
            } else if (mode == 9 || mode == 10) {
                memcpy(hMapIn, pageIn.data(), NB);
                if (mode == 10) { k_work<<<N / 256, 256, 0, s>>>(dMapIn, dOut, N, SPIN); k_work<<<N / 256, 256, 0, s>>>(dIn, dOut, N, SPIN); }
                k_work_pub2<<<N / 256, 256, 0, s>>>(dMapIn, dMapOut, N, SPIN, dCounter, dFlag, ++seq);
                spinwait(seq);
                memcpy(pageOut.data(), hMapOut, OUTB);
            } else if (mode == 11) {
                memcpy(hPinIn, pageIn.data(), NB);
                hipMemcpyAsync(dIn, hPinIn, NB, hipMemcpyHostToDevice, s);
                k_work_pub2<<<N / 256, 256, 0, s>>>(dIn, dMapOut, N, SPIN, dCounter, dFlag, ++seq);
                spinwait(seq);
                memcpy(pageOut.data(), hMapOut, OUTB);
            }
            const auto t1 = std::chrono::steady_clock::now();
            if (it >= 20) t.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
        }
        hipStreamSynchronize(s);
        double mean = 0; for (double x : t) mean += x; mean /= t.size();
        printf("%-110s median %7.1f us  mean %7.1f us\n", names[mode], med(t), mean);
    }
    // the kernel alone
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, s);
    for (int i = 0; i < 100; i++) k_work<<<N / 256, 256, 0, s>>>(dIn, dOut, N, SPIN);
    hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("k_work back to back: %.2f us per launch\n", ms * 10);
    return 0;
}
