// micro-benchmark (tools only): issue rate of the VALU op classes the extractor / matcher kernels are made of, on gfx950 (wave64).
//
// Every class runs as 8 independent dependency chains per wave, 8 waves per SIMD (2048 workgroups of 256 threads on 256 CUs x 4 SIMDs),
// so the number that comes out is ISSUE throughput, not latency.  Cycles are SHADER cycles read with s_memtime inside the kernel (lane 0
// of every wave, first to last instruction of its loop), wall time is s_memrealtime (100 MHz, constant): the shader clock the run really
// had is their ratio, so nothing here assumes a clock.  Output (profiles/r04_valu_issue.txt):
//   class, cycles per wave-instruction per SIMD (from the kernel's duration), chip rate in G wave-instr/s, clock, the waves' own loop time
// bench.py's roofline_valu takes its peak from this file's "integer byte/packed" classes; the guide's figure for
// comparison is 2 cycles per wave64 instruction per SIMD (MI355X_MICROARCH.md, "Wave scheduling").
// Round 5 added 36 opcodes (f32 / f16 min and max, the 16-bit integer forms with two and three operands, SDWA byte selects, v_cndmask, v_addc, ...).
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_valu tools/ubench_valu.hip && tools/ubench_valu
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pkmin(unsigned a, unsigned b) { union { unsigned u; u16x2 v; } x, y, z; x.u = a; y.u = b; z.v = __builtin_elementwise_min(x.v, y.v); return z.u; }
__device__ __forceinline__ unsigned udot2(unsigned a, unsigned b, unsigned c) { union { unsigned u; u16x2 v; } x, y; x.u = a; y.u = b; return __builtin_amdgcn_udot2(x.v, y.v, c, false); }

enum { OP_BCNT = 0, OP_XOR, OP_ADD, OP_MIN3, OP_PKMIN, OP_PERM, OP_ALIGN, OP_DOT4, OP_DOT2, OP_SAD, OP_LSHLOR, OP_MAD24, OP_FADD, OP_FFMA, OP_PKFMA,
       OP_AND, OP_LSHL, OP_MIN, OP_MED3, OP_BFE, OP_ADD3, OP_LSHLADD, OP_ANDOR, OP_PKSUB, OP_PKMAX, OP_MULLO, OP_SUB,
       OP_FMIN, OP_FMAX, OP_PKMINH, OP_PKMAXH, OP_PKADDU16, OP_PKADDH, OP_OR, OP_MOV, OP_LSHR, OP_MAXU, OP_MINH, OP_MINU16, OP_NOT, OP_CVTUB, OP_FMUL, OP_PKFMAH, OP_MAX3F, OP_ALIGNBIT,
       OP_MIN3U16, OP_MAX3U16, OP_MED3U16, OP_MAXU16, OP_MAXI16, OP_SUBU16, OP_ADDU16, OP_MADU16, OP_LSHLB16, OP_CNDMASK, OP_ADDC, OP_BFI, OP_MINI32, OP_MINU16SDWA, OP_ADDSDWA, OP_MULU16, OP_MIN3F16, OP_CMPCND, OP_COUNT };
static const char *kNames[OP_COUNT] = {"v_bcnt_u32_b32", "v_xor_b32", "v_add_u32", "v_min3_u32", "v_pk_min_u16", "v_perm_b32", "v_alignbyte_b32", "v_dot4_u32_u8", "v_dot2_u32_u16",
                                       "v_sad_u8", "v_lshl_or_b32", "v_mad_u32_u24", "v_add_f32", "v_fma_f32", "v_pk_fma_f32",
                                       "v_and_b32", "v_lshlrev_b32", "v_min_u32", "v_med3_i32", "v_bfe_u32", "v_add3_u32", "v_lshl_add_u32", "v_and_or_b32", "v_pk_sub_i16", "v_pk_max_i16",
                                       "v_mul_lo_u32", "v_sub_u32",
                                       "v_min_f32", "v_max_f32", "v_pk_min_f16", "v_pk_max_f16", "v_pk_add_u16", "v_pk_add_f16", "v_or_b32", "v_mov_b32", "v_lshrrev_b32", "v_max_u32",
                                       "v_min_f16", "v_min_u16", "v_not_b32", "v_cvt_f32_ubyte0", "v_mul_f32", "v_pk_fma_f16", "v_max3_f32", "v_alignbit_b32",
                                       "v_min3_u16", "v_max3_u16", "v_med3_u16", "v_max_u16", "v_max_i16", "v_sub_u16", "v_add_u16", "v_mad_u16", "v_lshlrev_b16", "v_cndmask_b32", "v_addc_co_u32",
                                       "v_bfi_b32", "v_min_i32", "v_min_u16_sdwa(bytes)", "v_add_u32_sdwa(bytes)", "v_mul_lo_u16", "v_min3_f16", "v_cmp_gt_u16+v_cndmask"};

// every class is ONE named instruction (inline asm: the optimiser neither folds the chains nor picks another opcode)
#define ASM2(NAME) { unsigned r; asm volatile(NAME " %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
#define ASM3(NAME) { unsigned r; asm volatile(NAME " %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
template <int OP> __device__ __forceinline__ unsigned op(unsigned a, unsigned b, unsigned c)
{
    if (OP == OP_BCNT) ASM2("v_bcnt_u32_b32")
    if (OP == OP_XOR) ASM2("v_xor_b32")
    if (OP == OP_ADD) ASM2("v_add_u32")
    if (OP == OP_MIN3) ASM3("v_min3_u32")
    if (OP == OP_PKMIN) ASM2("v_pk_min_u16")
    if (OP == OP_PERM) ASM3("v_perm_b32")
    if (OP == OP_ALIGN) { unsigned r; asm volatile("v_alignbyte_b32 %0, %1, %2, 1" : "=v"(r) : "v"(a), "v"(b)); return r; }
    if (OP == OP_DOT4) ASM3("v_dot4_u32_u8")
    if (OP == OP_DOT2) ASM3("v_dot2_u32_u16")
    if (OP == OP_SAD) ASM3("v_sad_u8")
    if (OP == OP_LSHLOR) { unsigned r; asm volatile("v_lshl_or_b32 %0, %1, 8, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
    if (OP == OP_MAD24) ASM3("v_mad_u32_u24")
    if (OP == OP_FADD) ASM2("v_add_f32")
    if (OP == OP_FFMA) ASM3("v_fma_f32")
    if (OP == OP_AND) ASM2("v_and_b32")
    if (OP == OP_LSHL) { unsigned r; asm volatile("v_lshlrev_b32 %0, 3, %1" : "=v"(r) : "v"(a)); return r ^ 0; }
    if (OP == OP_MIN) ASM2("v_min_u32")
    if (OP == OP_MED3) ASM3("v_med3_i32")
    if (OP == OP_BFE) { unsigned r; asm volatile("v_bfe_u32 %0, %1, 3, 8" : "=v"(r) : "v"(a)); return r; }
    if (OP == OP_ADD3) ASM3("v_add3_u32")
    if (OP == OP_LSHLADD) { unsigned r; asm volatile("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
    if (OP == OP_ANDOR) ASM3("v_and_or_b32")
    if (OP == OP_PKSUB) ASM2("v_pk_sub_i16")
    if (OP == OP_PKMAX) ASM2("v_pk_max_i16")
    if (OP == OP_MULLO) ASM2("v_mul_lo_u32")
    if (OP == OP_SUB) ASM2("v_sub_u32")
    if (OP == OP_FMIN) ASM2("v_min_f32")
    if (OP == OP_FMAX) ASM2("v_max_f32")
    if (OP == OP_PKMINH) ASM2("v_pk_min_f16")
    if (OP == OP_PKMAXH) ASM2("v_pk_max_f16")
    if (OP == OP_PKADDU16) ASM2("v_pk_add_u16")
    if (OP == OP_PKADDH) ASM2("v_pk_add_f16")
    if (OP == OP_OR) ASM2("v_or_b32")
    if (OP == OP_MOV) { unsigned r; asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(a)); return r; }
    if (OP == OP_LSHR) { unsigned r; asm volatile("v_lshrrev_b32 %0, 1, %1" : "=v"(r) : "v"(a)); return r; }
    if (OP == OP_MAXU) ASM2("v_max_u32")
    if (OP == OP_MINH) ASM2("v_min_f16")
    if (OP == OP_MINU16) ASM2("v_min_u16")
    if (OP == OP_NOT) { unsigned r; asm volatile("v_not_b32 %0, %1" : "=v"(r) : "v"(a)); return r; }
    if (OP == OP_CVTUB) { unsigned r; asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(r) : "v"(a)); return r; }
    if (OP == OP_FMUL) ASM2("v_mul_f32")
    if (OP == OP_PKFMAH) ASM3("v_pk_fma_f16")
    if (OP == OP_MAX3F) ASM3("v_max3_f32")
    if (OP == OP_MIN3U16) ASM3("v_min3_u16")
    if (OP == OP_MAX3U16) ASM3("v_max3_u16")
    if (OP == OP_MED3U16) ASM3("v_med3_u16")
    if (OP == OP_MAXU16) ASM2("v_max_u16")
    if (OP == OP_MAXI16) ASM2("v_max_i16")
    if (OP == OP_SUBU16) ASM2("v_sub_u16")
    if (OP == OP_ADDU16) ASM2("v_add_u16")
    if (OP == OP_MADU16) ASM3("v_mad_u16")
    if (OP == OP_LSHLB16) { unsigned r; asm volatile("v_lshlrev_b16 %0, 3, %1" : "=v"(r) : "v"(a)); return r; }
    if (OP == OP_CNDMASK) { unsigned r; const unsigned long long sel = 0x5555aaaa3333ccccull; asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(sel)); return r; }
    if (OP == OP_ADDC) { unsigned r; asm volatile("v_addc_co_u32 %0, vcc, %1, %2, vcc" : "=v"(r) : "v"(a), "v"(b) : "vcc"); return r; }
    if (OP == OP_BFI) ASM3("v_bfi_b32")
    if (OP == OP_MINI32) ASM2("v_min_i32")
    if (OP == OP_MINU16SDWA) { unsigned r; asm volatile("v_min_u16_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2" : "=v"(r) : "v"(a), "v"(b)); return r; }
    if (OP == OP_ADDSDWA) { unsigned r; asm volatile("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2" : "=v"(r) : "v"(a), "v"(b)); return r; }
    if (OP == OP_MULU16) ASM2("v_mul_lo_u16")
    if (OP == OP_MIN3F16) ASM3("v_min3_f16")
    if (OP == OP_CMPCND) { unsigned r; asm volatile("v_cmp_gt_u16 vcc, %1, %2\n\tv_cndmask_b32 %0, %1, %2, vcc" : "=v"(r) : "v"(a), "v"(b) : "vcc"); return r; }
    if (OP == OP_ALIGNBIT) { unsigned r; asm volatile("v_alignbit_b32 %0, %1, %2, 1" : "=v"(r) : "v"(a), "v"(b)); return r; }
    return a;
}
__device__ __forceinline__ unsigned long long pkfma(unsigned long long a, unsigned long long b, unsigned long long c)
{
    unsigned long long r;
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

template <int OP> __global__ __launch_bounds__(256, 8) void k(unsigned *out, unsigned long long *stamps, int iters)
{
    unsigned v[8];
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = (threadIdx.x + 1) * 2654435761u + 0x9e3779b9u * i;
    unsigned long long pv[8];
#pragma unroll
    for (int i = 0; i < 8; i++) pv[i] = ((unsigned long long)__float_as_uint(1.0f + i) << 32) | __float_as_uint(0.5f + threadIdx.x);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            if (OP == OP_PKFMA) {
#pragma unroll
                for (int i = 0; i < 8; i++) pv[i] = pkfma(pv[i], pv[(i + 1) & 7], pv[(i + 2) & 7]);
            } else {
                unsigned n[8];
#pragma unroll
                for (int i = 0; i < 8; i++) n[i] = op<OP>(v[i], v[(i + 1) & 7], v[(i + 2) & 7]);      // 8 independent instructions
#pragma unroll
                for (int i = 0; i < 8; i++) v[i] = n[i];
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    unsigned s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += v[i] + (unsigned)(pv[i] >> 32) + (unsigned)pv[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
        stamps[2 * w] = t1 - t0; stamps[2 * w + 1] = r1 - r0;
    }
}

template <int OP> void run(unsigned *out, unsigned long long *stamps, std::vector<unsigned long long> &host)
{
    const int iters = 4000, blocks = 256 * 8, waves = blocks * 4;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(out, stamps, 50);
    hipEventRecord(e0);
    k<OP><<<blocks, 256>>>(out, stamps, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(host.data(), stamps, sizeof(unsigned long long) * 2 * waves, hipMemcpyDeviceToHost);
    std::vector<double> cyc(waves), clk(waves);
    for (int w = 0; w < waves; w++) { cyc[w] = (double)host[2 * w]; clk[w] = host[2 * w + 1] ? (double)host[2 * w] / ((double)host[2 * w + 1] / 100e6) : 0; }
    std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
    const double instrPerWave = (double)iters * 16 * 8, medCyc = cyc[waves / 2], medClk = clk[waves / 2];
    // every SIMD executes 8 waves' worth of the loop, in however many rounds the residency allows: cycles per wave-instruction per SIMD from
    // the KERNEL's duration (HIP events x the clock the waves measured); the waves' own loop time is the cross-check (= 8 x instr x cpi when
    // all 8 waves of a SIMD are resident together)
    const double cpi = (double)ms * 1e-3 * medClk / (8.0 * instrPerWave);
    const double chip = 1024.0 * medClk / cpi / 1e9;       // 256 CUs x 4 SIMDs
    printf("%-18s %6.3f cycles per wave-instr per SIMD  chip %7.1f G wave-instr/s  (kernel %.3f ms, clock %.3f GHz, wave loop %9.0f cycles = %.2f of the kernel, %9.0f instr/wave)\n",
           kNames[OP], cpi, chip, ms, medClk / 1e9, medCyc, medCyc / ((double)ms * 1e-3 * medClk), instrPerWave);
}

int main()
{
    const int blocks = 256 * 8, waves = blocks * 4;
    unsigned *out;
    unsigned long long *stamps;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipMalloc(&stamps, sizeof(unsigned long long) * 2 * waves);
    std::vector<unsigned long long> host(2 * waves);
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("# %s, %d CUs, clockRate %d kHz (driver), 8 waves per SIMD, 8 independent chains per wave; cycles = s_memtime, clock = s_memtime / s_memrealtime(100 MHz)\n", p.name,
           p.multiProcessorCount, p.clockRate);
    run<OP_BCNT>(out, stamps, host); run<OP_XOR>(out, stamps, host); run<OP_ADD>(out, stamps, host); run<OP_MIN3>(out, stamps, host); run<OP_PKMIN>(out, stamps, host);
    run<OP_PERM>(out, stamps, host); run<OP_ALIGN>(out, stamps, host); run<OP_DOT4>(out, stamps, host); run<OP_DOT2>(out, stamps, host); run<OP_SAD>(out, stamps, host);
    run<OP_LSHLOR>(out, stamps, host); run<OP_MAD24>(out, stamps, host); run<OP_FADD>(out, stamps, host); run<OP_FFMA>(out, stamps, host); run<OP_PKFMA>(out, stamps, host);
    run<OP_AND>(out, stamps, host); run<OP_LSHL>(out, stamps, host); run<OP_MIN>(out, stamps, host); run<OP_MED3>(out, stamps, host); run<OP_BFE>(out, stamps, host);
    run<OP_ADD3>(out, stamps, host); run<OP_LSHLADD>(out, stamps, host); run<OP_ANDOR>(out, stamps, host); run<OP_PKSUB>(out, stamps, host); run<OP_PKMAX>(out, stamps, host);
    run<OP_MULLO>(out, stamps, host); run<OP_SUB>(out, stamps, host);
    run<OP_FMIN>(out, stamps, host); run<OP_FMAX>(out, stamps, host); run<OP_PKMINH>(out, stamps, host); run<OP_PKMAXH>(out, stamps, host); run<OP_PKADDU16>(out, stamps, host);
    run<OP_PKADDH>(out, stamps, host); run<OP_OR>(out, stamps, host); run<OP_MOV>(out, stamps, host); run<OP_LSHR>(out, stamps, host); run<OP_MAXU>(out, stamps, host);
    run<OP_MINH>(out, stamps, host); run<OP_MINU16>(out, stamps, host); run<OP_NOT>(out, stamps, host); run<OP_CVTUB>(out, stamps, host); run<OP_FMUL>(out, stamps, host);
    run<OP_PKFMAH>(out, stamps, host); run<OP_MAX3F>(out, stamps, host); run<OP_ALIGNBIT>(out, stamps, host);
    run<OP_MIN3U16>(out, stamps, host); run<OP_MAX3U16>(out, stamps, host); run<OP_MED3U16>(out, stamps, host); run<OP_MAXU16>(out, stamps, host); run<OP_MAXI16>(out, stamps, host);
    run<OP_SUBU16>(out, stamps, host); run<OP_ADDU16>(out, stamps, host); run<OP_MADU16>(out, stamps, host); run<OP_LSHLB16>(out, stamps, host); run<OP_CNDMASK>(out, stamps, host);
    run<OP_ADDC>(out, stamps, host); run<OP_BFI>(out, stamps, host); run<OP_MINI32>(out, stamps, host); run<OP_MINU16SDWA>(out, stamps, host); run<OP_ADDSDWA>(out, stamps, host);
    run<OP_MULU16>(out, stamps, host); run<OP_MIN3F16>(out, stamps, host); run<OP_CMPCND>(out, stamps, host);
    return 0;
}
