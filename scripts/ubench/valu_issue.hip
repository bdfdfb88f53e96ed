// VALU issue-rate microbenchmark for gfx950: how many cycles does one SIMD need per wave64
// vector instruction at 1, 2, 4, 8 resident waves per SIMD?  No memory traffic inside the timed loop.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/valu_issue.hip -o /tmp/valu_issue && /tmp/valu_issue
// Prints one JSON object: per mode and per waves/SIMD the cycles per wave-instruction per SIMD
// (= median over waves of (s_memtime delta) / (instructions per wave) / waves per SIMD) and the wall-clock figure.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

enum { M_FMA_INDEP = 0, M_FMA_DEP, M_PK_FMA, M_EXP, M_FMA_SGPR, M_CMP_CNDMASK, M_DPP, M_FMA_MFMA_F32, M_FMA_MFMA_BF16,
       M_MFMA_F32_ONLY, M_MFMA_BF16_ONLY, M_ADD_F64, M_FMA_SALU, M_COUNT };
static const char* kNames[M_COUNT] = {"v_fma_f32 x16 independent", "v_fma_f32 dependent chain", "v_pk_fma_f32 x8 independent (2 flop-lanes)",
    "v_exp_f32 x16 independent", "v_fma_f32 with SGPR operand", "v_cmp_lt_f32 + v_cndmask_b32 pairs", "v_add_f32 DPP row_shr:1",
    "16 v_fma_f32 + 1 v_mfma_f32_16x16x4_f32", "16 v_fma_f32 + 1 v_mfma_f32_16x16x32_bf16", "v_mfma_f32_16x16x4_f32 only (x4 acc)",
    "v_mfma_f32_16x16x32_bf16 only (x4 acc)", "v_add_f64 x8 independent", "16 v_fma_f32 + 8 s_add_u32"};
// vector instructions per loop trip (what the cycles are divided by)
static const int kPerTrip[M_COUNT] = {64, 64, 32, 64, 64, 64, 64, 68, 68, 16, 16, 32, 64};

#define R16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int MODE>
__global__ __launch_bounds__(256) void bench(float* out, unsigned long long* dt, unsigned long long* rt, int trips, float seed) {
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = seed + threadIdx.x * 1e-3f + i;
    float b = 1.0000001f, c = 1e-7f;
    float sb = __builtin_amdgcn_readfirstlane(b);
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    bf16x8 bfa = {1, 2, 3, 4, 5, 6, 7, 8}, bfb = {8, 7, 6, 5, 4, 3, 2, 1};
    double d[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) d[i] = seed + i;
    unsigned s0 = blockIdx.x, s1 = 1;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    for (int t = 0; t < trips; ++t) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if constexpr (MODE == M_FMA_INDEP) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                R16(X)
#undef X
            } else if constexpr (MODE == M_FMA_DEP) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(b), "v"(c));
                R16(X)
#undef X
            } else if constexpr (MODE == M_PK_FMA) {
#define X(i) if (i < 8) { typedef float f2 __attribute__((ext_vector_type(2))); f2 v = {a[2 * (i & 7)], a[2 * (i & 7) + 1]}; f2 bb = {b, b}, cc = {c, c}; \
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(bb), "v"(cc)); a[2 * (i & 7)] = v.x; a[2 * (i & 7) + 1] = v.y; }
                R16(X)
#undef X
            } else if constexpr (MODE == M_EXP) {
#define X(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
                R16(X)
#undef X
            } else if constexpr (MODE == M_FMA_SGPR) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(sb), "v"(c));
                R16(X)
#undef X
            } else if constexpr (MODE == M_CMP_CNDMASK) {
#define X(i) if (i < 8) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(a[i]) : "v"(b), "v"(c) : "vcc");
                R16(X) R16(X)
#undef X
            } else if constexpr (MODE == M_DPP) {
#define X(i) asm volatile("v_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(c));
                R16(X)
#undef X
            } else if constexpr (MODE == M_FMA_MFMA_F32) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                R16(X)
#undef X
                acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, c, acc[u], 0, 0, 0);
            } else if constexpr (MODE == M_FMA_MFMA_BF16) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                R16(X)
#undef X
                acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfa, bfb, acc[u], 0, 0, 0);
            } else if constexpr (MODE == M_MFMA_F32_ONLY) {
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, c, acc[q], 0, 0, 0);
            } else if constexpr (MODE == M_MFMA_BF16_ONLY) {
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfa, bfb, acc[q], 0, 0, 0);
            } else if constexpr (MODE == M_ADD_F64) {
#define X(i) if (i < 8) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
                R16(X)
#undef X
            } else if constexpr (MODE == M_FMA_SALU) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); \
             if (i & 1) asm volatile("s_add_u32 %0, %0, %1" : "+s"(s0) : "s"(s1) : "scc");
                R16(X)
#undef X
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
#pragma unroll
    for (int q = 0; q < 4; ++q) s += acc[q].x + acc[q].y + acc[q].z + acc[q].w;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += (float)d[i];
    s += (float)s0;
    if (s == 12345.678f) out[0] = s;                      // keep everything alive
    if ((threadIdx.x & 63) == 0) {
        int w = blockIdx.x * 4 + (threadIdx.x >> 6);
        dt[w] = t1 - t0;
        rt[w] = r1 - r0;
    }
}

typedef void (*kern_t)(float*, unsigned long long*, unsigned long long*, int, float);

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    kern_t ks[M_COUNT] = {bench<0>, bench<1>, bench<2>, bench<3>, bench<4>, bench<5>, bench<6>, bench<7>, bench<8>, bench<9>, bench<10>,
                          bench<11>, bench<12>};
    const int trips = 4000;
    const int maxw = cus * 4 * 8;
    float* out; unsigned long long *dt, *rt;
    CK(hipMalloc(&out, 64)); CK(hipMalloc(&dt, maxw * 8)); CK(hipMalloc(&rt, maxw * 8));
    std::vector<unsigned long long> h(maxw), hr(maxw);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("{\"device\": \"%s\", \"cus\": %d, \"trips\": %d, \"note\": \"cycles = s_memtime ticks (shader clock); per_simd = median wave delta / instr per wave / waves per SIMD; "
           "blocks of 256 threads = one wave per SIMD each, waves/SIMD = blocks per CU\", \"modes\": [\n", prop.name, cus, trips);
    for (int m = 0; m < M_COUNT; ++m) {
        printf("  {\"mode\": \"%s\", \"vector_instr_per_trip\": %d, \"by_waves_per_simd\": {", kNames[m], kPerTrip[m]);
        int wlist[4] = {1, 2, 4, 8};
        for (int wi = 0; wi < 4; ++wi) {
            int w = wlist[wi];
            int blocks = cus * w;
            for (int rep = 0; rep < 2; ++rep) {               // first rep warms up
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(ks[m], dim3(blocks), dim3(256), 0, 0, out, dt, rt, trips, 1.0f);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
            }
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(h.data(), dt, blocks * 4 * 8, hipMemcpyDeviceToHost));
            CK(hipMemcpy(hr.data(), rt, blocks * 4 * 8, hipMemcpyDeviceToHost));
            std::sort(h.begin(), h.begin() + blocks * 4);
            std::sort(hr.begin(), hr.begin() + blocks * 4);
            double med = (double)h[blocks * 2], medr = (double)hr[blocks * 2];
            double n = (double)trips * kPerTrip[m];
            double cyc_per_instr_wave = med / n;                  // what ONE wave sees
            double cyc_per_instr_simd = med / n / w;              // SIMD throughput with w waves resident
            double ghz = med / (medr * 10.0);                     // s_memrealtime ticks at 100 MHz
            printf("%s\"%d\": {\"cyc_per_instr_one_wave\": %.3f, \"cyc_per_instr_per_simd\": %.3f, \"shader_clock_ghz\": %.3f, \"wall_ms\": %.4f, "
                   "\"wall_cyc_per_instr_per_simd_at_that_clock\": %.3f}", wi ? ", " : "", w, cyc_per_instr_wave, cyc_per_instr_simd, ghz, ms,
                   ms * 1e-3 * ghz * 1e9 / (n * w));
        }
        printf("}}%s\n", m + 1 < M_COUNT ? "," : "");
    }
    printf("]}\n");
    return 0;
}
