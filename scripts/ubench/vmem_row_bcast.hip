// How expensive is it to fetch a 16-lane-row-uniform record with VECTOR loads?  (Per-row entry lists for the blend
// kernels need 4 different records per wave instruction, one per DPP row -- see DESIGN.md "issue-rate price list".)
//   mode 0: global_load_dwordx4, every lane of a 16-lane row reads the SAME 16 bytes, 4 rows -> 4 random records
//   mode 1: global_load_dwordx4, all 64 lanes the same 16 bytes (wave-uniform record through the vector path)
//   mode 2: global_load_dwordx4, plain coalesced stream (lane l reads base + 16*l): the usual best case
//   mode 3: ds_read_b128, 4 rows -> 4 records staged in LDS
// Records: 80 bytes, table of `nrec` records per workgroup region (L1 / L2 resident), 5 loads per "entry".
// Output: CU cycles per load instruction (wall x clock x CUs... per CU) at 1/2/4/8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;

template <int MODE>
__global__ __launch_bounds__(256) void bench(const float* __restrict__ table, float* out, u64* dt, u64* rt, int trips, int nrec) {
    __shared__ f32x4 lds[5 * 256];          // 256 records x 80 B
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 5 * 256; i += 256) lds[i] = f32x4{(float)i, 1.f, 2.f, 3.f};
    __syncthreads();
    const float* base = table + (size_t)(blockIdx.x % 64) * nrec * 20;      // 64 regions of nrec records
    unsigned seed = blockIdx.x * 977u + wave * 131u + (MODE == 0 || MODE == 3 ? (lane >> 4) * 7919u : 0u);
    f32x4 acc = {0, 0, 0, 0};
    u64 t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    for (int t = 0; t < trips; ++t) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            seed = seed * 1664525u + 1013904223u;
            const unsigned idx = (seed >> 10) % (unsigned)nrec;
            if constexpr (MODE == 3) {
                const f32x4* p = lds + (idx & 255) * 5;
#pragma unroll
                for (int k = 0; k < 5; ++k) acc += p[k];
            } else {
                const f32x4* p = MODE == 2 ? (const f32x4*)base + ((idx * 5) & ~63u) + lane : (const f32x4*)(base + (size_t)idx * 20);
#pragma unroll
                for (int k = 0; k < 5; ++k) acc += MODE == 2 ? p[k * 64] : p[k];
            }
        }
    }
    u64 t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
    if (lane == 0) { dt[blockIdx.x * 4 + wave] = t1 - t0; rt[blockIdx.x * 4 + wave] = r1 - r0; }
}
typedef void (*kern_t)(const float*, float*, u64*, u64*, int, int);

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount, trips = 500;
    const char* names[4] = {"global_load_dwordx4, 4 records per wave instr (one per 16-lane row)", "global_load_dwordx4, 1 record per wave instr (all lanes same address)",
                            "global_load_dwordx4, coalesced 1 KiB per wave instr", "ds_read_b128, 4 records per wave instr (LDS)"};
    kern_t ks[4] = {bench<0>, bench<1>, bench<2>, bench<3>};
    float *table, *out; u64 *dt, *rt;
    const int maxrec = 4096;
    CK(hipMalloc(&table, (size_t)64 * maxrec * 80 + (1 << 20))); CK(hipMemset(table, 0, (size_t)64 * maxrec * 80 + (1 << 20)));
    CK(hipMalloc(&out, 64)); CK(hipMalloc(&dt, cus * 32 * 8)); CK(hipMalloc(&rt, cus * 32 * 8));
    std::vector<u64> h(cus * 32), hr(cus * 32);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("{\"device\": \"%s\", \"note\": \"cu_cycles_per_load_instr = wall x shader clock / (load instructions per wave x waves per CU); 80-byte records, 5 x 16-byte loads per record\", \"results\": [\n", prop.name);
    bool first = true;
    for (int m = 0; m < 4; ++m)
        for (int nrec : {256, 4096}) {
            if (m == 3 && nrec != 256) continue;
            printf("%s  {\"mode\": \"%s\", \"records_per_region\": %d, \"region_KiB\": %d, \"cu_cycles_per_load_instr\": {", first ? "" : ",\n", names[m], nrec, nrec * 80 / 1024);
            first = false;
            int wl[4] = {1, 2, 4, 8};
            for (int wi = 0; wi < 4; ++wi) {
                const int w = wl[wi], blocks = cus * w;
                float best = 1e30f;
                for (int rep = 0; rep < 3; ++rep) {
                    CK(hipEventRecord(e0));
                    hipLaunchKernelGGL(ks[m], dim3(blocks), dim3(256), 0, 0, table, out, dt, rt, trips, nrec);
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
                }
                CK(hipMemcpy(h.data(), dt, blocks * 4 * 8, hipMemcpyDeviceToHost));
                CK(hipMemcpy(hr.data(), rt, blocks * 4 * 8, hipMemcpyDeviceToHost));
                std::sort(h.begin(), h.begin() + blocks * 4); std::sort(hr.begin(), hr.begin() + blocks * 4);
                const double ghz = (double)h[blocks * 2] / ((double)hr[blocks * 2] * 10.0);
                const double loads_per_wave = (double)trips * 4 * 5;
                printf("%s\"%d\": %.2f", wi ? ", " : "", w, (best - 0.004) * 1e-3 * ghz * 1e9 / (loads_per_wave * w * 4));
            }
            printf("}}");
        }
    printf("\n]}\n");
    return 0;
}
