// microbench.hip -- two measurements that decide how the traversal kernels are read:
//   1. VALU issue: shader cycles per wave64 v_fma_f32 with 1, 2, 4 waves per SIMD (independent chains), which
//      turns SQ_INSTS_VALU into a share of the SIMDs' issue capacity;
//   2. dependent gathers: wave-iterations per microsecond and CU when every lane chases pointers through a
//      table of 64-byte records (the BVH access pattern), for several fetch shapes: 4, 2 or 1 dwordx4 per lane,
//      half of the lanes masked off, and four adjacent lanes sharing one record (one dwordx4 each).
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/microbench tools/microbench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_valu(float* out, int iters, unsigned long long* cyc) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float b = 1.0001f, c = 0.5f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            asm volatile("v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\t"
                         "v_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                         : "v"(b), "v"(c));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

// Issue cost of other VALU instructions the traversal kernels are made of, 4 waves per SIMD, independent chains.
#define VALU_KERNEL(NAME, ASM)                                                                                    \
    __global__ void NAME(float* out, int iters, unsigned long long* cyc) {                                        \
        float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        const float b = 1.0001f, c = 0.5f;                                                                         \
        const unsigned long long t0 = __builtin_readcyclecounter();                                               \
        for (int i = 0; i < iters; ++i) {                                                                         \
            _Pragma("unroll") for (int r = 0; r < 8; ++r) {                                                       \
                asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                               \
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)     \
                             : "v"(b), "v"(c) : "vcc");                                                          \
            }                                                                                                      \
        }                                                                                                          \
        const unsigned long long t1 = __builtin_readcyclecounter();                                               \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                       \
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;            \
    }
#define A_ADD(k) "v_add_f32 %" #k ", %" #k ", %8\n\t"
#define A_MUL(k) "v_mul_f32 %" #k ", %" #k ", %8\n\t"
#define A_MAX3(k) "v_max3_f32 %" #k ", %" #k ", %8, %9\n\t"
#define A_CND(k) "v_cndmask_b32 %" #k ", %" #k ", %8, vcc\n\t"
#define A_CMP(k) "v_cmp_lt_f32 vcc, %" #k ", %8\n\t"
#define A_CMPS(k) "v_cmp_lt_f32 s[20:21], %" #k ", %8\n\t"
#define A_CVT(k) "v_cvt_f32_ubyte1 %" #k ", %" #k "\n\t"
#define A_AND(k) "v_and_b32 %" #k ", %" #k ", %8\n\t"
#define A_LSH(k) "v_lshlrev_b32 %" #k ", 1, %" #k "\n\t"
#define A_MOV(k) "v_mov_b32 %" #k ", %8\n\t"
#define A_FMA(k) "v_fma_f32 %" #k ", %" #k ", %8, %9\n\t"
#define A_RCP(k) "v_rcp_f32 %" #k ", %" #k "\n\t"
#define A_CNDS(k) "v_cndmask_b32 %" #k ", %" #k ", %8, s[22:23]\n\t"
VALU_KERNEL(k_add, A_ADD)
VALU_KERNEL(k_mul, A_MUL)
VALU_KERNEL(k_max3, A_MAX3)
VALU_KERNEL(k_cnd, A_CND)
VALU_KERNEL(k_cmp, A_CMP)
VALU_KERNEL(k_cvt, A_CVT)
VALU_KERNEL(k_and, A_AND)
VALU_KERNEL(k_lsh, A_LSH)
VALU_KERNEL(k_mov, A_MOV)
VALU_KERNEL(k_fma, A_FMA)
VALU_KERNEL(k_rcp, A_RCP)
#define A_MAX(k) "v_max_f32 %" #k ", %" #k ", %8\n\t"
#define A_PAIR(k) "v_cmp_lt_f32 vcc, %" #k ", %9\n\tv_cndmask_b32 %" #k ", %" #k ", %8, vcc\n\t"
#define A_CNDZ(k) "v_cndmask_b32 %" #k ", %8, %9, vcc\n\t"
VALU_KERNEL(k_max, A_MAX)
VALU_KERNEL(k_pair, A_PAIR)
VALU_KERNEL(k_cndz, A_CNDZ)
__global__ void k_cnds(float* out, int iters, unsigned long long* cyc) {   // select on an SGPR pair (VOP3 form)
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float b = 1.0001f, c = 0.5f;
    asm volatile("s_mov_b64 s[22:23], 0x55555555" ::: "s22", "s23");
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            asm volatile(A_CNDS(0) A_CNDS(1) A_CNDS(2) A_CNDS(3) A_CNDS(4) A_CNDS(5) A_CNDS(6) A_CNDS(7)
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "s22", "s23");
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
__global__ void k_cmps(float* out, int iters, unsigned long long* cyc) {   // compare into an SGPR pair, then a select that reads one
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float b = 1.0001f, c = 0.5f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            asm volatile(A_CMPS(0) A_CMPS(1) A_CMPS(2) A_CMPS(3) A_CMPS(4) A_CMPS(5) A_CMPS(6) A_CMPS(7)
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "s20", "s21");
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

// mode 0: 4 x dwordx4 per lane (64 B record), 1: 2 x dwordx4, 2: 1 x dwordx4, 3: mode 0 with odd lanes idle,
// 4: four adjacent lanes fetch the four quarters of ONE record (one dwordx4 each; 16 records per wave-instruction)
// 5: mode 0 with the records of a wave's lanes adjacent in memory (coherent rays)
// 6: every lane gets its own 64-byte record, fetched quad-cooperatively: load j brings, for each quad, the record of
//    the quad's lane j (each lane one quarter), then a 4x4 transpose inside the quad (DPP) hands every lane its record
// 7: mode 6 with the records of a wave's lanes adjacent in memory
template <int CTRL>
__device__ __forceinline__ uint32_t qperm(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}
template <int CTRL>
__device__ __forceinline__ void quad_exchange(uint32_t& ra, uint32_t& rb, bool upper) {
    const uint32_t send = upper ? ra : rb;
    const uint32_t recv = qperm<CTRL>(send);
    if (upper) ra = recv; else rb = recv;
}
__device__ __forceinline__ void quad_fetch64(const void* rec, uint4& a, uint4& b, uint4& c, uint4& d) {
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t p = (uint64_t)rec;
    const uint32_t lo = (uint32_t)p, hi = (uint32_t)(p >> 32), piece = (lane & 3u) << 4;
    const uint64_t p0 = ((uint64_t)qperm<0x00>(hi) << 32) | (qperm<0x00>(lo) | piece);
    const uint64_t p1 = ((uint64_t)qperm<0x55>(hi) << 32) | (qperm<0x55>(lo) | piece);
    const uint64_t p2 = ((uint64_t)qperm<0xAA>(hi) << 32) | (qperm<0xAA>(lo) | piece);
    const uint64_t p3 = ((uint64_t)qperm<0xFF>(hi) << 32) | (qperm<0xFF>(lo) | piece);
    asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %5, off\n\t"
                 "global_load_dwordx4 %2, %6, off\n\tglobal_load_dwordx4 %3, %7, off\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(p0), "v"(p1), "v"(p2), "v"(p3) : "memory");
    const bool b0 = lane & 1, b1 = lane & 2;
    quad_exchange<0xB1>(a.x, b.x, b0); quad_exchange<0xB1>(a.y, b.y, b0); quad_exchange<0xB1>(a.z, b.z, b0); quad_exchange<0xB1>(a.w, b.w, b0);
    quad_exchange<0xB1>(c.x, d.x, b0); quad_exchange<0xB1>(c.y, d.y, b0); quad_exchange<0xB1>(c.z, d.z, b0); quad_exchange<0xB1>(c.w, d.w, b0);
    quad_exchange<0x4E>(a.x, c.x, b1); quad_exchange<0x4E>(a.y, c.y, b1); quad_exchange<0x4E>(a.z, c.z, b1); quad_exchange<0x4E>(a.w, c.w, b1);
    quad_exchange<0x4E>(b.x, d.x, b1); quad_exchange<0x4E>(b.y, d.y, b1); quad_exchange<0x4E>(b.z, d.z, b1); quad_exchange<0x4E>(b.w, d.w, b1);
}
template <int MODE>
__global__ __launch_bounds__(256) void k_gather(const uint4* __restrict__ table, uint32_t mask, int iters, uint32_t* out) {
    const uint32_t lane = threadIdx.x & 63;
    uint32_t idx = (blockIdx.x * 256 + threadIdx.x) * 2654435761u;
    if (MODE == 4) idx = (blockIdx.x * 64 + (threadIdx.x >> 2)) * 2654435761u;
    uint32_t acc = 0;
    bool wrong = false;
    const bool idle = (MODE == 3) && (lane & 1);
    for (int i = 0; i < iters && !idle; ++i) {
        uint32_t rec = idx & mask;
        if (MODE == 5 || MODE == 7) rec = ((idx & mask) & ~63u) | lane;
        const uint4* p = table + (size_t)rec * 4;
        uint4 a, b = make_uint4(0, 0, 0, 0), c = b, d = b;
        if (MODE == 0 || MODE == 3 || MODE == 5) {
            asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:16\n\t"
                         "global_load_dwordx4 %2, %4, off offset:32\n\tglobal_load_dwordx4 %3, %4, off offset:48\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(p) : "memory");
        } else if (MODE == 6 || MODE == 7) {
            quad_fetch64(p, a, b, c, d);
            // check: the record's 16 words are hash(rec * 16 + w); word 5 must be there after the transpose
            if (b.y != (uint32_t)(((uint64_t)(rec * 16u + 5u) * 0x9E3779B97F4A7C15ull) >> 32)) wrong = true;
        } else if (MODE == 1) {
            asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:16\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(a), "=&v"(b) : "v"(p) : "memory");
        } else if (MODE == 2) {
            asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(a) : "v"(p) : "memory");
        } else {
            const uint4* q = p + (lane & 3);
            asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(a) : "v"(q) : "memory");
            // all four lanes of the quad continue with the same record: take the first quarter's word
            a.x = __shfl(a.x, lane & ~3u);
        }
        acc += a.y ^ b.z ^ c.w ^ d.x;
        idx = a.x * 2654435761u + (uint32_t)i;     // the next record depends on the data just fetched
    }
    out[blockIdx.x * 256 + threadIdx.x] = wrong ? 0xDEADBEEFu : ((acc + idx) & 0x7fffffffu);
}

template <int MODE>
int run_gather(const char* name, const uint4* table, uint32_t mask, int blocks_per_cu, int cus, uint32_t* out) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int grid = blocks_per_cu * cus;
    hipLaunchKernelGGL(k_gather<MODE>, dim3(grid), dim3(256), 0, 0, table, mask, 200, out);
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_gather<MODE>, dim3(grid), dim3(256), 0, 0, table, mask, iters, out);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double wave_iters = (double)grid * 4 * iters;
    printf("  %-46s %d WG/CU: %7.3f ms  %8.1f wave-iterations/us/CU  = one per %6.1f cycles at 2.4 GHz\n", name, blocks_per_cu, ms,
           wave_iters / (ms * 1e3) / cus, 2400.0 * cus * ms * 1e3 / wave_iters);
    return 0;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device: %s, %d CUs, clock %d kHz\n", prop.name, cus, prop.clockRate);
    // ---- 1. VALU issue ----
    float* out;
    unsigned long long* cyc;
    CHECK(hipMalloc(&out, (size_t)cus * 8 * 1024 * sizeof(float)));
    CHECK(hipMalloc(&cyc, (size_t)cus * 8 * 16 * sizeof(unsigned long long)));
    const int iters = 20000;
    for (int threads : {256, 512, 1024}) {
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k_valu, dim3(cus), dim3(threads), 0, 0, out, iters, cyc);
        CHECK(hipDeviceSynchronize());
        std::vector<unsigned long long> h((size_t)cus * threads / 64);
        CHECK(hipMemcpy(h.data(), cyc, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        double sum = 0;
        for (auto v : h) sum += (double)v;
        const double per_wave = sum / h.size() / ((double)iters * 64);
        const int waves_per_simd = threads / 256;
        printf("VALU: %d wave(s) per SIMD: %.2f cycles per v_fma_f32 per wave, %.2f cycles of the SIMD per wave-instruction\n", waves_per_simd,
               per_wave, per_wave / waves_per_simd);
    }
    {
        struct { const char* name; void (*fn)(float*, int, unsigned long long*); } ops[] = {
            {"v_fma_f32", k_fma}, {"v_add_f32", k_add}, {"v_mul_f32", k_mul}, {"v_max3_f32", k_max3}, {"v_cndmask_b32 (vcc)", k_cnd},
            {"v_cmp_lt_f32 -> vcc", k_cmp}, {"v_cmp_lt_f32 -> sgpr pair", k_cmps}, {"v_cvt_f32_ubyte1", k_cvt}, {"v_and_b32", k_and},
            {"v_lshlrev_b32", k_lsh}, {"v_mov_b32", k_mov}, {"v_rcp_f32", k_rcp}, {"v_max_f32", k_max},
            {"v_cmp + v_cndmask pair (per pair)", k_pair}, {"v_cndmask_b32 no dep on dst", k_cndz}, {"v_cndmask_b32 (sgpr pair)", k_cnds}};
        for (auto& op : ops) {
            for (int threads : {256, 1024}) {
                for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(op.fn, dim3(cus), dim3(threads), 0, 0, out, 4000, cyc);
                CHECK(hipDeviceSynchronize());
                std::vector<unsigned long long> h((size_t)cus * threads / 64);
                CHECK(hipMemcpy(h.data(), cyc, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
                double sum = 0;
                for (auto v : h) sum += (double)v;
                const double per_wave = sum / h.size() / (4000.0 * 64);
                printf("VALU op %-28s %d wave(s)/SIMD: %.2f cycles per instruction per wave, %.2f of the SIMD\n", op.name, threads / 256, per_wave,
                       per_wave / (threads / 256));
            }
        }
    }
    // ---- 2. dependent gathers ----
    for (size_t kb : {16, 1024, 24 * 1024, 192 * 1024, 2048 * 1024}) {
        const size_t recs = (kb << 10) / 64;
        uint32_t mask = 1;
        while ((size_t)mask * 2 <= recs) mask *= 2;
        mask -= 1;
        uint4* table;
        CHECK(hipMalloc(&table, ((size_t)mask + 1) * 64));
        std::vector<uint32_t> host(((size_t)mask + 1) * 16);
        for (size_t w = 0; w < host.size(); ++w) host[w] = (uint32_t)(((uint64_t)(uint32_t)w * 0x9E3779B97F4A7C15ull) >> 32);
        CHECK(hipMemcpy(table, host.data(), host.size() * 4, hipMemcpyHostToDevice));
        uint32_t* gout;
        CHECK(hipMalloc(&gout, (size_t)cus * 8 * 256 * 4));
        printf("table of %zu KiB (%u records of 64 B), random dependent fetches:\n", (((size_t)mask + 1) * 64) >> 10, mask + 1);
        for (int wg : {1, 2, 4, 8}) {
            if (run_gather<0>("4 x dwordx4 per lane (64 B)", table, mask, wg, cus, gout)) return 1;
        }
        if (run_gather<1>("2 x dwordx4 per lane (32 B)", table, mask, 4, cus, gout)) return 1;
        if (run_gather<2>("1 x dwordx4 per lane (16 B)", table, mask, 4, cus, gout)) return 1;
        if (run_gather<3>("4 x dwordx4, odd lanes idle", table, mask, 4, cus, gout)) return 1;
        if (run_gather<4>("quad shares a record, 1 x dwordx4 per lane", table, mask, 4, cus, gout)) return 1;
        if (run_gather<5>("4 x dwordx4, a wave's records adjacent", table, mask, 4, cus, gout)) return 1;
        for (int wg : {2, 4, 6}) {
            if (run_gather<6>("quad-cooperative 64 B per lane + transpose", table, mask, wg, cus, gout)) return 1;
        }
        if (run_gather<7>("quad-cooperative, a wave's records adjacent", table, mask, 4, cus, gout)) return 1;
        {   // the transpose must hand every lane its own record
            std::vector<uint32_t> h((size_t)cus * 4 * 256);
            CHECK(hipMemcpy(h.data(), gout, h.size() * 4, hipMemcpyDeviceToHost));
            size_t bad = 0;
            for (auto v : h) bad += (v == 0xDEADBEEFu);
            printf("  quad-cooperative transpose check: %zu of %zu lanes got a wrong record\n", bad, h.size());
        }
        CHECK(hipFree(table));
        CHECK(hipFree(gout));
    }
    return 0;
}
