// icache_cold.hip -- microbenchmark (not product code): what does a wavefront pay for instructions it executes for the first
// time in a launch?  A straight-line body of N independent-chain VALU instructions (8-byte encodings, like most of the greedy
// kernel's), run TWICE inside one launch through the same addresses; per pass the wavefront's own cycle count (s_memtime) --
// pass 1 fetches the code afresh (the instruction caches are invalidated at every dispatch), pass 2 finds it cached.
//     hipcc --offload-arch=gfx950 -O3 -o icache_cold icache_cold.hip && ./icache_cold
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

// 8 instructions x 8 bytes = one 64-byte line per I8
#define I8 "v_add3_u32 %0, %0, %4, %5\n v_add3_u32 %1, %1, %4, %5\n v_add3_u32 %2, %2, %4, %5\n v_add3_u32 %3, %3, %4, %5\n" \
           "v_add3_u32 %0, %0, %5, %4\n v_add3_u32 %1, %1, %5, %4\n v_add3_u32 %2, %2, %5, %4\n v_add3_u32 %3, %3, %5, %4\n"
#define I64 I8 I8 I8 I8 I8 I8 I8 I8
#define I512 I64 I64 I64 I64 I64 I64 I64 I64

template <int KB>  // body size in KiB (0.5 KiB per I64)
__global__ __launch_bounds__(64) void k(uint32_t *out, unsigned long long *cyc)
{
    uint32_t a = threadIdx.x, b = a * 3u, c = a * 5u, d = a * 7u, x = a + 11u, y = a + 13u;
    unsigned long long t[3];
#pragma nounroll
    for (int pass = 0; pass < 2; ++pass) {
        t[pass] = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (KB >= 4) asm volatile(I512 : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(x), "v"(y));
        if (KB >= 8) asm volatile(I512 : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(x), "v"(y));
        if (KB >= 16) asm volatile(I512 I512 : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(x), "v"(y));
        if (KB >= 32) asm volatile(I512 I512 I512 I512 : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(x), "v"(y));
        asm volatile("s_nop 0" ::: "memory");
    }
    t[2] = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + threadIdx.x] = a ^ b ^ c ^ d;
    if (threadIdx.x == 0) {
        cyc[blockIdx.x * 2] = t[1] - t[0];
        cyc[blockIdx.x * 2 + 1] = t[2] - t[1];
    }
}

// The same with JUMPS: 32 segments of one executed line followed by `skip` lines that are jumped over -- does a taken branch to
// a line nobody has fetched yet cost more than running into it?
#define SEG(SKIP) I8 "s_branch 1f\n" SKIP "1:\n"
#define SEG8(SKIP) SEG(SKIP) SEG(SKIP) SEG(SKIP) SEG(SKIP) SEG(SKIP) SEG(SKIP) SEG(SKIP) SEG(SKIP)
template <int SKIPLINES>
__global__ __launch_bounds__(64) void kj(uint32_t *out, unsigned long long *cyc)
{
    uint32_t a = threadIdx.x, b = a * 3u, c = a * 5u, d = a * 7u, x = a + 11u, y = a + 13u;
    unsigned long long t[3];
#pragma nounroll
    for (int pass = 0; pass < 2; ++pass) {
        t[pass] = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (SKIPLINES == 1) asm volatile(SEG8(I8) SEG8(I8) SEG8(I8) SEG8(I8) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(x), "v"(y));
        if (SKIPLINES == 8) asm volatile(SEG8(I64) SEG8(I64) SEG8(I64) SEG8(I64) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(x), "v"(y));
        if (SKIPLINES == 32) asm volatile(SEG8(I64 I64 I64 I64) SEG8(I64 I64 I64 I64) SEG8(I64 I64 I64 I64) SEG8(I64 I64 I64 I64)
                                          : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(x), "v"(y));
        asm volatile("s_nop 0" ::: "memory");
    }
    t[2] = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + threadIdx.x] = a ^ b ^ c ^ d;
    if (threadIdx.x == 0) {
        cyc[blockIdx.x * 2] = t[1] - t[0];
        cyc[blockIdx.x * 2 + 1] = t[2] - t[1];
    }
}

template <int SKIPLINES>
static void runj(int cus, int waves_per_cu)
{
    const int grid = cus * waves_per_cu;
    uint32_t *out;
    unsigned long long *cyc;
    CK(hipMalloc(&out, (size_t)grid * 64 * 4));
    CK(hipMalloc(&cyc, (size_t)grid * 2 * 8));
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(kj<SKIPLINES>, dim3(grid), dim3(64), 0, 0, out, cyc);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(grid * 2);
    CK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<unsigned long long> p1, p2;
    for (int i = 0; i < grid; ++i) { p1.push_back(h[2 * i]); p2.push_back(h[2 * i + 1]); }
    std::sort(p1.begin(), p1.end()); std::sort(p2.begin(), p2.end());
    const double m1 = (double)p1[grid / 2], m2 = (double)p2[grid / 2];
    printf("32 x (one line executed, %2d lines jumped over), %d wavefront(s) per CU: first pass %7.0f cycles, second %7.0f  ->  %6.1f extra cycles per jump\n",
           SKIPLINES, waves_per_cu, m1, m2, (m1 - m2) / 32);
    CK(hipFree(out)); CK(hipFree(cyc));
}

template <int KB>
static void run(int cus, int waves_per_cu)
{
    const int grid = cus * waves_per_cu;
    uint32_t *out;
    unsigned long long *cyc;
    CK(hipMalloc(&out, (size_t)grid * 64 * 4));
    CK(hipMalloc(&cyc, (size_t)grid * 2 * 8));
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k<KB>, dim3(grid), dim3(64), 0, 0, out, cyc);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(grid * 2);
    CK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<unsigned long long> p1, p2;
    for (int i = 0; i < grid; ++i) { p1.push_back(h[2 * i]); p2.push_back(h[2 * i + 1]); }
    std::sort(p1.begin(), p1.end()); std::sort(p2.begin(), p2.end());
    const double m1 = (double)p1[grid / 2], m2 = (double)p2[grid / 2];
    printf("%2d KiB of code (%4d lines), %d wavefront(s) per CU: first pass %7.0f cycles, second %7.0f  ->  %5.1f extra cycles per 64-byte line\n",
           KB, KB * 16, waves_per_cu, m1, m2, (m1 - m2) / (KB * 16));
    CK(hipFree(out)); CK(hipFree(cyc));
}

int main()
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("# %s, %d CUs; s_memtime ticks of the third of three back-to-back launches (median over wavefronts)\n", prop.gcnArchName, cus);
    for (int w : {1, 4}) {
        run<4>(cus, w);
        run<8>(cus, w);
        run<16>(cus, w);
        run<32>(cus, w);
    }
    for (int w : {1, 4}) {
        runj<1>(cus, w);
        runj<8>(cus, w);
        runj<32>(cus, w);
    }
    return 0;
}
