// valu_mix.hip -- microbenchmark (not product code): do the per-instruction SIMD costs of valu_rates.hip ADD UP in a mixed
// stream?  valu_rates times pure streams of one instruction: two-operand logic / add / right shift on VGPRs cost a SIMD 1.2
// cycles with four wavefronts on it, three-operand and "complex" integer instructions 2.4-3.2, and a 32-bit literal or an SGPR
// operand moves a simple instruction into the expensive class.  The greedy pair evaluation (436 VALU instructions, a third of
// them v_bitop3 with an SGPR constant) measures 1 530 cycles per SIMD (reply_rate.hip), 1.7x the sum of those rates.  This
// file times mixed streams -- 8 independent chains, events around the launch, wavefronts per SIMD 1, 2, 4, 8 -- to find out
// which forms are worth steering the compiler towards.
//     hipcc --offload-arch=gfx950 -O3 -o valu_mix valu_mix.hip && ./valu_mix
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

constexpr int kIters = 2048;
#define B8(M) M("%0") M("%1") M("%2") M("%3") M("%4") M("%5") M("%6") M("%7")

// name, instructions per chain step, asm for one chain step
#define C_and_vgpr(A) "v_and_b32 " A ", " A ", %[x]\n"
#define C_and_inline(A) "v_and_b32 " A ", 15, " A "\n"
#define C_and_lit(A) "v_and_b32 " A ", 0x7f7f7f7f, " A "\n"
#define C_and_sgpr(A) "v_and_b32 " A ", s20, " A "\n"
#define C_lshr_inline(A) "v_lshrrev_b32 " A ", 3, " A "\n"
#define C_lshl_inline(A) "v_lshlrev_b32 " A ", 3, " A "\n"
#define C_add_self(A) "v_add_u32 " A ", " A ", " A "\n"
#define C_bitop3_vgpr(A) "v_bitop3_b32 " A ", " A ", %[x], %[y] bitop3:0x96\n"
#define C_bitop3_sgpr(A) "v_bitop3_b32 " A ", " A ", s20, %[y] bitop3:0x96\n"
#define C_simple6(A) "v_and_b32 " A ", " A ", %[x]\n v_add_u32 " A ", " A ", %[y]\n v_lshrrev_b32 " A ", 3, " A "\n v_or_b32 " A ", " A ", %[y]\n v_xor_b32 " A ", " A ", %[x]\n v_sub_u32 " A ", " A ", %[y]\n"
#define C_lit3(A) "v_and_b32 " A ", 0x7f7f7f7f, " A "\n v_add_u32 " A ", 0x1010101, " A "\n v_or_b32 " A ", 0x40404040, " A "\n"
#define C_bitop_and(A) "v_bitop3_b32 " A ", " A ", %[x], %[y] bitop3:0x96\n v_and_b32 " A ", " A ", %[x]\n"
#define C_bitop_3and(A) "v_bitop3_b32 " A ", " A ", %[x], %[y] bitop3:0x96\n v_and_b32 " A ", " A ", %[x]\n v_or_b32 " A ", " A ", %[y]\n v_add_u32 " A ", " A ", %[x]\n"
#define C_lshl_and(A) "v_lshlrev_b32 " A ", 3, " A "\n v_and_b32 " A ", " A ", %[x]\n"
#define C_bitopS_and(A) "v_bitop3_b32 " A ", " A ", s20, %[y] bitop3:0x96\n v_and_b32 " A ", " A ", %[x]\n"
#define C_andlit_add(A) "v_and_b32 " A ", 0x7f7f7f7f, " A "\n v_add_u32 " A ", " A ", %[x]\n"
#define C_or3_and(A) "v_or3_b32 " A ", " A ", %[x], %[y]\n v_and_b32 " A ", " A ", %[x]\n"
#define C_xad(A) "v_xad_u32 " A ", " A ", %[x], %[y]\n"
#define C_mov_and(A) "v_mov_b32 " A ", %[x]\n v_and_b32 " A ", " A ", %[y]\n"
// different destination than sources (the compiler's usual form), sources two other chains' registers: no dependency stalls
#define C_and_3addr(A) "v_and_b32 " A ", %[x], %[y]\n"

#define CASES(X) \
    X(and_vgpr, 1) X(and_inline, 1) X(and_lit, 1) X(and_sgpr, 1) X(lshr_inline, 1) X(lshl_inline, 1) X(add_self, 1) \
    X(bitop3_vgpr, 1) X(bitop3_sgpr, 1) X(simple6, 6) X(lit3, 3) X(bitop_and, 2) X(bitop_3and, 4) X(lshl_and, 2) \
    X(bitopS_and, 2) X(andlit_add, 2) X(or3_and, 2) X(xad, 1) X(mov_and, 2) X(and_3addr, 1)

enum Case {
#define X(n, c) K_##n,
    CASES(X)
#undef X
    K_COUNT
};
static const char *kNames[] = {
#define X(n, c) #n,
    CASES(X)
#undef X
};
static const int kPer[] = {
#define X(n, c) c,
    CASES(X)
#undef X
};

template <int OP>
__global__ __launch_bounds__(256) void k_mix(uint32_t *out, uint32_t seed)
{
    uint32_t a[8];
    uint32_t x = seed * 2654435761u + threadIdx.x * 40503u + 12345u, y = (x >> 7) | 1u;
#pragma unroll
    for (int c = 0; c < 8; ++c) a[c] = x + 977u * c;
    asm volatile("s_mov_b32 s20, 0x3fffffff" ::: "s20");
    for (int it = 0; it < kIters; ++it) {
        switch (OP) {
#define X(n, c)                                                                                                               \
    case K_##n:                                                                                                               \
        asm volatile(B8(C_##n)                                                                                                \
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])        \
                     : [x] "v"(x), [y] "v"(y)                                                                                 \
                     : "s20");                                                                                                \
        break;
            CASES(X)
#undef X
        }
    }
    uint32_t acc = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) acc ^= a[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

typedef void (*Kern)(uint32_t *, uint32_t);
template <int OP>
struct Table {
    static void fill(Kern *k) { k[OP] = k_mix<OP>; Table<OP + 1>::fill(k); }
};
template <>
struct Table<K_COUNT> {
    static void fill(Kern *) {}
};

int main()
{
    Kern kern[K_COUNT];
    Table<0>::fill(kern);
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    uint32_t *out;
    CK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double ghz = 2.2;  // (shader clock under load, as in reply_rate.hip; the columns compare like with like)
    printf("# %s, %d CUs; SIMD cycles (at %.1f GHz) per wave64 instruction, workgroups of 4 wavefronts (one per SIMD)\n", prop.gcnArchName, cus, ghz);
    printf("%-14s %10s %10s %10s %10s\n", "stream", "1 wave", "2 waves", "4 waves", "8 waves");
    for (int op = 0; op < K_COUNT; ++op) {
        printf("%-14s", kNames[op]);
        for (int wps = 1; wps <= 8; wps *= 2) {
            hipLaunchKernelGGL(kern[op], dim3(cus * wps), dim3(256), 0, 0, out, 1u);
            CK(hipEventRecord(e0, 0));
            for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(kern[op], dim3(cus * wps), dim3(256), 0, 0, out, 2u + r);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double instr = (double)kIters * 8 * kPer[op] * wps;  // per SIMD
            printf(" %10.2f", ms / 3 * 1e6 * ghz / instr);
        }
        printf("\n");
    }
    return 0;
}
