// valu_rates.hip -- microbenchmark (not product code): issue cost, in shader cycles per wave64 instruction, of
// the integer instructions the Gobblet kernels are made of, measured with s_memtime around a long stream of
// instructions (8 independent dependency chains, ONE asm block per 8 so that the compiler pads nothing in
// between) -- for one wavefront alone on its SIMD, and for two and four sharing it.
//     hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip && ./valu_rates
// Used to choose between instruction forms (32-bit multiplies vs v_mad_u64_u32 vs 24-bit multiplies, dot4
// gathers vs shifts, ...); results in profiles/r02/valu_rates.txt.   (generated table: edit the lists below)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

constexpr int kIters = 512, kChains = 8;
#define BLOCK8(M) M("%0") M("%1") M("%2") M("%3") M("%4") M("%5") M("%6") M("%7")

#define I_v_add_u32(A) "v_add_u32 " A ", " A ", %[x]\n"
#define I_v_and_b32(A) "v_and_b32 " A ", " A ", %[x]\n"
#define I_v_or_b32(A) "v_or_b32 " A ", " A ", %[x]\n"
#define I_v_xor_b32(A) "v_xor_b32 " A ", " A ", %[x]\n"
#define I_v_sub_u32(A) "v_sub_u32 " A ", " A ", %[x]\n"
#define I_v_not_b32(A) "v_not_b32 " A ", " A "\n"
#define I_v_mov_b32(A) "v_mov_b32 " A ", %[x]\n"
#define I_v_lshrrev(A) "v_lshrrev_b32 " A ", %[x], " A "\n"
#define I_v_min_u32(A) "v_min_u32 " A ", " A ", %[x]\n"
#define I_v_max_u32(A) "v_max_u32 " A ", " A ", %[x]\n"
#define I_v_and_const(A) "v_and_b32 " A ", 0x7f7f7f7f, " A "\n"
#define I_v_add_const(A) "v_add_u32 " A ", 0x7f7f7f7f, " A "\n"
#define I_v_and_sgpr(A) "v_and_b32 " A ", s20, " A "\n"
#define I_v_cmp_only(A) "v_cmp_lt_u32 vcc, " A ", %[x]\n"
#define I_s_and_b32(A) "s_and_b32 s20, s20, s21\n"
#define I_s_add_u32(A) "s_add_u32 s20, s20, s21\n"
#define I_s_mul_i32(A) "s_mul_i32 s20, s20, s21\n"
#define I_s_bcnt1_b64(A) "s_bcnt1_i32_b64 s20, s[20:21]\n"
#define I_valu_salu_mix(A) "v_and_or_b32 " A ", " A ", %[x], %[y]\n s_and_b32 s20, s20, s21\n"
#define I_v_bitop3(A) "v_bitop3_b32 " A ", " A ", %[x], %[y] bitop3:0x96\n"
#define I_v_or3(A) "v_or3_b32 " A ", " A ", %[x], %[y]\n"
#define I_v_and_or(A) "v_and_or_b32 " A ", " A ", %[x], %[y]\n"
#define I_v_lshl_or(A) "v_lshl_or_b32 " A ", " A ", %[x], %[y]\n"
#define I_v_lshl_add(A) "v_lshl_add_u32 " A ", " A ", %[x], %[y]\n"
#define I_v_add3(A) "v_add3_u32 " A ", " A ", %[x], %[y]\n"
#define I_v_bfe_u32(A) "v_bfe_u32 " A ", " A ", %[x], %[y]\n"
#define I_v_bfi(A) "v_bfi_b32 " A ", " A ", %[x], %[y]\n"
#define I_v_alignbyte(A) "v_alignbyte_b32 " A ", " A ", %[x], %[y]\n"
#define I_v_alignbit(A) "v_alignbit_b32 " A ", " A ", %[x], %[y]\n"
#define I_v_perm(A) "v_perm_b32 " A ", " A ", %[x], %[y]\n"
#define I_v_bcnt(A) "v_bcnt_u32_b32 " A ", %[x], " A "\n"
#define I_v_ffbl(A) "v_ffbl_b32 " A ", " A "\n"
#define I_v_ffbh(A) "v_ffbh_u32 " A ", " A "\n"
#define I_v_mul_u24(A) "v_mul_u32_u24 " A ", " A ", %[x]\n"
#define I_v_mad_u24(A) "v_mad_u32_u24 " A ", " A ", %[x], %[y]\n"
#define I_v_mul_lo_u32(A) "v_mul_lo_u32 " A ", " A ", %[x]\n"
#define I_v_mul_hi_u32(A) "v_mul_hi_u32 " A ", " A ", %[x]\n"
#define I_v_dot4_u8(A) "v_dot4_u32_u8 " A ", %[x], %[y], " A "\n"
#define I_v_dot8_u4(A) "v_dot8_u32_u4 " A ", %[x], %[y], " A "\n"
#define I_v_sad_u8(A) "v_sad_u8 " A ", %[x], %[y], " A "\n"
#define I_v_lshlrev(A) "v_lshlrev_b32 " A ", %[x], " A "\n"
#define I_v_cndmask(A) "v_cndmask_b32 " A ", " A ", %[x], vcc\n"
#define I_v_cmp_cnd(A) "v_cmp_lt_u32 vcc, " A ", %[x]\n v_cndmask_b32 " A ", " A ", %[y], vcc\n"
#define I_v_cmp_sgpr_cnd(A) "v_cmp_lt_u32 s[20:21], " A ", %[x]\n v_cndmask_b32 " A ", " A ", %[y], s[20:21]\n"
#define I_v_mov_dpp(A) "v_mov_b32_dpp " A ", %[x] row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_v_add_dpp(A) "v_add_u32_dpp " A ", %[x], " A " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_v_and_sdwa(A) "v_and_b32_sdwa " A ", " A ", %[x] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
#define I_v_mbcnt_lo(A) "v_mbcnt_lo_u32_b32 " A ", %[x], " A "\n"
#define I_v_readlane_add(A) "v_readlane_b32 s20, " A ", 3\n s_nop 3\n v_add_u32 " A ", s20, " A "\n"
#define I_ds_bpermute(A) "ds_bpermute_b32 " A ", %[x], " A "\n"
#define I_ds_swizzle(A) "ds_swizzle_b32 " A ", " A " offset:swizzle(SWAP,1)\n"
#define I_s_nop0(A) "s_nop 0\n"
#define I_v_mad_u64_u32(A) "v_mad_u64_u32 " A ", vcc, %[x], %[y], " A "\n"
#define I_v_lshrrev_b64(A) "v_lshrrev_b64 " A ", %[x], " A "\n"
#define I_v_lshl_add_u64(A) "v_lshl_add_u64 " A ", " A ", 1, " A "\n"

enum Op {
    OP_v_add_u32,
    OP_v_and_b32,
    OP_v_or_b32,
    OP_v_xor_b32,
    OP_v_sub_u32,
    OP_v_not_b32,
    OP_v_mov_b32,
    OP_v_lshrrev,
    OP_v_min_u32,
    OP_v_max_u32,
    OP_v_and_const,
    OP_v_add_const,
    OP_v_and_sgpr,
    OP_v_cmp_only,
    OP_s_and_b32,
    OP_s_add_u32,
    OP_s_mul_i32,
    OP_s_bcnt1_b64,
    OP_valu_salu_mix,
    OP_v_bitop3,
    OP_v_or3,
    OP_v_and_or,
    OP_v_lshl_or,
    OP_v_lshl_add,
    OP_v_add3,
    OP_v_bfe_u32,
    OP_v_bfi,
    OP_v_alignbyte,
    OP_v_alignbit,
    OP_v_perm,
    OP_v_bcnt,
    OP_v_ffbl,
    OP_v_ffbh,
    OP_v_mul_u24,
    OP_v_mad_u24,
    OP_v_mul_lo_u32,
    OP_v_mul_hi_u32,
    OP_v_dot4_u8,
    OP_v_dot8_u4,
    OP_v_sad_u8,
    OP_v_lshlrev,
    OP_v_cndmask,
    OP_v_cmp_cnd,
    OP_v_cmp_sgpr_cnd,
    OP_v_mov_dpp,
    OP_v_add_dpp,
    OP_v_and_sdwa,
    OP_v_mbcnt_lo,
    OP_v_readlane_add,
    OP_ds_bpermute,
    OP_ds_swizzle,
    OP_s_nop0,
    OP_v_mad_u64_u32,
    OP_v_lshrrev_b64,
    OP_v_lshl_add_u64,
    OP_COUNT
};
static const char *kNames[] = {"v_add_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_sub_u32", "v_not_b32", "v_mov_b32", "v_lshrrev", "v_min_u32", "v_max_u32", "v_and_const", "v_add_const", "v_and_sgpr", "v_cmp_only", "s_and_b32", "s_add_u32", "s_mul_i32", "s_bcnt1_b64", "valu_salu_mix", "v_bitop3", "v_or3", "v_and_or", "v_lshl_or", "v_lshl_add", "v_add3", "v_bfe_u32", "v_bfi", "v_alignbyte", "v_alignbit", "v_perm", "v_bcnt", "v_ffbl", "v_ffbh", "v_mul_u24", "v_mad_u24", "v_mul_lo_u32", "v_mul_hi_u32", "v_dot4_u8", "v_dot8_u4", "v_sad_u8", "v_lshlrev", "v_cndmask", "v_cmp_cnd", "v_cmp_sgpr_cnd", "v_mov_dpp", "v_add_dpp", "v_and_sdwa", "v_mbcnt_lo", "v_readlane_add", "ds_bpermute", "ds_swizzle", "s_nop0", "v_mad_u64_u32", "v_lshrrev_b64", "v_lshl_add_u64"};
static const int kPerBlock[] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 1, 1, 1, 1, 3, 1, 1, 1, 1, 1, 1};

template <int OP>
__global__ __launch_bounds__(256) void k_rate(uint32_t *out, unsigned long long *cycles, uint32_t seed)
{
    uint32_t a[kChains];
    uint64_t w[kChains];
    uint32_t x = seed * 2654435761u + threadIdx.x * 40503u + 12345u, y = (x >> 7) | 1u;
#pragma unroll
    for (int c = 0; c < kChains; ++c) {
        a[c] = x + 977u * c;
        w[c] = ((uint64_t)a[c] << 32) | (y + c);
    }
    asm volatile("v_cmp_lt_u32 vcc, %0, %1" ::"v"(x), "v"(y) : "vcc");
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int it = 0; it < kIters; ++it) {
#define RUN32(n) RUN32W(n, "")
#define RUN32DS(n) RUN32W(n, "s_waitcnt lgkmcnt(0)\n")
#define RUN32W(n, WAIT)                                                                                             \
    case OP_##n:                                                                                                    \
        asm volatile(BLOCK8(I_##n) WAIT                                                         \
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) \
                     : [x] "v"(x), [y] "v"(y)                                                                       \
                     : "vcc", "scc", "s20", "s21");                                                                        \
        break;
#define RUN64(n)                                                                                                    \
    case OP_##n:                                                                                                    \
        asm volatile(BLOCK8(I_##n)                                                                                  \
                     : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]) \
                     : [x] "v"(x), [y] "v"(y)                                                                       \
                     : "vcc");                                                                                      \
        break;
        switch (OP) {
            RUN32(v_add_u32)
            RUN32(v_and_b32)
            RUN32(v_or_b32)
            RUN32(v_xor_b32)
            RUN32(v_sub_u32)
            RUN32(v_not_b32)
            RUN32(v_mov_b32)
            RUN32(v_lshrrev)
            RUN32(v_min_u32)
            RUN32(v_max_u32)
            RUN32(v_and_const)
            RUN32(v_add_const)
            RUN32(v_and_sgpr)
            RUN32(v_cmp_only)
            RUN32(s_and_b32)
            RUN32(s_add_u32)
            RUN32(s_mul_i32)
            RUN32(s_bcnt1_b64)
            RUN32(valu_salu_mix)
            RUN32(v_bitop3)
            RUN32(v_or3)
            RUN32(v_and_or)
            RUN32(v_lshl_or)
            RUN32(v_lshl_add)
            RUN32(v_add3)
            RUN32(v_bfe_u32)
            RUN32(v_bfi)
            RUN32(v_alignbyte)
            RUN32(v_alignbit)
            RUN32(v_perm)
            RUN32(v_bcnt)
            RUN32(v_ffbl)
            RUN32(v_ffbh)
            RUN32(v_mul_u24)
            RUN32(v_mad_u24)
            RUN32(v_mul_lo_u32)
            RUN32(v_mul_hi_u32)
            RUN32(v_dot4_u8)
            RUN32(v_dot8_u4)
            RUN32(v_sad_u8)
            RUN32(v_lshlrev)
            RUN32(v_cndmask)
            RUN32(v_cmp_cnd)
            RUN32(v_cmp_sgpr_cnd)
            RUN32(v_mov_dpp)
            RUN32(v_add_dpp)
            RUN32(v_and_sdwa)
            RUN32(v_mbcnt_lo)
            RUN32(v_readlane_add)
            RUN32DS(ds_bpermute)
            RUN32DS(ds_swizzle)
            RUN32(s_nop0)
            RUN64(v_mad_u64_u32)
            RUN64(v_lshrrev_b64)
            RUN64(v_lshl_add_u64)
        }
    }
    asm volatile("s_nop 0" ::: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    uint32_t acc = 0;
#pragma unroll
    for (int c = 0; c < kChains; ++c) acc ^= a[c] ^ (uint32_t)w[c] ^ (uint32_t)(w[c] >> 32);
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

typedef void (*Kern)(uint32_t *, unsigned long long *, uint32_t);

template <int OP>
struct Table {
    static void fill(Kern *k)
    {
        k[OP] = k_rate<OP>;
        Table<OP + 1>::fill(k);
    }
};
template <>
struct Table<OP_COUNT> {
    static void fill(Kern *) {}
};

int main()
{
    Kern kern[OP_COUNT];
    Table<0>::fill(kern);
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    uint32_t *out;
    unsigned long long *cyc;
    CK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
    CK(hipMalloc(&cyc, (size_t)cus * 8 * 4 * 8));
    std::vector<unsigned long long> h(cus * 8 * 4);
    printf("# %s, %d CUs; shader cycles per wave64 instruction (median over waves), %d instructions per wave\n",
           prop.gcnArchName, cus, kIters * kChains);
    printf("# the last column is the SIMD's cost per instruction with 4 waves sharing it (= column 3 / 4)\n");
    printf("%-20s %12s %12s %12s %12s\n", "instruction", "1 wave/SIMD", "2 waves/SIMD", "4 waves/SIMD", "SIMD cyc/instr");
    for (int op = 0; op < OP_COUNT; ++op) {
        double res[3];
        for (int v = 0; v < 3; ++v) {
            const int blocks_per_cu = 1 << v;  // 256-thread blocks: 4 waves, one per SIMD
            const int grid = cus * blocks_per_cu;
            for (int rep = 0; rep < 2; ++rep) {
                hipLaunchKernelGGL(kern[op], dim3(grid), dim3(256), 0, 0, out, cyc, 7u + rep);
                CK(hipDeviceSynchronize());
            }
            CK(hipMemcpy(h.data(), cyc, (size_t)grid * 4 * 8, hipMemcpyDeviceToHost));
            std::vector<unsigned long long> s(h.begin(), h.begin() + grid * 4);
            std::sort(s.begin(), s.end());
            res[v] = (double)s[s.size() / 2] / (kIters * kChains * kPerBlock[op]);
        }
        printf("%-20s %12.2f %12.2f %12.2f %12.2f\n", kNames[op], res[0], res[1], res[2], res[2] / 4);
        fflush(stdout);
    }
    return 0;
}
