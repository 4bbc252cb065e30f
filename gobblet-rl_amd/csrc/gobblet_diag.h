// gobblet_diag.h -- DIAGNOSTIC BUILDS ONLY.  With -DGBL_STAMPS (scripts/build_variant.sh stamps -DGBL_STAMPS) the
// kernels record per-wavefront s_memtime / s_memrealtime stamps of their phases in a side buffer that nothing else
// reads (scripts/microbench/phase_stamps.py, greedy_stamps.py).  In the product build every macro below expands to
// nothing and no symbol is added.
#pragma once
struct TileStamps {
    unsigned long long t[3];  // phase stamps taken inside greedy_tile (diagnostic builds only; unused otherwise)
};
#ifdef GBL_STAMPS
__device__ unsigned long long g_stamps[1 << 17][12];
__device__ unsigned long long g_wave_stamps[1024][16][12];  // per block and wavefront: phase stamps inside greedy_tile
#define GBL_STAMP(i) unsigned long long st_##i = __builtin_amdgcn_s_memtime()
#define GBL_STAMP_REAL(i) unsigned long long rt_##i = __builtin_amdgcn_s_memrealtime()
#define GBL_STAMP_DECL(i) unsigned long long st_##i = 0
#define GBL_STAMP_SET(i) st_##i = __builtin_amdgcn_s_memtime()
#define GBL_STAMP_DEP(i, v)                                  \
    asm volatile("" ::"v"(v));                               \
    unsigned long long st_##i = __builtin_amdgcn_s_memtime()
#define GBL_STAMP_VAL(i, v) unsigned long long st_##i = (v)
#define GBL_TILE_STAMP(ts, i) (ts).t[i] = __builtin_amdgcn_s_memtime()
#define GBL_STAMP_DRAIN(i)                                   \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         \
    unsigned long long st_##i = __builtin_amdgcn_s_memtime()
#define GBL_WAVE_STAMP(i)                                                                                   \
    if ((threadIdx.x & 63u) == 0 && blockIdx.x < 1024)                                                      \
        g_wave_stamps[blockIdx.x][threadIdx.x >> 6][i] = __builtin_amdgcn_s_memtime()
// small_role: cycles per phase of the ply loop, summed over the plies of a launch (a stamp waits for lgkmcnt(0): it perturbs
// the LDS round trips it brackets, not the VALU chain)
#define GBL_PHASE_DECL unsigned long long ph_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pt_ = __builtin_amdgcn_s_memtime()
#define GBL_PHASE(i)                                                       \
    {                                                                      \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();      \
        ph_[i] += now_ - pt_;                                              \
        pt_ = now_;                                                        \
    }
#define GBL_PHASE_DEP(i, v)                                                \
    asm volatile("" ::"v"(v));                                             \
    GBL_PHASE(i)
#define GBL_PHASE_FLUSH(wave)                                                                              \
    if ((threadIdx.x & 63u) == 0 && blockIdx.x < 1024)                                                      \
        for (int i_ = 0; i_ < 8; ++i_) g_wave_stamps[blockIdx.x][wave][i_] = ph_[i_]
#define GBL_STAMP_FLUSH(tile)                                                                              \
    if (threadIdx.x == 0 && (tile) < (1 << 17)) {                                                          \
        unsigned long long *o_ = g_stamps[tile];                                                           \
        o_[0] = st_0; o_[1] = st_1; o_[2] = st_2; o_[3] = st_3; o_[4] = st_4; o_[5] = st_5;               \
        o_[6] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)); /* HW_REG_HW_ID */                   \
        o_[7] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11)); /* HW_REG_XCC_ID */                 \
        o_[8] = rt_0; o_[9] = __builtin_amdgcn_s_memrealtime(); /* 100 MHz, chip-wide */                  \
    }
#else
#define GBL_STAMP(i)
#define GBL_STAMP_REAL(i)
#define GBL_STAMP_DECL(i)
#define GBL_STAMP_SET(i)
#define GBL_STAMP_DEP(i, v)
#define GBL_STAMP_VAL(i, v)
#define GBL_TILE_STAMP(ts, i)
#define GBL_STAMP_DRAIN(i)
#define GBL_STAMP_FLUSH(tile)
#define GBL_WAVE_STAMP(i)
#define GBL_PHASE_DECL
#define GBL_PHASE(i)
#define GBL_PHASE_DEP(i, v)
#define GBL_PHASE_FLUSH(wave)
#endif

#ifdef GBL_STAMPS
extern "C" int gbl_debug_wave_stamps(unsigned long long *host_out)
{
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_wave_stamps), sizeof(unsigned long long) * 1024 * 16 * 12);
}
extern "C" int gbl_debug_stamps(unsigned long long *host_out, int64_t ntiles)
{
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_stamps), (size_t)ntiles * 12 * sizeof(unsigned long long));
}
#endif
