// pattern_bench.hip -- microbenchmark (not product code): how fast does an MI355X move the Gobblet step
// kernel's MEMORY PATTERN with no game logic at all?  Gives the ceiling the real kernels are read
// against and prices individual features of the pattern (scalar side streams, tile->block map,
// workgroup size, non-temporal stores).   hipcc --offload-arch=gfx950 -O3 -o pattern_bench pattern_bench.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

typedef uint32_t __attribute__((ext_vector_type(4))) v4u;

struct Args {
    int8_t *state, *mask, *obs, *to_move, *done, *winner, *reward;
    int32_t *action;
    int64_t ntiles;
    int64_t obs_stride, state_stride;  // bytes per tile: 7488 / 1728 as in the product, or padded to 128-byte lines
};

template <int ROWB, bool NT>
__device__ __forceinline__ void tile_out(int8_t *g, const uint32_t *lds, int lane)
{
    constexpr int NV = 64 * ROWB / 16, FULL = NV / 64, REM = NV % 64;
    v4u *gv = reinterpret_cast<v4u *>(g);
    const v4u *lv = reinterpret_cast<const v4u *>(lds);
    v4u v[FULL + 1];
#pragma unroll
    for (int i = 0; i < FULL; ++i) v[i] = lv[lane + 64 * i];
    if (REM && lane < REM) v[FULL] = lv[lane + 64 * FULL];
#pragma unroll
    for (int i = 0; i < FULL; ++i) {
        if (NT) __builtin_nontemporal_store(v[i], &gv[lane + 64 * i]); else gv[lane + 64 * i] = v[i];
    }
    if (REM && lane < REM) {
        if (NT) __builtin_nontemporal_store(v[FULL], &gv[lane + 64 * FULL]); else gv[lane + 64 * FULL] = v[FULL];
    }
}

// one tile: read 27-byte rows, write 27 + 54 + 117 (+ scalar side streams)
template <bool SCALARS, bool NT, bool OBS, bool MASK, bool STATE_RW>
__device__ __forceinline__ void do_tile(const Args &a, int64_t tile, int lane, uint32_t *img)
{
    if (STATE_RW) {
        const v4u *gv = reinterpret_cast<const v4u *>(a.state + tile * a.state_stride);
        v4u *lv = reinterpret_cast<v4u *>(img);
        v4u x0 = gv[lane], x1 = {0, 0, 0, 0};
        if (lane < 44) x1 = gv[lane + 64];
        lv[lane] = x0;
        if (lane < 44) lv[lane + 64] = x1;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        uint32_t t = img[(lane * 27) >> 2];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        img[lane] = t + 1;  // keep a data dependency load -> stores
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (OBS) tile_out<117, NT>(a.obs + tile * a.obs_stride, img, lane);
    if (MASK) tile_out<54, NT>(a.mask + tile * 3456, img, lane);
    if (STATE_RW) tile_out<27, false>(a.state + tile * a.state_stride, img, lane);
    if (SCALARS) {
        int64_t b = tile * 64 + lane;
        uint32_t t = img[lane];
        a.to_move[b] = (int8_t)t; a.done[b] = (int8_t)(t >> 8); a.winner[b] = (int8_t)(t >> 16);
        reinterpret_cast<uint16_t *>(a.reward)[b] = (uint16_t)t;
        a.action[b] = (int32_t)t;
    }
}

template <int WAVES, bool XCD, bool SCALARS, bool NT, bool OBS, bool MASK, bool STATE_RW, int TILES_PER_WAVE>
__global__ __launch_bounds__(64 * WAVES) void k_pattern(Args a)
{
    __shared__ uint32_t s_img[WAVES][7488 / 4 + 8];
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int64_t nunits = (a.ntiles + WAVES * TILES_PER_WAVE - 1) / (WAVES * TILES_PER_WAVE);
    int64_t unit;
    if (XCD) { int64_t chunk = (nunits + 7) >> 3; unit = (int64_t)(blockIdx.x & 7) * chunk + (blockIdx.x >> 3); }
    else unit = blockIdx.x;
    if (unit >= nunits) return;
#pragma unroll 1
    for (int t = 0; t < TILES_PER_WAVE; ++t) {
        int64_t tile = (unit * WAVES + wave) * TILES_PER_WAVE + t;
        if (tile < a.ntiles) do_tile<SCALARS, NT, OBS, MASK, STATE_RW>(a, tile, lane, s_img[wave]);
    }
}

template <int WAVES, bool XCD, bool SCALARS, bool NT, bool OBS, bool MASK, bool STATE_RW, int TPW>
void run(const char *name, Args a, int64_t boards, double bytes_per_board)
{
    int64_t nunits = (a.ntiles + WAVES * TPW - 1) / (WAVES * TPW);
    uint32_t grid = XCD ? (uint32_t)(((nunits + 7) / 8) * 8) : (uint32_t)nunits;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k_pattern<WAVES, XCD, SCALARS, NT, OBS, MASK, STATE_RW, TPW>), dim3(grid), dim3(64 * WAVES), 0, 0, a);
    CK(hipDeviceSynchronize());
    const int iters = 50;
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((k_pattern<WAVES, XCD, SCALARS, NT, OBS, MASK, STATE_RW, TPW>), dim3(grid), dim3(64 * WAVES), 0, 0, a);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double us = ms * 1e3 / iters;
    printf("%-44s boards %8ld  %8.2f us/launch  %7.0f GB/s (%.0f B/board)\n", name, (long)boards, us, bytes_per_board * boards / us / 1e3, bytes_per_board);
}

int main(int argc, char **argv)
{
    for (int64_t boards : {(int64_t)1 << 20, (int64_t)1 << 22}) {
        Args a;
        a.ntiles = boards / 64;
        a.obs_stride = 7488; a.state_stride = 1728;
        CK(hipMalloc(&a.state, boards * 28)); CK(hipMalloc(&a.mask, boards * 54)); CK(hipMalloc(&a.obs, boards * 118));
        CK(hipMalloc(&a.to_move, boards)); CK(hipMalloc(&a.done, boards)); CK(hipMalloc(&a.winner, boards));
        CK(hipMalloc(&a.reward, boards * 2)); CK(hipMalloc(&a.action, boards * 4));
        CK(hipMemset(a.state, 0, boards * 28));
        const double FULL = 27 + 27 + 54 + 117 + 1 + 1 + 1 + 2 + 4, NOSC = 27 + 27 + 54 + 117;
        //   WAVES XCD  SCAL  NT    OBS   MASK  ST    TPW
        run<1, true, true, true, true, true, true, 1>("V0 full pattern (as k_rollout), NT", a, boards, FULL);
        run<1, true, true, false, true, true, true, 1>("V0b full pattern, plain stores", a, boards, FULL);
        run<1, true, false, true, true, true, true, 1>("V1 no scalar side streams", a, boards, NOSC);
        run<4, true, true, true, true, true, true, 1>("V2 256-thread blocks (4 tiles)", a, boards, FULL);
        run<1, false, true, true, true, true, true, 1>("V3 no XCD remap (block = tile)", a, boards, FULL);
        run<1, true, true, true, true, true, true, 2>("V5 2 consecutive tiles per wave", a, boards, FULL);
        run<1, true, true, true, true, true, true, 4>("V5b 4 consecutive tiles per wave", a, boards, FULL);
        run<4, true, true, true, true, true, true, 2>("V6 256 threads x 2 tiles per wave", a, boards, FULL);
        run<1, true, false, true, true, false, false, 1>("V4 obs only (pure 117 B/board write), NT", a, boards, 117);
        run<1, true, false, false, true, false, false, 1>("V4b obs only, plain stores", a, boards, 117);
        run<1, true, false, true, true, true, false, 1>("V7 obs+mask write only, NT", a, boards, 171);
        // what would 128-byte aligned tiles be worth?  (hypothetical layout: the product's rows are unpadded)
        a.obs_stride = 7552; a.state_stride = 1792;
        run<1, true, true, true, true, true, true, 1>("V8 full pattern, tiles padded to 128 B lines", a, boards, FULL);
        run<1, true, false, true, true, false, false, 1>("V8b obs only, tiles padded to 128 B lines", a, boards, 117);
        a.obs_stride = 7488; a.state_stride = 1728;
        run<1, true, true, true, true, true, true, 1>("V0 again", a, boards, FULL);
        CK(hipFree(a.state)); CK(hipFree(a.mask)); CK(hipFree(a.obs)); CK(hipFree(a.to_move)); CK(hipFree(a.done));
        CK(hipFree(a.winner)); CK(hipFree(a.reward)); CK(hipFree(a.action));
        printf("\n");
    }
    return 0;
}
