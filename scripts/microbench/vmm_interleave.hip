// vmm_interleave.hip -- microbenchmark (not product code): can a process build an array whose physical backing
// alternates between the device's memory classes (see write_classes.hip), so that ANY single write stream runs at the
// multi-class rate?  Physical chunks from hipMemCreate, grouped in pools of consecutively created chunks (with spacers
// between pools so that they come from different places), pools sorted into classes by pair tests, then one virtual
// range mapped chunk by chunk alternately from two pools of different classes.
//   hipcc --offload-arch=gfx950 -O3 -o vmm_interleave vmm_interleave.hip ; ./vmm_interleave [chunk MiB] [pools] [spacer GiB]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s (%d) at line %d\n", hipGetErrorString(e_), (int)e_, __LINE__); exit(1);} } while (0)

typedef uint32_t __attribute__((ext_vector_type(4))) v4u;

template <int ROWB>
__device__ __forceinline__ void tile_out(int8_t *g, const uint32_t *lds, int lane)
{
    constexpr int NV = 64 * ROWB / 16, FULL = NV / 64, REM = NV % 64;
    const v4u *lv = reinterpret_cast<const v4u *>(lds);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(g, 0, 64 * ROWB, 0x00020000);
    v4u v[FULL + 1];
#pragma unroll
    for (int i = 0; i < FULL; ++i) v[i] = lv[lane + 64 * i];
    if (REM && lane < REM) v[FULL] = lv[lane + 64 * FULL];
#pragma unroll
    for (int i = 0; i < FULL; ++i) __builtin_amdgcn_raw_buffer_store_b128(v[i], rs, (lane + 64 * i) * 16, 0, 18);
    if (REM && lane < REM) __builtin_amdgcn_raw_buffer_store_b128(v[FULL], rs, (lane + 64 * FULL) * 16, 0, 18);
}

// gbl_collect's store pattern: T plies, slot t of the obs array (64 x 117-byte rows per tile) and of the mask array
__global__ __launch_bounds__(64) void k_write(int8_t *obs, int8_t *mask, int64_t ntiles, int plies)
{
    __shared__ uint32_t img[64 * 117 / 4 + 4];
    const int lane = threadIdx.x;
    const int64_t tile = blockIdx.x;
    for (int i = lane; i < 64 * 117 / 4 + 4; i += 64) img[i] = 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int t = 0; t < plies; ++t) {
        const int64_t cell = ((int64_t)t * ntiles + tile) * 64;
        if (obs) tile_out<117>(obs + cell * 117, img, lane);
        if (mask) tile_out<54>(mask + cell * 54, img, lane);
    }
}

static hipEvent_t e0, e1;

static float run(int8_t *obs, int8_t *mask, int64_t ntiles, int plies, int reps = 5)
{
    float best = 1e30f;
    for (int r = -1; r < reps; ++r) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_write, dim3((uint32_t)ntiles), dim3(64), 0, 0, obs, mask, ntiles, plies);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 0) best = std::min(best, ms);
    }
    return best * 1e3f;
}

int main(int argc, char **argv)
{
    const size_t chunk = (size_t)(argc > 1 ? atoll(argv[1]) : 32) << 20;
    const int npools = argc > 2 ? atoi(argv[2]) : 6;
    const size_t spacer = (size_t)(argc > 3 ? atoll(argv[3]) : 24) << 30;
    const int64_t boards = 1 << 20, ntiles = boards / 64;
    const int T = 8;
    const size_t obs_bytes = (size_t)T * boards * 117, mask_bytes = (size_t)T * boards * 54;
    CK(hipSetDevice(0));
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    printf("granularity %zu KiB, chunk %zu MiB\n", gran >> 10, chunk >> 20);
    const size_t per_pool = (obs_bytes + mask_bytes + 2 * chunk + chunk - 1) / chunk;  // chunks per pool: one obs + one mask array
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    struct Pool { std::vector<hipMemGenericAllocationHandle_t> h; int8_t *va; };
    std::vector<Pool> pools(npools);
    std::vector<void *> spacers;
    for (int p = 0; p < npools; ++p) {
        hipEvent_t t0, t1;
        pools[p].h.resize(per_pool);
        for (size_t i = 0; i < per_pool; ++i) CK(hipMemCreate(&pools[p].h[i], chunk, &prop, 0));
        void *va = nullptr;
        CK(hipMemAddressReserve(&va, per_pool * chunk, 0, nullptr, 0));
        for (size_t i = 0; i < per_pool; ++i) CK(hipMemMap((char *)va + i * chunk, chunk, 0, pools[p].h[i], 0));
        CK(hipMemSetAccess(va, per_pool * chunk, &acc, 1));
        pools[p].va = (int8_t *)va;
        (void)t0; (void)t1;
        if (p + 1 < npools && spacer) {
            void *s = nullptr;
            CK(hipMalloc(&s, spacer));
            spacers.push_back(s);
        }
    }
    printf("%d pools of %zu chunks (%.0f MiB each), %zu GiB spacers between them\n", npools, per_pool, per_pool * chunk / 1048576.0, spacer >> 30);
    const size_t mask_off = ((obs_bytes + chunk - 1) / chunk) * chunk;
    // a plain hipMalloc for reference
    int8_t *plain;
    CK(hipMalloc(&plain, obs_bytes));
    printf("obs stream alone: hipMalloc %.2f us per ply; pools:", run(plain, nullptr, ntiles, T) / T);
    for (int p = 0; p < npools; ++p) printf(" %.2f", run(pools[p].va, nullptr, ntiles, T) / T);
    printf("\n");
    // classes of the pools
    std::vector<int> cls(npools, -1), rep;
    for (int p = 0; p < npools; ++p) {
        for (size_t k = 0; k < rep.size() && cls[p] < 0; ++k) {
            const float both = run(pools[rep[k]].va, pools[p].va + mask_off, ntiles, T);
            const float a = run(pools[rep[k]].va, nullptr, ntiles, T), b = run(nullptr, pools[p].va + mask_off, ntiles, T);
            printf("  pool %d vs pool %d: ratio %.3f\n", p, rep[k], both / (a + b));
            if (both / (a + b) > 0.92f) cls[p] = (int)k;
        }
        if (cls[p] < 0) { cls[p] = (int)rep.size(); rep.push_back(p); }
    }
    printf("classes of the pools:");
    for (int p = 0; p < npools; ++p) printf(" %c", 'A' + cls[p]);
    printf("\n");
    if (rep.size() < 2) { printf("one class only\n"); return 0; }
    // interleave: a fresh virtual range, chunk i from pool rep[i % k]'s chunk i (remapped: unmap there first)
    for (int ways = 2; ways <= (int)std::min<size_t>(rep.size(), 3); ++ways) {
        void *va = nullptr;
        const size_t nch = per_pool;
        CK(hipMemAddressReserve(&va, nch * chunk, 0, nullptr, 0));
        for (size_t i = 0; i < nch; ++i) {
            Pool &src = pools[rep[i % ways]];
            CK(hipMemUnmap(src.va + i * chunk, chunk));
            CK(hipMemMap((char *)va + i * chunk, chunk, 0, src.h[i], 0));
        }
        CK(hipMemSetAccess(va, nch * chunk, &acc, 1));
        int8_t *arr = (int8_t *)va;
        const float o = run(arr, nullptr, ntiles, T), m = run(nullptr, arr + mask_off, ntiles, T), both = run(arr, arr + mask_off, ntiles, T);
        printf("%d-way interleaved array (chunks of %zu MiB): obs alone %.2f us per ply (%.2f TB/s), mask alone %.2f (%.2f TB/s), obs + mask %.2f (%.2f TB/s)\n",
               ways, chunk >> 20, o / T, boards * 117.0 * T / o / 1e6, m / T, boards * 54.0 * T / m / 1e6, both / T, boards * 171.0 * T / both / 1e6);
        // put the chunks back
        for (size_t i = 0; i < nch; ++i) {
            Pool &src = pools[rep[i % ways]];
            CK(hipMemUnmap((char *)va + i * chunk, chunk));
            CK(hipMemMap(src.va + i * chunk, chunk, 0, src.h[i], 0));
        }
        for (int k = 0; k < ways; ++k) CK(hipMemSetAccess(pools[rep[k]].va, per_pool * chunk, &acc, 1));
        CK(hipMemAddressFree(va, nch * chunk));
    }
    const float same = run(pools[rep[0]].va, pools[rep[0]].va + mask_off, ntiles, T), two = run(pools[rep[0]].va, pools[rep[1]].va + mask_off, ntiles, T);
    printf("for comparison: obs + mask in one pool %.2f us per ply, obs in pool %d / mask in pool %d: %.2f\n", same / T, rep[0], rep[1], two / T);
    return 0;
}
