// write_classes.hip -- microbenchmark (not product code): how do concurrent WRITE streams share an MI355X's HBM?
// One arena; its 8 GiB chunks are sorted into classes by pair tests (two streams in one class take the sum of their
// single times, in two classes they overlap); then gbl_collect's store pattern (a wavefront writes 64 x 117-byte rows
// and 64 x 54-byte rows per ply, T plies on the same tile, `nt sc1`) with its arrays dealt over one, two and three
// classes -- including the observation array itself split by ply parity over two classes.
//   hipcc --offload-arch=gfx950 -O3 -o write_classes write_classes.hip ; ./write_classes [arena GiB] [boards] [T]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

typedef uint32_t __attribute__((ext_vector_type(4))) v4u;

// AUX: cache-policy bits of the buffer store (gfx940 encoding: sc0 = 1, nt = 2, sc1 = 16); 0 = plain
template <int ROWB, int AUX>
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
    for (int i = 0; i < FULL; ++i) __builtin_amdgcn_raw_buffer_store_b128(v[i], rs, (lane + 64 * i) * 16, 0, AUX);
    if (REM && lane < REM) __builtin_amdgcn_raw_buffer_store_b128(v[FULL], rs, (lane + 64 * FULL) * 16, 0, AUX);
}

// obs rows of ply t go to obs[t & 1], mask rows to mask[t & 1] (either may be null)
struct Streams {
    int8_t *obs[2], *mask[2];
};

template <int AUX>
__global__ __launch_bounds__(64) void k_write(Streams s, int64_t ntiles, int plies)
{
    __shared__ uint32_t img[64 * 117 / 4 + 4];
    const int lane = threadIdx.x;
    const int64_t tile = blockIdx.x;
    for (int i = lane; i < 64 * 117 / 4 + 4; i += 64) img[i] = 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int t = 0; t < plies; ++t) {
        // every (array, slot) is written once per launch: the two parities share an array -> slot t, else slot t / 2
        const int64_t cell_o = ((int64_t)(s.obs[0] == s.obs[1] ? t : t >> 1) * ntiles + tile) * 64;
        const int64_t cell_m = ((int64_t)(s.mask[0] == s.mask[1] ? t : t >> 1) * ntiles + tile) * 64;
        if (s.obs[t & 1]) tile_out<117, AUX>(s.obs[t & 1] + cell_o * 117, img, lane);
        if (s.mask[t & 1]) tile_out<54, AUX>(s.mask[t & 1] + cell_m * 54, img, lane);
    }
}


// Model of a wavefront that has `think` x 64 clocks of game logic per ply in front of its stores (SPLIT = false), and of
// the same work dealt over two wavefronts of a workgroup: wavefront 0 thinks, wavefront 1 stores the previous ply
// (SPLIT = true) -- does decoupling store issue from the computing wavefront pay at small grids?
template <bool SPLIT>
__global__ __launch_bounds__(SPLIT ? 128 : 64) void k_think(Streams s, int64_t ntiles, int plies, int think)
{
    __shared__ uint32_t img[64 * 117 / 4 + 4];
    const int lane = threadIdx.x & 63, role = threadIdx.x >> 6;
    const int64_t tile = blockIdx.x;
    for (int i = threadIdx.x; i < 64 * 117 / 4 + 4; i += SPLIT ? 128 : 64) img[i] = 0;
    __syncthreads();
    for (int t = 0; t < plies + (SPLIT ? 1 : 0); ++t) {
        if (!SPLIT || role == 0) {
            if (t < plies)
                for (int i = 0; i < think; ++i) __builtin_amdgcn_s_sleep(1);
        }
        if (!SPLIT || role == 1) {
            const int tt = SPLIT ? t - 1 : t;
            if (tt >= 0) {
                const int64_t cell = ((int64_t)tt * ntiles + tile) * 64;
                if (s.obs[0]) tile_out<117, 18>(s.obs[0] + cell * 117, img, lane);
                if (s.mask[0]) tile_out<54, 18>(s.mask[0] + cell * 54, img, lane);
            }
        }
        if (SPLIT) __syncthreads();
    }
}

static hipEvent_t e0, e1;

static int g_aux = 18;

static float run(const Streams &s, int64_t ntiles, int plies, int reps = 5)
{
    float best = 1e30f;
    for (int r = -1; r < reps; ++r) {
        CK(hipEventRecord(e0, 0));
        switch (g_aux) {
        case 0: hipLaunchKernelGGL(k_write<0>, dim3((uint32_t)ntiles), dim3(64), 0, 0, s, ntiles, plies); break;
        case 2: hipLaunchKernelGGL(k_write<2>, dim3((uint32_t)ntiles), dim3(64), 0, 0, s, ntiles, plies); break;
        case 16: hipLaunchKernelGGL(k_write<16>, dim3((uint32_t)ntiles), dim3(64), 0, 0, s, ntiles, plies); break;
        case 17: hipLaunchKernelGGL(k_write<17>, dim3((uint32_t)ntiles), dim3(64), 0, 0, s, ntiles, plies); break;
        case 1: hipLaunchKernelGGL(k_write<1>, dim3((uint32_t)ntiles), dim3(64), 0, 0, s, ntiles, plies); break;
        case 19: hipLaunchKernelGGL(k_write<19>, dim3((uint32_t)ntiles), dim3(64), 0, 0, s, ntiles, plies); break;
        default: hipLaunchKernelGGL(k_write<18>, dim3((uint32_t)ntiles), dim3(64), 0, 0, s, ntiles, plies); break;
        }
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
    const int64_t gib = argc > 1 ? atoll(argv[1]) : 160;
    const int64_t boards = argc > 2 ? atoll(argv[2]) : 1 << 20;
    const int T = argc > 3 ? atoi(argv[3]) : 8;
    const int64_t GiB = 1ll << 30, ntiles = boards / 64;
    int8_t *arena;
    CK(hipMalloc(&arena, gib * GiB));
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int nchunk = (int)(gib / 8);
    auto chunk = [&](int c, int64_t off = 0) { return arena + (int64_t)c * 8 * GiB + off; };
    const int64_t obs_bytes = (int64_t)T * boards * 117, mask_bytes = (int64_t)T * boards * 54;
    const int64_t mask_off = (obs_bytes + (2 << 20)) & ~(int64_t)((2 << 20) - 1);  // the mask array behind the obs array of a chunk
    if (mask_off + mask_bytes > 8 * GiB) { printf("arrays do not fit a chunk\n"); return 1; }
    printf("arena %lld GiB, %d chunks of 8 GiB; boards %lld, T %d: obs %.0f MiB, mask %.0f MiB per array\n", (long long)gib, nchunk,
           (long long)boards, T, obs_bytes / 1048576.0, mask_bytes / 1048576.0);
    // classes: chunk c belongs to the class of the first earlier representative it conflicts with
    std::vector<int> cls(nchunk, -1), rep;
    auto pair_ratio = [&](int ca, int cb) {
        Streams both{{chunk(ca), chunk(ca)}, {chunk(cb, mask_off), chunk(cb, mask_off)}};
        Streams a{{chunk(ca), chunk(ca)}, {nullptr, nullptr}}, b{{nullptr, nullptr}, {chunk(cb, mask_off), chunk(cb, mask_off)}};
        return run(both, ntiles, T) / (run(a, ntiles, T) + run(b, ntiles, T));
    };
    for (int c = 0; c < nchunk; ++c) {
        for (size_t k = 0; k < rep.size() && cls[c] < 0; ++k)
            if (pair_ratio(rep[k], c) > 0.92f) cls[c] = (int)k;
        if (cls[c] < 0) { cls[c] = (int)rep.size(); rep.push_back(c); }
    }
    printf("classes of the chunks:");
    for (int c = 0; c < nchunk; ++c) printf(" %c", 'A' + cls[c]);
    printf("   (%zu classes)\n", rep.size());
    if (rep.size() < 2) { printf("one class only inside this arena\n"); return 0; }
    if (rep.size() < 3) { printf("(two classes only inside this arena: the lines with class C use B instead)\n"); rep.push_back(rep[1]); }
    // two chunks of each of the first three classes, where available
    int first[3], second[3];
    for (int k = 0; k < 3; ++k) {
        first[k] = rep[k];
        second[k] = -1;
        for (int c = 0; c < nchunk; ++c)
            if (cls[c] == cls[rep[k]] && c != first[k]) { second[k] = c; break; }
    }
    const int A = first[0], B = first[1], Cc = first[2], A2 = second[0] >= 0 ? second[0] : first[0];
    const double step_bytes = (double)boards * (117 + 54);
    auto report = [&](const char *what, const Streams &s, double bytes_per_ply) {
        const float us = run(s, ntiles, T);
        printf("%-78s %8.1f us = %6.2f us per ply, %5.2f TB/s\n", what, us, us / T, bytes_per_ply * T / us / 1e6);
    };
    const int64_t half = 4 * GiB;  // a second array inside the same chunk
    report("obs alone, one array", Streams{{chunk(A), chunk(A)}, {nullptr, nullptr}}, boards * 117.0);
    report("obs alone, plies alternate between two arrays of ONE class (other chunk)", Streams{{chunk(A), chunk(A2)}, {nullptr, nullptr}}, boards * 117.0);
    report("obs alone, plies alternate between two arrays in TWO classes", Streams{{chunk(A), chunk(B)}, {nullptr, nullptr}}, boards * 117.0);
    report("mask alone, one array", Streams{{nullptr, nullptr}, {chunk(A, mask_off), chunk(A, mask_off)}}, boards * 54.0);
    report("mask alone, plies alternate between two classes", Streams{{nullptr, nullptr}, {chunk(A, mask_off), chunk(B, mask_off)}}, boards * 54.0);
    report("obs + mask, same class (same chunk)", Streams{{chunk(A), chunk(A)}, {chunk(A, mask_off), chunk(A, mask_off)}}, step_bytes);
    report("obs + mask, same class (different chunks)", Streams{{chunk(A), chunk(A)}, {chunk(A2, mask_off), chunk(A2, mask_off)}}, step_bytes);
    report("obs in A, mask in B", Streams{{chunk(A), chunk(A)}, {chunk(B, mask_off), chunk(B, mask_off)}}, step_bytes);
    report("obs plies alternate A / C, mask in B", Streams{{chunk(A), chunk(Cc)}, {chunk(B, mask_off), chunk(B, mask_off)}}, step_bytes);
    report("obs plies alternate A / B, mask plies alternate B / A", Streams{{chunk(A), chunk(B)}, {chunk(B, mask_off), chunk(A, mask_off)}}, step_bytes);
    report("obs plies alternate A / B, mask plies alternate A / B (each ply in one class)", Streams{{chunk(A), chunk(B)}, {chunk(A, mask_off), chunk(B, mask_off)}}, step_bytes);
    report("obs plies alternate A / C, mask plies alternate B / C", Streams{{chunk(A), chunk(Cc)}, {chunk(B, mask_off), chunk(Cc, mask_off)}}, step_bytes);
    report("obs plies alternate A / B, mask in C", Streams{{chunk(A), chunk(B)}, {chunk(Cc, mask_off), chunk(Cc, mask_off)}}, step_bytes);
    (void)half;
    printf("\nthink time per ply in front of the stores (x 64 clocks), obs in A / mask in B, us per ply: one wavefront per tile | a computing and a storing wavefront per tile\n");
    for (int think : {0, 15, 30, 45, 60, 90}) {
        float best[2] = {1e30f, 1e30f};
        Streams st{{chunk(A), chunk(A)}, {chunk(B, mask_off), chunk(B, mask_off)}};
        for (int r = -1; r < 5; ++r)
            for (int split = 0; split < 2; ++split) {
                CK(hipEventRecord(e0, 0));
                if (split) hipLaunchKernelGGL(k_think<true>, dim3((uint32_t)ntiles), dim3(128), 0, 0, st, ntiles, T, think);
                else hipLaunchKernelGGL(k_think<false>, dim3((uint32_t)ntiles), dim3(64), 0, 0, st, ntiles, T, think);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (r >= 0) best[split] = std::min(best[split], ms);
            }
        printf("  think %3d: %6.2f | %6.2f\n", think, best[0] * 1e3f / T, best[1] * 1e3f / T);
    }
    printf("\nstore cache policy (sc0 = 1, nt = 2, sc1 = 16) x placement, obs + mask, us per ply:\n");
    for (int aux : {0, 2, 18, 16, 1, 17, 19}) {
        g_aux = aux;
        const float same = run(Streams{{chunk(A), chunk(A)}, {chunk(A2, mask_off), chunk(A2, mask_off)}}, ntiles, T);
        const float two = run(Streams{{chunk(A), chunk(A)}, {chunk(B, mask_off), chunk(B, mask_off)}}, ntiles, T);
        const float three = run(Streams{{chunk(A), chunk(Cc)}, {chunk(B, mask_off), chunk(B, mask_off)}}, ntiles, T);
        printf("  policy %2d%s%s%s: one class %6.2f (%5.2f TB/s)   two classes %6.2f (%5.2f TB/s)   three %6.2f (%5.2f TB/s)\n", aux,
               aux & 2 ? " nt" : "", aux & 16 ? " sc1" : "", aux & 1 ? " sc0" : "", same / T, step_bytes * T / same / 1e6, two / T,
               step_bytes * T / two / 1e6, three / T, step_bytes * T / three / 1e6);
    }
    return 0;
}
