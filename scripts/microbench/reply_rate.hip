// reply_rate.hip -- what does ONE depth-2 pair evaluation (greedy_reply: moved + legal54 + outcomes54 + summary) cost a SIMD,
// with 1, 2, 4 wavefronts per SIMD and nothing else in the way (no LDS, no barriers)?  Every lane evaluates candidates of
// its own board in a loop; cycles per evaluation per SIMD = launch time x clock / (evaluations per wavefront x wavefronts per SIMD).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I gobblet-rl_amd/csrc -o reply_rate reply_rate.hip && ./reply_rate
#include "gobblet_device.h"

#include <stdio.h>
#include <vector>

using namespace gbl;

template <int WHAT>
__global__ __launch_bounds__(256) void k(const uint32_t *__restrict__ planes, uint32_t *__restrict__ out, int iters)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    Planes p{planes[3 * i], planes[3 * i + 1], planes[3 * i + 2]};
    const int me = i & 1;
    const uint64_t legal = legal54(p, me);
    uint32_t acc = 0, a = (uint32_t)(i % 54);
    for (int it = 0; it < iters; ++it) {
        if (WHAT == 0) acc += greedy_reply(p, me, legal, a);
        if (WHAT == 1) { uint64_t w, l; outcomes54<true>(p, me, w, l); acc += (uint32_t)w ^ (uint32_t)(l >> 7); p.odd ^= acc & 1u; }
        if (WHAT == 2) acc += (uint32_t)greedy_undefused(p, me, a);
        if (WHAT == 3) { uint64_t w, l; outcomes54<false>(p, me, w, l); acc += (uint32_t)w ^ (uint32_t)(l >> 7); p.odd ^= acc & 1u; }
        if (WHAT == 4) acc += greedy_reply<true>(p, me, legal, a) + (greedy_pair_is_plain(p, me, a) ? 1u : 0u);
        a = a + 7 + (acc & 1u);
        a = a >= 54 ? a - 54 : a;
    }
    out[i] = acc;
}

int main()
{
    const int cus = 256, max_wg = 8;  // workgroups of 4 wavefronts: one per SIMD
    const int n = cus * max_wg * 256;
    std::vector<uint32_t> h(3 * n);
    uint32_t s = 12345;
    for (int i = 0; i < n; ++i) {  // plausible boards: a few pieces per level, no piece twice is not required for timing
        s = s * 1664525u + 1013904223u; uint32_t nz = s & (s >> 5) & (s >> 11) & 0x7FFFFFFu;
        s = s * 1664525u + 1013904223u; h[3 * i] = nz; h[3 * i + 1] = s & nz; h[3 * i + 2] = (s >> 9) & nz;
    }
    uint32_t *dp, *dout;
    (void)hipMalloc(&dp, h.size() * 4); (void)hipMalloc(&dout, n * 4);
    (void)hipMemcpy(dp, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const char *names[5] = {"greedy_reply (ordered form)", "outcomes54<QUIET>", "greedy_undefused (item)", "outcomes54 (general)",
                            "greedy_reply (fast form + test)"};
    const int iters = 200;
    for (int what = 0; what < 5; ++what)
        for (int wps = 1; wps <= 8; wps *= 2) {  // wavefronts per SIMD
            const int grid = cus * wps;
            auto launch = [&] {
                if (what == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, dp, dout, iters);
                if (what == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, dp, dout, iters);
                if (what == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, dp, dout, iters);
                if (what == 3) hipLaunchKernelGGL(k<3>, dim3(grid), dim3(256), 0, 0, dp, dout, iters);
                if (what == 4) hipLaunchKernelGGL(k<4>, dim3(grid), dim3(256), 0, 0, dp, dout, iters);
            };
            launch(); launch();
            (void)hipEventRecord(e0, 0);
            for (int r = 0; r < 5; ++r) launch();
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            const double us = ms * 1e3 / 5, cycles = us * 2.2e3;  // (shader clock ~2.2 GHz under load)
            printf("%-32s %d wavefront(s) per SIMD: %8.1f us per launch = %7.1f cycles per evaluation per wavefront, %7.1f per SIMD\n",
                   names[what], wps, us, cycles / iters, cycles / iters / wps);
        }
    return 0;
}
