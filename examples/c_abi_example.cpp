// c_abi_example.cpp -- the C-ABI of include/gobblet_hip.h used with no Python and no torch: plain
// hipMalloc'ed buffers, one stream, masked-random play of N boards, tallies read back.
//   hipcc --offload-arch=gfx950 -O2 -I include examples/c_abi_example.cpp \
//         -L gobblet-rl_amd/csrc -lgobblet_hip -Wl,-rpath,$PWD/gobblet-rl_amd/csrc -o c_abi_example
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#include "gobblet_hip.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s\n", hipGetErrorString(e_)); return 2; } } while (0)
#define GBL_OK_(x) do { int r_ = (x); if (r_ != GBL_OK) { fprintf(stderr, "gbl error %d: %s\n", r_, gbl_last_error()); return 3; } } while (0)

int main(int argc, char **argv)
{
    const int64_t n = argc > 1 ? atoll(argv[1]) : 100000;
    const uint32_t plies = argc > 2 ? (uint32_t)atoi(argv[2]) : 50;
    int8_t *state, *to_move, *done, *winner, *reward, *mask, *obs;
    int32_t *actions;
    int64_t *counters;
    HIP_OK(hipMalloc(&state, n * GBL_CELLS)); HIP_OK(hipMalloc(&to_move, n)); HIP_OK(hipMalloc(&done, n));
    HIP_OK(hipMalloc(&winner, n)); HIP_OK(hipMalloc(&reward, 2 * n)); HIP_OK(hipMalloc(&mask, n * GBL_ACTIONS));
    HIP_OK(hipMalloc(&obs, n * GBL_OBS_BYTES)); HIP_OK(hipMalloc(&actions, 4 * n));
    const size_t cbytes = sizeof(int64_t) * GBL_COUNTER_STRIPES * GBL_COUNTER_STRIDE;
    HIP_OK(hipMalloc(&counters, cbytes)); HIP_OK(hipMemset(counters, 0, cbytes));
    hipStream_t s;
    HIP_OK(hipStreamCreate(&s));

    GBL_OK_(gbl_reset(state, to_move, done, winner, n, s));                       // raw_env.reset
    GBL_OK_(gbl_legal_mask(state, to_move, mask, n, s));                          // first action_mask
    for (uint32_t t = 0; t < plies; ++t) {                                        // sample + step, two launches per ply
        GBL_OK_(gbl_sample(mask, actions, n, /*seed*/ 0, /*env_base*/ 0, t, s));
        GBL_OK_(gbl_step(state, to_move, done, actions, winner, reward, mask, obs, nullptr, n, GBL_ILLEGAL_NOOP,
                         /*auto_reset*/ 1, s));
    }
    GBL_OK_(gbl_rollout(state, to_move, done, actions, winner, reward, mask, obs, n, 0, 0, plies, plies,  // the same,
                        GBL_ILLEGAL_NOOP, counters, nullptr, s));                                       // fused, counted
    HIP_OK(hipStreamSynchronize(s));

    std::vector<int64_t> c(GBL_COUNTER_STRIPES * GBL_COUNTER_STRIDE);
    HIP_OK(hipMemcpy(c.data(), counters, cbytes, hipMemcpyDeviceToHost));
    int64_t tot[4] = {0, 0, 0, 0};
    for (int st = 0; st < GBL_COUNTER_STRIPES; ++st)
        for (int k = 0; k < 4; ++k) tot[k] += c[st * GBL_COUNTER_STRIDE + k];
    std::vector<int8_t> m(GBL_ACTIONS);
    HIP_OK(hipMemcpy(m.data(), mask, GBL_ACTIONS, hipMemcpyDeviceToHost));
    int legal0 = 0;
    for (int a = 0; a < GBL_ACTIONS; ++a) legal0 += m[a];
    printf("boards %lld: %u + %u plies; rollout tallies: plies %lld games %lld p1 %lld p2 %lld; board 0 has %d legal moves\n",
           (long long)n, plies, plies, (long long)tot[0], (long long)tot[1], (long long)tot[2], (long long)tot[3], legal0);
    if (gbl_winner(state + 1, winner, n, s) != GBL_ERR_ALIGN) return 4;          // misaligned row buffer is refused
    printf("misaligned pointer refused: %s\n", gbl_last_error());
    return (tot[0] == n * (int64_t)plies && tot[1] == tot[2] + tot[3] && legal0 >= 10) ? 0 : 5;
}
