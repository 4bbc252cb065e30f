// c_abi_example.cpp -- the C-ABI of include/gobblet_hip.h used with no Python and no torch: plain
// hipMalloc'ed buffers, one stream, masked-random play of N boards, tallies read back.
//   hipcc --offload-arch=gfx950 -O2 -I include examples/c_abi_example.cpp \
//         -L gobblet-rl_amd/csrc -lgobblet_hip -Wl,-rpath,$PWD/gobblet-rl_amd/csrc -o c_abi_example
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#include "gobblet_hip.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s\n", hipGetErrorString(e_)); return 2; } } while (0)
#define GBL_OK_(x) do { int r_ = (x); if (r_ != GBL_OK) { fprintf(stderr, "gbl error %d: %s\n", r_, gbl_last_error()); return 3; } } while (0)

// Where the two large trajectory arrays of gbl_collect go (speed only, see gbl_placement_probe in the header): either
// array at the head of a hipMalloc block of its own of at least 2 GiB; while the pair shares one of the device's three
// 96 GiB memory classes (probe ratio both / (a + b) near 1.0), take another block for one of the two -- behind a gap that
// doubles after every plain conflict -- and probe it against the other array's first block; stop at the first clean
// pair (ratio <= 0.83), else keep the best seen.  Never hold more than 64 GiB or a quarter of the free memory, leave
// 4 GiB alone, stop when the device refuses a block; rejected blocks and gaps are freed before returning.
// (The same search as gobblet-rl_amd/placement.py.)  Returns the ratio of the pair handed back in *a / *b.
static double place_pair(int64_t bytes_a, int64_t bytes_b, int64_t slot_boards, int plies, void **a, void **b, hipStream_t s)
{
    const int64_t GiB = 1ll << 30, kMinBlock = 2 * GiB, kReserve = 4 * GiB, kMaxGap = 32 * GiB;
    auto block = [&](int64_t n) { return std::max(kMinBlock, (n + (2 << 20) - 1) / (2 << 20) * (2 << 20)); };
    const int64_t blk[2] = {block(bytes_a), block(bytes_b)};
    int64_t free0 = 0;
    if (gbl_device_memory(&free0, nullptr) != GBL_OK || free0 < blk[0] + blk[1] + kReserve) return -1.0;
    const int64_t cap = std::max(std::min(64 * GiB, free0 / 4), blk[0] + blk[1]);
    std::vector<void *> pool[2], gaps;
    void *p = nullptr;
    for (int k = 0; k < 2; ++k) {
        if (gbl_block_alloc(blk[k], &p) != GBL_OK) { for (void *q : pool[0]) gbl_block_free(q); return -1.0; }
        pool[k].push_back(p);
    }
    int64_t held = blk[0] + blk[1], gap = kMinBlock;
    double best = 1e9, last = 0;
    size_t ia = 0, ib = 0;
    auto probe = [&](size_t i, size_t j) {
        float both = 0, ua = 0, ub = 0;
        if (gbl_placement_probe(pool[0][i], bytes_a, pool[1][j], bytes_b, slot_boards, plies, &both, &ua, &ub, s) != GBL_OK) return;
        last = both / (ua + ub);
        if (last < best) { best = last; ia = i; ib = j; }
    };
    probe(0, 0);
    for (int grow = 1, probes = 1; best > 0.83 && probes < 16; grow ^= 1, ++probes) {
        int64_t room = 0;
        gbl_device_memory(&room, nullptr);
        if (held + blk[grow] > cap || room < blk[grow] + kReserve) break;
        if (last > 0.96 && held + gap + blk[grow] <= cap && room >= gap + blk[grow] + kReserve) {  // plain conflict: a gap first
            if (gbl_block_alloc(gap, &p) != GBL_OK) break;
            gaps.push_back(p); held += gap; gap = std::min(2 * gap, kMaxGap);
        }
        if (gbl_block_alloc(blk[grow], &p) != GBL_OK) break;
        pool[grow].push_back(p); held += blk[grow];
        grow == 0 ? probe(pool[0].size() - 1, 0) : probe(0, pool[1].size() - 1);
    }
    *a = pool[0][ia]; *b = pool[1][ib];
    for (void *q : gaps) gbl_block_free(q);
    for (int k = 0; k < 2; ++k)
        for (void *q : pool[k]) if (q != *a && q != *b) gbl_block_free(q);
    return best;
}

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
    // ... the same in ONE launch per ply (gbl_step_ex, ABI version 2): the step draws the next mover's action from the mask it stores,
    // over the action array itself, and flags what became of every action -- here a garbage index on board 0 first
    int8_t *status;
    HIP_OK(hipMalloc(&status, n));
    GBL_OK_(gbl_sample(mask, actions, n, 0, 0, plies, s));
    const int32_t garbage = 1000;
    HIP_OK(hipMemcpyAsync(actions, &garbage, 4, hipMemcpyHostToDevice, s));
    for (uint32_t t = plies; t < plies + 4; ++t)
        GBL_OK_(gbl_step_ex(state, to_move, done, actions, winner, reward, mask, obs, nullptr, nullptr, nullptr, nullptr,
                            t == plies ? status : nullptr, /*next_actions_out = */ actions, 0, 0, t + 1, nullptr, n,
                            GBL_ILLEGAL_NOOP, 1, s));
    int8_t status0 = -1;
    HIP_OK(hipMemcpyAsync(&status0, status, 1, hipMemcpyDeviceToHost, s));
    HIP_OK(hipStreamSynchronize(s));
    if (status0 != (GBL_STATUS_ILLEGAL | GBL_STATUS_OUT_OF_RANGE)) { fprintf(stderr, "status byte of a garbage index: %d\n", status0); return 8; }
    int32_t info[6];
    GBL_OK_(gbl_layout_info(info));
    if (info[0] != GBL_ABI_VERSION) return 9;
    GBL_OK_(gbl_rollout(state, to_move, done, actions, winner, reward, mask, obs, n, 0, 0, plies + 4, plies,  // the same,
                        GBL_ILLEGAL_NOOP, counters, nullptr, s));                                           // fused, counted
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
    // a collected trajectory (T plies in one launch, every ply kept) into arrays placed by the probe
    const uint32_t T = 16;
    const int64_t slot = (n + 127) / 128 * 128;
    void *obs_traj = nullptr, *mask_traj = nullptr;
    int32_t *act_traj;
    HIP_OK(hipMalloc(&act_traj, 4 * T * slot));
    const double ratio = place_pair(T * slot * GBL_OBS_BYTES, T * slot * GBL_ACTIONS, slot, (int)T, &obs_traj, &mask_traj, s);
    if (ratio < 0) { fprintf(stderr, "placement: not enough free device memory\n"); return 6; }
    GBL_OK_(gbl_collect(state, to_move, done, act_traj, nullptr, nullptr, nullptr, nullptr, (int8_t *)mask_traj, (int8_t *)obs_traj, n,
                        slot, 64, 0, 0, 2 * plies + 4, nullptr, T, GBL_ILLEGAL_NOOP, counters, nullptr, s));
    HIP_OK(hipStreamSynchronize(s));
    std::vector<int32_t> a0(T * slot);
    std::vector<int8_t> m0(GBL_ACTIONS * 2);
    HIP_OK(hipMemcpy(a0.data(), act_traj, 4 * T * slot, hipMemcpyDeviceToHost));
    int played_legal = 1;
    for (uint32_t t = 0; t < T; ++t) played_legal &= a0[t * slot] >= 0 && a0[t * slot] < GBL_ACTIONS;
    printf("trajectory arrays placed: probe ratio %.2f (near 0.8: different memory classes; near 1.0: the same); "
           "kernel variant %d; board 0 played %u legal actions\n", ratio, gbl_collect_variant(n, T, 1, 1), T);
    GBL_OK_(gbl_block_free(obs_traj)); GBL_OK_(gbl_block_free(mask_traj));
    if (!played_legal) return 7;
    if (gbl_winner(state + 1, winner, n, s) != GBL_ERR_ALIGN) return 4;          // misaligned row buffer is refused
    printf("misaligned pointer refused: %s\n", gbl_last_error());
    return (tot[0] == n * (int64_t)plies && tot[1] == tot[2] + tot[3] && legal0 >= 10) ? 0 : 5;
}
