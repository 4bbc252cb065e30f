// winner_lanes.hip -- microbenchmark (not product code): the 8-line win reduction of Board.check_for_winner
// (board.py:183-194) in the two shapes that were on the table:
//   A  one board per lane   -- the product's shape: both colours' eight line tests as packed arithmetic in the lane
//                              (three lines per word in 10-bit fields), no cross-lane traffic;
//   B  eight lanes per board -- north_star's suggestion: lane l of a group of eight tests line l, one __ballot per
//                              colour gathers the eight match bits of every board of the wavefront, the byte of
//                              the ballot that belongs to the board is compared as an integer (highest line wins).
// Input per board: the tops as two 9-bit sets (player_1 | player_2 << 16), resident in HBM; output int8 winner.
// Both kernels read 4 B and write 1 B per board; B needs eight times the wavefronts.
//     hipcc --offload-arch=gfx950 -O3 -o winner_lanes winner_lanes.hip && ./winner_lanes
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

__constant__ uint32_t c_line[8] = {0x007u, 0x038u, 0x1C0u, 0x049u, 0x092u, 0x124u, 0x111u, 0x054u};  // board.py:135-153

// A: the packed form of gobblet_device.h winner_of(), on tops
__global__ __launch_bounds__(256) void k_lane_per_board(const uint32_t *__restrict__ tops, int8_t *__restrict__ win, int n)
{
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n) return;
    uint32_t t = tops[b], t1 = t & 0x1FFu, t2 = (t >> 16) & 0x1FFu;
    constexpr uint32_t L[8] = {0x007u, 0x038u, 0x1C0u, 0x049u, 0x092u, 0x124u, 0x111u, 0x054u};
    constexpr uint32_t LOW3 = 0x00100401u, G3 = LOW3 << 9, F3 = LOW3 * 0x1FFu;
    constexpr uint32_t WA = L[0] | (L[3] << 10) | (L[6] << 20), WB = L[1] | (L[4] << 10) | (L[7] << 20);
    constexpr uint32_t WC = L[2] | (L[5] << 10), NONE = 1u << 20;
    uint32_t n1 = ~(t1 | (t1 << 10) | (t1 << 20)), n2 = ~(t2 | (t2 << 10) | (t2 << 20));
    uint32_t m1 = ((G3 & ~((WA & n1) + F3)) >> 9) | ((G3 & ~((WB & n1) + F3)) >> 8) | ((G3 & ~(((WC & n1) | NONE) + F3)) >> 7);
    uint32_t m2 = ((G3 & ~((WA & n2) + F3)) >> 9) | ((G3 & ~((WB & n2) + F3)) >> 8) | ((G3 & ~(((WC & n2) | NONE) + F3)) >> 7);
    win[b] = m2 > m1 ? -1 : (m1 > m2 ? 1 : 0);
}

// B: eight lanes per board, ballot
__global__ __launch_bounds__(256) void k_eight_lanes(const uint32_t *__restrict__ tops, int8_t *__restrict__ win, int n)
{
    int g = blockIdx.x * blockDim.x + threadIdx.x;  // lane index over all boards x 8
    int b = g >> 3, l = g & 7;
    uint32_t t = b < n ? tops[b] : 0u, t1 = t & 0x1FFu, t2 = (t >> 16) & 0x1FFu;
    uint32_t line = c_line[l];
    unsigned long long a1 = __ballot((t1 & line) == line), a2 = __ballot((t2 & line) == line);
    int grp = (threadIdx.x & 63) >> 3;  // which byte of the ballot is this board's
    uint32_t m1 = (uint32_t)(a1 >> (8 * grp)) & 0xFFu, m2 = (uint32_t)(a2 >> (8 * grp)) & 0xFFu;
    if (l == 0 && b < n) win[b] = m2 > m1 ? -1 : (m1 > m2 ? 1 : 0);
}

static int host_winner(uint32_t t)
{
    static const uint32_t L[8] = {0x007u, 0x038u, 0x1C0u, 0x049u, 0x092u, 0x124u, 0x111u, 0x054u};
    uint32_t t1 = t & 0x1FFu, t2 = (t >> 16) & 0x1FFu;
    int w = 0;
    for (int l = 0; l < 8; ++l) {  // no early exit: the last matching line decides
        if ((t1 & L[l]) == L[l]) w = 1;
        if ((t2 & L[l]) == L[l]) w = -1;
    }
    return w;
}

int main()
{
    const int n = 1 << 20, reps = 200;
    std::vector<uint32_t> h(n);
    uint64_t s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) {  // random tops: every square empty / player_1 / player_2
        uint32_t t1 = 0, t2 = 0;
        for (int q = 0; q < 9; ++q) {
            s ^= s << 13; s ^= s >> 7; s ^= s << 17;
            int v = (int)(s % 3);
            if (v == 1) t1 |= 1u << q;
            if (v == 2) t2 |= 1u << q;
        }
        h[i] = t1 | (t2 << 16);
    }
    uint32_t *d_t;
    int8_t *d_a, *d_b;
    CK(hipMalloc(&d_t, n * 4)); CK(hipMalloc(&d_a, n)); CK(hipMalloc(&d_b, n));
    CK(hipMemcpy(d_t, h.data(), n * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms[2];
    for (int v = 0; v < 2; ++v) {
        for (int r = 0; r < reps + 10; ++r) {
            if (r == 10) CK(hipEventRecord(e0));
            if (v == 0) hipLaunchKernelGGL(k_lane_per_board, dim3(n / 256), dim3(256), 0, 0, d_t, d_a, n);
            else hipLaunchKernelGGL(k_eight_lanes, dim3(n * 8 / 256), dim3(256), 0, 0, d_t, d_b, n);
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms[v], e0, e1));
    }
    std::vector<int8_t> a(n), b(n);
    CK(hipMemcpy(a.data(), d_a, n, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), d_b, n, hipMemcpyDeviceToHost));
    int bad = 0, cnt[3] = {0, 0, 0};
    for (int i = 0; i < n; ++i) {
        int w = host_winner(h[i]);
        bad += (a[i] != w) + (b[i] != w);
        cnt[w + 1]++;
    }
    printf("# 2^20 boards, tops resident in HBM, %d launches each; winners -1/0/+1: %d/%d/%d; mismatches vs host: %d\n", reps,
           cnt[0], cnt[1], cnt[2], bad);
    printf("A one board per lane  (packed line arithmetic): %8.2f us per launch  %6.1f G boards/s\n", ms[0] / reps * 1e3, n / (ms[0] / reps * 1e-3) / 1e9);
    printf("B eight lanes per board (__ballot per colour)  : %8.2f us per launch  %6.1f G boards/s   (%.1fx A)\n", ms[1] / reps * 1e3,
           n / (ms[1] / reps * 1e-3) / 1e9, ms[1] / ms[0]);
    return bad != 0;
}
