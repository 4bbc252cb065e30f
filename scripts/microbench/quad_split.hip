// quad_split.hip -- round 6, VERDICT r05 item 3: does splitting ONE board's ply over the lanes of a quad (DPP combines) shorten the
// serial chain of a lone wavefront?  The chain of the T-plies-per-launch kernels -- legal mask -> k-th-bit pick -> move -> winner ->
// reset -- played `iters` times by one wavefront per CU (nothing stored inside the loop: "chain-only"), in three forms that must
// reach the same final checksum per board:
//   V = 0  one lane per board: legal54 + pick54 + winner_of                               (k_collect / k_collect2 / k_collect3's player)
//   V = 1  lane PAIRS: the winner test split by colour over one DPP exchange              (the role kernel as shipped)
//   V = 2  QUADS: the pick split four ways (14 bits per lane, counts combined with DPP quad_perm reads, the hit OR-reduced) and the
//          winner split by colour AND by line word (two lanes per colour, partial match masks OR-ed with one DPP) -- what
//          north_star's "ballot / shfl across the lanes of a board" comes to once the sub-results have to be combined
// Prints shader cycles per ply (s_memtime, median over the wavefronts) and, from the ISA, nothing: count instructions with
//   /opt/rocm/lib/llvm/bin/llvm-objdump -d --offloading ... (profiles/r06/quad_split.txt has both).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I gobblet-rl_amd/csrc -o /tmp/quad_split scripts/microbench/quad_split.hip
#include "gobblet_device.h"

#include <algorithm>
#include <stdio.h>
#include <vector>

using namespace gbl;

template <int CTRL>
__device__ __forceinline__ uint32_t quad_read(uint32_t x)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xf, 0xf, false);
}

// k-th legal action over a QUAD: lane j holds bits [14 j, 14 j + 14) of the mask
__device__ __forceinline__ int pick54_quad(uint64_t m, uint32_t r, int j)
{
    const uint32_t part = (uint32_t)(m >> (14 * j)) & 0x3FFFu;
    const uint32_t cnt = __popc(part);
    const uint32_t c0 = quad_read<0x00>(cnt), c1 = quad_read<0x55>(cnt), c2 = quad_read<0xAA>(cnt), c3 = quad_read<0xFF>(cnt);
    const uint32_t n = c0 + c1 + c2 + c3;
    const uint32_t pre = (j > 0 ? c0 : 0u) + (j > 1 ? c1 : 0u) + (j > 2 ? c2 : 0u);
    const uint32_t k = __umulhi(r, n), kk = k - pre;
    const bool mine = kk < cnt;  // (unsigned: k < pre wraps)
    uint32_t w = part, q = mine ? kk : 0u, pos = 0;
#pragma unroll
    for (int s = 8; s >= 1; s >>= 1) {
        const uint32_t c = __popc(w & ((1u << s) - 1u));
        const bool up = q >= c;
        q = up ? q - c : q;
        w = up ? (w >> s) : w;
        pos += up ? s : 0;
    }
    uint32_t a = mine ? 14u * (uint32_t)j + pos : 0u;
    a |= quad_read<0xB1>(a);  // [1, 0, 3, 2]
    a |= quad_read<0x4E>(a);  // [2, 3, 0, 1]
    return n ? (int)a : -1;
}

// winner over a QUAD: lane j walks colour j & 1, line words by j >> 1 (WA + half of the rest / WB + WC), partial masks OR-ed
__device__ __forceinline__ int winner_of_quad(const Planes &p, int j)
{
    const uint32_t flip = (j & 1) ? 0u : ~0u;
    const uint32_t side = p.nz & (p.neg ^ flip), nz = p.nz;
    const uint32_t o1 = (nz >> 9) & 0x1FFu, o2 = (nz >> 18) & 0x1FFu;
    const uint32_t t = ((side >> 18) & 0x1FFu) | (~o2 & (((side >> 9) & 0x1FFu) | (~o1 & (side & 0x1FFu))));
    constexpr uint32_t L[8] = {0x007u, 0x038u, 0x1C0u, 0x049u, 0x092u, 0x124u, 0x111u, 0x054u};
    constexpr uint32_t LOW3 = 0x00100401u, G3 = LOW3 << 9, F3 = LOW3 * 0x1FFu;
    constexpr uint32_t WA = L[0] | (L[3] << 10) | (L[6] << 20), WB = L[1] | (L[4] << 10) | (L[7] << 20);
    constexpr uint32_t WC = L[2] | (L[5] << 10), NONE = 1u << 20;
    const uint32_t n = ~(t | (t << 10) | (t << 20));
    uint32_t mine;
    if (j & 2) mine = ((G3 & ~((WB & n) + F3)) >> 8) | ((G3 & ~(((WC & n) | NONE) + F3)) >> 7);
    else mine = (G3 & ~((WA & n) + F3)) >> 9;
    mine |= quad_read<0x4E>(mine);                       // the other half of my colour's lines: [2, 3, 0, 1]
    const uint32_t theirs = quad_read<0xB1>(mine);       // the other colour's: [1, 0, 3, 2]
    const int w = mine > theirs ? 1 : (theirs > mine ? -1 : 0);
    return (j & 1) ? -w : w;
}

template <int V>
__global__ __launch_bounds__(64) void k_chain(uint32_t *sums, unsigned long long *cycles, int iters)
{
    constexpr int LPB = V == 0 ? 1 : V == 1 ? 2 : 4;
    const int lane = threadIdx.x, j = lane & (LPB - 1);
    const uint32_t id = blockIdx.x * 1024u + (uint32_t)(lane / LPB);  // (the lanes of a board share its id; index i < 16 exists in every form)
    Planes p{0u, 0u, 0u};
    int mover = 0;
    uint32_t word = id * 2654435761u + 12345u, sum = 0;
    uint64_t legal = kLegalEmpty;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < iters; ++t) {
        int a;
        if constexpr (V == 2) a = pick54_quad(legal, word, j);
        else a = pick54(legal, word);
        word = word * 1664525u + 1013904223u;
        move_planes(p, mover, (uint32_t)a);
        mover ^= 1;
        int w;
        if constexpr (V == 0) w = winner_of(p);
        else if constexpr (V == 1) w = winner_of_pair(p, j);
        else w = winner_of_quad(p, j);
        uint64_t next = legal54(p, mover);
        if (w != 0) {
            p = Planes{0u, 0u, 0u};
            mover = 0;
            next = kLegalEmpty;
        }
        legal = next;
        sum = sum * 31u + (uint32_t)a + (uint32_t)(w + 1) * 64u;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (j == 0) sums[blockIdx.x * 64 + lane / LPB] = sum;
    if (lane == 0) cycles[blockIdx.x] = t1 - t0;
}

static double g_us;  // wall clock of the timed launch per ply (HIP events)

template <int V>
double run(int blocks, int iters, std::vector<uint32_t> &out)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    uint32_t *ds;
    unsigned long long *dc;
    hipMalloc(&ds, blocks * 64 * 4);
    hipMalloc(&dc, blocks * 8);
    hipMemset(ds, 0, blocks * 64 * 4);
    k_chain<V><<<blocks, 64>>>(ds, dc, iters);  // warm
    hipEventRecord(e0);
    k_chain<V><<<blocks, 64>>>(ds, dc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    g_us = ms * 1e3 / iters;
    out.resize(blocks * 64);
    std::vector<unsigned long long> c(blocks);
    hipMemcpy(out.data(), ds, blocks * 64 * 4, hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), dc, blocks * 8, hipMemcpyDeviceToHost);
    std::sort(c.begin(), c.end());
    hipFree(ds); hipFree(dc);
    return (double)c[blocks / 2] / iters;
}

int main()
{
    const int blocks = 256, iters = 4096;
    std::vector<uint32_t> s0, s1, s2;
    const double c0 = run<0>(blocks, iters, s0), u0 = g_us, c1 = run<1>(blocks, iters, s1), u1 = g_us, c2 = run<2>(blocks, iters, s2), u2 = g_us;
    // boards of the same id: form 0 has 64 per block, form 1: 32, form 2: 16 -- ids are block * 1024 + index, so index i agrees
    int bad = 0;
    for (int b = 0; b < blocks; ++b)
        for (int i = 0; i < 16; ++i) {
            if (s0[b * 64 + i] != s1[b * 64 + i]) ++bad;
            if (s0[b * 64 + i] != s2[b * 64 + i]) ++bad;
        }
    printf("chain-only, one wavefront per CU, %d plies: s_memtime ticks per ply (median wavefront) | us per ply (HIP events, whole launch)\n", iters);
    printf("  one lane per board  (pick54, winner_of)            %8.1f | %.3f\n", c0, u0);
    printf("  lane pairs          (pick54, winner_of_pair)       %8.1f | %.3f\n", c1, u1);
    printf("  quads               (pick54_quad, winner_of_quad)  %8.1f | %.3f\n", c2, u2);
    printf("  checksums of the three forms: %s\n", bad ? "DIFFER" : "identical");
    return bad != 0;
}
