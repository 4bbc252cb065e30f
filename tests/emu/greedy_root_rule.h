// greedy_root_rule.h -- test-side only: the LOOP forms of the root rule (gobblet_device.h: greedy_root, greedy_undefused --
// the rule, its proof and the table forms the kernel uses live there since round 3), kept as the cross-check the
// emulation compares the kernel's table forms and the exact evaluation with (emu_greedy_root_rule, test_device_emulation.py).
#pragma once
#include "../../gobblet-rl_amd/csrc/gobblet_device.h"

namespace gbl {

// The candidate sets of greedy_replay_closed for all placements from hand at once (mask with the resolved candidates).
__device__ __forceinline__ GreedyHandSets greedy_hand_sets(const Planes &p, int me, uint64_t replies, uint64_t legal_me)
{
    GreedyHandSets s{0ull, 0ull, 0ull, 0ull};
    for (uint64_t it = replies; it; it &= it - 1) {  // ascending: the reference's reply order
        const uint32_t a2 = (uint32_t)__builtin_ctzll(it);
        const uint64_t u = greedy_undefused(p, me, a2);
        s.second |= u & s.threat;
        if ((legal_me >> a2) & 1ull) {
            s.flegal |= u & ~s.threat;  // the FIRST winning reply is a legal move of ours
            s.block |= u;
        }
        s.threat |= u;
    }
    return s;
}

// the 16-bit summary (see below) of ONE resolved placement from hand
__device__ __forceinline__ uint32_t greedy_hand_summary(const Planes &p, int me, uint64_t replies, uint64_t legal_me, uint32_t a)
{
    uint64_t ow = 0;
    for (uint64_t it = replies; it; it &= it - 1) {
        const uint32_t a2 = (uint32_t)__builtin_ctzll(it);
        if ((greedy_undefused(p, me, a2) >> a) & 1ull) ow |= 1ull << a2;
    }
    const uint64_t block = ow & legal_me;
    uint32_t s = ow ? 1u : 0u;
    s |= (ow ? (uint32_t)__builtin_ctzll(ow) : 0u) << 1;
    s |= (ow & (ow - 1)) ? 1u << 7 : 0u;
    s |= block ? 1u << 8 : 0u;
    s |= (block ? (uint32_t)__builtin_ctzll(block) : 0u) << 9;
    return s;
}

}  // namespace gbl
