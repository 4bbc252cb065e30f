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

// ---- moves of PLACED pieces through a VIRTUAL root (round 3: proven, pinned here, measured in a kernel, not shipped) ---------
// A move of our placed piece i from its square to q leaves the opponent exactly "P with piece i lifted off the board, then
// piece i placed from hand on q": with P_i := P minus piece i as the root, the root rule (gobblet_device.h, above GreedyRoot)
// applies verbatim -- every winning reply to the move is one of the opponent's winning moves R_i on P_i, minus those q
// defuses, provided q is not a risky square OF P_i.  (What the lift exposes is part of P_i and so of R_i; a P_i on which the
// opponent holds a line has nearly every move in R_i and falls out of the rule by its size.)  `legal_me` in the summaries stays
// the REAL root's (greedy_policy.py:141 tests is_legal on the board the policy was handed).  One evaluation of P_i -- as
// expensive as one pair -- settles ALL moves of that piece: on the masked-random mix 3.3 of the 5.5 evaluated pairs per board
// go for 0.6 virtual roots and 1.3 more items per board (the largest movable piece alone: 2.5 for 0.43 and 0.9).
// emu_greedy_vroot_rule checks every candidate it settles against the exact evaluation.  A kernel built on it (virtual
// roots evaluated by the non-owner wavefronts while the owners walk depth 1, a merge phase after the items) gave bit-exact
// decisions on the GPU and shortened the pair phase by a third, but LOST overall -- 65 536 boards 14.7 -> 15.5 us (one virtual
// root per board) / 17.7 us (three), 2^20 boards 178 -> 174 / 221 us: the owners' extra planning and listing, the merge phase
// with its barrier and the heavier summary lookup cost more than the pairs saved.
__device__ __forceinline__ Planes greedy_lifted(const Planes &p, int me, uint32_t pi)
{
    const uint32_t k = pi >> 1;
    const uint32_t mine = me ? (p.nz & p.neg) : (p.nz & ~p.neg);
    const uint32_t ploc = mine & ((pi & 1u) ? ~p.odd : p.odd) & (0x1FFu << (9u * k));
    return Planes{p.nz & ~ploc, p.neg & ~ploc, p.odd & ~ploc};
}

}  // namespace gbl
