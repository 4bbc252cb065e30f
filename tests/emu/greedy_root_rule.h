// greedy_root_rule.h -- test-side only (not part of the product): a rule about Gobblet that would let k_greedy settle
// two thirds of its (board, candidate) pairs from the root position alone.  It is PROVEN below and pinned here against
// the exact evaluation on every candidate of the emulation's test boards (emu_greedy_root_rule, test_device_emulation.py);
// a first kernel built on it lost (the per-board loop over the opponent's winning moves runs at the pace of a
// wavefront's slowest lane: 65 536 boards 16.8 -> 21.3 us) -- the pooled form of that loop is the next step.
#pragma once
#include "../../gobblet-rl_amd/csrc/gobblet_device.h"

namespace gbl {

// ---- placements from hand: resolved from the root alone -------------------------------------------------------------
// Let R be the OPPONENT's winning moves on the root, were it to move.  After a placement `a` of ours from hand on q
// every winning reply is in R: a reply that wins after `a` is legal on the root too (we only covered q) and finds there
// the same tops except that q is what it was instead of ours -- which only helps the opponent -- or, if it gobbles our
// new piece, exactly the same tops.  And a2 = (piece pj, square q2) in R stops winning ("is defused") exactly if
//   q is where pj stands (we gobbled it: it cannot move), or
//   q = q2 and our piece is at least as large as pj (the reply is no longer legal), or
//   q != q2 lies on EVERY line the opponent holds after a2 (each of them now has our piece on it),
// provided no lift by the opponent can hand US a line (reply_is_plain), which with "have" = our tops and our pieces
// directly under the opponent's is the case iff q does not complete a line inside have ("risky" squares).  So for the
// placements from hand on non-risky squares the whole summary -- first / second winning reply, the first one we could
// play ourselves -- follows from R and a few set operations per member of R, with no depth-2 evaluation at all: two
// thirds of the (board, candidate) pairs of the masked-random mix.  Placements on risky squares go to the exact
// evaluation.  (R empty -- half of the boards -- means: none of these placements has a winning reply.)
struct GreedyRoot {
    uint64_t replies;    // R: the opponent's winning moves on the root
    uint32_t risky;      // 9 bits: squares where a placement of ours is not "plain" (0x1FF if have already holds a line)
};

__device__ __forceinline__ uint64_t spread9(uint32_t squares)  // a 9-bit set of squares under all six pieces
{
    uint64_t x = squares & 0x1FFu;
    x |= x << 9;
    return x | x << 18 | x << 36;
}

__device__ __forceinline__ GreedyRoot greedy_root(const Planes &p, int me)
{
    GreedyRoot g;
    uint64_t ow, ol;
    outcomes54(p, 1 - me, ow, ol);  // (the general form: the root itself need not be free of lines)
    g.replies = ow & legal54(p, 1 - me);
    const uint32_t mine = me ? (p.nz & p.neg) : (p.nz & ~p.neg);
    const uint32_t othr = me ? (p.nz & ~p.neg) : (p.nz & p.neg);
    // have: as in reply_is_plain(d1, opponent) -- our tops, and ours directly below a piece of the opponent's
    const uint32_t t0 = mine & 0x1FFu, t1 = (mine >> 9) & 0x1FFu, t2 = (mine >> 18) & 0x1FFu;
    const uint32_t m1 = (othr >> 9) & 0x1FFu, m2 = (othr >> 18) & 0x1FFu;
    const uint32_t o1 = m1 | t1, o2 = m2 | t2;
    const uint32_t To = t2 | (~o2 & (t1 | (~o1 & t0)));
    const uint32_t X = (m1 & t0) | (m2 & (t1 | (~o1 & t0)));
    const uint32_t T = (To | X) & 0x1FFu;
    constexpr uint32_t L[8] = {0x007u, 0x038u, 0x1C0u, 0x049u, 0x092u, 0x124u, 0x111u, 0x054u};
    uint32_t risky = 0;
    bool full = false;
#pragma unroll
    for (int l = 0; l < 8; ++l) {
        const uint32_t miss = L[l] & ~T;
        full = full || miss == 0;
        if ((miss & (miss - 1)) == 0) risky |= miss;  // exactly one square missing: that square completes the line
    }
    g.risky = full ? 0x1FFu : risky;
    return g;
}

// the placements from hand (of any of our pieces, on any square: mask with the candidates) that do NOT defuse the
// opponent's winning move a2 of the root
__device__ __forceinline__ uint64_t greedy_undefused(const Planes &p, int me, uint32_t a2)
{
    const int opp = 1 - me;
    const uint32_t pj = (a2 * 57u) >> 9, q2 = a2 - 9u * pj, kj = pj >> 1, first = (~pj) & 1u;
    const uint32_t theirs = me ? (p.nz & ~p.neg) : (p.nz & p.neg);
    const uint32_t stands = ((theirs & (first ? p.odd : ~p.odd)) >> (9u * kj)) & 0x1FFu;  // where pj stands (0: in hand)
    const Planes d = moved(p, opp, a2);
    const uint32_t ours_d = me ? (d.nz & d.neg) : (d.nz & ~d.neg), theirs_d = me ? (d.nz & ~d.neg) : (d.nz & d.neg);
    const uint32_t t0 = theirs_d & 0x1FFu, t1 = (theirs_d >> 9) & 0x1FFu, t2 = (theirs_d >> 18) & 0x1FFu;
    const uint32_t m1 = (ours_d >> 9) & 0x1FFu, m2 = (ours_d >> 18) & 0x1FFu;
    const uint32_t o1 = m1 | t1, o2 = m2 | t2;
    const uint32_t Tt = t2 | (~o2 & (t1 | (~o1 & t0)));  // the opponent's tops after a2
    constexpr uint32_t L[8] = {0x007u, 0x038u, 0x1C0u, 0x049u, 0x092u, 0x124u, 0x111u, 0x054u};
    uint32_t every = 0x1FFu;  // squares on every line the opponent then holds
#pragma unroll
    for (int l = 0; l < 8; ++l) every &= (L[l] & ~Tt) == 0 ? L[l] : 0x1FFu;
    const uint32_t q2bit = 1u << q2, defuse = stands | (every & ~q2bit);
    uint64_t u = 0;
#pragma unroll
    for (uint32_t k = 0; k < 3; ++k) {
        const uint64_t uk = ~(defuse | (k >= kj ? q2bit : 0u)) & 0x1FFu;
        u |= uk << (18u * k) | uk << (18u * k + 9u);
    }
    return u;
}

// The candidate sets of greedy_replay_closed for all placements from hand at once (mask with the resolved candidates).
struct GreedyHandSets {
    uint64_t threat, second, block, flegal;
};

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
