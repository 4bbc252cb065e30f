// gobblet_ab.h -- EXPERIMENT BUILDS ONLY (-DGBL_AB_BUILD, scripts/build_variant.sh).  Never part of the product: the product
// build includes gobblet_knobs.h's constants instead and refuses to compile with any of the macros below set.
//
//   -DGBL_X_GREEDY_SKIP=bits      FLOOR builds (scripts/greedy_floor.sh; results are WRONG, timing only): leaves phases of the greedy
//                                 decision out -- 1 the owners' tail (merge, replay), 2 the chunk phase (the evaluations), 4 the list
//                                 phase, 8 the B phase (depth-1 walk, root, nonplain set), 16 the candidate rows of gbl_greedy's output
//   -DGBL_X_GREEDY_PAIR_CAP=n     FLOOR build (round 6; WRONG results): the depth-2 pairs of a block beyond the first n are dropped --
//                                 what a block would take with fewer pairs per board, without paying for any rule that removes them
//   -DGBL_FORCE_NT=1|3            store policy of the one-ply kernels pinned
//   -DGBL_FORCE_COLLECT_NT=0|1    gbl_collect's trajectory stores plain / streamed (0 instantiates k_collect with plain stores)
//   -DGBL_FORCE_COLLECT_PAIR=0|1  k_collect2 off / on whatever the grid
//   -DGBL_FORCE_COLLECT_SMALL=f   the role kernel's form (100 LA + 10 KO + MERGE; 0 = never, 3 = k_collect3)
//   -DGBL_AB_COLLECT_CFG          a menu of role-kernel forms in one library, picked at run time with gbl_ab_collect_cfg(cfg)
//   -DGBL_FORCE_GREEDY_SHAPE=s    the greedy kernels' block shape (11, 14, 18, 26, 28, 48, 56)
//   -DGBL_COLLECT_WAVES_PER_EU=a,b / -DGBL_CP_WAVES_PER_EU=n   occupancy pins of k_collect / k_collect_policy
#pragma once

#ifndef GBL_X_GREEDY_SKIP
#define GBL_X_GREEDY_SKIP 0
#endif
#ifndef GBL_X_GREEDY_PAIR_CAP
#define GBL_X_GREEDY_PAIR_CAP 0
#endif
#ifndef GBL_FORCE_NT
#define GBL_FORCE_NT 0
#endif
#ifndef GBL_FORCE_COLLECT_NT
#define GBL_FORCE_COLLECT_NT -1
#define GBL_KNOB_COLLECT_STREAM_OR_PLAIN(M, O, D) GBL_COLLECT_KN(M, O, D, true)
#else
#define GBL_KNOB_COLLECT_STREAM_OR_PLAIN(M, O, D)   \
    if (nt) GBL_COLLECT_KN(M, O, D, true);          \
    else GBL_COLLECT_KN(M, O, D, false)
#endif
#ifndef GBL_FORCE_COLLECT_PAIR
#define GBL_FORCE_COLLECT_PAIR -1
#endif
#ifndef GBL_FORCE_COLLECT_SMALL
#define GBL_FORCE_COLLECT_SMALL -1
#endif
#ifndef GBL_FORCE_GREEDY_SHAPE
#define GBL_FORCE_GREEDY_SHAPE 0
#endif

namespace gbl {
namespace knob {
constexpr int kGreedySkip = GBL_X_GREEDY_SKIP;
constexpr int kGreedyPairCap = GBL_X_GREEDY_PAIR_CAP;
constexpr int kForcedNt = GBL_FORCE_NT;
constexpr int kForcedCollectNt = GBL_FORCE_COLLECT_NT;
constexpr int kForcedCollectPair = GBL_FORCE_COLLECT_PAIR;
constexpr int kForcedCollectSmall = GBL_FORCE_COLLECT_SMALL;
constexpr int kForcedGreedyShape = GBL_FORCE_GREEDY_SHAPE;
#ifdef GBL_AB_COLLECT_CFG
inline int g_ab_collect_cfg = -1;  // gbl_ab_collect_cfg() picks the form at run time (one library, many forms)
inline int collect_cfg_override() { return g_ab_collect_cfg; }
#else
inline int collect_cfg_override() { return -1; }
#endif
}  // namespace knob
}  // namespace gbl

#ifdef GBL_COLLECT_WAVES_PER_EU
#define GBL_KNOB_COLLECT_WAVES_PER_EU GBL_COLLECT_WAVES_PER_EU
#else
#define GBL_KNOB_COLLECT_WAVES_PER_EU 4, 4
#endif
#ifdef GBL_CP_WAVES_PER_EU
#define GBL_KNOB_CP_WAVES_PER_EU GBL_CP_WAVES_PER_EU
#else
#define GBL_KNOB_CP_WAVES_PER_EU 4
#endif

#ifdef GBL_AB_COLLECT_CFG
#define GBL_KNOB_SMALL_FORMS                                                                                                     \
    GBL_SMALL_CFG(1, 2, false) /* (the product's form for 8 193 ... 16 384 boards until k_collect3's scalars left its player) */ \
    GBL_SMALL_CFG(1, 1, false)                                                                                                   \
    GBL_SMALL_CFG(4, 1, false)                                                                                                   \
    GBL_SMALL_CFG(1, 4, false)                                                                                                   \
    GBL_SMALL_CFG(1, 4, true)                                                                                                    \
    GBL_SMALL_CFG(1, 2, true)                                                                                                    \
    GBL_SMALL_CFG(1, 1, true)                                                                                                    \
    GBL_SMALL_CFG(2, 2, true)                                                                                                    \
    GBL_SMALL_CFG(2, 1, true)                                                                                                    \
    GBL_SMALL_CFG(4, 1, true)                                                                                                    \
    GBL_SMALL_CFG(4, 0, true)                                                                                                    \
    GBL_SMALL_CFG(2, 0, true)                                                                                                    \
    GBL_SMALL_CFG(1, 0, true)
#define GBL_KNOB_EXTRA_ENTRY_POINTS                                                                         \
    extern "C" int gbl_ab_collect_cfg(int cfg) /* (not part of the ABI: -1 = the library's own choice) */  \
    {                                                                                                       \
        gbl::knob::g_ab_collect_cfg = cfg;                                                                  \
        return 0;                                                                                           \
    }
#else
#define GBL_KNOB_SMALL_FORMS
#define GBL_KNOB_EXTRA_ENTRY_POINTS
#endif

#if GBL_FORCE_GREEDY_SHAPE == 48
#define GBL_KNOB_CP_SHAPES else if (shape == 48) GBL_CP(4, 8);
#define GBL_KNOB_GREEDY_SHAPES else if (shape == 48) GBL_GREEDY(4, 8);
#elif GBL_FORCE_GREEDY_SHAPE == 18
#define GBL_KNOB_CP_SHAPES else if (shape == 18) GBL_CP(1, 8);
#define GBL_KNOB_GREEDY_SHAPES else if (shape == 18) GBL_GREEDY(1, 8);
#else
#define GBL_KNOB_CP_SHAPES
#define GBL_KNOB_GREEDY_SHAPES
#endif
