// gobblet_knobs.h -- the ONE place where an experiment build may differ from the product.
//
// The product build (what `_native.build()` / `__graft_entry__.build()` compile and what ships) takes the constants below: its
// kernel source has no A/B switch, no forced dispatch and no build that leaves work out.  An EXPERIMENT build
// (scripts/build_variant.sh: -DGBL_AB_BUILD plus knob macros) includes csrc/gobblet_ab.h instead, which maps the -D macros of
// the measurement scripts (profiles/README.md says which script used which) onto the same names.  A product build with one of
// those macros set is a mistake and does not compile.
#pragma once
#include <stdint.h>

#ifdef GBL_AB_BUILD
#include "gobblet_ab.h"
#else

#if defined(GBL_X_GREEDY_SKIP) || defined(GBL_FORCE_NT) || defined(GBL_FORCE_COLLECT_NT) || defined(GBL_FORCE_COLLECT_PAIR) || \
    defined(GBL_FORCE_COLLECT_SMALL) || defined(GBL_AB_COLLECT_CFG) || defined(GBL_FORCE_GREEDY_SHAPE) ||                     \
    defined(GBL_COLLECT_WAVES_PER_EU) || defined(GBL_CP_WAVES_PER_EU) || defined(GBL_X_GREEDY_PAIR_CAP)
#error "experiment knobs need -DGBL_AB_BUILD (csrc/gobblet_ab.h): the product build has none"
#endif

namespace gbl {
namespace knob {
constexpr int kGreedySkip = 0;            // phases of the greedy decision left out (floor builds: WRONG results, timing only)
constexpr int kGreedyPairCap = 0;         // depth-2 pairs of a block beyond this many dropped (0: none; timing only)
constexpr int kForcedNt = 0;              // store policy of the one-ply kernels pinned (0: by batch size)
constexpr int kForcedCollectNt = -1;      // gbl_collect's trajectory stores: -1 = streamed (the product has no plain-store form)
constexpr int kForcedCollectPair = -1;    // k_collect2 forced on / off (-1: by grid size)
constexpr int kForcedCollectSmall = -1;   // the role kernel's form forced (-1: by batch size)
constexpr int kForcedGreedyShape = 0;     // the greedy kernels' block shape forced (0: by batch size)
inline int collect_cfg_override() { return -1; }  // a form picked at run time (gbl_ab_collect_cfg): none
}  // namespace knob
}  // namespace gbl

// occupancy pins (see k_collect / k_collect_policy)
#define GBL_KNOB_COLLECT_WAVES_PER_EU 4, 4
#define GBL_KNOB_CP_WAVES_PER_EU 4
// template instantiations only experiment builds carry: more forms of the role kernel, k_collect with plain stores, more block shapes
#define GBL_KNOB_SMALL_FORMS
#define GBL_KNOB_COLLECT_STREAM_OR_PLAIN(M, O, D) GBL_COLLECT_KN(M, O, D, true)
#define GBL_KNOB_CP_SHAPES
#define GBL_KNOB_GREEDY_SHAPES
#define GBL_KNOB_EXTRA_ENTRY_POINTS

#endif  // GBL_AB_BUILD
