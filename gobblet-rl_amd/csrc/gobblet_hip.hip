// gobblet_hip.hip -- gfx950 kernels + the C-ABI of include/gobblet_hip.h.
// Building blocks and the execution shape are described in gobblet_device.h.
#include "gobblet_device.h"

#include <stdio.h>
#include <algorithm>
#include <type_traits>
#include <stdlib.h>
#include <string.h>

#include "../../include/gobblet_hip.h"
#include "gobblet_diag.h"
#include "gobblet_knobs.h"  // the product's constants -- or, in an experiment build (-DGBL_AB_BUILD), the A/B knobs of gobblet_ab.h

using namespace gbl;

namespace {

thread_local char g_err[256] = "";

int fail(int code, const char *msg)
{
    snprintf(g_err, sizeof g_err, "%s", msg);
    return code;
}

int hip_fail(hipError_t e, const char *what)
{
    snprintf(g_err, sizeof g_err, "%s: %s", what, hipGetErrorString(e));
    return GBL_ERR_HIP;
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

struct Geometry {
    int64_t ntiles;
    uint32_t grid;
};

// waves: wavefronts per workgroup, each with a tile of its own (the step kernels; see kStepWaves)
inline Geometry geometry(int64_t n, int waves = 1)
{
    Geometry g;
    g.ntiles = (n + kTile - 1) / kTile;
    int64_t groups = (g.ntiles + waves - 1) / waves;
    int64_t chunk = (groups + 7) / 8;
    g.grid = (uint32_t)(chunk * 8);
    return g;
}

// blocks of nt tiles per workgroup (the greedy kernels): ntiles counts TILES, the grid blocks
inline Geometry block_geometry(int64_t n, int nt)
{
    Geometry g;
    g.ntiles = (n + kTile - 1) / kTile;
    const int64_t blocks = (g.ntiles + nt - 1) / nt;
    g.grid = (uint32_t)(((blocks + 7) / 8) * 8);
    return g;
}

// Wavefronts per workgroup of the step kernels.  The wavefronts of a workgroup are independent (a tile and an LDS
// image each, no barrier); a workgroup is only the unit of dispatch.
constexpr int kStepWaves = 1;  // (128- and 256-thread workgroups: +-1-2 %, round 1)

// Non-temporal store policy of the step kernels (see store_rows): stream the observation always, the
// mask too once one ply's footprint exceeds the 256 MiB Infinity Cache.  A pure function of the batch size: the
// library reads no environment and keeps no state (experiment builds pin a policy: gobblet_ab.h).
inline int nt_policy(int64_t n)
{
    if (knob::kForcedNt) return knob::kForcedNt == 3 ? 3 : 1;
    return n * 234 > ((int64_t)256 << 20) ? 3 : 1;
}

// gbl_collect always stores its trajectory rows non-temporally.  Plain (cached) stores win 5-12 % where the trajectory fits
// the Infinity Cache AND the batch is 131 072 - 262 144 boards, but lose 15 % at 65 536 boards and 30 % once it does not fit
// (see k_collect): the product build has no such path (experiment builds: gobblet_ab.h).

// Which kernel a gbl_collect call runs (also reported by gbl_collect_variant): GBL_COLLECT_PAIR = k_collect2 (grids of up
// to kCollect2MaxTiles tiles that stream), GBL_COLLECT_STREAM / GBL_COLLECT_CACHED = k_collect with non-temporal / plain
// stores.  A pure function of the call's shape (and of the -D macros of an A/B build).
// (round 5: the kernel needs 79 VGPRs since its ragged path stopped costing registers, and ten workgroups per CU are co-resident:
//  147 456 boards 4.08 against k_collect's 4.24 us per ply, 163 840: 4.39 / 4.53; 196 608 = 3 072 tiles: 5.67 / 5.30 -- until then
//  the limit was 2 048 tiles, eight workgroups of 122 VGPRs per CU)
constexpr int64_t kCollect2MaxTiles = 2560;

// Which form of the role kernel (k_collect_small<LA, KO, MERGE>, see there) a batch of n boards runs, as 100 LA + 10 KO + MERGE;
// 0 = none (k_collect2 / k_collect).  Measured, us per ply, FULL outputs, 32 plies per launch (scripts/experiments/ab_roles.sh, round 5):
// see the table in DESIGN.md 5.2.

// k_collect's occupancy, pinned: FOUR wavefronts per SIMD.  Until round 5 that was an accident of the register allocation (111
// VGPRs, most of them the ragged tile's byte loop's); when the ragged paths were rewritten the kernel needed 83, a fifth wavefront
// fitted, and the 2^20-board launch lost 2 % (27.19 -> 27.69 us per ply at 8 plies per launch, 26.86 -> 27.49 at 20; min 3 / max 3:
// 27.21 / 26.81; profiles/r05/collect_occupancy.txt) -- more tiles open at once is more write streams for the same DRAM pages.
constexpr int64_t kTrioHandMaxTiles = 8192;  // k_collect3<..., HAND>: the first row wavefront stores the scalars up to 524 288 boards


inline int small_cfg(int64_t n, bool with_mask, bool with_obs)
{
    (void)with_mask;
    if (knob::collect_cfg_override() >= 0) return knob::collect_cfg_override();  // (experiment builds only)
    if (knob::kForcedCollectSmall >= 0) return knob::kForcedCollectSmall;
    // (round 5, scripts/experiments/ab_roles.sh; DESIGN.md 5.2 has the table; 3 = k_collect3)
    // MASK_ONLY is bound by what ONE wavefront issues, so the player + mask-row pair of k_collect3 keeps paying far
    // into the HBM regime: 163 840 boards 1.79 against k_collect's 2.19 us per ply, 2^20: 10.39 against 10.93, 2^21: 19.9 against
    // 20.7; 2^22: 45.1 against 44.5 (profiles/r05/ab_roles_long_launches.txt)
    // (round 6: k_collect5 -- 32-board groups behind a hand-over ring, cfg 6 -- wherever its wavefronts find a SIMD each: MASK_ONLY 4 096 /
    //  8 192 boards 0.418 / 0.419 us per ply against the role kernel's 0.546 / 0.548, 16 384 ... 32 768: 0.50-0.55 against k_collect3's
    //  0.58-0.65, 49 152: 0.81 against 0.73; it needs the mask rows' wavefront: profiles/r06/ab_group32.txt)
    if (!with_obs) return n <= 32768 && with_mask ? 6 : n <= 8192 ? 210 : n <= 3 * (int64_t)(1 << 20) ? 3 : 0;
    // (with the trajectory arrays placed across HBM's memory classes: 40 960 boards k_collect3 1.31 against k_collect2's 1.40 us per
    //  ply, 49 152: 1.49 against 1.41 -- profiles/r05/placed_forms.txt)
    // (late round 5: with its scalars stored by the first row wavefront k_collect3 runs 0.634 us per ply at 8 192 ... 16 384 boards,
    //  256 plies per launch -- <1,2>: 0.643 at 12 288, 0.72 at 16 384; <2,2>: 0.558 at 8 192: profiles/r05/ab_trio_scalars.txt)
    // (round 6: FULL 4 096 / 8 192 boards k_collect5 0.440 / 0.439 us per ply against the role kernel's 0.525 / 0.527, 9 216: 0.59 against
    //  k_collect3's 0.63, 10 240: a tie -- from 257 groups on two of them share a CU)
    if (with_mask && n <= 9216) return 6;
    return n <= 8192 ? 220 : n <= 45056 ? 3 : 0;
}

inline int collect_variant(int64_t n, uint32_t plies, bool with_mask, bool with_obs)
{
    (void)plies;
    const bool nt = knob::kForcedCollectNt != 0;
    const bool pair = knob::kForcedCollectPair >= 0 ? knob::kForcedCollectPair != 0
                                                    : (n + kTile - 1) / kTile <= kCollect2MaxTiles && nt && (with_mask || with_obs);
    const int cfg = nt ? small_cfg(n, with_mask, with_obs) : 0;
    if (cfg == 3) return (with_mask || with_obs) ? GBL_COLLECT_TRIO : GBL_COLLECT_STREAM;
    if (cfg == 6) return with_mask ? GBL_COLLECT_GROUP32 : with_obs ? GBL_COLLECT_TRIO : GBL_COLLECT_STREAM;  // k_collect5 (needs the mask rows' wavefront)
    return cfg ? GBL_COLLECT_ROLES(cfg / 100, (cfg / 10) % 10, cfg % 10) : pair ? GBL_COLLECT_PAIR : nt ? GBL_COLLECT_STREAM : GBL_COLLECT_CACHED;
}

// LDS words for a tile image of ROWB-byte rows (+ slack for row_load's look-ahead dword)
template <int ROWB>
constexpr int image_words() { return kTile * ROWB / 4 + 4; }

// A wavefront's index in its workgroup, as a SCALAR: threadIdx.x >> 6 is the same on all lanes, but the compiler does not know it
// and keeps everything derived from it -- the tile index, every tile base address, the role tests -- in vector registers (in
// gbl_collect_policy: 64-bit per-lane addresses of seven output arrays, hoisted out of the ply loop and spilled to scratch).
__device__ __forceinline__ int wave_index()
{
#ifndef GBL_HOST_EMU
    return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
#else
    return (int)(threadIdx.x >> 6);
#endif
}

// -------------------------------------------------------------------------------------------
// Per-lane state of one board inside a kernel.
struct Lane {
    int64_t tile, b;
    int lane, rows;
    bool valid;
};

template <int W = 1, bool XCD_REMAP = true>
__device__ __forceinline__ bool lane_setup(Lane &L, int64_t n, int64_t ntiles)
{
    if (W == 1) {
        L.tile = XCD_REMAP ? xcd_tile(blockIdx.x, ntiles) : (int64_t)blockIdx.x;
        L.lane = threadIdx.x;
    } else {  // W consecutive tiles per workgroup, one per wavefront
        L.tile = xcd_tile(blockIdx.x, (ntiles + W - 1) / W) * W + (threadIdx.x >> 6);
        L.lane = threadIdx.x & 63;
    }
    if (L.tile >= ntiles) return false;
    int64_t left = n - L.tile * kTile;
    L.rows = left < kTile ? (int)left : kTile;
    L.valid = L.lane < L.rows;
    L.b = L.tile * kTile + L.lane;
    return true;
}

template <typename Between = NoWork>
__device__ __forceinline__ void load_state(const int8_t *state, uint32_t *img, const Lane &L, uint32_t (&r)[7],
                                           Between between = Between())
{
    tile_in<kCells>(state + L.tile * (kTile * kCells), img, L.lane, L.rows, between);
    wave_lds_fence();
    row_load<kCells>(img, L.lane, r);
    r[6] &= 0x00FFFFFFu;  // (lanes past the end of a ragged last tile hold whatever the image held: see planes_of)
}

// The board's bit planes; lanes past the end of a ragged last tile get an EMPTY board (one select on the occupancy
// plane -- the other two are only read where it is set -- instead of seven on the row's dwords).  Nothing such a
// lane computes is ever stored: ragged tiles are written back byte-granular, valid rows only.
__device__ __forceinline__ Planes planes_of(const Lane &L, const uint32_t (&r)[7])
{
    Planes p = make_planes(r);
    p.nz = L.valid ? p.nz : 0u;
    return p;
}

// A lane's 54 mask bytes into its row of a tile's LDS image.  STAGED: the row is shifted onto aligned dwords in registers
// (row_stage: ~45 v_alignbyte / select instructions and a DPP fetch of the neighbour's bytes), every LDS access an aligned dword.
// Otherwise: 14 nibble expansions and four UNALIGNED LDS stores (rows are 2-byte aligned; gfx950 takes them) -- 45 VALU
// instructions less, more LDS passes.  Measured (round 5, profiles/r05/ab_maskrow.txt): the one-wavefront kernels, which are bound
// by what ONE wavefront issues, gain -- k_collect MASK_ONLY at 2^20 boards 10.88 -> 10.06 us per ply (0.80 -> 0.87 of the HBM peak),
// FULL and the one-ply kernels +-1 % -- while k_collect2, whose storing wavefront reads the image back beside the player's writes,
// loses (MASK_ONLY 0.92 -> 1.13 us per ply at 65 536 boards, 1.39 -> 1.69 at 131 072): it and the greedy kernels stay staged.
template <bool STAGED>
__device__ __forceinline__ void mask_to_image(uint32_t *img, int lane, uint64_t bits)
{
    if constexpr (STAGED) {
        uint32_t d[14];
        mask_row(bits, d);
        row_stage<kActions>(img, lane, d);
    } else {
        mask_row_part<1>(reinterpret_cast<uint8_t *>(img) + lane * kActions, bits, 0);
    }
}

// -------------------------------------------------------------------------------------------
// Board-level kernels (one reference function each)

__global__ __launch_bounds__(64) void k_legal_mask(const int8_t *__restrict__ state, const int8_t *__restrict__ to_move,
                                                   int8_t *__restrict__ mask, int64_t n, int64_t ntiles)
{
    __shared__ uint32_t s_state[image_words<kCells>()];
    __shared__ uint32_t s_mask[image_words<kActions>()];
    Lane L;
    if (!lane_setup(L, n, ntiles)) return;
    uint32_t r[7];
    load_state(state, s_state, L, r);
    Planes p = planes_of(L, r);
    int mover = L.valid ? to_move[L.b] : 0;
    mask_to_image<false>(s_mask, L.lane, legal54(p, mover != 0));
    wave_lds_fence();
    tile_out<kActions>(mask + L.tile * (kTile * kActions), s_mask, L.lane, L.rows);
}

__global__ __launch_bounds__(64) void k_is_legal(const int8_t *__restrict__ state, const int8_t *__restrict__ agent,
                                                 const int32_t *__restrict__ actions, int8_t *__restrict__ out,
                                                 int64_t n, int64_t ntiles)
{
    __shared__ uint32_t s_state[image_words<kCells>()];
    Lane L;
    if (!lane_setup(L, n, ntiles)) return;
    uint32_t r[7];
    load_state(state, s_state, L, r);
    if (!L.valid) return;
    Planes p = planes_of(L, r);
    int a = actions[L.b];
    uint64_t m = legal54(p, agent[L.b] != 0);
    bool ok = (uint32_t)a < (uint32_t)kActions && ((m >> (a & 63)) & 1ull);
    out[L.b] = ok ? 1 : 0;
}

__global__ __launch_bounds__(64) void k_play_turn(int8_t *__restrict__ state, const int8_t *__restrict__ agent,
                                                  const int32_t *__restrict__ actions, int64_t n, int64_t ntiles)
{
    __shared__ uint32_t s_state[image_words<kCells>()];
    Lane L;
    if (!lane_setup(L, n, ntiles)) return;
    uint32_t r[7];
    load_state(state, s_state, L, r);
    Planes p = planes_of(L, r);
    int a = L.valid ? actions[L.b] : 0;
    int mover = L.valid ? (agent[L.b] != 0) : 0;
    uint64_t m = legal54(p, mover);
    bool ok = L.valid && (uint32_t)a < (uint32_t)kActions && ((m >> (a & 63)) & 1ull);
    if (ok) apply_move(p, r, mover, (uint32_t)a);
    row_stage<kCells>(s_state, L.lane, r);
    wave_lds_fence();
    tile_out<kCells>(state + L.tile * (kTile * kCells), s_state, L.lane, L.rows);
}

__global__ __launch_bounds__(64) void k_winner(const int8_t *__restrict__ state, int8_t *__restrict__ winner, int64_t n,
                                               int64_t ntiles)
{
    __shared__ uint32_t s_state[image_words<kCells>()];
    Lane L;
    if (!lane_setup(L, n, ntiles)) return;
    uint32_t r[7];
    load_state(state, s_state, L, r);
    if (!L.valid) return;
    winner[L.b] = (int8_t)winner_of(planes_of(L, r));
}

__global__ __launch_bounds__(64) void k_flatboard(const int8_t *__restrict__ state, int8_t *__restrict__ flat, int64_t n,
                                                  int64_t ntiles)
{
    __shared__ uint32_t s_state[image_words<kCells>()];
    __shared__ uint32_t s_flat[image_words<9>()];
    Lane L;
    if (!lane_setup(L, n, ntiles)) return;
    uint32_t r[7];
    load_state(state, s_state, L, r);
    uint32_t d[3];
    flat_row(planes_of(L, r), r, d);
    row_stage<9>(s_flat, L.lane, d);
    wave_lds_fence();
    tile_out<9>(flat + L.tile * (kTile * 9), s_flat, L.lane, L.rows);
}

__global__ __launch_bounds__(64) void k_covered(const int8_t *__restrict__ state, int8_t *__restrict__ cov, int64_t n,
                                                int64_t ntiles)
{
    __shared__ uint32_t s_state[image_words<kCells>()];
    Lane L;
    if (!lane_setup(L, n, ntiles)) return;
    uint32_t r[7];
    load_state(state, s_state, L, r);
    uint32_t d[7];
    covered_row(planes_of(L, r), d);
    wave_lds_fence();  // every lane has read its row before the image is reused
    row_stage<kCells>(s_state, L.lane, d);
    wave_lds_fence();
    tile_out<kCells>(cov + L.tile * (kTile * kCells), s_state, L.lane, L.rows);
}

__global__ __launch_bounds__(64) void k_observe(const int8_t *__restrict__ state, const int8_t *__restrict__ to_move,
                                                int agent_sel, int8_t *__restrict__ obs, int64_t n, int64_t ntiles)
{
    __shared__ uint32_t s_state[image_words<kCells>()];
    __shared__ uint32_t s_obs[image_words<kObs>()];
    Lane L;
    if (!lane_setup(L, n, ntiles)) return;
    uint32_t r[7];
    load_state(state, s_state, L, r);
    int who = agent_sel >= 0 ? agent_sel : (L.valid ? to_move[L.b] : 0);
    obs_image_zero(s_obs, L.lane);
    wave_lds_fence();
    obs_scatter(s_obs, L.lane, planes_of(L, r), who != 0);
    wave_lds_fence();
    tile_out<kObs, kStoreStreamDrop>(obs + L.tile * (kTile * kObs), s_obs, L.lane, L.rows);
}

// gbl_board_eval: optional Board.play_turn, then one record per board with everything the reference derives
// from the position (the seven board-level kernels above in one launch; the single-environment facade's ply).
// Every field of a record starts on a dword, records are 432 bytes apart, so a lane writes its record into
// the LDS image with plain dword stores; the observations are scattered sparsely as in k_observe.
static_assert(GBL_REC_BYTES % 16 == 0 && GBL_REC_OBS1 + kObs <= GBL_REC_BYTES, "record layout");
static_assert(GBL_REC_SQUARES % 4 == 0 && GBL_REC_WINNER % 4 == 0 && GBL_REC_FLAT % 4 == 0 && GBL_REC_COVERED % 4 == 0 &&
              GBL_REC_MASK0 % 4 == 0 && GBL_REC_MASK1 % 4 == 0 && GBL_REC_OBS0 % 4 == 0 && GBL_REC_OBS1 % 4 == 0,
              "record fields are dword aligned");

__global__ __launch_bounds__(64) void k_board_eval(int8_t *__restrict__ state, const int8_t *__restrict__ agent,
                                                   const int32_t *__restrict__ actions, int8_t *__restrict__ rec,
                                                   int64_t n, int64_t ntiles)
{
    __shared__ uint32_t s_state[image_words<kCells>()];
    __shared__ uint32_t s_rec[image_words<GBL_REC_BYTES>()];
    Lane L;
    if (!lane_setup(L, n, ntiles)) return;
    uint32_t r[7];
    load_state(state, s_state, L, r);
    Planes p = planes_of(L, r);
    if (actions) {  // board.py:118-132
        const int a = L.valid ? actions[L.b] : -1;
        const int mover = L.valid ? (agent[L.b] != 0) : 0;
        const uint64_t m = legal54(p, mover);
        if ((uint32_t)a < (uint32_t)kActions && ((m >> (a & 63)) & 1ull)) apply_move(p, r, mover, (uint32_t)a);
        wave_lds_fence();  // every lane has read its row before the image is rebuilt
        row_stage<kCells>(s_state, L.lane, r);
        wave_lds_fence();
        tile_out<kCells>(state + L.tile * (kTile * kCells), s_state, L.lane, L.rows);
    }
    {  // zero the record image (padding bytes and the observation planes' zeros)
        constexpr int NV = kTile * GBL_REC_BYTES / 16;
        static_assert(NV % 64 == 0, "whole vectors per lane");
        uint4 *lv = reinterpret_cast<uint4 *>(s_rec);
        const uint4 z = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < NV / 64; ++i) lv[L.lane + 64 * i] = z;
    }
    wave_lds_fence();
    uint32_t *row = s_rec + L.lane * (GBL_REC_BYTES / 4);
    r[6] &= 0x00FFFFFFu;
#pragma unroll
    for (int j = 0; j < 7; ++j) row[GBL_REC_SQUARES / 4 + j] = r[j];
    row[GBL_REC_WINNER / 4] = (uint32_t)winner_of(p) & 0xFFu;
    uint32_t f[3], c[7], m0[14], m1[14];
    flat_row(p, r, f);
    covered_row(p, c);
    mask_row(legal54(p, 0), m0);
    mask_row(legal54(p, 1), m1);
#pragma unroll
    for (int j = 0; j < 3; ++j) row[GBL_REC_FLAT / 4 + j] = f[j];
#pragma unroll
    for (int j = 0; j < 7; ++j) row[GBL_REC_COVERED / 4 + j] = c[j];
#pragma unroll
    for (int j = 0; j < 14; ++j) {
        row[GBL_REC_MASK0 / 4 + j] = m0[j];
        row[GBL_REC_MASK1 / 4 + j] = m1[j];
    }
    obs_scatter_row(reinterpret_cast<uint8_t *>(row) + GBL_REC_OBS0, p, 0);
    obs_scatter_row(reinterpret_cast<uint8_t *>(row) + GBL_REC_OBS1, p, 1);
    wave_lds_fence();
    // records out: 16-byte vectors, four in flight per lane (tile_out would hold all 27 in registers)
    const int nv = L.rows * (GBL_REC_BYTES / 16);
    uint4 *gv = reinterpret_cast<uint4 *>(rec + L.tile * (int64_t)(kTile * GBL_REC_BYTES));
    const uint4 *lv = reinterpret_cast<const uint4 *>(s_rec);
    for (int i0 = 0; i0 < nv; i0 += 256) {
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = lv[min(i0 + 64 * u + L.lane, nv - 1)];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i0 + 64 * u + L.lane < nv) gv[i0 + 64 * u + L.lane] = v[u];
    }
}

// Rows of a tile -> HBM through ONE LDS image that is reused.  On entry it holds the tile's state rows
// where the load put them, already patched by the plies (ImageRow): they go out as they are.  Then the
// image is free (the LDS reads of tile_out are issued before the next writes) for the 117-byte
// observation rows, and after those for the mask rows.  7.5 KB of LDS per wave instead of 12.7 KB: more
// resident waves per CU to cover the store latency.
template <bool WITH_MASK, bool WITH_OBS>
constexpr int out_image_words()
{
    return WITH_OBS ? image_words<kObs>() : WITH_MASK ? image_words<kActions>() : image_words<kCells>();
}
static_assert(image_words<kObs>() >= image_words<kActions>() && image_words<kActions>() >= image_words<kCells>(),
              "the largest row stream of a variant sizes its image");
static_assert(image_words<kObs>() % 4 == 0 && image_words<kActions>() % 4 == 0 && image_words<kCells>() % 4 == 0,
              "images of the wavefronts of a workgroup lie back to back, 16-byte aligned");

// NT: which row streams are stored with the non-temporal hint -- bit 0 observation, bit 1 mask, bit 2
// state.  Measured (profiles/r01): the write-once observation stream always gains from it; the mask
// stream gains once a ply's footprint (234 B per board) no longer fits the 256 MiB Infinity Cache
// (2^22 boards: 169 -> 141 us) and loses ~2 % below that; the state rows are re-read next ply and stay
// cached.  The host picks the variant from the batch size (nt_policy()).
// the tile's observation rows (raw_env.observe of `observer`) through the image `img`; obs_tile: the tile's first row
template <int NT>
__device__ __forceinline__ void store_obs(uint32_t *img, const Lane &L, const Planes &p, int observer,
                                          int8_t *__restrict__ obs_tile)
{
    obs_image_zero(img, L.lane);
    wave_lds_fence();
    obs_scatter(img, L.lane, p, observer);
    wave_lds_fence();
    tile_out<kObs, NT>(obs_tile, img, L.lane, L.rows);
    wave_lds_fence();
}

// the tile's mask rows from 54-bit sets; mask_tile: the tile's first row
template <int NT>
__device__ __forceinline__ void store_mask(uint32_t *img, const Lane &L, uint64_t legal, int8_t *__restrict__ mask_tile)
{
    mask_to_image<false>(img, L.lane, legal);
    wave_lds_fence();
    tile_out<kActions, NT>(mask_tile, img, L.lane, L.rows);
    wave_lds_fence();
}

// Returns the 54-bit set behind the mask rows it stored (WITH_MASK; 0 otherwise): gbl_step_ex draws the next action from it.
template <bool WITH_MASK, bool WITH_OBS, int NT>
__device__ __forceinline__ uint64_t store_rows(uint32_t *img, const Lane &L, bool mask_zero, const Planes &p, int observer,
                                               int8_t *__restrict__ state, int8_t *__restrict__ mask_out,
                                               int8_t *__restrict__ obs_out)
{
    wave_lds_fence();  // every lane's byte patches are in the image
    tile_out<kCells, (NT & 4) ? kStoreStream : kStorePlain>(state + L.tile * (kTile * kCells), img, L.lane, L.rows);
    wave_lds_fence();
    if (WITH_OBS) store_obs<(NT & 1) ? kStoreStreamDrop : kStorePlain>(img, L, p, observer, obs_out + L.tile * (kTile * kObs));
    // the next mover's legal mask is computed only now, behind the state and observation stores: the
    // sooner a wave's first stores are in flight, the shorter the launch's ramp-up
    uint64_t legal = 0ull;
    if (WITH_MASK) {
        legal = mask_zero ? 0ull : legal54(p, observer);
        store_mask<(NT & 2) ? kStoreStreamDrop : kStorePlain>(img, L, legal, mask_out + L.tile * (kTile * kActions));
    }
    return legal;
}

// (GBL_STAMP* : per-wavefront phase stamps of the diagnostic build, gobblet_diag.h; they expand to nothing here)

// gbl_step: fused raw_env.step + observe(next mover) over a tile of boards.
// (argument order: what a wavefront needs first comes first -- the first 16 dwords are preloaded into SGPRs at
// wave launch, -amdgpu-kernarg-preload-count, so the tile loads do not wait for a kernel-argument fetch)
// EXT: gbl_step_into / gbl_step_ex -- the ply's scalars also go to a trajectory slot, the status byte of every action, and the
// NEXT mover's masked-uniform draw (the gbl_sample rule on the mask this launch stores: the sampler's own launch and its 58 bytes
// per board of traffic leave an external policy's pipeline).  A template parameter: as run-time tests of three more pointers the
// plain step paid 0.5 us at 2^20 boards.
struct StepExt {
    int32_t *actions_copy;
    int8_t *done_copy, *to_move_copy, *status;
    int32_t *next_actions;       // may alias `actions`: a lane reads its board's action before it writes the next one
    uint64_t seed, env_base;
    const uint32_t *ply_dev;
    uint32_t ply;                // the draw's ply index (+ *ply_dev)
};

template <bool WITH_MASK, bool WITH_OBS, int NT, bool EXT>
__global__ __launch_bounds__(64 * kStepWaves) void k_step(int8_t *__restrict__ state, int8_t *__restrict__ to_move,
                                             int8_t *__restrict__ done, const int32_t *actions,
                                             int64_t n, int64_t ntiles, int illegal_mode, int auto_reset,
                                             int8_t *__restrict__ winner_out, int8_t *__restrict__ reward_out,
                                             int8_t *__restrict__ mask_out, int8_t *__restrict__ obs_out,
                                             int32_t *__restrict__ turn, StepExt X)
{
    constexpr int kImg = out_image_words<WITH_MASK, WITH_OBS>();
    __shared__ uint32_t s_imgs[kStepWaves * kImg];
    uint32_t *const s_img = s_imgs + (kStepWaves > 1 ? (threadIdx.x >> 6) * kImg : 0);
    Lane L;
    if (!lane_setup<kStepWaves>(L, n, ntiles)) return;
    // Per-board scalars first, from a clamped index and with no branch around them (a branch would
    // pin their s_waitcnt to the load): they are in flight together with the tile's loads -- one HBM
    // round trip per wave instead of two.
    const int64_t bs = L.valid ? L.b : n - 1;
    int mover = to_move[bs], was_done = done[bs], action = actions[bs];
    uint32_t r[7];
    uint32_t word = 0;  // EXT: the next mover's 32 random bits -- they depend on nothing the tile holds, so they are drawn under its loads
    load_state(state, s_img, L, r, [&] {
        if (EXT && X.next_actions) {
            const uint32_t ply = X.ply + (X.ply_dev ? *X.ply_dev : 0u);
            word = draw32(X.seed, X.env_base + (uint64_t)L.b, ply);
        }
    });
    mover = L.valid && mover != 0;
    was_done = L.valid && !auto_reset && was_done != 0;
    action = L.valid ? action : 0;
    Planes p = planes_of(L, r);
    Ply y;
    int dn;
    step_lane(ImageRow{reinterpret_cast<uint8_t *>(s_img) + L.lane * kCells}, p, mover, was_done, action, illegal_mode,
              auto_reset, dn, y);
    // gobblet.py:209: the mask belongs to the agent to move; a frozen board has nobody to move
    const bool frozen = dn && !auto_reset;
    const uint64_t stored = store_rows<WITH_MASK, WITH_OBS, NT>(s_img, L, frozen, p, mover, state, mask_out, obs_out);
    if (L.valid) {
        to_move[L.b] = (int8_t)mover;
        done[L.b] = (int8_t)dn;
        if (winner_out) winner_out[L.b] = (int8_t)y.winner;
        if (reward_out)
            reinterpret_cast<uint16_t *>(reward_out)[L.b] = (uint16_t)((y.r0 & 0xFF) | ((y.r1 & 0xFF) << 8));
        if (turn) turn[L.b] = next_turn(turn[L.b], y, auto_reset);
        if (EXT) {
            if (X.actions_copy) X.actions_copy[L.b] = action;
            if (X.done_copy) X.done_copy[L.b] = (int8_t)dn;
            if (X.to_move_copy) X.to_move_copy[L.b] = (int8_t)mover;
            if (X.status) X.status[L.b] = (int8_t)(was_done ? 0 : action_status(y.ok, action));  // (a frozen board consumes no action)
            if (X.next_actions) {  // gbl_sample's rule on the mask just stored (-1 where nobody is to move)
                const uint64_t legal = WITH_MASK ? stored : (frozen ? 0ull : legal54(p, mover));
                X.next_actions[L.b] = pick54(legal, word);
            }
        }
    }
}

// gbl_rollout: `plies` masked-random plies (sample + step + auto-reset) per launch; the board lives
// in registers between plies and the outputs of the LAST ply are stored.  plies = 1 is the fused
// "sample + step" ply of the benchmark pipeline.
// DEV_PLY: the ply index is ply0 + *ply_dev (gbl_rollout_at, graph replay).  A template parameter because even
// the never-taken runtime test costs the by-value kernel 0.3 us: it sits in front of the hoisted draw.
// ONE_PLY: plies == 1 known at compile time (the fused ply of a ply-by-ply pipeline): no loop, no second draw.
template <bool WITH_MASK, bool WITH_OBS, int NT, bool DEV_PLY, bool ONE_PLY>
__global__ __launch_bounds__(64 * kStepWaves) void k_rollout(int8_t *__restrict__ state, int8_t *__restrict__ to_move,
                                                int64_t n, int64_t ntiles, uint64_t seed, uint64_t env_base,
                                                const uint32_t *__restrict__ ply_dev, uint32_t ply0, uint32_t plies,
                                                int8_t *__restrict__ done, int32_t *__restrict__ actions_out,
                                                int8_t *__restrict__ winner_out, int8_t *__restrict__ reward_out,
                                                int8_t *__restrict__ mask_out, int8_t *__restrict__ obs_out,
                                                int illegal_mode, int64_t *__restrict__ counters,
                                                int32_t *__restrict__ turn)
{
    constexpr int kImg = out_image_words<WITH_MASK, WITH_OBS>();
    __shared__ uint32_t s_imgs[kStepWaves * kImg];
    uint32_t *const s_img = s_imgs + (kStepWaves > 1 ? (threadIdx.x >> 6) * kImg : 0);
    GBL_STAMP(0);
    GBL_STAMP_REAL(0);
    if (DEV_PLY) ply0 += *ply_dev;
    Lane L;
    if (!lane_setup<kStepWaves>(L, n, ntiles)) return;
    int mover = to_move[L.valid ? L.b : n - 1];  // issued before the tile loads, branch-free (see k_step)
    uint32_t r[7];
    // The first ply's random draw does not depend on the board: it is computed while the tile's loads
    // are in flight (between their issue and the LDS commit), off the wave's serial path.
    Draw4 block{{0u, 0u, 0u, 0u}};
    load_state(state, s_img, L, r, [&] { block = draw_block(seed, env_base + (uint64_t)L.b, ply0); });
    mover = L.valid && mover != 0;
    GBL_STAMP_DEP(1, r[0] + (uint32_t)mover);
    Planes p = planes_of(L, r);
    const ImageRow row{reinterpret_cast<uint8_t *>(s_img) + L.lane * kCells};  // the board's row, patched in place
    uint32_t games = 0, w1 = 0, w2 = 0;  // wave-uniform tallies (ballot + popcount)
    Ply y{0, 0, 0, false, false};
    int dn = 0, action = -1, tcount = 0;  // tcount: turn delta, or the absolute turn once a reset happened
    bool treset = false;
    if (ONE_PLY) plies = 1;
    for (uint32_t t = 0; t < plies; ++t) {
        const uint32_t ply = ply0 + t;
        action = pick54(legal54(p, mover), draw_word(block, ply));
        if (!ONE_PLY && t + 1 < plies && ((ply + 1) & 3u) == 0) block = draw_block(seed, env_base + (uint64_t)L.b, ply + 1);
        step_lane(row, p, mover, 0, action, illegal_mode, 1, dn, y);
        tcount = next_turn(tcount, y, 1);
        treset = treset || y.terminal;
        if (counters) {
            games += __popcll(__ballot(L.valid && y.terminal));
            w1 += __popcll(__ballot(L.valid && y.winner == 1));
            w2 += __popcll(__ballot(L.valid && y.winner == -1));
        }
    }
    GBL_STAMP_DEP(2, p.nz + (uint32_t)action);
    store_rows<WITH_MASK, WITH_OBS, NT>(s_img, L, false, p, mover, state, mask_out, obs_out);
    GBL_STAMP(3);
    if (L.valid) {
        to_move[L.b] = (int8_t)mover;
        done[L.b] = (int8_t)dn;
        if (actions_out) actions_out[L.b] = action;
        if (winner_out) winner_out[L.b] = (int8_t)y.winner;
        if (reward_out)
            reinterpret_cast<uint16_t *>(reward_out)[L.b] = (uint16_t)((y.r0 & 0xFF) | ((y.r1 & 0xFF) << 8));
        if (turn) turn[L.b] = treset ? tcount : turn[L.b] + tcount;
    }
    // Tallies: one stripe (its own 128-byte line) per tile mod GBL_COUNTER_STRIPES, so that the
    // device-scope atomics of concurrently finishing waves go to different lines.
    if (counters && L.lane == 0) {
        unsigned long long *c = reinterpret_cast<unsigned long long *>(counters) +
                                (size_t)(L.tile % GBL_COUNTER_STRIPES) * GBL_COUNTER_STRIDE;
        atomicAdd(c + 0, (unsigned long long)L.rows * plies);
        if (games) atomicAdd(c + 1, (unsigned long long)games);
        if (w1) atomicAdd(c + 2, (unsigned long long)w1);
        if (w2) atomicAdd(c + 3, (unsigned long long)w2);
    }
    GBL_STAMP(4);
    GBL_STAMP_DRAIN(5);
    GBL_STAMP_FLUSH(L.tile);
}

// gbl_collect: `plies` masked-random plies per launch with EVERY ply's outputs materialised -- ply t writes its
// action, winner, reward, done, next mover, legal mask and observation into slot t of trajectory arrays
// ([plies][slot_boards][...]), exactly what `plies` launches of gbl_rollout(plies = 1) with advancing output
// pointers leave behind.  The tile's state is loaded once, lives in its LDS image (patched per move) and in the
// lane's bit planes for all plies, and is stored once; a wavefront's stores of ply t drain while it computes
// ply t + 1, there is no kernel boundary, no launch ramp and no state traffic between plies, and the legal mask
// stored for the next mover is the one the next ply samples from (computed once).
// NT: the trajectory rows are stored with the non-temporal hint (a trajectory larger than the Infinity Cache is a
// write-once stream to HBM: 2^20 boards x 8 plies 34.6 vs 45.0 us per ply) or plainly (a trajectory that fits the
// 256 MiB cache stays there for whoever reads it next and is overwritten there by the next launch: 131 072 boards x 8
// plies 4.15 vs 4.76 us per ply, 262 144 x 4: 8.5 vs 9.5 -- but 65 536 x 16: 2.44 vs 2.12); the host decides by the
// footprint in A/B builds only (gpurun_out/ab5, scripts/sweep_sizes.py with -DGBL_FORCE_COLLECT_NT=0|1); the product streams.
template <bool WITH_MASK, bool WITH_OBS, bool DEV_PLY, bool NT>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(GBL_KNOB_COLLECT_WAVES_PER_EU))) void k_collect(int8_t *__restrict__ state, int8_t *__restrict__ to_move, int64_t n,
                                                int64_t ntiles, uint64_t seed, uint64_t env_base,
                                                const uint32_t *__restrict__ ply_dev, uint32_t ply0, uint32_t plies,
                                                int8_t *__restrict__ done, int64_t ply_stride, int64_t tile_stride,
                                                int32_t *__restrict__ actions_t, int8_t *__restrict__ winner_t,
                                                int8_t *__restrict__ reward_t, int8_t *__restrict__ done_t,
                                                int8_t *__restrict__ to_move_t, int8_t *__restrict__ mask_t,
                                                int8_t *__restrict__ obs_t, int illegal_mode,
                                                int64_t *__restrict__ counters, int32_t *__restrict__ turn,
                                                const int32_t *__restrict__ first_actions, int8_t *__restrict__ first_status)
{
    __shared__ uint32_t s_state[image_words<kCells>()];
    __shared__ uint32_t s_out[out_image_words<true, WITH_OBS>()];
    GBL_STAMP(0);
    GBL_STAMP_REAL(0);
    if (DEV_PLY) ply0 += *ply_dev;
    Lane L;
    // identity block -> tile map: the trajectory is written once and streams to HBM, where ONE write front per
    // array beats eight (one per XCD) -- unlike the single-ply kernels, whose outputs are rewritten every launch
    if (!lane_setup<1, false>(L, n, ntiles)) return;
    int mover = to_move[L.valid ? L.b : n - 1];
    // gbl_collect_from: the first ply plays the caller's actions (an external policy), the others are sampled
    int given = first_actions ? first_actions[L.valid ? L.b : n - 1] : 0;
    uint32_t r[7];
    Draw4 block{{0u, 0u, 0u, 0u}};
    load_state(state, s_state, L, r, [&] { block = draw_block(seed, env_base + (uint64_t)L.b, ply0); });
    mover = L.valid && mover != 0;
    Planes p = planes_of(L, r);
    const ImageRow row{reinterpret_cast<uint8_t *>(s_state) + L.lane * kCells};
    uint32_t games = 0, w1 = 0, w2 = 0;
    Ply y{0, 0, 0, false, false};
    int dn = 0, tcount = 0;
    bool treset = false;
    uint64_t legal = legal54(p, mover);
    if (first_status && L.valid) first_status[L.b] = (int8_t)action_status_of(legal, given);  // (gbl_collect_from_ex; outside the ply loop)
    for (uint32_t t = 0; t < plies; ++t) {
        const uint32_t ply = ply0 + t;
        int action = pick54(legal, draw_word(block, ply));
        if (first_actions && t == 0) action = given;
        if (t + 1 < plies && ((ply + 1) & 3u) == 0) block = draw_block(seed, env_base + (uint64_t)L.b, ply + 1);
        {  // step_lane with the mover's mask at hand (a sampled action is legal by construction: only gbl_collect_from's given ones are tested)
            if (first_actions && t == 0) y = play_ply(p, row, mover, legal, action, illegal_mode);
            else y = play_ply<true>(p, row, mover, legal, action, illegal_mode);
            dn = y.terminal ? 1 : 0;
            if (y.terminal) {  // raw_env.reset, gobblet.py:275-290
                p = Planes{0u, 0u, 0u};
                mover = 0;
                row.reset();
            }
        }
        tcount = next_turn(tcount, y, 1);
        treset = treset || y.terminal;
        if (counters) {
            games += __popcll(__ballot(L.valid && y.terminal));
            w1 += __popcll(__ballot(L.valid && y.winner == 1));
            w2 += __popcll(__ballot(L.valid && y.winner == -1));
        }
        // ply t of this tile: boards [cell, cell + 64) of the trajectory arrays
        const int64_t cell = (int64_t)t * ply_stride + L.tile * tile_stride;
        if (L.valid) {
            const int64_t at = cell + L.lane;
            if (actions_t) actions_t[at] = action;
            if (winner_t) winner_t[at] = (int8_t)y.winner;
            if (reward_t) reinterpret_cast<uint16_t *>(reward_t)[at] = (uint16_t)((y.r0 & 0xFF) | ((y.r1 & 0xFF) << 8));
            if (done_t) done_t[at] = (int8_t)dn;
            if (to_move_t) to_move_t[at] = (int8_t)mover;
        }
        constexpr int kPolicy = NT ? kStoreStreamDrop : kStorePlain;
        if (WITH_OBS) store_obs<kPolicy>(s_out, L, p, mover, obs_t + cell * kObs);
        legal = legal54(p, mover);  // the next mover's: stored now, sampled from next ply
        if (WITH_MASK) store_mask<kPolicy>(s_out, L, legal, mask_t + cell * kActions);
    }
    wave_lds_fence();  // every lane's byte patches are in the state image
    tile_out<kCells>(state + L.tile * (kTile * kCells), s_state, L.lane, L.rows);
    if (L.valid) {
        to_move[L.b] = (int8_t)mover;
        done[L.b] = (int8_t)dn;
        if (turn) turn[L.b] = treset ? tcount : turn[L.b] + tcount;
    }
    if (counters && L.lane == 0) {
        unsigned long long *c = reinterpret_cast<unsigned long long *>(counters) +
                                (size_t)(L.tile % GBL_COUNTER_STRIPES) * GBL_COUNTER_STRIDE;
        atomicAdd(c + 0, (unsigned long long)L.rows * plies);
        if (games) atomicAdd(c + 1, (unsigned long long)games);
        if (w1) atomicAdd(c + 2, (unsigned long long)w1);
        if (w2) atomicAdd(c + 3, (unsigned long long)w2);
    }
    GBL_STAMP(1); GBL_STAMP(2); GBL_STAMP(3); GBL_STAMP(4);
    GBL_STAMP_DRAIN(5);
    GBL_STAMP_FLUSH(L.tile);
}

// gbl_collect for grids that fit the chip in one batch (up to kCollect2MaxTiles tiles): TWO wavefronts per tile.  A
// wavefront whose stores are held back at issue cannot compute, and with eight or sixteen wavefronts per CU all in step
// nothing else fills the gap (the think-time model).  So the roles are dealt out: wavefront 0 plays the
// game -- pick, move, winner, auto-reset, tallies, the observation image, the next mask, its image -- and never
// issues a trajectory store; wavefront 1 takes the finished images over into registers (after which wavefront 0
// rebuilds them for the next ply; the observation image comes back zeroed) and does nothing but store.  Two barriers per ply: "images ready" and "images
// taken".  Bit for bit the trajectories of k_collect.
template <bool WITH_MASK, bool WITH_OBS, bool DEV_PLY>
__global__ __launch_bounds__(128) void k_collect2(int8_t *__restrict__ state, int8_t *__restrict__ to_move, int64_t n,
                                                 int64_t ntiles, uint64_t seed, uint64_t env_base,
                                                 const uint32_t *__restrict__ ply_dev, uint32_t ply0, uint32_t plies,
                                                 int8_t *__restrict__ done, int64_t ply_stride, int64_t tile_stride,
                                                 int32_t *__restrict__ actions_t, int8_t *__restrict__ winner_t,
                                                 int8_t *__restrict__ reward_t, int8_t *__restrict__ done_t,
                                                 int8_t *__restrict__ to_move_t, int8_t *__restrict__ mask_t,
                                                 int8_t *__restrict__ obs_t, int illegal_mode,
                                                 int64_t *__restrict__ counters, int32_t *__restrict__ turn,
                                                 const int32_t *__restrict__ first_actions, int8_t *__restrict__ first_status)
{
    __shared__ uint32_t s_state[image_words<kCells>()];
    __shared__ uint32_t s_obs[WITH_OBS ? image_words<kObs>() : 4];
    __shared__ uint32_t s_mask[WITH_MASK ? image_words<kActions>() : 4];
    __shared__ uint32_t s_small[kTile][2];  // per board: the action; winner | reward << 8 | done << 24 | to_move << 25
    if (DEV_PLY) ply0 += *ply_dev;
    // (Round 5 tried to alternate the playing wavefront between the workgroups that share a CU -- wavefront 0 in workgroups w and
    //  w + 256, wavefront 1 in w + 512 and w + 768 -- on the theory that all players of a CU sit on two of its four SIMDs: WORSE,
    //  MASK_ONLY 0.92 -> 1.16 us per ply at 65 536 boards, 1.29 -> 1.91 at 131 072; FULL +0 ... +6 %: profiles/r05/pair_flip.txt.)
    const int role = wave_index();
    Lane L;
    L.tile = (int64_t)blockIdx.x;
    L.lane = threadIdx.x & 63;
    if (L.tile >= ntiles) return;  // (the same for both wavefronts of the workgroup)
    const int64_t left = n - L.tile * kTile;
    L.rows = left < kTile ? (int)left : kTile;
    L.valid = L.lane < L.rows;
    L.b = L.tile * kTile + L.lane;
    constexpr int kPolicy = kStoreStreamDrop;
    if (role == 1) {
        // ---- the storing wavefront ---------------------------------------------------------------------------------
        for (uint32_t t = 0; t < plies; ++t) {
            const int64_t cell = (int64_t)t * ply_stride + L.tile * tile_stride;
            uint4 vo[kTile * kObs / 16 / 64 + 1], vm[kTile * kActions / 16 / 64 + 1];
            pair_barrier();  // images ready
            const uint32_t act = s_small[L.lane][0], sc = s_small[L.lane][1];
            if (WITH_OBS) tile_fetch<kObs>(s_obs, L.lane, vo);
            if (WITH_MASK) tile_fetch<kActions>(s_mask, L.lane, vm);
            if (L.rows != kTile) {  // the ragged last tile: the last few bytes of its rows straight from the images, now
                if (WITH_OBS) sub_tail(obs_t + cell * kObs, s_obs, L.lane, L.rows * kObs);
                if (WITH_MASK) sub_tail(mask_t + cell * kActions, s_mask, L.lane, L.rows * kActions);
            }
            if (WITH_OBS) {  // hand the observation image back zeroed: off the playing wavefront's path
                wave_lds_fence();
                obs_image_zero(s_obs, L.lane);
            }
            pair_barrier();  // images taken
            if (L.valid) {
                const int64_t at = cell + L.lane;
                if (actions_t) actions_t[at] = (int32_t)act;
                if (winner_t) winner_t[at] = (int8_t)(sc & 0xFFu);
                if (reward_t) reinterpret_cast<uint16_t *>(reward_t)[at] = (uint16_t)((sc >> 8) & 0xFFFFu);
                if (done_t) done_t[at] = (int8_t)((sc >> 24) & 1u);
                if (to_move_t) to_move_t[at] = (int8_t)((sc >> 25) & 1u);
            }
            // (a ragged tile: the whole vectors of its rows -- the descriptor ends there)
            if (WITH_OBS) tile_store<kObs, kPolicy>(obs_t + cell * kObs, vo, L.lane, (L.rows * kObs) & ~15);
            if (WITH_MASK) tile_store<kActions, kPolicy>(mask_t + cell * kActions, vm, L.lane, (L.rows * kActions) & ~15);
        }
        return;
    }
    // ---- the playing wavefront (k_collect's loop without its trajectory stores) -------------------------------------
    int mover = to_move[L.valid ? L.b : n - 1];
    int given = first_actions ? first_actions[L.valid ? L.b : n - 1] : 0;  // (gbl_collect_from, see k_collect)
    uint32_t r[7];
    Draw4 block{{0u, 0u, 0u, 0u}};
    load_state(state, s_state, L, r, [&] { block = draw_block(seed, env_base + (uint64_t)L.b, ply0); });
    mover = L.valid && mover != 0;
    Planes p = planes_of(L, r);
    const ImageRow row{reinterpret_cast<uint8_t *>(s_state) + L.lane * kCells};
    uint32_t games = 0, w1 = 0, w2 = 0;
    Ply y{0, 0, 0, false, false};
    int dn = 0, tcount = 0;
    bool treset = false;
    uint64_t legal = legal54(p, mover);
    if (first_status && L.valid) first_status[L.b] = (int8_t)action_status_of(legal, given);  // (gbl_collect_from_ex)
    if (WITH_OBS) {  // the first ply's observation image (later ones come back zeroed from the storing wavefront)
        obs_image_zero(s_obs, L.lane);
        wave_lds_fence();
    }
    for (uint32_t t = 0; t < plies; ++t) {
        const uint32_t ply = ply0 + t;
        int action = pick54(legal, draw_word(block, ply));
        if (first_actions && t == 0) action = given;
        if (t + 1 < plies && ((ply + 1) & 3u) == 0) block = draw_block(seed, env_base + (uint64_t)L.b, ply + 1);
        if (first_actions && t == 0) y = play_ply(p, row, mover, legal, action, illegal_mode);
        else y = play_ply<true>(p, row, mover, legal, action, illegal_mode);  // (sampled: legal by construction)
        dn = y.terminal ? 1 : 0;
        if (y.terminal) {  // raw_env.reset, gobblet.py:275-290
            p = Planes{0u, 0u, 0u};
            mover = 0;
            row.reset();
        }
        tcount = next_turn(tcount, y, 1);
        treset = treset || y.terminal;
        if (counters) {
            games += __popcll(__ballot(L.valid && y.terminal));
            w1 += __popcll(__ballot(L.valid && y.winner == 1));
            w2 += __popcll(__ballot(L.valid && y.winner == -1));
        }
        legal = legal54(p, mover);  // the next mover's: stored now, sampled from next ply
        if (t) pair_barrier();      // the images of the previous ply have been taken
        s_small[L.lane][0] = (uint32_t)action;
        s_small[L.lane][1] = ((uint32_t)y.winner & 0xFFu) | (((uint32_t)y.r0 & 0xFFu) << 8) | (((uint32_t)y.r1 & 0xFFu) << 16) |
                             ((uint32_t)dn << 24) | ((uint32_t)mover << 25);
        if (WITH_OBS) obs_scatter(s_obs, L.lane, p, mover);  // (into the zeroed image)
        if (WITH_MASK) mask_to_image<true>(s_mask, L.lane, legal);
        pair_barrier();  // images ready
    }
    pair_barrier();  // (the last ply's images taken: pairs with the storing wavefront's second barrier)
    wave_lds_fence();
    tile_out<kCells>(state + L.tile * (kTile * kCells), s_state, L.lane, L.rows);
    if (L.valid) {
        to_move[L.b] = (int8_t)mover;
        done[L.b] = (int8_t)dn;
        if (turn) turn[L.b] = treset ? tcount : turn[L.b] + tcount;
    }
    if (counters && L.lane == 0) {
        unsigned long long *c = reinterpret_cast<unsigned long long *>(counters) +
                                (size_t)(L.tile % GBL_COUNTER_STRIPES) * GBL_COUNTER_STRIDE;
        atomicAdd(c + 0, (unsigned long long)L.rows * plies);
        if (games) atomicAdd(c + 1, (unsigned long long)games);
        if (w1) atomicAdd(c + 2, (unsigned long long)w1);
        if (w2) atomicAdd(c + 3, (unsigned long long)w2);
    }
}

// gbl_collect for batches that do NOT fill the chip: ROLE wavefronts that share NOTHING, and the game played REDUNDANTLY wherever
// that saves a hand-over.  At 4 096 boards k_collect2's 64 workgroups leave three CUs in four idle and a ply lasts as long as
// its playing wavefront's serial path; a lone wavefront issues one instruction per 5-7 cycles whatever it is, so what counts is
// the number of instructions ONE wavefront executes per ply.  The game itself -- sample, move, winner, auto-reset, next legal
// mask: the CHAIN, ~250 instructions, every one depending on the ply before -- cannot be dealt out (0.72 us per ply with nothing
// stored, scripts/experiments/ab_floor.sh); everything else can.  A role wavefront loads a SUB-TILE of 64 / LPB boards (LPB lanes per board,
// all playing alike), plays every ply, and materialises one share of the outputs, 1 / LPB of a row per lane --
//   scalars: the five scalars of a ply, the tallies, and at the end the state (the only role that patches a state image);
//   mask:    the mask rows (lane j: bytes [64 j / LPB, 64 (j + 1) / LPB) of its board's row);
//   obs:     the observation rows (lane j: channels j, j + LPB, ...).
// No LDS hand-over, no cross-lane instruction; the redundant arithmetic runs on SIMDs that would idle.  A role's image of ply t
// is read back into registers at the end of its iteration and stored in the NEXT one, behind the sample and the move: the LDS
// round trip is off the chain.  Bit for bit the trajectories of k_collect (same sampler keys, same arithmetic).
//
// What a role costs a lone wavefront per ply (round 5, scripts/experiments/ab_roles.sh): the chain 0.72 us, + scalars ~0.05, + mask rows ~0.10
// whatever LPB, + observation rows 0.15 / 0.30 / 0.58 at 16 / 32 / 64 boards per wavefront: the observation role is the long one,
// and it alone gains from more lanes per board.  So a WORKGROUP is a group of 64 / LA boards with
//   one scalars wavefront (which also builds the mask rows when MERGE) and, unless MERGE, one mask wavefront, LA lanes per board,
//   KO observation wavefronts of LA * KO lanes per board, each over 1 / KO of the group,
// <LA, KO, MERGE> chosen by the batch size (collect_variant): every redundant chain is SIMD time that saturated SIMDs do not have.
// cell of sub-tile s (LPB lanes per board) at ply t: t * ply_stride + (s / LPB) * tile_stride + (s % LPB) * (64 / LPB).
// ONE barrier per launch (not per ply): the roles read the group's state, movers and first actions on their own and the scalars
// role overwrites them at the end -- the rendezvous behind the loads keeps a late role from reading what an early one wrote back
// (with one ply per launch the scalars role reaches its write-back after ~300 instructions).
constexpr int kRoleScalars = 1, kRoleMask = 2, kRoleObs = 4;  // (bits: a wavefront may hold several roles)

struct SmallArgs {
    int8_t *state, *to_move, *done;
    int64_t n;
    uint64_t seed, env_base;
    uint32_t ply0, plies;
    int64_t ply_stride, tile_stride;
    int32_t *actions_t;
    int8_t *winner_t, *reward_t, *done_t, *to_move_t, *mask_t, *obs_t;
    int illegal_mode;
    int64_t *counters;
    int32_t *turn;
    const int32_t *first_actions;
    int8_t *first_status;
};

template <int ROLE, int LPB, bool SYNC>
__device__ __forceinline__ void small_role(const SmallArgs &A, uint32_t *img, uint32_t *mask_img, uint32_t *obs_img, uint32_t *draw_buf,
                                           int64_t sub)
{
    constexpr int BPS = kTile / LPB, SH = LPB == 4 ? 2 : LPB == 2 ? 1 : 0;
    constexpr bool SC = (ROLE & kRoleScalars) != 0, MK = (ROLE & kRoleMask) != 0, OB = (ROLE & kRoleObs) != 0;
    const int lane = (int)(threadIdx.x & 63u), bq = lane >> SH, j = lane & (LPB - 1);
    const int64_t left = A.n - sub * BPS;
    if (left <= 0) {  // (an observation wavefront whose share of a ragged last group is empty: it still meets the others once)
#ifndef GBL_HOST_EMU
        if (SYNC) __syncthreads();
#endif
        return;
    }
    const int rows = left < BPS ? (int)left : BPS;
    const bool valid = bq < rows, full = rows == BPS;
    const int64_t b = sub * BPS + bq, bs = valid ? b : A.n - 1;
    int mover = A.to_move[bs];
    int given = A.first_actions ? A.first_actions[bs] : 0;  // (gbl_collect_from: the first ply plays the caller's actions)
    // The sampler's words.  One lane per board: a generator block per four plies, in registers, as k_collect.  LPB lanes per board:
    // the generator leaves the ply's serial chain -- lane j of a board generates block (base + j), the words of 4 LPB plies go
    // through LDS (16 bytes per lane), and a ply reads its word back one ply ahead: ~5 instead of ~30 instructions per ply.
    Draw4 block{{0u, 0u, 0u, 0u}};
    uint32_t base = A.ply0 >> 2;                            // the first generator block in draw_buf (wave-uniform)
    const uint32_t *const my_draws = draw_buf + (lane & ~(LPB - 1)) * 4;
    auto refill = [&](uint32_t blk) {                       // blocks blk .. blk + LPB - 1 of this wavefront's boards -> draw_buf
        const Draw4 d = draw_block(A.seed, A.env_base + (uint64_t)b, (blk + (uint32_t)j) << 2);
        reinterpret_cast<uint4 *>(draw_buf)[lane] = uint4{d.w[0], d.w[1], d.w[2], d.w[3]};
    };
    sub_in<kCells, BPS>(A.state + sub * (BPS * kCells), img, lane, rows, [&] {
        if constexpr (LPB == 1) block = draw_block(A.seed, A.env_base + (uint64_t)b, A.ply0);
        else refill(base);
    });
    wave_lds_fence();
    uint32_t r[7];
    row_load<kCells>(img, bq, r);
    r[6] &= 0x00FFFFFFu;
    uint32_t word = 0;                                      // the current ply's 32 random bits
    if constexpr (LPB > 1) word = my_draws[A.ply0 & 3u];
    if (SYNC) {
#ifndef GBL_HOST_EMU
        // every role's loads of the group have landed before the scalars role may write anything back (waits vmcnt(0) + lgkmcnt(0))
        asm volatile("" : "+v"(mover), "+v"(given));
        __syncthreads();
#endif
    }
    mover = valid && mover != 0;
    Planes p = make_planes(r);
    p.nz = valid ? p.nz : 0u;
    const ImageRow row{reinterpret_cast<uint8_t *>(img) + bq * kCells};
    uint32_t games = 0, w1 = 0, w2 = 0;
    Ply y{0, 0, 0, false, false};
    int dn = 0, tcount = 0;
    bool treset = false;
    uint64_t legal = legal54(p, mover);
    if (SC && A.first_status && valid && j == 0) A.first_status[b] = (int8_t)action_status_of(legal, given);  // (gbl_collect_from_ex)
    constexpr int kRowPolicy = kStoreStreamDrop;  // trajectory slots are written once: streamed
    // the images of the previous ply, on their way out
    SubVecs<MK ? sub_vectors<kActions, BPS>() : 0> vm{};
    SubVecs<OB ? sub_vectors<kObs, BPS>() : 0> vo{};
    int8_t *mdst = nullptr, *odst = nullptr;
    const int obytes = rows * kObs, mbytes = rows * kActions;  // what leaves of this sub-tile's images per ply
    const uint32_t plies = A.plies;
    const int64_t cell0 = (sub >> SH) * A.tile_stride + (sub & (LPB - 1)) * BPS;
    // this ply's rows of the two row streams: advanced by a ply's stride per ply (no 64-bit multiply inside the chain)
    int8_t *obs_at = OB ? A.obs_t + cell0 * kObs : nullptr, *mask_at = MK ? A.mask_t + cell0 * kActions : nullptr;
    const int64_t obs_step = A.ply_stride * kObs, mask_step = A.ply_stride * kActions;
    int64_t at = 0;  // ... and this ply's cell of the scalar arrays, relative to ply 0's (sc_* below)
    constexpr bool PAIR = LPB > 1;  // the lanes of a board split the winner test (winner_of_pair)
    // The scalars role's per-lane output arrays (NULL: this lane stores nothing there), at ply 0's cell of its board:
    //   4 lanes per board: lane 0 action + next mover, 1 reward, 2 winner, 3 done  (one byte store serves three arrays)
    //   2 lanes:           lane 0 action + next mover + winner, 1 reward + done
    //   1 lane:            everything
    int32_t *sc_act = nullptr;
    uint16_t *sc_rw = nullptr;
    int8_t *sc_b0 = nullptr, *sc_b1 = nullptr, *sc_b2 = nullptr;
    bool b0_is_mover = false, b0_is_winner = false;
    if (SC && valid) {
        const int64_t at0 = cell0 + bq;
        if (j == 0 && A.actions_t) sc_act = A.actions_t + at0;
        if (j == (LPB > 1 ? 1 : 0) && A.reward_t) sc_rw = reinterpret_cast<uint16_t *>(A.reward_t) + at0;
        if constexpr (LPB == 4) {
            b0_is_mover = j == 0;
            b0_is_winner = j == 2;
            int8_t *const arr = j == 0 ? A.to_move_t : j == 2 ? A.winner_t : j == 3 ? A.done_t : nullptr;
            if (arr) sc_b0 = arr + at0;
        } else if constexpr (LPB == 2) {
            b0_is_mover = j == 0;
            int8_t *const arr = j == 0 ? A.to_move_t : A.done_t;
            if (arr) sc_b0 = arr + at0;
            if (j == 0 && A.winner_t) sc_b1 = A.winner_t + at0;
        } else {
            if (A.winner_t) sc_b0 = A.winner_t + at0;
            if (A.done_t) sc_b1 = A.done_t + at0;
            if (A.to_move_t) sc_b2 = A.to_move_t + at0;
        }
    }
    GBL_PHASE_DECL;
    for (uint32_t t = 0; t < plies; ++t) {
        const uint32_t ply = A.ply0 + t;
        GBL_PHASE(0);  // loop overhead, the previous ply's tail
        if constexpr (LPB == 1) word = draw_word(block, ply);
        int action = pick54(legal, word);
        GBL_PHASE_DEP(1, action);  // the pick
        if (t + 1 < plies) {  // the next ply's word (nothing here depends on the game)
            if constexpr (LPB == 1) {
                if (((ply + 1) & 3u) == 0) block = draw_block(A.seed, A.env_base + (uint64_t)b, ply + 1);
            } else {
                uint32_t rel = ply + 1 - 4u * base;
                if (rel == 4u * LPB) {
                    wave_lds_fence();  // (every lane has taken this ply's word)
                    base += LPB;
                    refill(base);
                    wave_lds_fence();
                    rel = 0;
                }
                word = my_draws[rel];
            }
        }
        // (a sampled action is legal by construction: only the caller's actions of gbl_collect_from's first ply are tested)
        if (A.first_actions && t == 0) {
            action = given;
            if (SC)
                y = play_ply<false, PAIR>(p, row, mover, legal, action, A.illegal_mode, j);
            else
                y = play_ply<false, PAIR>(p, NoRow{}, mover, legal, action, A.illegal_mode, j);
        } else {
            if (SC)
                y = play_ply<true, PAIR>(p, row, mover, legal, action, A.illegal_mode, j);
            else
                y = play_ply<true, PAIR>(p, NoRow{}, mover, legal, action, A.illegal_mode, j);
        }
        GBL_PHASE_DEP(2, (uint32_t)y.winner + p.nz);  // the next word, move, winner
        dn = y.terminal ? 1 : 0;
        // the next mover's legal mask, of the moved position: it does not wait for the winner test (the two interleave on the lone
        // wavefront); a reset swaps in the empty board's
        uint64_t legal_next = legal54(p, mover);
        if (y.terminal) {  // raw_env.reset, gobblet.py:275-290
            p = Planes{0u, 0u, 0u};
            mover = 0;
            legal_next = kLegalEmpty;
            if (SC) row.reset();
        }
        GBL_PHASE_DEP(3, p.nz);  // reset
        if (t) {  // ply t - 1's rows (a ragged sub-tile's: the whole vectors of its rows)
            if constexpr (OB) sub_store<kObs, kRowPolicy, BPS>(odst, vo, lane, obytes & ~15);
            if constexpr (MK) sub_store<kActions, kRowPolicy, BPS>(mdst, vm, lane, mbytes & ~15);
        }
        if constexpr (SC) {
            tcount = next_turn(tcount, y, 1);
            treset = treset || y.terminal;
            if (A.counters) {  // (one lane per board counts)
                games += __popcll(__ballot(valid && j == 0 && y.terminal));
                w1 += __popcll(__ballot(valid && j == 0 && y.winner == 1));
                w2 += __popcll(__ballot(valid && j == 0 && y.winner == -1));
            }
            // the five scalars of a board, dealt over its lanes: per-lane array pointers fixed before the loop (sc_*), no branch
            // on the lane's index inside it
            {
                const uint16_t rw = (uint16_t)((y.r0 & 0xFF) | ((y.r1 & 0xFF) << 8));
                if (sc_act) sc_act[at] = action;
                if (sc_rw) sc_rw[at] = rw;
                if (sc_b0) sc_b0[at] = (int8_t)(LPB == 1 ? y.winner : b0_is_mover ? mover : b0_is_winner ? y.winner : dn);
                if constexpr (LPB <= 2) {
                    if (sc_b1) sc_b1[at] = (int8_t)(LPB == 1 ? dn : y.winner);
                }
                if constexpr (LPB == 1) {
                    if (sc_b2) sc_b2[at] = (int8_t)mover;
                }
                at += A.ply_stride;
            }
        }
        GBL_PHASE(4);  // the previous ply's row stores, this ply's scalars
        if constexpr (OB) {
            sub_obs_zero<BPS>(obs_img, lane);
            wave_lds_fence();
            obs_scatter_part<LPB>(reinterpret_cast<uint8_t *>(obs_img) + bq * kObs, p, mover, j);
            wave_lds_fence();
            odst = obs_at;
            obs_at += obs_step;
            sub_fetch<kObs, BPS>(obs_img, lane, vo);
            if (!full) sub_tail(odst, obs_img, lane, obytes);
            wave_lds_fence();
        }
        GBL_PHASE(5);  // the observation image
        legal = legal_next;  // stored now, sampled from next ply
        GBL_PHASE_DEP(6, (uint32_t)legal);  // the next legal mask
        if constexpr (MK) {
            mask_row_part<LPB>(reinterpret_cast<uint8_t *>(mask_img) + bq * kActions, legal, j);
            wave_lds_fence();
            mdst = mask_at;
            mask_at += mask_step;
            sub_fetch<kActions, BPS>(mask_img, lane, vm);
            if (!full) sub_tail(mdst, mask_img, lane, mbytes);
            wave_lds_fence();
        }
        GBL_PHASE(7);  // the mask image
    }
    GBL_PHASE(7);  // (the last ply's mask image; phase 7 = the mask image of every ply but the last, see below)
    GBL_PHASE_FLUSH(wave_index());
    if (plies) {  // the last ply's rows
        if constexpr (OB) sub_store<kObs, kRowPolicy, BPS>(odst, vo, lane, obytes & ~15);
        if constexpr (MK) sub_store<kActions, kRowPolicy, BPS>(mdst, vm, lane, mbytes & ~15);
    }
    if constexpr (SC) {
        wave_lds_fence();  // every board's byte patches are in the state image
        sub_out<kCells, kStorePlain, BPS>(A.state + sub * (BPS * kCells), img, lane, rows);
        if (valid && j == 0) {
            A.to_move[b] = (int8_t)mover;
            A.done[b] = (int8_t)dn;
            if (A.turn) A.turn[b] = treset ? tcount : A.turn[b] + tcount;
        }
        if (A.counters && lane == 0) {
            unsigned long long *c = reinterpret_cast<unsigned long long *>(A.counters) +
                                    (size_t)((sub >> SH) % GBL_COUNTER_STRIPES) * GBL_COUNTER_STRIDE;
            atomicAdd(c + 0, (unsigned long long)rows * plies);
            if (games) atomicAdd(c + 1, (unsigned long long)games);
            if (w1) atomicAdd(c + 2, (unsigned long long)w1);
            if (w2) atomicAdd(c + 3, (unsigned long long)w2);
        }
    }
}

// gbl_collect between the role kernel and the HBM-bound regime: k_collect3 -- ONE wavefront plays a tile's game, and hands every
// ply's position (planes, mover, next legal mask: six dwords per board, double-buffered in LDS, ONE barrier per ply) to a mask-row
// wavefront and an observation-row wavefront, which build, read back and store their rows.  Against k_collect2 (one wavefront
// plays AND builds both images, one takes them over and stores: two barriers per ply, ~360 against ~60 instructions per ply) the
// work is dealt evenly -- player ~235 (chain + scalars), mask ~80, observation ~200 -- and nothing is played twice, so it keeps
// paying where the role kernel's redundant chains run out of idle SIMDs (from ~32 768 boards, three wavefronts per tile and SIMD).
// Bit for bit the trajectories of k_collect.
// HAND: the first row wavefront stores the ply's five scalars, which ride over with the legal mask -- ~25 instructions off the playing
// wavefront, whose chain the launch waits for (MASK_ONLY 12 288 boards 0.63 -> 0.56 us per ply, 32 768: 0.67 -> 0.61, 262 144: 3.02 -> 2.75,
// 524 288: 5.6 -> 5.2; FULL 20 480 ... 45 056: -2 ... -3 %).  Not beyond 524 288 boards: there MASK_ONLY runs at the chip's write rate on
// every wavefront it can get, and the 1 KB of LDS the wider hand-over costs a workgroup is 3 % (786 432 boards 7.48 -> 7.72, 2^20:
// 9.81 -> 10.07 whoever stores the scalars; profiles/r05/ab_trio_scalars.txt).
template <bool WITH_MASK, bool WITH_OBS, bool DEV_PLY, bool HAND>
__global__ __launch_bounds__(64 * (1 + (WITH_MASK ? 1 : 0) + (WITH_OBS ? 1 : 0))) void k_collect3(
    int8_t *__restrict__ state, int8_t *__restrict__ to_move, int64_t n, int64_t ntiles, uint64_t seed, uint64_t env_base,
    const uint32_t *__restrict__ ply_dev, uint32_t ply0, uint32_t plies, int8_t *__restrict__ done, int64_t ply_stride,
    int64_t tile_stride, int32_t *__restrict__ actions_t, int8_t *__restrict__ winner_t, int8_t *__restrict__ reward_t,
    int8_t *__restrict__ done_t, int8_t *__restrict__ to_move_t, int8_t *__restrict__ mask_t, int8_t *__restrict__ obs_t,
    int illegal_mode, int64_t *__restrict__ counters, int32_t *__restrict__ turn, const int32_t *__restrict__ first_actions,
    int8_t *__restrict__ first_status)
{
    constexpr int WAVES = 1 + (WITH_MASK ? 1 : 0) + (WITH_OBS ? 1 : 0);
    __shared__ uint32_t s_state[image_words<kCells>()];
    __shared__ uint32_t s_obs[WITH_OBS ? image_words<kObs>() : 4];
    __shared__ uint32_t s_mask[WITH_MASK ? image_words<kActions>() : 4];
    __shared__ uint4 s_hand[2][kTile];      // per ply parity and board: nz, neg, odd, mover
    // ... the next mover's legal mask (x, y) and, HAND, the ply's scalars for the first row wavefront to store: z = the action played,
    // w = winner | r0 << 8 | r1 << 16 | done << 24 | next mover << 25
    using LegalSlot = typename std::conditional<HAND, uint4, uint2>::type;
    __shared__ LegalSlot s_legal[2][kTile];
    if (DEV_PLY) ply0 += *ply_dev;
    const int role = WAVES > 1 ? wave_index() : 0;
    const int lane = (int)(threadIdx.x & 63u);
    const int64_t tile = (int64_t)blockIdx.x;
    if (tile >= ntiles) return;  // (the same for every wavefront of the workgroup)
    const int64_t left = n - tile * kTile;
    const int rows = left < kTile ? (int)left : kTile;
    const bool valid = lane < rows, full = rows == kTile;
    const int64_t b = tile * kTile + lane;
    constexpr int kPolicy = kStoreStreamDrop;
    const int64_t cell0 = tile * tile_stride;
    constexpr bool hand_scalars = HAND && WAVES > 1;  // (a launch without mask and observation arrays has nobody to hand them to)
    if (role != 0) {
        // ---- a row wavefront: the mask rows (role 1 when there are any) or the observation rows -----------------------------
        const bool is_mask = WITH_MASK && role == 1;
        SubVecs<WITH_MASK ? sub_vectors<kActions, kTile>() : 0> vm{};
        SubVecs<WITH_OBS ? sub_vectors<kObs, kTile>() : 0> vo{};
        int8_t *dst = nullptr;
        // this ply's rows: advanced by a ply's stride per ply (no 64-bit multiply per ply)
        int8_t *row_at = is_mask ? mask_t + cell0 * kActions : obs_t + cell0 * kObs;
        const int64_t row_step = ply_stride * (is_mask ? kActions : kObs);
        int64_t scell = cell0 + lane;  // (role 1) this lane's cell of the scalar arrays at ply t
        for (uint32_t t = 0; t < plies; ++t) {
            pair_barrier();  // ply t's positions are in s_hand[t & 1] (and the player is free to go on with ply t + 1)
            if constexpr (hand_scalars) {
                if (role == 1) {  // the ply's five scalars leave from here
                    const uint4 sc = s_legal[t & 1u][lane];
                    if (valid) {
                        if (actions_t) actions_t[scell] = (int32_t)sc.z;
                        if (winner_t) winner_t[scell] = (int8_t)(sc.w & 0xFFu);
                        if (reward_t) reinterpret_cast<uint16_t *>(reward_t)[scell] = (uint16_t)((sc.w >> 8) & 0xFFFFu);
                        if (done_t) done_t[scell] = (int8_t)((sc.w >> 24) & 1u);
                        if (to_move_t) to_move_t[scell] = (int8_t)((sc.w >> 25) & 1u);
                    }
                    scell += ply_stride;
                }
            }
            if (t) {  // ply t - 1's rows, read back at the end of the last iteration (a ragged tile's: the whole vectors of its rows)
                if constexpr (WITH_MASK) {
                    if (is_mask) sub_store<kActions, kPolicy, kTile>(dst, vm, lane, (rows * kActions) & ~15);
                }
                if constexpr (WITH_OBS) {
                    if (!is_mask) sub_store<kObs, kPolicy, kTile>(dst, vo, lane, (rows * kObs) & ~15);
                }
            }
            if constexpr (WITH_MASK) {
                if (is_mask) {
                    const LegalSlot lg = s_legal[t & 1u][lane];
                    mask_row_part<1>(reinterpret_cast<uint8_t *>(s_mask) + lane * kActions, ((uint64_t)lg.y << 32) | lg.x, 0);
                    wave_lds_fence();
                    dst = row_at;
                    sub_fetch<kActions, kTile>(s_mask, lane, vm);
                    if (!full) sub_tail(dst, s_mask, lane, rows * kActions);
                    wave_lds_fence();
                }
            }
            if constexpr (WITH_OBS) {
                if (!is_mask) {
                    const uint4 h = s_hand[t & 1u][lane];
                    sub_obs_zero<kTile>(s_obs, lane);
                    wave_lds_fence();
                    obs_scatter_row(reinterpret_cast<uint8_t *>(s_obs) + lane * kObs, Planes{h.x, h.y, h.z}, (int)h.w);
                    wave_lds_fence();
                    dst = row_at;
                    sub_fetch<kObs, kTile>(s_obs, lane, vo);
                    if (!full) sub_tail(dst, s_obs, lane, rows * kObs);
                    wave_lds_fence();
                }
            }
            row_at += row_step;
        }
        if (plies) {  // the last ply's rows
            if constexpr (WITH_MASK) {
                if (is_mask) sub_store<kActions, kPolicy, kTile>(dst, vm, lane, (rows * kActions) & ~15);
            }
            if constexpr (WITH_OBS) {
                if (!is_mask) sub_store<kObs, kPolicy, kTile>(dst, vo, lane, (rows * kObs) & ~15);
            }
        }
        return;
    }
    // ---- the playing wavefront: k_collect's loop with the scalars, without the rows -----------------------------------------
    // (s_setprio 3 on the player, so that it wins the issue arbitration against the row wavefronts of its SIMD: no effect at any size,
    //  profiles/r05/player_prio.txt)
    Lane L;
    L.tile = tile; L.lane = lane; L.rows = rows; L.valid = valid; L.b = b;
    int mover = to_move[valid ? b : n - 1];
    int given = first_actions ? first_actions[valid ? b : n - 1] : 0;  // (gbl_collect_from, see k_collect)
    uint32_t r[7];
    Draw4 block{{0u, 0u, 0u, 0u}};
    load_state(state, s_state, L, r, [&] { block = draw_block(seed, env_base + (uint64_t)b, ply0); });
    mover = valid && mover != 0;
    Planes p = planes_of(L, r);
    const ImageRow row{reinterpret_cast<uint8_t *>(s_state) + lane * kCells};
    uint32_t games = 0, w1 = 0, w2 = 0;
    Ply y{0, 0, 0, false, false};
    int dn = 0, tcount = 0;
    bool treset = false;
    uint64_t legal = legal54(p, mover);
    if (first_status && valid) first_status[b] = (int8_t)action_status_of(legal, given);  // (gbl_collect_from_ex)
    int64_t cell = cell0;  // ply t's cell of the tile in the scalar arrays
    for (uint32_t t = 0; t < plies; ++t) {
        const uint32_t ply = ply0 + t;
        int action = pick54(legal, draw_word(block, ply));
        if (t + 1 < plies && ((ply + 1) & 3u) == 0) block = draw_block(seed, env_base + (uint64_t)b, ply + 1);
        if (first_actions && t == 0) {
            action = given;
            y = play_ply(p, row, mover, legal, action, illegal_mode);
        } else {
            y = play_ply<true>(p, row, mover, legal, action, illegal_mode);  // (sampled: legal by construction)
        }
        dn = y.terminal ? 1 : 0;
        legal = legal54(p, mover);  // the next mover's (beside the winner test, see small_role): stored now, sampled from next ply
        if (y.terminal) {  // raw_env.reset, gobblet.py:275-290
            p = Planes{0u, 0u, 0u};
            mover = 0;
            legal = kLegalEmpty;
            row.reset();
        }
        s_hand[t & 1u][lane] = uint4{p.nz, p.neg, p.odd, (uint32_t)mover};
        if constexpr (HAND)
            s_legal[t & 1u][lane] = uint4{(uint32_t)legal, (uint32_t)(legal >> 32), (uint32_t)action,
                                          ((uint32_t)y.winner & 0xFFu) | (((uint32_t)y.r0 & 0xFFu) << 8) | (((uint32_t)y.r1 & 0xFFu) << 16) |
                                              ((uint32_t)dn << 24) | ((uint32_t)mover << 25)};
        else
            s_legal[t & 1u][lane] = uint2{(uint32_t)legal, (uint32_t)(legal >> 32)};
        if (WAVES > 1) pair_barrier();  // ply t handed over
        tcount = next_turn(tcount, y, 1);
        treset = treset || y.terminal;
        if (counters) {
            games += __popcll(__ballot(valid && y.terminal));
            w1 += __popcll(__ballot(valid && y.winner == 1));
            w2 += __popcll(__ballot(valid && y.winner == -1));
        }
        if (!hand_scalars && valid) {
            const int64_t at = cell + lane;
            if (actions_t) actions_t[at] = action;
            if (winner_t) winner_t[at] = (int8_t)y.winner;
            if (reward_t) reinterpret_cast<uint16_t *>(reward_t)[at] = (uint16_t)((y.r0 & 0xFF) | ((y.r1 & 0xFF) << 8));
            if (done_t) done_t[at] = (int8_t)dn;
            if (to_move_t) to_move_t[at] = (int8_t)mover;
        }
        cell += ply_stride;
    }
    wave_lds_fence();
    tile_out<kCells>(state + tile * (kTile * kCells), s_state, lane, rows);
    if (valid) {
        to_move[b] = (int8_t)mover;
        done[b] = (int8_t)dn;
        if (turn) turn[b] = treset ? tcount : turn[b] + tcount;
    }
    if (counters && lane == 0) {
        unsigned long long *c = reinterpret_cast<unsigned long long *>(counters) +
                                (size_t)(tile % GBL_COUNTER_STRIPES) * GBL_COUNTER_STRIDE;
        atomicAdd(c + 0, (unsigned long long)rows * plies);
        if (games) atomicAdd(c + 1, (unsigned long long)games);
        if (w1) atomicAdd(c + 2, (unsigned long long)w1);
        if (w2) atomicAdd(c + 3, (unsigned long long)w2);
    }
}

// wavefronts of a workgroup of the role kernel: the scalars wavefront, the mask wavefront unless merged, KO observation wavefronts
// (KO = 0, MERGE: ONE wavefront per group holds all three roles -- A/B builds only, see scripts/experiments/ab_oneply.py)
template <bool WITH_MASK, bool WITH_OBS, int KO, bool MERGE>
constexpr int small_waves() { return 1 + ((WITH_MASK && !MERGE) ? 1 : 0) + (WITH_OBS ? KO : 0); }

// (The one-ply entry points -- gbl_rollout with plies = 1, gbl_step -- were routed here too and gained nothing: 3.34-3.47 us per
// launch against k_rollout's 3.29-3.36 at 1 024 - 4 096 boards, scripts/experiments/ab_ply.sh: a one-ply launch is its tile load, the chain
// and the launch boundary, none of which a smaller tile shortens.  They stay on k_rollout / k_step.)
template <bool WITH_MASK, bool WITH_OBS, bool DEV_PLY, int LA, int KO, bool MERGE>
__global__ __launch_bounds__((64 * small_waves<WITH_MASK, WITH_OBS, KO, MERGE>())) void k_collect_small(
    int8_t *__restrict__ state, int8_t *__restrict__ to_move, int64_t n, int64_t ngroups, uint64_t seed, uint64_t env_base,
    const uint32_t *__restrict__ ply_dev, uint32_t ply0, uint32_t plies, int8_t *__restrict__ done, int64_t ply_stride,
    int64_t tile_stride, int32_t *__restrict__ actions_t, int8_t *__restrict__ winner_t, int8_t *__restrict__ reward_t,
    int8_t *__restrict__ done_t, int8_t *__restrict__ to_move_t, int8_t *__restrict__ mask_t, int8_t *__restrict__ obs_t,
    int illegal_mode, int64_t *__restrict__ counters, int32_t *__restrict__ turn, const int32_t *__restrict__ first_actions,
    int8_t *__restrict__ first_status)
{
    static_assert(LA * KO <= 4 && (KO > 0 || MERGE), "at most four lanes per board");
    constexpr int WAVES = small_waves<WITH_MASK, WITH_OBS, KO, MERGE>(), NA = WAVES - (WITH_OBS ? KO : 0);
    constexpr int GB = kTile / LA, LO = LA * (KO ? KO : 1), OBB = kTile / LO;  // boards per group; observation wavefronts: LO lanes per board, OBB boards
    constexpr int kStateWords = GB * kCells / 4 + 4, kObsWords = OBB * kObs / 4 + 4, kMaskWords = GB * kActions / 4 + 4;
    __shared__ uint32_t s_state[WAVES][kStateWords];
    __shared__ uint32_t s_obs[WITH_OBS && KO ? KO : 1][WITH_OBS ? kObsWords : 4];
    __shared__ uint32_t s_mask[WITH_MASK ? kMaskWords : 4];
    __shared__ uint4 s_draw[LO > 1 ? WAVES : 1][LO > 1 ? 64 : 1];  // the sampler's words of 4 LPB plies, 16 bytes per lane
    if (DEV_PLY) ply0 += *ply_dev;
    const int64_t group = (int64_t)blockIdx.x;
    if (group >= ngroups) return;
    const int wave = WAVES > 1 ? wave_index() : 0;
    const SmallArgs A{state, to_move, done, n, seed, env_base, ply0, plies, ply_stride, tile_stride, actions_t, winner_t, reward_t,
                      done_t, to_move_t, mask_t, obs_t, illegal_mode, counters, turn, first_actions, first_status};
    constexpr bool SYNC = WAVES > 1;
    if (wave == 0) {
        small_role<kRoleScalars | ((WITH_MASK && MERGE) ? kRoleMask : 0) | ((WITH_OBS && KO == 0) ? kRoleObs : 0), LA, SYNC>(
            A, s_state[0], s_mask, s_obs[0], reinterpret_cast<uint32_t *>(s_draw[0]), group);
        return;
    }
    if constexpr (WITH_MASK && !MERGE) {
        if (wave == 1) {
            small_role<kRoleMask, LA, SYNC>(A, s_state[1], s_mask, nullptr, reinterpret_cast<uint32_t *>(s_draw[LO > 1 ? 1 : 0]), group);
            return;
        }
    }
    if constexpr (WITH_OBS && KO > 0)
        small_role<kRoleObs, LO, SYNC>(A, s_state[wave], nullptr, s_obs[wave - NA], reinterpret_cast<uint32_t *>(s_draw[LO > 1 ? wave : 0]),
                                       group * KO + (wave - NA));
}

// gbl_collect for batches that do not fill the chip, round 6: the role kernel's GEOMETRY -- groups of 32 boards, four wavefronts, each
// alone on a SIMD -- with k_collect3's HAND-OVER instead of redundant chains.  In the role kernel every role wavefront replays the
// whole game (150 instructions per ply) before it builds its share of the rows; here ONE wavefront plays a group -- two lanes per
// board, the winner test split over the pair -- and does nothing else: it leaves every ply's position, legal mask, action and results
// in an LDS ring of 2 x 4 plies and meets the row wavefronts once per four plies.  They take it from there:
//   wave 1   the mask rows (two lanes per board), the ply's five scalars dealt over the pair, the tallies and turn counters, and the
//            sampler's generator (one Philox block per board and group of four plies into a four-block ring, three ahead);
//   wave 2,3 the observation rows of 16 boards each, four lanes per board.
// The boards live in the player's planes for the whole launch; the 27-byte rows are rebuilt from them once, at the end
// (planes_to_row: bytes outside the state contract are not preserved).  Bit for bit the trajectories of k_collect.
constexpr int kGroupBoards = 32, kGroupRing = 4;

template <bool WITH_OBS, bool DEV_PLY>
__global__ __launch_bounds__(64 * (WITH_OBS ? 4 : 2)) void k_collect5(
    int8_t *__restrict__ state, int8_t *__restrict__ to_move, int64_t n, int64_t ngroups, uint64_t seed, uint64_t env_base,
    const uint32_t *__restrict__ ply_dev, uint32_t ply0, uint32_t plies, int8_t *__restrict__ done, int64_t ply_stride,
    int64_t tile_stride, int32_t *__restrict__ actions_t, int8_t *__restrict__ winner_t, int8_t *__restrict__ reward_t,
    int8_t *__restrict__ done_t, int8_t *__restrict__ to_move_t, int8_t *__restrict__ mask_t, int8_t *__restrict__ obs_t,
    int illegal_mode, int64_t *__restrict__ counters, int32_t *__restrict__ turn, const int32_t *__restrict__ first_actions,
    int8_t *__restrict__ first_status)
{
    constexpr int GB = kGroupBoards, OBB = 16;
    static_assert(kGroupRing == 4, "slot = t & 7, one generator block per group of plies");
    __shared__ uint32_t s_state[GB * kCells / 4 + 4];
    __shared__ uint32_t s_mask[GB * kActions / 4 + 4];
    __shared__ uint32_t s_obs[WITH_OBS ? 2 : 1][WITH_OBS ? OBB * kObs / 4 + 4 : 4];
    __shared__ uint4 s_hand[2 * kGroupRing][GB];   // per ply slot and board: nz, neg, odd, next mover
    __shared__ uint4 s_legal[2 * kGroupRing][GB];  // ... the next mover's legal mask (x, y), the action played (z), and in w: winner | r0 << 8 |
                                                   //     r1 << 16 | done << 24 | next mover << 25 | stepped << 26
    __shared__ uint4 s_draw[4][GB];                // the sampler's words: block m of board b (plies 4 m ... 4 m + 3) at [m & 3][b]
    if (DEV_PLY) ply0 += *ply_dev;
    const int role = wave_index();
    const int lane = (int)(threadIdx.x & 63u);
    const int64_t group = (int64_t)blockIdx.x;
    if (group >= ngroups) return;  // (the same for every wavefront of the workgroup)
    const int64_t left = n - group * GB;
    const int rows = left < GB ? (int)left : GB;
    constexpr int kPolicy = kStoreStreamDrop;
    const int64_t cell0 = (group >> 1) * tile_stride + (group & 1) * GB;  // the group's first cell of ply 0 (a tile is two groups)
    const uint32_t m0 = ply0 >> 2;
    const uint32_t groups_of_plies = (plies + kGroupRing - 1) / kGroupRing;
    if (role == 1) {
        // ---- the mask rows, the scalars, the tallies, the turn counters, the generator: two lanes per board ----------------------
        const int bq = lane >> 1, j = lane & 1;
        const bool valid = bq < rows;
        const int64_t b = group * GB + bq;
        SubVecs<sub_vectors<kActions, GB>()> vm{};
        int8_t *dst = nullptr, *row_at = mask_t + cell0 * kActions;
        const int64_t row_step = ply_stride * kActions;
        const int mbytes = rows * kActions;
        // the five scalars of a board dealt over its pair (as the role kernel's scalars role): lane 0 action + next mover + winner,
        // lane 1 reward + done; per-lane array pointers fixed here, no branch on the lane's index inside the loop
        int32_t *sc_act = nullptr;
        uint16_t *sc_rw = nullptr;
        int8_t *sc_b0 = nullptr, *sc_b1 = nullptr;
        if (valid) {
            const int64_t at0 = cell0 + bq;
            if (j == 0 && actions_t) sc_act = actions_t + at0;
            if (j == 1 && reward_t) sc_rw = reinterpret_cast<uint16_t *>(reward_t) + at0;
            int8_t *const arr = j == 0 ? to_move_t : done_t;
            if (arr) sc_b0 = arr + at0;
            if (j == 0 && winner_t) sc_b1 = winner_t + at0;
        }
        int64_t at = 0;
        uint32_t games = 0, w1 = 0, w2 = 0;
        int tcount = 0;
        bool treset = false;
        auto generate = [&](uint32_t m) {
            const Draw4 d = draw_block(seed, env_base + (uint64_t)b, m << 2);
            s_draw[m & 3u][bq] = uint4{d.w[0], d.w[1], d.w[2], d.w[3]};  // (both lanes of a board: the same words)
        };
        generate(m0);
        generate(m0 + 1u);
        generate(m0 + 2u);
        pair_barrier();  // setup: the first draws are in LDS
        for (uint32_t g = 0; g < groups_of_plies; ++g) {
            pair_barrier();  // plies 4 g ... 4 g + 3 are in slots (g & 1) * 4 ... + 3 (and the player is free to play the next four)
            // the generator, three blocks ahead of this wavefront's own consumption: block m0 + g + 3 replaces block m0 + g - 1, which
            // the player (now in plies 4 g + 4 ...: blocks m0 + g + 1, m0 + g + 2) has left behind
            generate(m0 + g + 3u);
            const uint32_t t_end = (g + 1u) * kGroupRing < plies ? (g + 1u) * kGroupRing : plies;
#pragma nounroll
            for (uint32_t t = g * kGroupRing; t < t_end; ++t) {
                const uint4 sc = s_legal[t & (2u * kGroupRing - 1u)][bq];
                {
                    const uint32_t winner = sc.w & 0xFFu, dn = (sc.w >> 24) & 1u, mv = (sc.w >> 25) & 1u;
                    if (sc_act) sc_act[at] = (int32_t)sc.z;
                    if (sc_rw) sc_rw[at] = (uint16_t)((sc.w >> 8) & 0xFFFFu);
                    if (sc_b0) sc_b0[at] = (int8_t)(j == 0 ? mv : dn);
                    if (sc_b1) sc_b1[at] = (int8_t)winner;
                    at += ply_stride;
                    const bool terminal = dn != 0;
                    tcount = terminal ? 0 : tcount + (int)((sc.w >> 26) & 1u);  // raw_env.turn: + 1 per step, 0 after a reset (next_turn)
                    treset = treset || terminal;
                    if (counters) {  // (one lane per board counts)
                        const int w = (int)(int8_t)winner;
                        games += __popcll(__ballot(valid && j == 0 && terminal));
                        w1 += __popcll(__ballot(valid && j == 0 && w == 1));
                        w2 += __popcll(__ballot(valid && j == 0 && w == -1));
                    }
                }
                if (t) sub_store<kActions, kPolicy, GB>(dst, vm, lane, mbytes & ~15);  // ply t - 1's rows (a ragged group's: the whole vectors)
                mask_row_part<2>(reinterpret_cast<uint8_t *>(s_mask) + bq * kActions, ((uint64_t)sc.y << 32) | sc.x, j);
                wave_lds_fence();
                dst = row_at;
                row_at += row_step;
                sub_fetch<kActions, GB>(s_mask, lane, vm);
                if (rows != GB) sub_tail(dst, s_mask, lane, mbytes);
                wave_lds_fence();
            }
        }
        if (plies) sub_store<kActions, kPolicy, GB>(dst, vm, lane, mbytes & ~15);
        if (valid && j == 0 && turn) turn[b] = treset ? tcount : turn[b] + tcount;
        if (counters && lane == 0) {
            unsigned long long *c = reinterpret_cast<unsigned long long *>(counters) +
                                    (size_t)((group >> 1) % GBL_COUNTER_STRIPES) * GBL_COUNTER_STRIDE;
            atomicAdd(c + 0, (unsigned long long)rows * plies);
            if (games) atomicAdd(c + 1, (unsigned long long)games);
            if (w1) atomicAdd(c + 2, (unsigned long long)w1);
            if (w2) atomicAdd(c + 3, (unsigned long long)w2);
        }
        return;
    }
    if (role >= 2) {
        if constexpr (WITH_OBS) {
            // ---- the observation rows of 16 boards: four lanes per board -------------------------------------------------------
            const int ow = role - 2, oq = lane >> 2, oj = lane & 3;
            const int orows_all = rows - ow * OBB, orows = orows_all < 0 ? 0 : orows_all > OBB ? OBB : orows_all;
            const int obytes = orows * kObs;
            uint32_t *const img = s_obs[ow];
            SubVecs<sub_vectors<kObs, OBB>()> vo{};
            int8_t *dst = nullptr, *row_at = obs_t + (cell0 + ow * OBB) * kObs;
            const int64_t row_step = ply_stride * kObs;
            pair_barrier();  // setup
            for (uint32_t g = 0; g < groups_of_plies; ++g) {
                pair_barrier();
                const uint32_t t_end = (g + 1u) * kGroupRing < plies ? (g + 1u) * kGroupRing : plies;
#pragma nounroll
                for (uint32_t t = g * kGroupRing; t < t_end; ++t) {
                    const uint4 h = s_hand[t & (2u * kGroupRing - 1u)][ow * OBB + oq];
                    if (t) sub_store<kObs, kPolicy, OBB>(dst, vo, lane, obytes & ~15);
                    sub_obs_zero<OBB>(img, lane);
                    wave_lds_fence();
                    obs_scatter_part<4>(reinterpret_cast<uint8_t *>(img) + oq * kObs, Planes{h.x, h.y, h.z}, (int)h.w, oj);
                    wave_lds_fence();
                    dst = row_at;
                    row_at += row_step;
                    sub_fetch<kObs, OBB>(img, lane, vo);
                    if (orows != OBB) sub_tail(dst, img, lane, obytes);
                    wave_lds_fence();
                }
            }
            if (plies) sub_store<kObs, kPolicy, OBB>(dst, vo, lane, obytes & ~15);
        }
        return;
    }
    // ---- the playing wavefront: the game and the hand-over, nothing else; two lanes per board -----------------------------------
    const int bq = lane >> 1, j = lane & 1;
    const bool valid = bq < rows;
    const int64_t b = group * GB + bq, bs = valid ? b : n - 1;
    int mover = to_move[bs];
    int given = first_actions ? first_actions[bs] : 0;  // (gbl_collect_from, see k_collect)
    sub_in<kCells, GB>(state + group * (GB * kCells), s_state, lane, rows);
    wave_lds_fence();
    uint32_t r[7];
    row_load<kCells>(s_state, bq, r);
    r[6] &= 0x00FFFFFFu;
    mover = valid && mover != 0;
    Planes p = make_planes(r);
    p.nz = valid ? p.nz : 0u;
    int dn = 0;
    uint64_t legal = legal54(p, mover);
    if (first_status && valid && j == 0) first_status[b] = (int8_t)action_status_of(legal, given);  // (gbl_collect_from_ex)
    pair_barrier();  // setup: the first draws are in LDS
    const uint32_t *const my_draws = reinterpret_cast<const uint32_t *>(&s_draw[0][bq]);  // word w of block m: [(m & 3) * 4 GB + w]
    uint32_t word = my_draws[(m0 & 3u) * (GB * 4) + (ply0 & 3u)];
    for (uint32_t t = 0; t < plies; ++t) {
        int action = pick54(legal, word);
        {  // the next ply's word (its block was generated at least a group of plies ago): the LDS round trip hides behind this ply
            const uint32_t nx = ply0 + t + 1u;
            word = my_draws[((nx >> 2) & 3u) * (GB * 4) + (nx & 3u)];
        }
        Ply y;
        if (first_actions && t == 0) {
            action = given;
            y = play_ply<false, true>(p, NoRow{}, mover, legal, action, illegal_mode, j);
        } else {
            y = play_ply<true, true>(p, NoRow{}, mover, legal, action, illegal_mode, j);  // (sampled: legal by construction)
        }
        dn = y.terminal ? 1 : 0;
        legal = legal54(p, mover);  // the next mover's (beside the winner test, see small_role): stored now, sampled from next ply
        if (y.terminal) {  // raw_env.reset, gobblet.py:275-290
            p = Planes{0u, 0u, 0u};
            mover = 0;
            legal = kLegalEmpty;
        }
        const uint32_t slot = t & (2u * kGroupRing - 1u);
        s_hand[slot][bq] = uint4{p.nz, p.neg, p.odd, (uint32_t)mover};  // (both lanes of a board: the same values)
        s_legal[slot][bq] = uint4{(uint32_t)legal, (uint32_t)(legal >> 32), (uint32_t)action,
                                  ((uint32_t)y.winner & 0xFFu) | (((uint32_t)y.r0 & 0xFFu) << 8) | (((uint32_t)y.r1 & 0xFFu) << 16) |
                                      ((uint32_t)dn << 24) | ((uint32_t)mover << 25) | ((y.stepped ? 1u : 0u) << 26)};
        if ((t & (kGroupRing - 1u)) == kGroupRing - 1u || t + 1u == plies) pair_barrier();  // four plies handed over
    }
    // the boards as they stand, rebuilt from the planes; lane 0 of every pair leaves its row in the image (27-byte rows: unaligned
    // LDS stores, as ImageRow::reset), the image goes out as the load brought it in
    planes_to_row(p, r);
    wave_lds_fence();
    if (j == 0) __builtin_memcpy(reinterpret_cast<uint8_t *>(s_state) + bq * kCells, r, kCells);
    wave_lds_fence();
    sub_out<kCells, kStorePlain, GB>(state + group * (GB * kCells), s_state, lane, rows);
    if (valid && j == 0) {
        to_move[b] = (int8_t)mover;
        done[b] = (int8_t)dn;
    }
}

// gbl_placement_probe: the write pattern of k_collect without the game -- tile i of `plies` slots stores 64 rows of
// 117 bytes into a and 64 rows of 54 bytes into b (zeros, `nt sc1` like the trajectory stream), one wavefront per
// tile, identity tile map.  Timed with both streams, with a alone and with b alone (see the header).
template <bool WITH_A, bool WITH_B>
__global__ __launch_bounds__(64) void k_probe(int8_t *__restrict__ a, int8_t *__restrict__ b, int64_t ntiles, int plies)
{
    __shared__ uint32_t s_img[image_words<kObs>()];
    const int lane = threadIdx.x;
    const int64_t tile = blockIdx.x;
    if (tile >= ntiles) return;
    obs_image_zero(s_img, lane);
    wave_lds_fence();
    for (int t = 0; t < plies; ++t) {
        const int64_t cell = ((int64_t)t * ntiles + tile) * kTile;
        if (WITH_A) tile_out<kObs, kStoreStreamDrop>(a + cell * kObs, s_img, lane, kTile);
        if (WITH_B) tile_out<kActions, kStoreStreamDrop>(b + cell * kActions, s_img, lane, kTile);
    }
}

// gbl_counter_add: the device-resident ply / call counter of the *_at entry points
__global__ void k_counter_add(uint32_t *ctr, uint32_t by) { *ctr += by; }

// gbl_sample: mask rows -> one action per board
__global__ __launch_bounds__(64) void k_sample(const int8_t *__restrict__ mask, int32_t *__restrict__ actions, int64_t n,
                                               int64_t ntiles, uint64_t seed, uint64_t env_base, uint32_t ply,
                                               const uint32_t *__restrict__ ply_dev)
{
    if (ply_dev) ply += *ply_dev;
    __shared__ uint32_t s_mask[image_words<kActions>()];
    Lane L;
    if (!lane_setup(L, n, ntiles)) return;
    uint32_t draw = 0;  // board-independent: computed while the tile is in flight
    tile_in<kActions>(mask + L.tile * (kTile * kActions), s_mask, L.lane, L.rows,
                      [&] { draw = draw32(seed, env_base + (uint64_t)L.b, ply); });
    wave_lds_fence();
    uint32_t d[14];
    row_load<kActions>(s_mask, L.lane, d);
    if (!L.valid) return;
    actions[L.b] = pick54(mask_bits(d), draw);
}

// gbl_decode_obs: greedy_policy.py:43-71
__global__ __launch_bounds__(64) void k_decode_obs(const int8_t *__restrict__ obs, int8_t *__restrict__ state,
                                                   int8_t *__restrict__ to_move, int64_t n, int64_t ntiles)
{
    __shared__ uint32_t s_obs[image_words<kObs>()];
    __shared__ uint32_t s_state[image_words<kCells>()];
    Lane L;
    if (!lane_setup(L, n, ntiles)) return;
    tile_in<kObs>(obs + L.tile * (kTile * kObs), s_obs, L.lane, L.rows);
    wave_lds_fence();
    uint32_t d[30];
    row_load<kObs>(s_obs, L.lane, d);
    uint32_t r[7];
    int agent = decode_obs_row(d, r);
    row_stage<kCells>(s_state, L.lane, r);
    wave_lds_fence();
    tile_out<kCells>(state + L.tile * (kTile * kCells), s_state, L.lane, L.rows);
    if (L.valid) to_move[L.b] = (int8_t)agent;
}

// gbl_validate: state-contract flags per board
__global__ __launch_bounds__(64) void k_validate(const int8_t *__restrict__ state, int8_t *__restrict__ flags, int64_t n,
                                                 int64_t ntiles)
{
    __shared__ uint32_t s_state[image_words<kCells>()];
    Lane L;
    if (!lane_setup(L, n, ntiles)) return;
    uint32_t r[7];
    load_state(state, s_state, L, r);
    if (!L.valid) return;
    flags[L.b] = (int8_t)validate_row(r);
}

// gbl_greedy: one decision per board.  Each lane owns a board (depth-1 walk, order-dependent replay,
// fallback test), but the depth-2 evaluations -- one moved + legal54 + outcomes54 per (board,
// candidate) pair, ~95 % of the work -- are pooled over the tile: the boards' candidate lists are
// laid back to back in LDS and lane l evaluates pairs l, l+64, ... whoever owns them, so a
// wavefront runs ceil(total / 64) evaluations instead of as many as its busiest board has candidates
// (~50 against a mean of ~33 on the masked-random mix).
// W wavefronts per workgroup share one tile: wavefront 0 owns the boards (loads, depth-1 walk, replay,
// outputs), all W evaluate pairs, so a tile's serial chain shrinks and every SIMD holds wavefronts in
// different phases.  W = 1 needs no barrier (depth 1 has no pairs and uses it).
// Orders the workgroup's LDS accesses (the only thing the wavefronts of a tile share).  Not __syncthreads(): that also
// waits for the caller's outstanding global stores (vmcnt), which in gbl_collect_policy would put the drain of a ply's
// trajectory stores on the path of the next decision.
template <int W>
__device__ __forceinline__ void pool_fence()
{
    if (W > 1)
        pair_barrier();
    else
        wave_lds_fence();
}

// ---- the greedy decision of a BLOCK of NT tiles (64 NT boards) by a workgroup of W wavefronts -----------------------------
// Wavefronts 0 .. NT-1 each OWN a tile (lane = board: loads, depth-1 walk, replay, outputs); wavefronts NT .. 2 NT - 1 look at
// the same tiles from the opponent's side while the owners walk depth 1; ALL W wavefronts lay out and evaluate the block's
// depth-2 work, whoever owns it (wavefronts beyond 2 NT do nothing else).  Shapes: <1, 16> / <1, 8> a lone tile spread over
// many wavefronts (small batches: the shortest serial chain), <1, 4> the same over four (large batches: the smallest
// workgroups), <2, 8> / <4, 8> / <4, 16>: blocks of tiles -- as many owners as helpers, so that fewer of a CU's wavefronts
// idle through the owners' phases (the tiles of a CU run in step) -- for batches of one generation of tiles; <1, 1>: depth 1
// only, no shared phase, no barrier.  greedy_shape() / policy_shape() hold the measurements behind the choice.
template <int NT, int W>
struct GreedyLds {
    static_assert(W == 1 || W >= 2 * NT, "a helper wavefront per owner");
    static constexpr int kBoards = kTile * NT;
    static_assert(kBoards <= 256, "(board << 8) | candidate fits 16 bits");
    // An owner's output image of gbl_collect_policy (scratch(), below) is staged in the pair list, which is padded so that NT images fit it
    static constexpr int kImageBytes = 4 * image_words<kObs>();
    static constexpr int kPairSlots = kBoards * kActions;
    static constexpr int kPairPad = NT * kImageBytes > 2 * kPairSlots ? (NT * kImageBytes - 2 * kPairSlots) / 2 : 0;
    alignas(16) uint16_t pair[kPairSlots + kPairPad]; // (board << 8) | candidate of every depth-2 evaluation of the block: those that
                                                      // take the fast form from the front, the ORDERED form (greedy_nonplain) from the back
                                                      // (the back = slot kPairSlots - 1; the padding behind it is scratch only)
    unsigned long long undef[kBoards][kRootItems];    // per board and member j of the root's replies: greedy_undefused
    uint16_t item[kBoards * kRootItems];              // (board << 8) | j of every member of a root's replies that is dealt out
    uint32_t board[kBoards][4];            // planes nz, neg, odd; bit 0: the agent to move, bit 1: the board wants depth 2
    uint64_t legal[kBoards];               // its legal moves on the root position
    unsigned long long work[kBoards];      // w0 = todo & ~dup: the candidates of a board that want a reply summary (see plan_of)
    unsigned long long replies[kBoards];   // the opponent's winning moves on the root (greedy_root); after the plan: those dealt out
    uint32_t risky[kBoards];               // 9 bits: squares where a placement from hand has to be evaluated (greedy_root)
    int npairs, nitems;                    // npairs: fast pairs in the low, ordered pairs in the high 16 bits
    uint16_t reply[kBoards][kActions];     // greedy_reply() of (board, candidate), where bit 0 is set
    unsigned long long threat[kBoards];    // candidates whose summary has bit 0 / bit 15 / bit 7 / bit 8,
    unsigned long long allwin[kBoards];    // and those whose first winning reply is a legal move of ours
    unsigned long long second[kBoards];    // (the sets greedy_replay_closed works on)
    unsigned long long block[kBoards];
    unsigned long long flegal[kBoards];
    unsigned long long nonplain[kBoards];  // greedy_nonplain of the board
    int job;                               // the next chunk of 64 pairs / items to hand out
    // The pair list is dead between two decisions (written after the third barrier of greedy_tile, read before its last one): a
    // kernel that decides in a loop stages its owners' output rows through it in between, owner w through bytes
    // [w, w + 1) * kImageBytes of it.  ONLY the pair list: the table `undef` is still read by every owner after the last
    // barrier (greedy_hand_merge), so an image laid over it -- round 3's layout: the list and the table as one region cut
    // in NT equal slices -- let a fast owner's next image overwrite a slower owner's rows (no barrier orders the two).
    static_assert(kImageBytes % 16 == 0 && NT * kImageBytes <= (int)sizeof(pair), "NT output images fit the (padded) pair list");
    __device__ __forceinline__ uint32_t *scratch(int owner_wave)
    {
        return reinterpret_cast<uint32_t *>(pair) + owner_wave * (kImageBytes / 4);
    }
};

// Inclusive prefix sum over the 64 lanes of a wavefront in six DPP adds (Hillis-Steele inside the rows of 16, then the rows'
// totals: lane 15 of a row into the next one, lane 31 into rows 2 and 3); scripts/microbench/dpp_scan.hip checks it on the GPU.
__device__ __forceinline__ uint32_t wave_incl_scan_add(uint32_t x)
{
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);  // row_shr:1
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);  // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);  // row_shr:4
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);  // row_shr:8
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
    return x;
}

// Compacts the (board, step) entries of a wavefront's lanes -- lane = board, `bits` = the steps (candidates / ranks) it has
// -- into a list shared by the whole block.  The lanes' counts are prefix-summed with six DPP adds (round 3: bit-sliced, one
// ballot + two mbcnt per bit of the largest possible count); the wavefront reserves its stretch of the list with ONE atomic,
// and a lane writes its few entries in a loop (as long as the busiest lane's count: 4-6 of 14 steps).  Round 3's first form --
// one ballot + mbcnt + store per STEP, in two passes around the atomic -- cost a wavefront 2 400 cycles (per-wavefront stamps),
// five barriers' worth.  (Every lane of the wavefront must be active: both callers are wave-uniform branches.)
template <int STEPS>
__device__ __forceinline__ void list_append(uint16_t *list, int *fill, uint32_t tag, int lane, uint32_t bits)
{
    static_assert(STEPS <= 31, "the steps of a lane fit a 32-bit set");
    const uint32_t cnt = (uint32_t)__popc(bits);
    const uint32_t inc = wave_incl_scan_add(cnt);
    uint32_t base = 0;
    if (lane == kTile - 1 && inc) base = (uint32_t)atomicAdd(fill, (int)inc);
    uint32_t at = (uint32_t)__builtin_amdgcn_readlane((int)base, kTile - 1) + inc - cnt;
    for (uint32_t rest = bits; rest; rest &= rest - 1u) list[at++] = (uint16_t)(tag + (uint32_t)__builtin_ctz(rest));
}

// list_append for two lists at once -- entries of `bits` go to `list` upwards, those of `bitsx` (few lanes have any) from
// `listx` DOWNWARDS (the two share one array) -- whose fill counts share one word (low / high 16 bits): one scan of the packed
// counts, one atomic per wavefront.
template <int STEPS>
__device__ __forceinline__ void list_append2(uint16_t *list, uint16_t *listx, int *fill, uint32_t tag, int lane, uint64_t bits, uint64_t bitsx)
{
    static_assert(STEPS <= 63, "the steps of a lane fit a 64-bit set");
    const uint32_t cnt = STEPS < 32 ? (uint32_t)__popc((uint32_t)bits) : (uint32_t)__popcll(bits);
    const uint32_t cntx = STEPS < 32 ? (uint32_t)__popc((uint32_t)bitsx) : (uint32_t)__popcll(bitsx);
    const uint32_t own = cnt | (cntx << 16);          // (a wavefront's totals are < 2^16 each: no carry between the halves)
    const uint32_t inc = wave_incl_scan_add(own);
    uint32_t base = 0;
    if (lane == kTile - 1 && inc) base = (uint32_t)atomicAdd(fill, (int)inc);
    const uint32_t start = (uint32_t)__builtin_amdgcn_readlane((int)base, kTile - 1) + inc - own;
    uint32_t at = start & 0xFFFFu, atx = start >> 16;
    if (STEPS < 32) {
        for (uint32_t rest = (uint32_t)bits; rest; rest &= rest - 1u) list[at++] = (uint16_t)(tag + (uint32_t)__builtin_ctz(rest));
        for (uint32_t rest = (uint32_t)bitsx; rest; rest &= rest - 1u) listx[-(int)(atx++)] = (uint16_t)(tag + (uint32_t)__builtin_ctz(rest));
    } else {
        for (uint64_t rest = bits; rest; rest &= rest - 1ull) list[at++] = (uint16_t)(tag + (uint32_t)__builtin_ctzll(rest));
        for (uint64_t rest = bitsx; rest; rest &= rest - 1ull) listx[-(int)(atx++)] = (uint16_t)(tag + (uint32_t)__builtin_ctzll(rest));
    }
}

// One decision per board of the block, by ALL 64 W threads of the workgroup (every thread calls, the barriers are inside).
// The owners (wave < NT, lane = board of their tile) pass their board, the agent to move, the legal mask handed to the
// policy (0: nothing to decide on this board -- an invalid lane, or in gbl_collect_policy a board whose mover plays at
// random), the board's depth (1, 2; 3 decides like 2) and the agent's last three actions; the other threads' arguments are
// ignored and they get an empty result.  deep: some board of SOME block of the launch may want depth 2 (workgroup-uniform:
// it decides whether the shared phases and their barriers exist at all; W == 1 has none).
//
// Depth 2 in five steps between barriers:  (A) the owners publish their boards;  (B) they walk depth 1 (greedy_head) while
// their helpers find the opponent's winning moves on the root and the risky squares (greedy_root) and a third wavefront per
// tile the moves of ours after which a reply could expose a line of ours (greedy_nonplain);  (C) the owners split their
// candidates (greedy_root_plan): placements from hand on non-risky squares are settled from the root's replies R, everything
// else is evaluated -- in the FAST form (threat squares) unless greedy_nonplain says otherwise;  (D) all wavefronts lay out the
// work -- the (board, candidate) pairs, fast ones from the front of the list and ordered ones from its back, and the (board,
// member of R) items;  (E) the work goes out in chunks of 64 through a counter -- ordered pairs, fast pairs (one moved +
// legal54 + wins54_plain / outcomes54 each: every reply's result at once), items (greedy_undefused) -- to whichever wavefront
// is free, leaving 16-bit summaries, candidate-set bits and table rows in LDS; then the owners replay the reference's depth-2
// loop in closed form over the sets (greedy_replay_closed).
// Safe to call in a loop: what the owners read last (replay) and write first (heads) is their own wavefront's business,
// and everybody else's reads of an iteration lie before its last barrier.
// (kSkip: 0 in the product.  The FLOOR builds of scripts/greedy_floor.sh -- experiment builds, gobblet_ab.h -- leave phases of the
// decision out, with WRONG results, so that what a block cannot go below is measured, not argued.)
template <int NT, int W>
__device__ __forceinline__ GreedyResult greedy_tile(GreedyLds<NT, W> &S, const Planes &p, int me, uint64_t mask, int depth,
                                                    bool deep, uint32_t prev3, TileStamps &ts)
{
    constexpr int kSkip = knob::kGreedySkip;
    const int lane = (int)(threadIdx.x & 63u), wave = wave_index();
    const bool owner = wave < NT;
    const int bi = (owner ? wave : wave - NT) * kTile + lane;  // the board this thread owns, or helps with (wave < 2 NT)
    GreedyHead h{0ull, 0ull, 0ull, 0ull, 0, -1};
    GreedyRootPlan plan{0ull, 0ull, 0ull};
    const bool two = owner && depth > 1 && mask != 0;  // this board takes part in the depth-2 round
    if constexpr (W == 1) {
        (void)S; (void)deep; (void)ts; (void)bi; (void)two;
        h = greedy_head(p, me, mask, 1);  // (one wavefront: depth 1 only -- the hosts never launch it for more)
        return greedy_finish(h, prev3);
    } else {
    if (owner && deep) {
        S.board[bi][0] = p.nz;
        S.board[bi][1] = p.neg;
        S.board[bi][2] = p.odd;
        S.board[bi][3] = (uint32_t)me | (two ? 2u : 0u);
        if (threadIdx.x == 0) {
            S.npairs = 0;
            S.nitems = 0;
            S.job = 0;
        }
    }
    if (deep) {
        GBL_WAVE_STAMP(8);
        pool_fence<W>();
        GBL_WAVE_STAMP(9);
        // (B) the helpers look at the root from the OPPONENT's side (as expensive as the depth-1 walk itself)
        if (!(kSkip & 8) && wave >= NT && wave < 2 * NT && (S.board[bi][3] & 2u)) {
            const Planes q{S.board[bi][0], S.board[bi][1], S.board[bi][2]};
            const GreedyRoot g = greedy_root(q, (int)(S.board[bi][3] & 1u));
            S.replies[bi] = g.replies;
            S.risky[bi] = g.risky;
        }
        // ... and a wavefront that would idle (blocks without one: the helpers, on top) finds the moves of ours after which
        // a reply could expose a line of ours: those pairs take the ordered form of the evaluation, all others the fast one
        constexpr int kPlainWave0 = W >= 3 * NT ? 2 * NT : NT;
        if (!(kSkip & 8) && wave >= kPlainWave0 && wave < kPlainWave0 + NT) {
            const int bn = (wave - kPlainWave0) * kTile + lane;
            if (S.board[bn][3] & 2u)
                S.nonplain[bn] = greedy_nonplain(Planes{S.board[bn][0], S.board[bn][1], S.board[bn][2]}, (int)(S.board[bn][3] & 1u));
        }
    }
    if (owner) {
        if (!(kSkip & 8)) h = greedy_head(p, me, mask, depth);  // empty mask: nothing to do
        if (deep) {
            S.legal[bi] = h.legal_me;
            S.threat[bi] = 0ull;
            S.allwin[bi] = 0ull;
            S.second[bi] = 0ull;
            S.block[bi] = 0ull;
            S.flegal[bi] = 0ull;
        }
    }
    // (C) The split of a board's candidates -- twin placements are not evaluated a second time; placements from hand on non-risky
    // squares not at all, they are settled from the root's replies; the rest in the fast or the ordered form -- needs the owner's
    // depth-1 result (w0 = todo & ~dup) and the helpers' findings.  Round 4 had the owners compute it alone between two barriers
    // (1.7 k cycles of a 21.8 k block with one wavefront per SIMD at work); round 5: the owners publish w0 with their other results
    // before barrier B, and whoever needs the split of a board afterwards -- the listing wavefronts for the boards they list, the
    // owners for their own -- derives it from LDS (~40 instructions, in the throughput-bound list phase): one barrier less.
    // Measured (scripts/ab_greedy.py, in-process, us per launch old -> new): <1,16> at 4 096 boards 8.96 -> 8.25, <2,8> at 32 768:
    // 10.65 -> 10.02 (gbl_collect_policy 10.67 -> 10.10 per ply); <4,16> at 65 536: 12.05 -> 12.25 and <1,4> at 2^20: 143.3 -> 144.8
    // -- blocks that fill their SIMDs pay for the sixteen (four) copies of the split more than the barrier cost them: they keep
    // round 4's flow (kListerPlan false).
    constexpr bool kListerPlan = (NT == 1 && W >= 8) || (NT == 2);
    if (kListerPlan && owner && deep) S.work[bi] = two ? (h.todo & ~h.dup) : 0ull;
    auto plan_of = [&](int bn) {
        const unsigned long long w0 = S.work[bn];
        const Planes q{S.board[bn][0], S.board[bn][1], S.board[bn][2]};
        const GreedyRoot gr{S.replies[bn], S.risky[bn]};
        const bool rule = __popcll(gr.replies) <= kRootItems;
        const uint64_t resolved = rule ? (w0 & greedy_from_hand(q, (int)(S.board[bn][3] & 1u)) & ~spread9(gr.risky)) : 0ull;
        return GreedyRootPlan{w0 & ~resolved, resolved, resolved ? gr.replies : 0ull};
    };
    GBL_TILE_STAMP(ts, 0);
    if (deep) {
        GBL_WAVE_STAMP(10);
        pool_fence<W>();  // the helpers' findings and the owners' depth-1 results are in
        if constexpr (kListerPlan) {
            if (two) plan = plan_of(bi);
        } else {
            if (owner) {  // the owners alone, between two barriers: S.work = the split's `eval` set itself
                if (two) plan = greedy_root_plan(h, p, me, GreedyRoot{S.replies[bi], S.risky[bi]});
                S.work[bi] = plan.eval;
                S.replies[bi] = plan.items;
            }
            GBL_WAVE_STAMP(0);
            pool_fence<W>();
        }
        GBL_WAVE_STAMP(1);
        // (D) the work lists: W / NT wavefronts share a tile's 54 candidate steps, and the owners take their tile's 6 rank
        // steps on top
        // (fewer listers with longer per-lane loops lose: two per tile +2 %, one per tile +12 % at 65 536 boards)
        constexpr int kWavesPerTile = W / NT, kSteps = (kActions + kWavesPerTile - 1) / kWavesPerTile;
        if (const int lw = wave - (W - NT * kWavesPerTile); lw >= 0 && !(kSkip & 4)) {  // (the last wavefronts: the owners come out of the plan last)
            const int g = lw / kWavesPerTile, c0 = (lw % kWavesPerTile) * kSteps;  // (a wavefront's steps stay inside one tile)
            if (c0 < kActions) {
                const int steps = kActions - c0 < kSteps ? kActions - c0 : kSteps;
                const int bn = g * kTile + lane;
                const uint64_t ev = kListerPlan ? plan_of(bn).eval : (uint64_t)S.work[bn];
                const uint64_t slow = ev ? S.nonplain[bn] : 0ull;  // (the nonplain set exists for boards that take part)
                const uint64_t wk = ((ev & ~slow) >> c0) & ((1ull << steps) - 1ull);
                const uint64_t wx = ((ev & slow) >> c0) & ((1ull << steps) - 1ull);
                list_append2<kSteps>(S.pair, S.pair + (GreedyLds<NT, W>::kBoards * kActions - 1), &S.npairs,
                                     ((uint32_t)bn << 8) + (uint32_t)c0, lane, wk, wx);
            }
        }
        GBL_WAVE_STAMP(6);
        // (blocks of tiles: by the owners -- the oldest wavefronts on their SIMDs win the issue arbitration and are through
        // first; a lone tile: by its helper, the owner's own path being the longest there)
        constexpr int kItemWave0 = NT * kWavesPerTile < W ? 0 : NT == 1 ? 1 : 0;  // (wavefronts that list no pairs, if any)
        if (!(kSkip & 4) && wave >= kItemWave0 && wave < kItemWave0 + NT) {
            const int g = wave - kItemWave0, bn = g * kTile + lane;
            // (the owners list their own boards; a lone tile's helper reads, or derives, its board's)
            const uint32_t nr = (uint32_t)__popcll(!kListerPlan ? (uint64_t)S.replies[bn] : (kItemWave0 == 0 ? plan : plan_of(bn)).items);
            list_append<kRootItems>(S.item, &S.nitems, (uint32_t)bn << 8, lane, (1u << nr) - 1u);
        }
        GBL_WAVE_STAMP(7);
        pool_fence<W>();
        GBL_WAVE_STAMP(2);
        // (E) the work in chunks of 64, handed out through a counter -- the ordered pairs first (an ordered evaluation is
        // twice a fast one), then the fast pairs, then the items: whoever is through takes the next chunk, so the oldest
        // wavefronts of a SIMD, which win the issue arbitration, simply take more
        int nfast = S.npairs & 0xFFFF, nslow = (int)((uint32_t)S.npairs >> 16);
        const int nitems = S.nitems;
        if constexpr (knob::kGreedyPairCap > 0) {  // (floor builds only: pairs beyond the cap are dropped -- WRONG results, timing only)
            nslow = nslow < knob::kGreedyPairCap ? nslow : knob::kGreedyPairCap;
            nfast = nfast < knob::kGreedyPairCap - nslow ? nfast : knob::kGreedyPairCap - nslow;
        }
        const int jslow = (nslow + kTile - 1) / kTile, jfast = (nfast + kTile - 1) / kTile, jitem = (nitems + kTile - 1) / kTile;
        auto record = [&](uint32_t o, uint32_t a, uint32_t sum) {
            if (sum & 1u) {
                S.reply[o][a] = (uint16_t)sum;
                atomicOr(&S.threat[o], 1ull << a);
                if (sum & (1u << 7)) atomicOr(&S.second[o], 1ull << a);
                if (sum & (1u << 8)) atomicOr(&S.block[o], 1ull << a);
                if ((S.legal[o] >> ((sum >> 1) & 63u)) & 1ull) atomicOr(&S.flegal[o], 1ull << a);
            }
            if (sum >> 15) atomicOr(&S.allwin[o], 1ull << a);
        };
        for (;;) {
            if (kSkip & 2) break;
            int j = 0;
            if (lane == 0) j = atomicAdd(&S.job, 1);
            j = __builtin_amdgcn_readfirstlane(j);
            if (j >= jslow + jfast + jitem) break;
            if (j < jslow) {
                const int g = j * kTile + lane;
                if (g < nslow) {
                    const uint32_t pair = S.pair[GreedyLds<NT, W>::kBoards * kActions - 1 - g], o = pair >> 8, a = pair & 0xFFu;
                    const Planes q{S.board[o][0], S.board[o][1], S.board[o][2]};
                    record(o, a, greedy_reply<false>(q, (int)(S.board[o][3] & 1u), S.legal[o], a));
                }
            } else if (j < jslow + jfast) {
                const int g = (j - jslow) * kTile + lane;
                if (g < nfast) {
                    const uint32_t pair = S.pair[g], o = pair >> 8, a = pair & 0xFFu;
                    const Planes q{S.board[o][0], S.board[o][1], S.board[o][2]};
                    record(o, a, greedy_reply<true>(q, (int)(S.board[o][3] & 1u), S.legal[o], a));
                }
            } else {
                const int g = (j - jslow - jfast) * kTile + lane;
                if (g < nitems) {
                    const uint32_t it = S.item[g], o = it >> 8, jr = it & 0xFFu;
                    const Planes q{S.board[o][0], S.board[o][1], S.board[o][2]};
                    const uint32_t a2 = kth_bit64(S.replies[o], jr);
                    S.undef[o][jr] = greedy_item_row(q, (int)(S.board[o][3] & 1u), S.legal[o], a2);  // (tagged: see greedy_hand_merge_tagged)
                }
            }
        }
        GBL_WAVE_STAMP(3);
        GBL_TILE_STAMP(ts, 1);
        GBL_WAVE_STAMP(4);
        pool_fence<W>();
        GBL_WAVE_STAMP(5);
    }
    GBL_TILE_STAMP(ts, 2);
    if (!owner) return GreedyResult{-1, 0ull, false};
    if (two && !(kSkip & 1)) {  // :103-157 on the owner's lane, in closed form over the candidate sets
        ReplySets r{S.threat[bi], S.allwin[bi], S.second[bi], S.block[bi], S.flegal[bi]};
        uint64_t undef[kRootItems] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull};
        const int nitems_b = __popcll(plan.items);
        if (plan.items) {  // the placements settled from the root: the table's (tagged) rows merged in the reference's reply order
#pragma unroll
            for (int j = 0; j < kRootItems; ++j) undef[j] = S.undef[bi][j];
            const GreedyHandSets hs = greedy_hand_merge_tagged(nitems_b, plan.resolved, undef);
            r.threat |= hs.threat;
            r.second |= hs.second;
            r.block |= hs.block;
            r.flegal |= hs.flegal;
        }
        greedy_replay_closed(h, r, [&](int a) {
            const uint32_t twin = ((h.dup >> a) & 1ull) ? (uint32_t)a - 9u : (uint32_t)a;  // twin placements share a summary
            return ((plan.resolved >> twin) & 1ull) ? greedy_hand_lookup_tagged(nitems_b, h.legal_me, undef, twin)
                                                    : (uint32_t)S.reply[bi][twin];
        });
    }
    return greedy_finish(h, prev3);
    }
}

// greedy_tile as a real function call, for the ply loop of gbl_collect_policy: inlined there, the decision's literal
// constants are hoisted out of the loop and stay live across the whole iteration (~160 VGPRs, or spills under the kernel's
// register budget); behind a call they live and die inside the callee.
template <int NT, int W>
__device__ __attribute__((noinline)) GreedyResult greedy_tile_call(GreedyLds<NT, W> &S, Planes p, int me, uint64_t mask, int depth,
                                                                   bool deep, uint32_t prev3)
{
    TileStamps ts{};
    return greedy_tile<NT, W>(S, p, me, mask, depth, deep, prev3, ts);
}

// the agent's last three actions (one per byte, 0xFF = none) from the 6 history bytes of a board, read as three 16-bit words
__device__ __forceinline__ uint32_t hist_prev3(uint32_t h0, uint32_t h1, uint32_t h2, int me)
{
    return me ? ((h1 >> 8) | (h2 << 8)) & 0x00FFFFFFu : (h0 | (h1 << 16)) & 0x00FFFFFFu;
}

// The tile of an owner wavefront of a block: tile = block * NT + wave; tiles past the end have no rows (their wavefront still
// takes part in the block's barriers).  False: the whole block lies past the end.
template <int NT>
__device__ __forceinline__ bool block_lane_setup(Lane &L, int64_t n, int64_t ntiles)
{
    const int wave = wave_index();
    if ((int64_t)blockIdx.x * NT >= ntiles) return false;
    L.tile = (int64_t)blockIdx.x * NT + (wave < NT ? wave : 0);
    L.lane = (int)(threadIdx.x & 63u);
    const int64_t left = n - L.tile * kTile;
    L.rows = (wave >= NT || left <= 0) ? 0 : left < kTile ? (int)left : kTile;
    L.valid = L.lane < L.rows;
    L.b = L.tile * kTile + L.lane;
    return true;
}

template <int NT, int W>
__global__ __launch_bounds__(64 * W) void k_greedy(const int8_t *__restrict__ state,
                                                   const int8_t *__restrict__ to_move,
                                                   const int8_t *__restrict__ mask_in,
                                                   const int8_t *__restrict__ hist, int depth,
                                                   int32_t *__restrict__ action_out, int8_t *__restrict__ cand_out,
                                                   int8_t *__restrict__ fallback_out, int64_t n, int64_t ntiles,
                                                   int8_t *__restrict__ hist_rw, int32_t *__restrict__ final_out,
                                                   uint64_t seed, uint64_t env_base, uint32_t call,
                                                   const uint32_t *__restrict__ call_dev)
{
    if (call_dev) call += *call_dev;
    // hist_rw != NULL: gbl_greedy_act -- the history is read from and appended to hist_rw, and final_out gets
    // the action the policy returns (the fallback draw included); hist is then unused.
    if (hist_rw) hist = hist_rw;
    __shared__ uint32_t s_states[NT][image_words<kCells>()];
    __shared__ uint32_t s_masks[NT][image_words<kActions>()];
    __shared__ GreedyLds<NT, W> S;
    GBL_STAMP(0);
    GBL_STAMP_REAL(0);
    Lane L;
    if (!block_lane_setup<NT>(L, n, ntiles)) return;  // the same for every thread of the workgroup
    const int wave = wave_index();
    const bool owner = wave < NT;
    uint32_t *const s_state = s_states[owner ? wave : 0], *const s_mask = s_masks[owner ? wave : 0];
    Planes p{0u, 0u, 0u};
    uint32_t prev3 = 0x00FFFFFFu, h0 = 0xFFFFu, h1 = 0xFFFFu, h2 = 0xFFFFu;
    int me = 0;
    uint64_t mask = 0;
    if (owner && L.rows > 0) {
        // per-board scalars first, branch-free from a clamped index: in flight together with the tile (see k_step);
        // the history of BOTH agents (6 bytes, 2-byte aligned), the mover picks its three below
        const int64_t bs = L.valid ? L.b : n - 1;
        const int tm = to_move[bs];
        if (hist) {
            const uint16_t *hp = reinterpret_cast<const uint16_t *>(hist + bs * 6);
            h0 = hp[0]; h1 = hp[1]; h2 = hp[2];
        }
        uint32_t r[7];
        load_state(state, s_state, L, r);
        p = planes_of(L, r);
        me = L.valid ? (tm != 0) : 0;
        if (mask_in) {
            tile_in<kActions>(mask_in + L.tile * (kTile * kActions), s_mask, L.lane, L.rows);
            wave_lds_fence();
            uint32_t d[14];
            row_load<kActions>(s_mask, L.lane, d);
            mask = mask_bits(d);
            wave_lds_fence();
        } else {
            mask = legal54(p, me);
        }
        if (!L.valid) mask = 0;
        if (hist && L.valid)  // bytes 0-2: player_1's last three actions, bytes 3-5: player_2's
            prev3 = hist_prev3(h0, h1, h2, me);
    }
    TileStamps ts{};
    const GreedyResult g = greedy_tile<NT, W>(S, p, me, mask, depth, depth > 1, prev3, ts);
    GBL_STAMP_VAL(1, ts.t[0]);
    GBL_STAMP_VAL(2, ts.t[1]);
    GBL_STAMP_VAL(3, ts.t[2]);
    if (!owner || L.rows == 0) return;
    if (cand_out && !(knob::kGreedySkip & 16)) {
        mask_to_image<true>(s_mask, L.lane, g.cands);
        wave_lds_fence();
        tile_out<kActions>(cand_out + L.tile * (kTile * kActions), s_mask, L.lane, L.rows);
    }
    if (L.valid) {
        if (action_out) action_out[L.b] = g.fallback ? -1 : g.chosen;
        if (fallback_out) fallback_out[L.b] = g.fallback ? 1 : 0;
        if (hist_rw) {
            // :211-217 with the library's sampler, then :219
            const int fin = g.fallback ? pick54(g.cands, draw32(seed, env_base + (uint64_t)L.b, call, kStreamGreedy)) : g.chosen;
            final_out[L.b] = fin;
            int8_t *hp = hist_rw + (L.b * 2 + me) * 3;
            hp[0] = (int8_t)(prev3 >> 8);
            hp[1] = (int8_t)(prev3 >> 16);
            hp[2] = (int8_t)fin;
        }
    }
    GBL_STAMP(4);
    GBL_WAVE_STAMP(11);
    GBL_STAMP_DRAIN(5);
    GBL_STAMP_FLUSH(L.tile);
}

// gbl_collect_policy: gbl_collect with a DEVICE-SIDE POLICY per side -- masked-random, or the greedy lookahead of
// GreedyGobbletPolicy.compute_action (depth 1 / 2) -- `plies` plies per launch, every ply materialised in its trajectory
// slot.  The reference plays whole games with the greedy policy on either or both sides (tutorials/GreedyAgent/
// tutorial_greedy.py:16-54: one policy object acting for both agents, the first two plies of a game drawn at random;
// greedy_policy_tianshou.py:63-84: greedy against a learner); here the decision (greedy_tile, all W wavefronts of the
// workgroup), the fallback draw (:211-217), the history append (:219), the move, winner, auto-reset and the next
// observation / mask all happen inside one launch, the block's boards living in LDS and registers between the plies.
// The owner wavefronts do everything but the shared depth-2 work.
// (Register budget: four wavefronts per SIMD = 128 VGPRs.  Left alone the compiler takes ~160 -- the ply loop keeps the
// decision's literal constants live across iterations -- which costs a wavefront of occupancy for nothing.)
template <int NT, int W>
__global__ __launch_bounds__(64 * W) __attribute__((amdgpu_waves_per_eu(W > 1 && NT == 1 ? GBL_KNOB_CP_WAVES_PER_EU : 1, 8))) void k_collect_policy(
    int8_t *__restrict__ state, int8_t *__restrict__ to_move, int64_t n, int64_t ntiles, uint64_t seed, uint64_t env_base,
    const uint32_t *__restrict__ ply_dev, uint32_t ply0, uint32_t plies, int8_t *__restrict__ done, int64_t ply_stride,
    int64_t tile_stride, int32_t *__restrict__ actions_t, int8_t *__restrict__ winner_t, int8_t *__restrict__ reward_t,
    int8_t *__restrict__ done_t, int8_t *__restrict__ to_move_t, int8_t *__restrict__ mask_t, int8_t *__restrict__ obs_t,
    int32_t *__restrict__ chosen_t, int8_t *__restrict__ how_t, int8_t *__restrict__ cand_t, int8_t *__restrict__ hist,
    int policy0, int policy1, int opening_plies, int illegal_mode, int64_t *__restrict__ counters, int32_t *__restrict__ turn)
{
    __shared__ uint32_t s_states[NT][image_words<kCells>()];
    __shared__ GreedyLds<NT, W> S;
    // The two generator blocks of a board (environment stream, fallback-draw stream: four plies' words each) are parked here, not
    // held in eight registers across the decision: a ply reads back the one word of each it needs.
    __shared__ uint32_t s_draws[NT][8][kTile];
    if (ply_dev) ply0 += *ply_dev;
    Lane L;
    if (!block_lane_setup<NT>(L, n, ntiles)) return;  // the same for every thread of the workgroup
    const int wave = wave_index();
    const bool owner = wave < NT, active = owner && L.rows > 0;
    uint32_t *const s_state = s_states[owner ? wave : 0], *const s_out = S.scratch(owner ? wave : 0);
    const bool deep = policy0 > 1 || policy1 > 1;
    Planes p{0u, 0u, 0u};
    int mover = 0, dn = 0, tabs = 0;
    uint32_t hp0 = 0x00FFFFFFu, hp1 = 0x00FFFFFFu;  // the two agents' last three actions, one per byte (0xFF = none)
    uint32_t (*const draws)[kTile] = s_draws[owner ? wave : 0];
    auto park_draws = [&](uint32_t ply) {  // the blocks that serve plies 4 * (ply >> 2) ... of this board: generated once per four plies
        const Draw4 block = draw_block(seed, env_base + (uint64_t)L.b, ply);
        Draw4 fblock{{0u, 0u, 0u, 0u}};
        if (deep || policy0 > 0 || policy1 > 0) fblock = draw_block(seed, env_base + (uint64_t)L.b, ply, kStreamGreedy);
        // (word w of a block serves the ply with ply & 3 == w: draw_word's order)
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            draws[w][L.lane] = block.w[w];
            draws[4 + w][L.lane] = fblock.w[w];
        }
    };
    const ImageRow row{reinterpret_cast<uint8_t *>(s_state) + L.lane * kCells};
    uint64_t legal = 0;
    uint32_t games = 0, w1 = 0, w2 = 0;
    if (active) {
        const int64_t bs = L.valid ? L.b : n - 1;
        const int tm = to_move[bs];
        if (turn) tabs = turn[bs];
        if (hist) {
            const uint16_t *hp = reinterpret_cast<const uint16_t *>(hist + bs * 6);
            const uint32_t h0 = hp[0], h1 = hp[1], h2 = hp[2];
            hp0 = hist_prev3(h0, h1, h2, 0);
            hp1 = hist_prev3(h0, h1, h2, 1);
        }
        uint32_t r[7];
        load_state(state, s_state, L, r, [&] { park_draws(ply0); });
        mover = L.valid && tm != 0;
        p = planes_of(L, r);
        legal = legal54(p, mover);
    }
    TileStamps ts{};
    for (uint32_t t = 0; t < plies; ++t) {
        const uint32_t ply = ply0 + t;
        const int pol = mover ? policy1 : policy0;
        // the greedy policy acts on this board at this ply (tutorial_greedy.py:34-49: not on a game's opening plies)
        const bool gre = active && L.valid && pol > 0 && tabs >= opening_plies;
        const uint32_t prev3 = mover ? hp1 : hp0;
        // (one tile per workgroup: behind a call, see greedy_tile_call -- 12.6 -> 11.6 us per ply at 16 384 boards, 79 -> 71 at
        // 262 144; blocks of tiles inlined: <2,8> has the registers, <4,16> exactly its 128 since round 4 -- table above policy_shape)
        GreedyResult g;
        if constexpr (NT == 1 && W > 1)  // (<4,16> inlined holds exactly its 128 VGPRs and is 5 % faster than behind the call)
            g = greedy_tile_call<NT, W>(S, p, mover, gre ? legal : 0ull, pol, deep, prev3);
        else
            g = greedy_tile<NT, W>(S, p, mover, gre ? legal : 0ull, pol, deep, prev3, ts);
        if (!active) continue;
        int action;
        {
            // :211-217 with the library's sampler on generator stream 1 (as gbl_greedy_act, keyed by the ply index)
            // (draw32(seed, board, ply, kStreamGreedy) is word ply & 3 of the stream's block for plies 4 * (ply >> 2) ...: one
            // generator call per four plies, like the environment's)
            const uint32_t wsel = draw_word_index(ply);
            const int greedy_action = g.fallback ? pick54(g.cands, draws[4 + wsel][L.lane]) : g.chosen;
            // (the masked-random pick only where some board of the tile needs one: none in greedy-vs-greedy play past the openings;
            //  the lanes past the end of a ragged tile then play their first legal move -- never an illegal one, which in TERMINATE
            //  mode would end and restart their phantom game every ply: nothing of theirs is stored or tallied, but their state
            //  stays that of a board that is being played)
            int random_action = (int)__builtin_ctzll(legal | (1ull << 63));
            if (wave_any(L.valid && !gre)) random_action = pick54(legal, draws[wsel][L.lane]);
            action = gre ? greedy_action : random_action;
            if (gre) {  // :219: the acting agent's history takes the returned action
                const uint32_t np3 = (prev3 >> 8) | (((uint32_t)action & 0xFFu) << 16);
                hp0 = mover ? hp0 : np3;
                hp1 = mover ? np3 : hp1;
            }
        }
        if (t + 1 < plies && ((ply + 1) & 3u) == 0) park_draws(ply + 1);
        const Ply y = play_ply(p, row, mover, legal, action, illegal_mode);
        dn = y.terminal ? 1 : 0;
        if (y.terminal) {  // raw_env.reset, gobblet.py:275-290
            p = Planes{0u, 0u, 0u};
            mover = 0;
            row.reset();
        }
        tabs = next_turn(tabs, y, 1);
        if (counters) {
            games += __popcll(__ballot(L.valid && y.terminal));
            w1 += __popcll(__ballot(L.valid && y.winner == 1));
            w2 += __popcll(__ballot(L.valid && y.winner == -1));
        }
        const int64_t cell = (int64_t)t * ply_stride + L.tile * tile_stride;
        if (L.valid) {
            const int64_t at = cell + L.lane;
            if (actions_t) actions_t[at] = action;
            if (winner_t) winner_t[at] = (int8_t)y.winner;
            if (reward_t) reinterpret_cast<uint16_t *>(reward_t)[at] = (uint16_t)((y.r0 & 0xFF) | ((y.r1 & 0xFF) << 8));
            if (done_t) done_t[at] = (int8_t)dn;
            if (to_move_t) to_move_t[at] = (int8_t)mover;
            if (chosen_t) chosen_t[at] = (gre && !g.fallback) ? g.chosen : -1;
            if (how_t) how_t[at] = (int8_t)(gre ? (g.fallback ? GBL_HOW_FALLBACK : GBL_HOW_GREEDY) : GBL_HOW_RANDOM);
        }
        constexpr int kPolicy = kStoreStreamDrop;
        // The lane index the output rows are staged and stored with is made opaque once per ply: everything derived from it (image
        // offsets, the `lane < REM` predicates, buffer offsets of up to eight vectors per tile) is loop-invariant, and hoisted out of
        // the ply loop it stayed live across the decision -- under the 128-VGPR budget of the <1, W> shapes that was 22 spilled
        // registers (96 B of scratch per lane and ply); recomputing it costs a few shifts and adds per ply.
        int ol = L.lane;
#ifndef GBL_HOST_EMU
        asm volatile("" : "+v"(ol));
#endif
        auto mask_rows = [&](uint64_t bits, int8_t *__restrict__ dst) {  // (store_mask / store_obs with few registers)
            mask_to_image<true>(s_out, ol, bits);
            wave_lds_fence();
            tile_out_narrow<kActions, kPolicy, 2>(dst, s_out, ol, L.rows);
            wave_lds_fence();
        };
        wave_lds_fence();
        if (cand_t) mask_rows(gre ? g.cands : 0ull, cand_t + cell * kActions);
        if (obs_t) {
            obs_image_zero(s_out, ol);
            wave_lds_fence();
            obs_scatter(s_out, ol, p, mover);
            wave_lds_fence();
            tile_out_narrow<kObs, kPolicy, 2>(obs_t + cell * kObs, s_out, ol, L.rows);
            wave_lds_fence();
        }
        legal = legal54(p, mover);  // the next mover's: stored now, the next ply's policy is handed it
        if (mask_t) mask_rows(legal, mask_t + cell * kActions);
    }
    if (!active) return;
    wave_lds_fence();  // every lane's byte patches are in the state image
    tile_out<kCells>(state + L.tile * (kTile * kCells), s_state, L.lane, L.rows);
    if (L.valid) {
        to_move[L.b] = (int8_t)mover;
        done[L.b] = (int8_t)dn;
        if (turn) turn[L.b] = tabs;
        if (hist) {
            uint16_t *hp = reinterpret_cast<uint16_t *>(hist + L.b * 6);
            hp[0] = (uint16_t)(hp0 & 0xFFFFu);
            hp[1] = (uint16_t)(((hp0 >> 16) & 0xFFu) | ((hp1 & 0xFFu) << 8));
            hp[2] = (uint16_t)((hp1 >> 8) & 0xFFFFu);
        }
    }
    if (counters && L.lane == 0) {
        unsigned long long *c = reinterpret_cast<unsigned long long *>(counters) +
                                (size_t)(L.tile % GBL_COUNTER_STRIPES) * GBL_COUNTER_STRIDE;
        atomicAdd(c + 0, (unsigned long long)L.rows * plies);
        if (games) atomicAdd(c + 1, (unsigned long long)games);
        if (w1) atomicAdd(c + 2, (unsigned long long)w1);
        if (w2) atomicAdd(c + 3, (unsigned long long)w2);
    }
}

// k_collect_small for a checked call (see gbl_collect_from); cfg = 100 LA + 10 KO + MERGE (collect_variant)
// (the product build instantiates the forms collect_variant() can return; an experiment build a whole menu: gobblet_ab.h)
bool launch_small(int cfg, int8_t *state, int8_t *to_move, int8_t *done, const int32_t *first_actions, int8_t *first_status, int32_t *actions_t, int8_t *winner_t,
                  int8_t *reward_t, int8_t *done_t, int8_t *to_move_t, int8_t *mask_t, int8_t *obs_t, int64_t n, int64_t ply_stride,
                  int64_t tile_stride, uint64_t seed, uint64_t env_base, uint32_t ply0, const uint32_t *ply_dev, uint32_t plies,
                  int illegal_mode, int64_t *counters, int32_t *turn, hipStream_t s)
{
    const int la = cfg / 100;
    const int64_t gb = kTile / la, ngroups = (n + gb - 1) / gb;
#define GBL_SMALL_K(M, O, D, LA, KO, MG)                                                                                        \
    hipLaunchKernelGGL((k_collect_small<M, O, D, LA, KO, MG>), dim3((uint32_t)ngroups), dim3(64 * small_waves<M, O, KO, MG>()), 0, s, \
                       state, to_move, n, ngroups, seed, env_base, ply_dev, ply0, plies, done, ply_stride, tile_stride, actions_t,      \
                       winner_t, reward_t, done_t, to_move_t, mask_t, obs_t, illegal_mode, counters, turn, first_actions, first_status)
#define GBL_SMALL_D(M, O, LA, KO, MG)                           \
    if (ply_dev) { GBL_SMALL_K(M, O, true, LA, KO, MG); }       \
    else { GBL_SMALL_K(M, O, false, LA, KO, MG); }
#define GBL_SMALL_CFG(LA, KO, MG)                                                                              \
    if (cfg == 100 * LA + 10 * KO + (MG ? 1 : 0)) {                                                            \
        if (mask_t && obs_t) { GBL_SMALL_D(true, true, LA, KO, MG); }                                          \
        else if (mask_t) { GBL_SMALL_D(true, false, LA, KO, MG); }                                             \
        else if (obs_t) { GBL_SMALL_D(false, true, LA, KO, MG); }                                              \
        else { GBL_SMALL_D(false, false, LA, KO, MG); }                                                        \
        return true;                                                                                           \
    }
    GBL_SMALL_CFG(2, 2, false)
    GBL_SMALL_CFG(2, 1, false)
    GBL_KNOB_SMALL_FORMS
#undef GBL_SMALL_CFG
#undef GBL_SMALL_D
#undef GBL_SMALL_K
    return false;
}

int greedy_shape(int depth, int64_t n);  // (defined with the greedy entry points below)
int policy_shape(int depth, int64_t n);

}  // namespace

// ===========================================================================================
// C-ABI

#define GBL_CHECK_N(n)                                      \
    do {                                                    \
        if ((n) < 0) return fail(GBL_ERR_ARG, "n < 0");     \
        if ((n) == 0) return GBL_OK;                        \
    } while (0)
#define GBL_NEED(p, name)                                                 \
    do {                                                                  \
        if (!(p)) return fail(GBL_ERR_ARG, name " must not be NULL");     \
    } while (0)
#define GBL_ALIGNED(p, name)                                                                   \
    do {                                                                                       \
        if ((p) && !aligned16(p)) return fail(GBL_ERR_ALIGN, name " must be 16-byte aligned"); \
    } while (0)
#define GBL_LAUNCHED(name)                                   \
    do {                                                     \
        hipError_t e_ = hipGetLastError();                   \
        if (e_ != hipSuccess) return hip_fail(e_, name);     \
        return GBL_OK;                                       \
    } while (0)

extern "C" {


const char *gbl_last_error(void) { return g_err; }

int gbl_layout_info(int32_t out[6])
{
    if (!out) return fail(GBL_ERR_ARG, "out must not be NULL");
    out[0] = GBL_ABI_VERSION; out[1] = kCells; out[2] = kActions; out[3] = kObs; out[4] = kTile; out[5] = 16;
    return GBL_OK;
}

int gbl_reset(int8_t *state, int8_t *to_move, int8_t *done, int8_t *winner, int64_t n, void *stream)
{
    GBL_CHECK_N(n);
    GBL_NEED(state, "state"); GBL_NEED(to_move, "to_move"); GBL_NEED(done, "done");
    hipStream_t s = (hipStream_t)stream;
    hipError_t e;
    if ((e = hipMemsetAsync(state, 0, (size_t)n * kCells, s)) != hipSuccess) return hip_fail(e, "gbl_reset");
    if ((e = hipMemsetAsync(to_move, 0, (size_t)n, s)) != hipSuccess) return hip_fail(e, "gbl_reset");
    if ((e = hipMemsetAsync(done, 0, (size_t)n, s)) != hipSuccess) return hip_fail(e, "gbl_reset");
    if (winner && (e = hipMemsetAsync(winner, 0, (size_t)n, s)) != hipSuccess) return hip_fail(e, "gbl_reset");
    return GBL_OK;
}

int gbl_legal_mask(const int8_t *state, const int8_t *to_move, int8_t *mask, int64_t n, void *stream)
{
    GBL_CHECK_N(n);
    GBL_NEED(state, "state"); GBL_NEED(to_move, "to_move"); GBL_NEED(mask, "mask");
    GBL_ALIGNED(state, "state"); GBL_ALIGNED(mask, "mask");
    Geometry g = geometry(n);
    hipLaunchKernelGGL(k_legal_mask, dim3(g.grid), dim3(64), 0, (hipStream_t)stream, state, to_move, mask, n, g.ntiles);
    GBL_LAUNCHED("gbl_legal_mask");
}

int gbl_is_legal(const int8_t *state, const int8_t *agent_index, const int32_t *actions, int8_t *out, int64_t n,
                 void *stream)
{
    GBL_CHECK_N(n);
    GBL_NEED(state, "state"); GBL_NEED(agent_index, "agent_index"); GBL_NEED(actions, "actions"); GBL_NEED(out, "out");
    GBL_ALIGNED(state, "state");
    Geometry g = geometry(n);
    hipLaunchKernelGGL(k_is_legal, dim3(g.grid), dim3(64), 0, (hipStream_t)stream, state, agent_index, actions, out, n,
                       g.ntiles);
    GBL_LAUNCHED("gbl_is_legal");
}

int gbl_play_turn(int8_t *state, const int8_t *agent_index, const int32_t *actions, int64_t n, void *stream)
{
    GBL_CHECK_N(n);
    GBL_NEED(state, "state"); GBL_NEED(agent_index, "agent_index"); GBL_NEED(actions, "actions");
    GBL_ALIGNED(state, "state");
    Geometry g = geometry(n);
    hipLaunchKernelGGL(k_play_turn, dim3(g.grid), dim3(64), 0, (hipStream_t)stream, state, agent_index, actions, n,
                       g.ntiles);
    GBL_LAUNCHED("gbl_play_turn");
}

int gbl_winner(const int8_t *state, int8_t *winner, int64_t n, void *stream)
{
    GBL_CHECK_N(n);
    GBL_NEED(state, "state"); GBL_NEED(winner, "winner");
    GBL_ALIGNED(state, "state");
    Geometry g = geometry(n);
    hipLaunchKernelGGL(k_winner, dim3(g.grid), dim3(64), 0, (hipStream_t)stream, state, winner, n, g.ntiles);
    GBL_LAUNCHED("gbl_winner");
}

int gbl_flatboard(const int8_t *state, int8_t *flat, int64_t n, void *stream)
{
    GBL_CHECK_N(n);
    GBL_NEED(state, "state"); GBL_NEED(flat, "flat");
    GBL_ALIGNED(state, "state"); GBL_ALIGNED(flat, "flat");
    Geometry g = geometry(n);
    hipLaunchKernelGGL(k_flatboard, dim3(g.grid), dim3(64), 0, (hipStream_t)stream, state, flat, n, g.ntiles);
    GBL_LAUNCHED("gbl_flatboard");
}

int gbl_covered(const int8_t *state, int8_t *cov, int64_t n, void *stream)
{
    GBL_CHECK_N(n);
    GBL_NEED(state, "state"); GBL_NEED(cov, "cov");
    GBL_ALIGNED(state, "state"); GBL_ALIGNED(cov, "cov");
    Geometry g = geometry(n);
    hipLaunchKernelGGL(k_covered, dim3(g.grid), dim3(64), 0, (hipStream_t)stream, state, cov, n, g.ntiles);
    GBL_LAUNCHED("gbl_covered");
}

int gbl_observe(const int8_t *state, const int8_t *to_move, int agent_sel, int8_t *obs, int64_t n, void *stream)
{
    GBL_CHECK_N(n);
    GBL_NEED(state, "state"); GBL_NEED(obs, "obs");
    if (agent_sel < -1 || agent_sel > 1) return fail(GBL_ERR_ARG, "agent_sel must be -1, 0 or 1");
    if (agent_sel < 0) GBL_NEED(to_move, "to_move (agent_sel == -1)");
    GBL_ALIGNED(state, "state"); GBL_ALIGNED(obs, "obs");
    Geometry g = geometry(n);
    hipLaunchKernelGGL(k_observe, dim3(g.grid), dim3(64), 0, (hipStream_t)stream, state, to_move, agent_sel, obs, n,
                       g.ntiles);
    GBL_LAUNCHED("gbl_observe");
}

int gbl_pinned_alloc(int64_t bytes, void **host_ptr, void **dev_ptr)
{
    if (bytes <= 0 || !host_ptr || !dev_ptr) return fail(GBL_ERR_ARG, "gbl_pinned_alloc: bytes > 0, host_ptr and dev_ptr required");
    void *h = nullptr, *d = nullptr;
    hipError_t e = hipHostMalloc(&h, (size_t)bytes, hipHostMallocMapped | hipHostMallocPortable);
    if (e != hipSuccess) return hip_fail(e, "gbl_pinned_alloc");
    e = hipHostGetDevicePointer(&d, h, 0);
    if (e != hipSuccess) {
        (void)hipHostFree(h);
        return hip_fail(e, "gbl_pinned_alloc (device pointer)");
    }
    memset(h, 0, (size_t)bytes);
    *host_ptr = h;
    *dev_ptr = d;
    return GBL_OK;
}

int gbl_pinned_free(void *host_ptr)
{
    if (!host_ptr) return GBL_OK;
    hipError_t e = hipHostFree(host_ptr);
    if (e != hipSuccess) return hip_fail(e, "gbl_pinned_free");
    return GBL_OK;
}

int gbl_block_alloc(int64_t bytes, void **dev_ptr)
{
    if (bytes <= 0 || !dev_ptr) return fail(GBL_ERR_ARG, "gbl_block_alloc: bytes > 0 and dev_ptr required");
    void *d = nullptr;
    hipError_t e = hipMalloc(&d, (size_t)bytes);
    if (e != hipSuccess) {
        (void)hipGetLastError();  // (an out-of-memory answer must not linger as the thread's "last error")
        return hip_fail(e, "gbl_block_alloc");
    }
    *dev_ptr = d;
    return GBL_OK;
}

int gbl_block_free(void *dev_ptr)
{
    if (!dev_ptr) return GBL_OK;
    hipError_t e = hipFree(dev_ptr);
    if (e != hipSuccess) return hip_fail(e, "gbl_block_free");
    return GBL_OK;
}

int gbl_device_memory(int64_t *free_bytes, int64_t *total_bytes)
{
    size_t f = 0, t = 0;
    hipError_t e = hipMemGetInfo(&f, &t);
    if (e != hipSuccess) return hip_fail(e, "gbl_device_memory");
    if (free_bytes) *free_bytes = (int64_t)f;
    if (total_bytes) *total_bytes = (int64_t)t;
    return GBL_OK;
}

int gbl_board_eval(int8_t *state, const int8_t *agent_index, const int32_t *actions, int8_t *record_out, int64_t n,
                   void *stream)
{
    GBL_CHECK_N(n);
    GBL_NEED(state, "state"); GBL_NEED(record_out, "record_out");
    if (actions) GBL_NEED(agent_index, "agent_index (with actions)");
    GBL_ALIGNED(state, "state"); GBL_ALIGNED(record_out, "record_out");
    if (actions && (reinterpret_cast<uintptr_t>(actions) & 3u)) return fail(GBL_ERR_ALIGN, "actions must be 4-byte aligned");
    Geometry g = geometry(n);
    hipLaunchKernelGGL(k_board_eval, dim3(g.grid), dim3(64), 0, (hipStream_t)stream, state, agent_index, actions,
                       record_out, n, g.ntiles);
    GBL_LAUNCHED("gbl_board_eval");
}

int gbl_step(int8_t *state, int8_t *to_move, int8_t *done, const int32_t *actions, int8_t *winner_out,
             int8_t *reward_out, int8_t *mask_out, int8_t *obs_out, int32_t *turn, int64_t n, int illegal_mode,
             int auto_reset, void *stream)
{
    return gbl_step_ex(state, to_move, done, actions, winner_out, reward_out, mask_out, obs_out, turn, nullptr, nullptr,
                       nullptr, nullptr, nullptr, 0, 0, 0, nullptr, n, illegal_mode, auto_reset, stream);
}

int gbl_step_into(int8_t *state, int8_t *to_move, int8_t *done, const int32_t *actions, int8_t *winner_out,
                  int8_t *reward_out, int8_t *mask_out, int8_t *obs_out, int32_t *turn, int32_t *actions_out,
                  int8_t *done_out, int8_t *to_move_out, int64_t n, int illegal_mode, int auto_reset, void *stream)
{
    return gbl_step_ex(state, to_move, done, actions, winner_out, reward_out, mask_out, obs_out, turn, actions_out, done_out,
                       to_move_out, nullptr, nullptr, 0, 0, 0, nullptr, n, illegal_mode, auto_reset, stream);
}

int gbl_step_ex(int8_t *state, int8_t *to_move, int8_t *done, const int32_t *actions, int8_t *winner_out,
                int8_t *reward_out, int8_t *mask_out, int8_t *obs_out, int32_t *turn, int32_t *actions_out,
                int8_t *done_out, int8_t *to_move_out, int8_t *status_out, int32_t *next_actions_out, uint64_t seed,
                uint64_t env_base, uint32_t ply, const uint32_t *ply_dev, int64_t n, int illegal_mode, int auto_reset,
                void *stream)
{
    GBL_CHECK_N(n);
    GBL_NEED(state, "state"); GBL_NEED(to_move, "to_move"); GBL_NEED(done, "done"); GBL_NEED(actions, "actions");
    if (illegal_mode != GBL_ILLEGAL_NOOP && illegal_mode != GBL_ILLEGAL_TERMINATE)
        return fail(GBL_ERR_ARG, "illegal_mode must be GBL_ILLEGAL_NOOP or GBL_ILLEGAL_TERMINATE");
    GBL_ALIGNED(state, "state"); GBL_ALIGNED(mask_out, "mask_out"); GBL_ALIGNED(obs_out, "obs_out");
    if (reward_out && (reinterpret_cast<uintptr_t>(reward_out) & 1u))
        return fail(GBL_ERR_ALIGN, "reward_out must be 2-byte aligned");
    if (turn && (reinterpret_cast<uintptr_t>(turn) & 3u)) return fail(GBL_ERR_ALIGN, "turn must be 4-byte aligned");
    if (reinterpret_cast<uintptr_t>(actions) & 3u) return fail(GBL_ERR_ALIGN, "actions must be 4-byte aligned");
    if (actions_out && (reinterpret_cast<uintptr_t>(actions_out) & 3u))
        return fail(GBL_ERR_ALIGN, "actions_out must be 4-byte aligned");
    if (next_actions_out && (reinterpret_cast<uintptr_t>(next_actions_out) & 3u))
        return fail(GBL_ERR_ALIGN, "next_actions_out must be 4-byte aligned");
    Geometry g = geometry(n, kStepWaves);
    hipStream_t s = (hipStream_t)stream;
    auto_reset = auto_reset != 0;
    const int nt = nt_policy(n);
    const bool ext = actions_out || done_out || to_move_out || status_out || next_actions_out;
    const StepExt X{actions_out, done_out, to_move_out, status_out, next_actions_out, seed, env_base, ply_dev, ply};
#define GBL_STEP_I(M, O, NT, I)                                                                                          \
    hipLaunchKernelGGL((k_step<M, O, NT, I>), dim3(g.grid), dim3(64 * kStepWaves), 0, s, state, to_move, done, actions, \
                       n, g.ntiles, illegal_mode, auto_reset, winner_out, reward_out, mask_out, obs_out, turn, X)
#define GBL_STEP_NT(M, O, NT)                                   \
    if (ext) GBL_STEP_I(M, O, NT, true);                        \
    else GBL_STEP_I(M, O, NT, false)
#define GBL_STEP(M, O)                                          \
    if (nt == 3) { GBL_STEP_NT(M, O, 3); }                      \
    else { GBL_STEP_NT(M, O, 1); }
    if (mask_out && obs_out) GBL_STEP(true, true)
    else if (mask_out) GBL_STEP(true, false)
    else if (obs_out) GBL_STEP(false, true)
    else GBL_STEP(false, false)
#undef GBL_STEP
#undef GBL_STEP_NT
#undef GBL_STEP_I
    GBL_LAUNCHED("gbl_step");
}

int gbl_sample_at(const int8_t *mask, int32_t *actions, int64_t n, uint64_t seed, uint64_t env_base, uint32_t ply,
                  const uint32_t *ply_dev, void *stream)
{
    GBL_CHECK_N(n);
    GBL_NEED(mask, "mask"); GBL_NEED(actions, "actions");
    GBL_ALIGNED(mask, "mask");
    Geometry g = geometry(n);
    hipLaunchKernelGGL(k_sample, dim3(g.grid), dim3(64), 0, (hipStream_t)stream, mask, actions, n, g.ntiles, seed,
                       env_base, ply, ply_dev);
    GBL_LAUNCHED("gbl_sample");
}

int gbl_sample(const int8_t *mask, int32_t *actions, int64_t n, uint64_t seed, uint64_t env_base, uint32_t ply,
               void *stream)
{
    return gbl_sample_at(mask, actions, n, seed, env_base, ply, nullptr, stream);
}

int gbl_counter_add(uint32_t *counter, uint32_t by, void *stream)
{
    GBL_NEED(counter, "counter");
    hipLaunchKernelGGL(k_counter_add, dim3(1), dim3(1), 0, (hipStream_t)stream, counter, by);
    GBL_LAUNCHED("gbl_counter_add");
}

int gbl_rollout(int8_t *state, int8_t *to_move, int8_t *done, int32_t *actions_out, int8_t *winner_out,
                int8_t *reward_out, int8_t *mask_out, int8_t *obs_out, int64_t n, uint64_t seed, uint64_t env_base,
                uint32_t ply0, uint32_t plies, int illegal_mode, int64_t *counters, int32_t *turn, void *stream)
{
    return gbl_rollout_at(state, to_move, done, actions_out, winner_out, reward_out, mask_out, obs_out, n, seed, env_base,
                          ply0, nullptr, plies, illegal_mode, counters, turn, stream);
}

int gbl_rollout_at(int8_t *state, int8_t *to_move, int8_t *done, int32_t *actions_out, int8_t *winner_out,
                   int8_t *reward_out, int8_t *mask_out, int8_t *obs_out, int64_t n, uint64_t seed, uint64_t env_base,
                   uint32_t ply0, const uint32_t *ply_dev, uint32_t plies, int illegal_mode, int64_t *counters,
                   int32_t *turn, void *stream)
{
    GBL_CHECK_N(n);
    GBL_NEED(state, "state"); GBL_NEED(to_move, "to_move"); GBL_NEED(done, "done");
    if (illegal_mode != GBL_ILLEGAL_NOOP && illegal_mode != GBL_ILLEGAL_TERMINATE)
        return fail(GBL_ERR_ARG, "illegal_mode must be GBL_ILLEGAL_NOOP or GBL_ILLEGAL_TERMINATE");
    if (plies == 0) return GBL_OK;
    GBL_ALIGNED(state, "state"); GBL_ALIGNED(mask_out, "mask_out"); GBL_ALIGNED(obs_out, "obs_out");
    if (reward_out && (reinterpret_cast<uintptr_t>(reward_out) & 1u))
        return fail(GBL_ERR_ALIGN, "reward_out must be 2-byte aligned");
    if (counters && (reinterpret_cast<uintptr_t>(counters) & 127u))
        return fail(GBL_ERR_ALIGN, "counters must be 128-byte aligned");
    if (turn && (reinterpret_cast<uintptr_t>(turn) & 3u)) return fail(GBL_ERR_ALIGN, "turn must be 4-byte aligned");
    if (actions_out && (reinterpret_cast<uintptr_t>(actions_out) & 3u))
        return fail(GBL_ERR_ALIGN, "actions_out must be 4-byte aligned");
    Geometry g = geometry(n, kStepWaves);
    hipStream_t s = (hipStream_t)stream;
    const int nt = nt_policy(n);
#define GBL_ROLL_K(M, O, NT, D, ONE)                                                                                \
    hipLaunchKernelGGL((k_rollout<M, O, NT, D, ONE>), dim3(g.grid), dim3(64 * kStepWaves), 0, s, state, to_move, n, \
                       g.ntiles, seed, env_base, ply_dev, ply0, plies, done, actions_out, winner_out, reward_out,   \
                       mask_out, obs_out, illegal_mode, counters, turn)
#define GBL_ROLL_D(M, O, NT, D)                                 \
    if (plies == 1) GBL_ROLL_K(M, O, NT, D, true);              \
    else GBL_ROLL_K(M, O, NT, D, false)
#define GBL_ROLL_NT(M, O, NT)                                   \
    if (ply_dev) { GBL_ROLL_D(M, O, NT, true); }                \
    else { GBL_ROLL_D(M, O, NT, false); }
#define GBL_ROLL(M, O)                                          \
    if (nt == 3) { GBL_ROLL_NT(M, O, 3); }                      \
    else { GBL_ROLL_NT(M, O, 1); }
    if (mask_out && obs_out) GBL_ROLL(true, true)
    else if (mask_out) GBL_ROLL(true, false)
    else if (obs_out) GBL_ROLL(false, true)
    else GBL_ROLL(false, false)
#undef GBL_ROLL
#undef GBL_ROLL_NT
#undef GBL_ROLL_D
#undef GBL_ROLL_K
    GBL_LAUNCHED("gbl_rollout");  // (also gbl_rollout_at)
}

int gbl_collect(int8_t *state, int8_t *to_move, int8_t *done, int32_t *actions_traj, int8_t *winner_traj,
                int8_t *reward_traj, int8_t *done_traj, int8_t *to_move_traj, int8_t *mask_traj, int8_t *obs_traj,
                int64_t n, int64_t ply_stride, int64_t tile_stride, uint64_t seed, uint64_t env_base, uint32_t ply0,
                const uint32_t *ply_dev, uint32_t plies, int illegal_mode, int64_t *counters, int32_t *turn, void *stream)
{
    return gbl_collect_from(state, to_move, done, nullptr, actions_traj, winner_traj, reward_traj, done_traj, to_move_traj,
                            mask_traj, obs_traj, n, ply_stride, tile_stride, seed, env_base, ply0, ply_dev, plies,
                            illegal_mode, counters, turn, stream);
}

int gbl_collect_from(int8_t *state, int8_t *to_move, int8_t *done, const int32_t *first_actions, int32_t *actions_traj,
                     int8_t *winner_traj, int8_t *reward_traj, int8_t *done_traj, int8_t *to_move_traj, int8_t *mask_traj,
                     int8_t *obs_traj, int64_t n, int64_t ply_stride, int64_t tile_stride, uint64_t seed,
                     uint64_t env_base, uint32_t ply0, const uint32_t *ply_dev, uint32_t plies, int illegal_mode,
                     int64_t *counters, int32_t *turn, void *stream)
{
    return gbl_collect_from_ex(state, to_move, done, first_actions, nullptr, actions_traj, winner_traj, reward_traj, done_traj,
                               to_move_traj, mask_traj, obs_traj, n, ply_stride, tile_stride, seed, env_base, ply0, ply_dev, plies,
                               illegal_mode, counters, turn, stream);
}

int gbl_collect_from_ex(int8_t *state, int8_t *to_move, int8_t *done, const int32_t *first_actions, int8_t *first_status,
                        int32_t *actions_traj, int8_t *winner_traj, int8_t *reward_traj, int8_t *done_traj, int8_t *to_move_traj,
                        int8_t *mask_traj, int8_t *obs_traj, int64_t n, int64_t ply_stride, int64_t tile_stride, uint64_t seed,
                        uint64_t env_base, uint32_t ply0, const uint32_t *ply_dev, uint32_t plies, int illegal_mode,
                        int64_t *counters, int32_t *turn, void *stream)
{
    GBL_CHECK_N(n);
    if (first_actions && (reinterpret_cast<uintptr_t>(first_actions) & 3u))
        return fail(GBL_ERR_ALIGN, "first_actions must be 4-byte aligned");
    if (first_status && !first_actions) return fail(GBL_ERR_ARG, "first_status without first_actions");
    GBL_NEED(state, "state"); GBL_NEED(to_move, "to_move"); GBL_NEED(done, "done");
    if (illegal_mode != GBL_ILLEGAL_NOOP && illegal_mode != GBL_ILLEGAL_TERMINATE)
        return fail(GBL_ERR_ARG, "illegal_mode must be GBL_ILLEGAL_NOOP or GBL_ILLEGAL_TERMINATE");
    if (plies == 0) return GBL_OK;
    {   // the (ply, tile) cells of 64 boards must start 16-byte aligned and must not overlap
        const int64_t tiles = (n + kTile - 1) / kTile;
        const bool aligned = ply_stride > 0 && tile_stride > 0 && !(ply_stride & 15) && !(tile_stride & 15);
        const bool time_major = tile_stride >= kTile && (plies == 1 || ply_stride >= (tiles - 1) * tile_stride + kTile);
        const bool tile_major = ply_stride >= kTile && (tiles == 1 || tile_stride >= ((int64_t)plies - 1) * ply_stride + kTile);
        if (!aligned || !(time_major || tile_major))
            return fail(GBL_ERR_ARG, "ply_stride / tile_stride: multiples of 16 boards that keep the (ply, tile) cells apart");
    }
    GBL_ALIGNED(state, "state"); GBL_ALIGNED(mask_traj, "mask_traj"); GBL_ALIGNED(obs_traj, "obs_traj");
    if (reward_traj && (reinterpret_cast<uintptr_t>(reward_traj) & 1u))
        return fail(GBL_ERR_ALIGN, "reward_traj must be 2-byte aligned");
    if (actions_traj && (reinterpret_cast<uintptr_t>(actions_traj) & 3u))
        return fail(GBL_ERR_ALIGN, "actions_traj must be 4-byte aligned");
    if (turn && (reinterpret_cast<uintptr_t>(turn) & 3u)) return fail(GBL_ERR_ALIGN, "turn must be 4-byte aligned");
    if (counters && (reinterpret_cast<uintptr_t>(counters) & 127u))
        return fail(GBL_ERR_ALIGN, "counters must be 128-byte aligned");
    Geometry g = geometry(n);
    hipStream_t s = (hipStream_t)stream;
    const int variant = collect_variant(n, plies, mask_traj != nullptr, obs_traj != nullptr);
    const bool pair = variant == GBL_COLLECT_PAIR;
    [[maybe_unused]] const bool nt = variant != GBL_COLLECT_CACHED;
    if (variant == GBL_COLLECT_TRIO) {
#define GBL_TRIO_KH(M, O, D, H)                                                                                                   \
    hipLaunchKernelGGL((k_collect3<M, O, D, H>), dim3((uint32_t)g.ntiles), dim3(64 * (1 + (M ? 1 : 0) + (O ? 1 : 0))), 0, s, state,   \
                       to_move, n, g.ntiles, seed, env_base, ply_dev, ply0, plies, done, ply_stride, tile_stride, actions_traj,   \
                       winner_traj, reward_traj, done_traj, to_move_traj, mask_traj, obs_traj, illegal_mode, counters, turn,      \
                       first_actions, first_status)
#define GBL_TRIO_K(M, O, D)                                     \
    if (g.ntiles <= kTrioHandMaxTiles) { GBL_TRIO_KH(M, O, D, true); } \
    else { GBL_TRIO_KH(M, O, D, false); }
#define GBL_TRIO(M, O)                                          \
    if (ply_dev) { GBL_TRIO_K(M, O, true); }                    \
    else { GBL_TRIO_K(M, O, false); }
        if (mask_traj && obs_traj) { GBL_TRIO(true, true); }
        else if (mask_traj) { GBL_TRIO(true, false); }
        else { GBL_TRIO(false, true); }
#undef GBL_TRIO
#undef GBL_TRIO_K
#undef GBL_TRIO_KH
        GBL_LAUNCHED("gbl_collect");
    }
    if (variant == GBL_COLLECT_GROUP32) {
        const int64_t ngroups = (n + kGroupBoards - 1) / kGroupBoards;
#define GBL_G32_K(O, D)                                                                                                            \
    hipLaunchKernelGGL((k_collect5<O, D>), dim3((uint32_t)ngroups), dim3(64 * (O ? 4 : 2)), 0, s, state, to_move, n, ngroups, seed,    \
                       env_base, ply_dev, ply0, plies, done, ply_stride, tile_stride, actions_traj, winner_traj, reward_traj,      \
                       done_traj, to_move_traj, mask_traj, obs_traj, illegal_mode, counters, turn, first_actions, first_status)
        if (obs_traj) {
            if (ply_dev) { GBL_G32_K(true, true); }
            else { GBL_G32_K(true, false); }
        } else {
            if (ply_dev) { GBL_G32_K(false, true); }
            else { GBL_G32_K(false, false); }
        }
#undef GBL_G32_K
        GBL_LAUNCHED("gbl_collect");
    }
    if (GBL_COLLECT_IS_ROLES(variant)) {
        if (!launch_small(variant - GBL_COLLECT_ROLES(0, 0, 0), state, to_move, done, first_actions, first_status, actions_traj, winner_traj, reward_traj,
                          done_traj, to_move_traj, mask_traj, obs_traj, n, ply_stride, tile_stride, seed, env_base, ply0, ply_dev, plies,
                          illegal_mode, counters, turn, s))
            return fail(GBL_ERR_ARG, "gbl_collect: this build has no such form of the role kernel");
        GBL_LAUNCHED("gbl_collect");
    }
    // (the plain-store instantiation of k_collect exists in experiment builds only)
#define GBL_COLLECT_K(M, O, D)                                  \
    if (pair) GBL_COLLECT_K2(M, O, D);                          \
    else GBL_KNOB_COLLECT_STREAM_OR_PLAIN(M, O, D)
#define GBL_COLLECT_K2(M, O, D)                                                                                         \
    hipLaunchKernelGGL((k_collect2<M, O, D>), dim3((uint32_t)g.ntiles), dim3(128), 0, s, state, to_move, n, g.ntiles, seed, \
                       env_base, ply_dev, ply0, plies, done, ply_stride, tile_stride, actions_traj, winner_traj,        \
                       reward_traj, done_traj, to_move_traj, mask_traj, obs_traj, illegal_mode, counters, turn,         \
                       first_actions, first_status)
#define GBL_COLLECT_KN(M, O, D, N)                                                                                      \
    hipLaunchKernelGGL((k_collect<M, O, D, N>), dim3(g.grid), dim3(64), 0, s, state, to_move, n, g.ntiles, seed, env_base, \
                       ply_dev, ply0, plies, done, ply_stride, tile_stride, actions_traj, winner_traj, reward_traj,     \
                       done_traj,                                                                                      \
                       to_move_traj, mask_traj, obs_traj, illegal_mode, counters, turn, first_actions, first_status)
#define GBL_COLLECT(M, O)                                       \
    if (ply_dev) { GBL_COLLECT_K(M, O, true); }                 \
    else { GBL_COLLECT_K(M, O, false); }
    if (mask_traj && obs_traj) { GBL_COLLECT(true, true); }
    else if (mask_traj) { GBL_COLLECT(true, false); }
    else if (obs_traj) { GBL_COLLECT(false, true); }
    else { GBL_COLLECT(false, false); }
#undef GBL_COLLECT
#undef GBL_COLLECT_K
#undef GBL_COLLECT_KN
#undef GBL_COLLECT_K2
    GBL_LAUNCHED("gbl_collect");
}

int gbl_collect_policy(int8_t *state, int8_t *to_move, int8_t *done, int8_t *hist, int32_t *actions_traj,
                       int8_t *winner_traj, int8_t *reward_traj, int8_t *done_traj, int8_t *to_move_traj, int8_t *mask_traj,
                       int8_t *obs_traj, int32_t *chosen_traj, int8_t *how_traj, int8_t *cand_traj, int64_t n,
                       int64_t ply_stride, int64_t tile_stride, uint64_t seed, uint64_t env_base, uint32_t ply0,
                       const uint32_t *ply_dev, uint32_t plies, int policy0, int policy1, int opening_plies,
                       int illegal_mode, int64_t *counters, int32_t *turn, void *stream)
{
    GBL_CHECK_N(n);
    GBL_NEED(state, "state"); GBL_NEED(to_move, "to_move"); GBL_NEED(done, "done");
    if (illegal_mode != GBL_ILLEGAL_NOOP && illegal_mode != GBL_ILLEGAL_TERMINATE)
        return fail(GBL_ERR_ARG, "illegal_mode must be GBL_ILLEGAL_NOOP or GBL_ILLEGAL_TERMINATE");
    if (policy0 < GBL_POLICY_RANDOM || policy0 > GBL_POLICY_GREEDY3 || policy1 < GBL_POLICY_RANDOM || policy1 > GBL_POLICY_GREEDY3)
        return fail(GBL_ERR_ARG, "policy0 / policy1: GBL_POLICY_RANDOM, GBL_POLICY_GREEDY1, _GREEDY2 or _GREEDY3");
    if (opening_plies < 0) return fail(GBL_ERR_ARG, "opening_plies < 0");
    if (opening_plies > 0 && !turn) return fail(GBL_ERR_ARG, "opening_plies > 0 needs the per-board turn counter (turn must not be NULL)");
    if (plies == 0) return GBL_OK;
    {   // the (ply, tile) cells of 64 boards must start 16-byte aligned and must not overlap (as gbl_collect)
        const int64_t tiles = (n + kTile - 1) / kTile;
        const bool aligned = ply_stride > 0 && tile_stride > 0 && !(ply_stride & 15) && !(tile_stride & 15);
        const bool time_major = tile_stride >= kTile && (plies == 1 || ply_stride >= (tiles - 1) * tile_stride + kTile);
        const bool tile_major = ply_stride >= kTile && (tiles == 1 || tile_stride >= ((int64_t)plies - 1) * ply_stride + kTile);
        if (!aligned || !(time_major || tile_major))
            return fail(GBL_ERR_ARG, "ply_stride / tile_stride: multiples of 16 boards that keep the (ply, tile) cells apart");
    }
    GBL_ALIGNED(state, "state"); GBL_ALIGNED(mask_traj, "mask_traj"); GBL_ALIGNED(obs_traj, "obs_traj"); GBL_ALIGNED(cand_traj, "cand_traj");
    if (hist && (reinterpret_cast<uintptr_t>(hist) & 1u)) return fail(GBL_ERR_ALIGN, "hist must be 2-byte aligned");
    if (reward_traj && (reinterpret_cast<uintptr_t>(reward_traj) & 1u))
        return fail(GBL_ERR_ALIGN, "reward_traj must be 2-byte aligned");
    if ((actions_traj && (reinterpret_cast<uintptr_t>(actions_traj) & 3u)) || (chosen_traj && (reinterpret_cast<uintptr_t>(chosen_traj) & 3u)))
        return fail(GBL_ERR_ALIGN, "actions_traj / chosen_traj must be 4-byte aligned");
    if (turn && (reinterpret_cast<uintptr_t>(turn) & 3u)) return fail(GBL_ERR_ALIGN, "turn must be 4-byte aligned");
    if (counters && (reinterpret_cast<uintptr_t>(counters) & 127u))
        return fail(GBL_ERR_ALIGN, "counters must be 128-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const int depth = policy0 > policy1 ? policy0 : policy1;
    const int shape = policy_shape(depth, n);
#define GBL_CP(NT, W)                                                                                                      \
    do {                                                                                                                   \
        const Geometry g = block_geometry(n, NT);                                                                          \
        hipLaunchKernelGGL((k_collect_policy<NT, W>), dim3(g.grid), dim3(64 * W), 0, s, state, to_move, n, g.ntiles, seed,  \
                           env_base, ply_dev, ply0, plies, done, ply_stride, tile_stride, actions_traj, winner_traj,       \
                           reward_traj, done_traj, to_move_traj, mask_traj, obs_traj, chosen_traj, how_traj, cand_traj,    \
                           hist, policy0, policy1, opening_plies, illegal_mode, counters, turn);                           \
    } while (0)
    // (the product build instantiates the shapes policy_shape() can return; an A/B build the one it forces as well)
    if (shape == 56) GBL_CP(4, 16);
    else if (shape == 26) GBL_CP(1, 16);
    else if (shape == 28) GBL_CP(2, 8);
    else if (shape == 14) GBL_CP(1, 4);
    GBL_KNOB_CP_SHAPES
    else GBL_CP(1, 1);
#undef GBL_CP
    GBL_LAUNCHED("gbl_collect_policy");
}

int gbl_collect_variant(int64_t n, uint32_t plies, int with_mask, int with_obs)
{
    if (n < 0) return fail(GBL_ERR_ARG, "n < 0");
    return collect_variant(n, plies, with_mask != 0, with_obs != 0);
}

int gbl_placement_probe(void *a, int64_t a_bytes, void *b, int64_t b_bytes, int64_t slot_boards, int plies, float *us_both,
                        float *us_a, float *us_b, void *stream)
{
    GBL_NEED(a, "a"); GBL_NEED(b, "b"); GBL_NEED(us_both, "us_both"); GBL_NEED(us_a, "us_a"); GBL_NEED(us_b, "us_b");
    if ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 127u)
        return fail(GBL_ERR_ALIGN, "a and b must be 128-byte aligned");
    int kPlies = 4;
    int64_t tiles;
    if (slot_boards > 0) {  // the geometry of a time-major trajectory: `plies` slots of slot_boards boards
        if (plies < 1 || (slot_boards & 127)) return fail(GBL_ERR_ARG, "slot_boards: a multiple of 128 boards; plies >= 1");
        kPlies = plies;
        tiles = slot_boards / kTile;
        if (a_bytes < (int64_t)plies * slot_boards * kObs || b_bytes < (int64_t)plies * slot_boards * kActions)
            return fail(GBL_ERR_ARG, "buffers smaller than plies x slot_boards rows");
    } else {
        // an even number of tiles per slot: every slot of both arrays then starts on a 128-byte line
        tiles = std::min(a_bytes / ((int64_t)kPlies * kTile * kObs), b_bytes / ((int64_t)kPlies * kTile * kActions)) & ~(int64_t)1;
    }
    if (tiles < 2) return fail(GBL_ERR_ARG, "buffers too small to probe (a: 4 x 7488 bytes per tile, b: 4 x 3456)");
    if (tiles > 0x7fffffff) return fail(GBL_ERR_ARG, "buffers too large to probe in one launch");
    hipStream_t s = (hipStream_t)stream;
    int8_t *pa = static_cast<int8_t *>(a), *pb = static_cast<int8_t *>(b);
    auto launch = [&](int which) {
        if (which == 0) hipLaunchKernelGGL((k_probe<true, true>), dim3((uint32_t)tiles), dim3(64), 0, s, pa, pb, tiles, kPlies);
        else if (which == 1) hipLaunchKernelGGL((k_probe<true, false>), dim3((uint32_t)tiles), dim3(64), 0, s, pa, pb, tiles, kPlies);
        else hipLaunchKernelGGL((k_probe<false, true>), dim3((uint32_t)tiles), dim3(64), 0, s, pa, pb, tiles, kPlies);
        return hipGetLastError();
    };
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    hipError_t e = hipSuccess;
    for (int i = 0; i < 4 && e == hipSuccess; ++i) e = hipEventCreate(&ev[i]);
    // a probe often runs on an idle device (the caller has just allocated memory): bring the clocks up first, then
    // time the three variants back to back, several rounds, and keep each one's best
    for (int i = 0; i < 6 && e == hipSuccess; ++i) e = launch(0);
    float best[3] = {0.f, 0.f, 0.f};
    constexpr int kRounds = 4;
    for (int round = 0; round < kRounds && e == hipSuccess; ++round) {
        e = hipEventRecord(ev[0], s);
        for (int which = 1; which <= 3 && e == hipSuccess; ++which) {  // a alone, b alone, both
            e = launch(which % 3);
            if (e == hipSuccess) e = hipEventRecord(ev[which], s);
        }
        if (e == hipSuccess) e = hipEventSynchronize(ev[3]);
        for (int which = 1; which <= 3 && e == hipSuccess; ++which) {
            float ms = 0.f;
            e = hipEventElapsedTime(&ms, ev[which - 1], ev[which]);
            float &slot = best[which % 3];
            if (round == 0 || ms < slot) slot = ms;
        }
    }
    for (int i = 0; i < 4; ++i)
        if (ev[i]) (void)hipEventDestroy(ev[i]);
    if (e != hipSuccess) return hip_fail(e, "gbl_placement_probe");
    *us_both = best[0] * 1e3f;
    *us_a = best[1] * 1e3f;
    *us_b = best[2] * 1e3f;
    return GBL_OK;
}

int gbl_decode_obs(const int8_t *obs, int8_t *state, int8_t *to_move, int64_t n, void *stream)
{
    GBL_CHECK_N(n);
    GBL_NEED(obs, "obs"); GBL_NEED(state, "state"); GBL_NEED(to_move, "to_move");
    GBL_ALIGNED(obs, "obs"); GBL_ALIGNED(state, "state");
    Geometry g = geometry(n);
    hipLaunchKernelGGL(k_decode_obs, dim3(g.grid), dim3(64), 0, (hipStream_t)stream, obs, state, to_move, n, g.ntiles);
    GBL_LAUNCHED("gbl_decode_obs");
}

int gbl_validate(const int8_t *state, int8_t *flags, int64_t n, void *stream)
{
    GBL_CHECK_N(n);
    GBL_NEED(state, "state"); GBL_NEED(flags, "flags");
    GBL_ALIGNED(state, "state");
    Geometry g = geometry(n);
    hipLaunchKernelGGL(k_validate, dim3(g.grid), dim3(64), 0, (hipStream_t)stream, state, flags, n, g.ntiles);
    GBL_LAUNCHED("gbl_validate");
}

namespace {
// The shape of the greedy kernels' workgroups (see GreedyLds), as a code: 11 = <1,1>, 14 = <1,4>, 18 = <1,8>, 26 = <1,16>,
// 28 = <2,8>, 48 = <4,8>, 56 = <4,16>.
// Round 1 (one tile per workgroup; scripts/bench_greedy.py, 65 536 / 262 144 / 2^20 boards): 1 wavefront 49 / 155 / 518 us,
// 2: 43 / 139 / 487, 4: 41 / 121 / 451, 8: 40 / 136 / 520, 16: 50 / 176 / 679.  Round 2 (scripts/ab_greedy.py, in-process):
// eight wavefronts shorten a lone tile's serial chain -- 4 096 / 16 384 boards 11.2 / 11.4 -> 10.0 / 10.2 us -- and lose
// 2-4 % beyond 131 072 boards.  Round 3 (ab_greedy.py / ab_policy_collect.py): what decides is how many
// wavefronts a SIMD has to issue from WHILE THE PAIRS ARE EVALUATED (this code needs four to hide its own LDS round trips
// and dependency chains) against how many sit idle through the owners' phases:
//   gbl_greedy, us (with the two-form pair evaluation, r03 shapes sweep):
//                          4 096   16 384   32 768   49 152   65 536   98 304   131 072   196 608   262 144    2^20
//     <1,4>                  9.4     9.4      11.5     13.0     14.4     20.5      25.8      34.7      43.6     152
//     <1,8>                  8.7     8.9      11.2     13.0     17.3     24.4      31.2      43.3      55.0     201
//     <1,16>                 8.9     9.0      14.8     20.6     26.2       -         -         -         -        -
//     <2,8>                 10.1    10.0      10.2     13.8     13.9     20.4      24.6      35.1      45.5     167
//     <4,8>                 12.7    12.8      12.9     13.1     13.2     24.1      25.1      36.4      48.1     182
//     <4,16>                11.7    11.5      11.5     11.9     12.0     21.8      22.7      32.8      43.1     165
// (<1,16>: one workgroup per CU up to 256 tiles; <4,16>: four owners, four helpers, four wavefronts for the nonplain sets and
// four more that only evaluate -- 16 wavefronts per CU for one GENERATION of tiles, 65 536 boards; a launch of 1.5 generations
// takes as long as one of two, so between generations the smallest workgroups take over, and beyond four they win anyway.)
[[maybe_unused]] static bool whole_generations(int64_t n)  // the last generation of 65 536-board blocks is at least 80 % full
{
    const int64_t gens = (n + 65535) / 65536;
    return n * 5 >= gens * 65536 * 4;
}

int greedy_shape(int depth, int64_t n)
{
    if (knob::kForcedGreedyShape) return depth == 1 ? 11 : knob::kForcedGreedyShape;
    // (round 5, scripts/ab_greedy.py: at 262 144 boards <1,4> 39.8 us against <4,16>'s 40.8 since the depth-1 walk is a closed form)
    return depth == 1 ? 11 : n <= 16384 ? 26 : n <= 32768 ? 28 : n <= 65536 ? 56 : n <= 196608 && whole_generations(n) ? 56 : 14;
}

// ... and of gbl_collect_policy's, whose ply loop keeps more registers live: blocks of tiles only where they stay inlined
// within the register file.  us per self-play ply (r03 shapes sweep):
//                          4 096   16 384   32 768   65 536   131 072   262 144
//     <1,4>                 11.5    11.7      13.8     17.2      32.7      62.4
//     <1,8>                 10.6    10.8      13.3     25.3      49.4      96.7
//     <1,16>                10.6    10.7      21.0     41.8        -         -
//     <2,8>                 11.2    11.0      11.2     22.4      44.2      87.8
//     <4,8>                 15.0    14.9      15.0     15.5      30.5      60.4
//     <4,16> (behind a call) 16.5   16.7      16.7     16.9      33.3      66.2
// Round 4 (scripts/ab_policy_collect.py; the wave index a scalar, the output rows' lane index opaque per ply, the two generator
// blocks parked in LDS: <4,8> 209 -> 132 VGPRs, and <4,16> fits its 128 INLINED without a spill):
//                                            32 768   65 536   131 072   262 144
//     <2,8>  / <4,8>  (round 3's choice)       11.1     15.0      29.9      59.4
//     the same after the register work         10.9     14.7      29.0      57.7
//     <4,16> behind a call                     13.2     13.5      26.6      52.7
//     <4,16> inlined                           12.6     12.9      25.2      49.9
int policy_shape(int depth, int64_t n)
{
    if (knob::kForcedGreedyShape) return depth <= 1 ? 11 : knob::kForcedGreedyShape;
    return depth <= 1 ? 11 : n <= 16384 ? 26 : n <= 32768 ? 28 : n <= 65536 ? 56 : n <= 262144 && whole_generations(n) ? 56 : 14;
}

void launch_greedy(int shape, int64_t n, hipStream_t stream, const int8_t *state, const int8_t *to_move,
                   const int8_t *mask, const int8_t *hist, int depth, int32_t *action_out, int8_t *cand_mask_out,
                   int8_t *fallback_out, int8_t *hist_rw, int32_t *final_out, uint64_t seed,
                   uint64_t env_base, uint32_t call, const uint32_t *call_dev = nullptr)
{
#define GBL_GREEDY(NT, W)                                                                                              \
    do {                                                                                                               \
        const Geometry g = block_geometry(n, NT);                                                                      \
        hipLaunchKernelGGL((k_greedy<NT, W>), dim3(g.grid), dim3(64 * W), 0, stream, state, to_move, mask, hist, depth, \
                           action_out, cand_mask_out, fallback_out, n, g.ntiles, hist_rw, final_out, seed, env_base,   \
                           call, call_dev);                                                                            \
    } while (0)
    // (the product build instantiates the shapes greedy_shape() can return; an A/B build the one it forces as well)
    if (shape == 56) GBL_GREEDY(4, 16);
    else if (shape == 28) GBL_GREEDY(2, 8);
    else if (shape == 26) GBL_GREEDY(1, 16);
    else if (shape == 14) GBL_GREEDY(1, 4);
    GBL_KNOB_GREEDY_SHAPES
    else GBL_GREEDY(1, 1);
#undef GBL_GREEDY
}
}  // namespace

int gbl_greedy(const int8_t *state, const int8_t *to_move, const int8_t *mask, const int8_t *hist, int depth,
               int32_t *action_out, int8_t *cand_mask_out, int8_t *fallback_out, int64_t n, void *stream)
{
    GBL_CHECK_N(n);
    GBL_NEED(state, "state"); GBL_NEED(to_move, "to_move"); GBL_NEED(action_out, "action_out");
    if (depth < 1 || depth > 3) return fail(GBL_ERR_ARG, "depth must be 1, 2 or 3");
    GBL_ALIGNED(state, "state"); GBL_ALIGNED(mask, "mask"); GBL_ALIGNED(cand_mask_out, "cand_mask_out");
    if (hist && (reinterpret_cast<uintptr_t>(hist) & 1u)) return fail(GBL_ERR_ALIGN, "hist must be 2-byte aligned");
    launch_greedy(greedy_shape(depth, n), n, (hipStream_t)stream, state, to_move, mask, hist, depth, action_out,
                  cand_mask_out, fallback_out, nullptr, nullptr, 0, 0, 0);
    GBL_LAUNCHED("gbl_greedy");
}

int gbl_greedy_act(const int8_t *state, const int8_t *to_move, const int8_t *mask, int8_t *hist, int depth,
                   uint64_t seed, uint64_t env_base, uint32_t call, int32_t *action_out, int32_t *chosen_out,
                   int8_t *cand_mask_out, int8_t *fallback_out, int64_t n, void *stream)
{
    return gbl_greedy_act_at(state, to_move, mask, hist, depth, seed, env_base, call, nullptr, action_out, chosen_out,
                             cand_mask_out, fallback_out, n, stream);
}

int gbl_greedy_act_at(const int8_t *state, const int8_t *to_move, const int8_t *mask, int8_t *hist, int depth,
                      uint64_t seed, uint64_t env_base, uint32_t call, const uint32_t *call_dev, int32_t *action_out,
                      int32_t *chosen_out, int8_t *cand_mask_out, int8_t *fallback_out, int64_t n, void *stream)
{
    GBL_CHECK_N(n);
    GBL_NEED(state, "state"); GBL_NEED(to_move, "to_move"); GBL_NEED(hist, "hist"); GBL_NEED(action_out, "action_out");
    if (depth < 1 || depth > 3) return fail(GBL_ERR_ARG, "depth must be 1, 2 or 3");
    GBL_ALIGNED(state, "state"); GBL_ALIGNED(mask, "mask"); GBL_ALIGNED(cand_mask_out, "cand_mask_out");
    if (hist && (reinterpret_cast<uintptr_t>(hist) & 1u)) return fail(GBL_ERR_ALIGN, "hist must be 2-byte aligned");
    launch_greedy(greedy_shape(depth, n), n, (hipStream_t)stream, state, to_move, mask, nullptr, depth, chosen_out,
                  cand_mask_out, fallback_out, hist, action_out, seed, env_base, call, call_dev);
    GBL_LAUNCHED("gbl_greedy_act");
}

}  // extern "C"

GBL_KNOB_EXTRA_ENTRY_POINTS  // (nothing in the product; an experiment build's run-time form switch: gobblet_ab.h)
