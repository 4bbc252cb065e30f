#include <stdlib.h>
#include <stdio.h>
// emu_device.cpp -- TEST INFRASTRUCTURE: compiles gobblet-rl_amd/csrc/gobblet_device.h for the
// HOST, with shims for the handful of AMDGPU builtins it uses, and walks the same
// tile -> LDS image -> row-per-lane -> lane body -> LDS image -> tile sequence the gfx950
// kernels run (64 "lanes" executed one after the other between the points where the kernels
// synchronise).  It lets the CPU test suite check the device code's bit logic and its
// staging index arithmetic against the oracle without a GPU.  It is NOT a product path and
// nothing under gobblet-rl_amd/ references it.
#define GBL_HOST_EMU
#include <stdint.h>
#include <string.h>
#include <type_traits>
#include <vector>

#define __device__
#define __forceinline__ inline
struct uint4 { uint32_t x, y, z, w; };

static inline uint32_t emu_alignbyte(uint32_t hi, uint32_t lo, uint32_t n)
{
    uint64_t v = ((uint64_t)hi << 32) | lo;
    return (uint32_t)(v >> (8 * (n & 3u)));
}
static inline uint32_t emu_udot4(uint32_t a, uint32_t b, uint32_t c, bool)
{
    for (int i = 0; i < 4; ++i) c += ((a >> (8 * i)) & 0xFFu) * ((b >> (8 * i)) & 0xFFu);
    return c;
}
static inline uint32_t emu_umul24(uint32_t a, uint32_t b) { return (uint32_t)((uint64_t)(a & 0xFFFFFFu) * (b & 0xFFFFFFu)); }
static inline uint32_t emu_umulhi(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }
static uint32_t g_next_d0[64];  // what lane l receives from __shfl_down(d[0], 1)
#define __builtin_amdgcn_alignbyte emu_alignbyte
#define __builtin_amdgcn_udot4 emu_udot4
#define __umul24 emu_umul24
#define __umulhi emu_umulhi
#define __popc __builtin_popcount
#define __popcll __builtin_popcountll
#define __shfl_down(v, delta) (g_next_d0[lane])

#include "../../gobblet-rl_amd/csrc/gobblet_device.h"
#include "greedy_root_rule.h"

using namespace gbl;

namespace {

template <int ROWB>
struct Image {
    std::vector<uint32_t> w;
    Image() : w(kTile * ROWB / 4 + 4, 0xDEADBEEFu) {}
    uint32_t *p() { return w.data(); }
};

struct TileCtx {
    int64_t tile;
    int rows;
};

template <int ROWB, int NW>
void stage_all(uint32_t *img, uint32_t (*rows)[NW])
{
    for (int l = 0; l < 64; ++l) g_next_d0[l] = rows[l < 63 ? l + 1 : l][0];
    for (int l = 0; l < 64; ++l) row_stage<ROWB>(img, l, rows[l]);
}

// the kernels' observation path: zero the image, then every lane scatters its bytes
void obs_all(uint32_t *img, const Planes *pl, const int *observer)
{
    for (int l = 0; l < 64; ++l) obs_image_zero(img, l);
    for (int l = 0; l < 64; ++l) obs_scatter(img, l, pl[l], observer[l]);
}

template <int ROWB>
void in_all(const int8_t *g, uint32_t *img, int rows)
{
    for (int l = 0; l < 64; ++l) tile_in<ROWB>(g, img, l, rows);
}

template <int ROWB>
void out_all(int8_t *g, const uint32_t *img, int rows)
{
    for (int l = 0; l < 64; ++l) tile_out<ROWB>(g, img, l, rows);
}

void load_rows(const int8_t *state, const TileCtx &t, uint32_t (*r)[7])
{
    Image<kCells> img;
    in_all<kCells>(state + t.tile * (kTile * kCells), img.p(), t.rows);
    for (int l = 0; l < 64; ++l) {
        row_load<kCells>(img.p(), l, r[l]);
        r[l][6] &= 0x00FFFFFFu;
        if (l >= t.rows)
            for (int j = 0; j < 7; ++j) r[l][j] = 0;
    }
}

// as the step kernels do: the tile's image stays, every lane reads its row from it (all lanes before
// any lane patches its row -- the wave runs in lockstep), and is later stored as it is
void load_rows_keep(const int8_t *state, const TileCtx &t, uint32_t (*r)[7], uint32_t *img)
{
    in_all<kCells>(state + t.tile * (kTile * kCells), img, t.rows);
    for (int l = 0; l < 64; ++l) {
        row_load<kCells>(img, l, r[l]);
        r[l][6] &= 0x00FFFFFFu;
        if (l >= t.rows)
            for (int j = 0; j < 7; ++j) r[l][j] = 0;
    }
}

struct RegRowNone {
    void apply(const MoveCells &) const {}
    void reset() const {}
};

template <typename F>
void for_tiles(int64_t n, F f)
{
    int64_t ntiles = (n + kTile - 1) / kTile;
    int64_t chunk = (ntiles + 7) / 8;
    for (uint32_t bid = 0; bid < (uint32_t)(chunk * 8); ++bid) {  // same block -> tile map as the kernels
        int64_t tile = xcd_tile(bid, ntiles);
        if (tile >= ntiles) continue;
        int64_t left = n - tile * kTile;
        f(TileCtx{tile, left < kTile ? (int)left : kTile});
    }
}

}  // namespace

extern "C" {

// k_collect5's write-back: state rows -> planes -> rows (planes_to_row), through the kernels' load / stage path
void emu_planes_roundtrip(const int8_t *state, int8_t *out, int64_t n)
{
    for_tiles(n, [&](TileCtx t) {
        uint32_t r[64][7], d[64][7];
        load_rows(state, t, r);
        for (int l = 0; l < 64; ++l) {
            Planes p = make_planes(r[l]);
            if (l >= t.rows) p.nz = 0;  // (lanes past a ragged tile's rows hold empty boards; neg / odd keep whatever the image held)
            planes_to_row(p, d[l]);
        }
        Image<kCells> img;
        stage_all<kCells, 7>(img.p(), d);
        out_all<kCells>(out + t.tile * (kTile * kCells), img.p(), t.rows);
    });
}

void emu_legal_mask(const int8_t *state, const int8_t *to_move, int8_t *mask, int64_t n)
{
    for_tiles(n, [&](TileCtx t) {
        uint32_t r[64][7], d[64][14];
        load_rows(state, t, r);
        for (int l = 0; l < 64; ++l) {
            int mover = l < t.rows ? to_move[t.tile * 64 + l] : 0;
            mask_row(legal54(make_planes(r[l]), mover != 0), d[l]);
        }
        Image<kActions> img;
        stage_all<kActions, 14>(img.p(), d);
        out_all<kActions>(mask + t.tile * (kTile * kActions), img.p(), t.rows);
    });
}

void emu_is_legal(const int8_t *state, const int8_t *agent, const int32_t *actions, int8_t *out, int64_t n)
{
    for_tiles(n, [&](TileCtx t) {
        uint32_t r[64][7];
        load_rows(state, t, r);
        for (int l = 0; l < t.rows; ++l) {
            int64_t b = t.tile * 64 + l;
            int a = actions[b];
            uint64_t m = legal54(make_planes(r[l]), agent[b] != 0);
            out[b] = ((uint32_t)a < 54u && ((m >> (a & 63)) & 1ull)) ? 1 : 0;
        }
    });
}

void emu_play_turn(int8_t *state, const int8_t *agent, const int32_t *actions, int64_t n)
{
    for_tiles(n, [&](TileCtx t) {
        uint32_t r[64][7];
        load_rows(state, t, r);
        for (int l = 0; l < 64; ++l) {
            bool valid = l < t.rows;
            int64_t b = t.tile * 64 + l;
            Planes p = make_planes(r[l]);
            int a = valid ? actions[b] : 0;
            int mover = valid ? (agent[b] != 0) : 0;
            uint64_t m = legal54(p, mover);
            if (valid && (uint32_t)a < 54u && ((m >> (a & 63)) & 1ull)) apply_move(p, r[l], mover, (uint32_t)a);
        }
        Image<kCells> img;
        stage_all<kCells, 7>(img.p(), r);
        out_all<kCells>(state + t.tile * (kTile * kCells), img.p(), t.rows);
    });
}

void emu_winner(const int8_t *state, int8_t *winner, int64_t n)
{
    for_tiles(n, [&](TileCtx t) {
        uint32_t r[64][7];
        load_rows(state, t, r);
        for (int l = 0; l < t.rows; ++l) winner[t.tile * 64 + l] = (int8_t)winner_of(make_planes(r[l]));
    });
}

void emu_validate(const int8_t *state, int8_t *flags, int64_t n)
{
    for_tiles(n, [&](TileCtx t) {
        uint32_t r[64][7];
        load_rows(state, t, r);
        for (int l = 0; l < t.rows; ++l) flags[t.tile * 64 + l] = (int8_t)validate_row(r[l]);
    });
}

void emu_flatboard(const int8_t *state, int8_t *flat, int64_t n)
{
    for_tiles(n, [&](TileCtx t) {
        uint32_t r[64][7], d[64][3];
        load_rows(state, t, r);
        for (int l = 0; l < 64; ++l) flat_row(make_planes(r[l]), r[l], d[l]);
        Image<9> img;
        stage_all<9, 3>(img.p(), d);
        out_all<9>(flat + t.tile * (kTile * 9), img.p(), t.rows);
    });
}

void emu_covered(const int8_t *state, int8_t *cov, int64_t n)
{
    for_tiles(n, [&](TileCtx t) {
        uint32_t r[64][7], d[64][7];
        load_rows(state, t, r);
        for (int l = 0; l < 64; ++l) covered_row(make_planes(r[l]), d[l]);
        Image<kCells> img;
        stage_all<kCells, 7>(img.p(), d);
        out_all<kCells>(cov + t.tile * (kTile * kCells), img.p(), t.rows);
    });
}

void emu_observe(const int8_t *state, const int8_t *to_move, int agent_sel, int8_t *obs, int64_t n)
{
    for_tiles(n, [&](TileCtx t) {
        uint32_t r[64][7];
        Planes pl[64];
        int who[64];
        load_rows(state, t, r);
        for (int l = 0; l < 64; ++l) {
            who[l] = (agent_sel >= 0 ? agent_sel : (l < t.rows ? to_move[t.tile * 64 + l] : 0)) != 0;
            pl[l] = make_planes(r[l]);
        }
        Image<kObs> img;
        obs_all(img.p(), pl, who);
        out_all<kObs>(obs + t.tile * (kTile * kObs), img.p(), t.rows);
    });
}

void emu_step(int8_t *state, int8_t *to_move, int8_t *done, const int32_t *actions, int8_t *winner_out,
              int8_t *reward_out, int8_t *mask_out, int8_t *obs_out, int64_t n, int illegal_mode, int auto_reset)
{
    for_tiles(n, [&](TileCtx t) {
        uint32_t r[64][7], dm[64][14];
        Planes pl[64];
        int who[64];
        Image<kCells> is;
        load_rows_keep(state, t, r, is.p());
        for (int l = 0; l < 64; ++l) {
            bool valid = l < t.rows;
            int64_t b = t.tile * 64 + l;
            int mover = 0, was_done = 0, action = 0;
            if (valid) {
                mover = to_move[b] != 0;
                was_done = auto_reset ? 0 : (done[b] != 0);
                action = actions[b];
            }
            Planes p = make_planes(r[l]);
            Ply y;
            int dn;
            step_lane(ImageRow{reinterpret_cast<uint8_t *>(is.p()) + l * kCells}, p, mover, was_done, action,
                      illegal_mode, auto_reset, dn, y);
            mask_row(next_mask(p, mover, dn, auto_reset), dm[l]);
            pl[l] = p; who[l] = mover;
            if (valid) {
                to_move[b] = (int8_t)mover;
                done[b] = (int8_t)dn;
                if (winner_out) winner_out[b] = (int8_t)y.winner;
                if (reward_out) { reward_out[2 * b] = (int8_t)y.r0; reward_out[2 * b + 1] = (int8_t)y.r1; }
            }
        }
        out_all<kCells>(state + t.tile * (kTile * kCells), is.p(), t.rows);
        if (mask_out) {
            Image<kActions> im;
            stage_all<kActions, 14>(im.p(), dm);
            out_all<kActions>(mask_out + t.tile * (kTile * kActions), im.p(), t.rows);
        }
        if (obs_out) {
            Image<kObs> io;
            obs_all(io.p(), pl, who);
            out_all<kObs>(obs_out + t.tile * (kTile * kObs), io.p(), t.rows);
        }
    });
}

void emu_rollout(int8_t *state, int8_t *to_move, int8_t *done, int32_t *actions_out, int8_t *winner_out,
                 int8_t *reward_out, int8_t *mask_out, int8_t *obs_out, int64_t n, uint64_t seed, uint64_t env_base,
                 uint32_t ply0, uint32_t plies, int illegal_mode, int64_t *counters)
{
    for_tiles(n, [&](TileCtx t) {
        uint32_t r[64][7], dm[64][14];
        Planes pl[64];
        int who[64];
        Image<kCells> is;
        load_rows_keep(state, t, r, is.p());
        for (int l = 0; l < 64; ++l) {
            bool valid = l < t.rows;
            int64_t b = t.tile * 64 + l;
            int mover = valid ? (to_move[b] != 0) : 0;
            Planes p = make_planes(r[l]);
            Ply y{0, 0, 0, false, false};
            int dn = 0, action = -1;
            for (uint32_t k = 0; k < plies; ++k) {
                uint64_t legal = legal54(p, mover);
                action = sample54(legal, seed, env_base + (uint64_t)b, ply0 + k);
                step_lane(ImageRow{reinterpret_cast<uint8_t *>(is.p()) + l * kCells}, p, mover, 0, action, illegal_mode,
                          1, dn, y);
                if (valid && counters) {
                    counters[0] += 1;
                    counters[1] += y.terminal;
                    counters[2] += y.winner == 1;
                    counters[3] += y.winner == -1;
                }
            }
            mask_row(legal54(p, mover), dm[l]);
            pl[l] = p; who[l] = mover;
            if (valid) {
                to_move[b] = (int8_t)mover;
                done[b] = (int8_t)dn;
                if (actions_out) actions_out[b] = action;
                if (winner_out) winner_out[b] = (int8_t)y.winner;
                if (reward_out) { reward_out[2 * b] = (int8_t)y.r0; reward_out[2 * b + 1] = (int8_t)y.r1; }
            }
        }
        out_all<kCells>(state + t.tile * (kTile * kCells), is.p(), t.rows);
        if (mask_out) {
            Image<kActions> im;
            stage_all<kActions, 14>(im.p(), dm);
            out_all<kActions>(mask_out + t.tile * (kTile * kActions), im.p(), t.rows);
        }
        if (obs_out) {
            Image<kObs> io;
            obs_all(io.p(), pl, who);
            out_all<kObs>(obs_out + t.tile * (kTile * kObs), io.p(), t.rows);
        }
    });
}

// The role kernel's walk (k_collect_small<.., LPB>, gobblet_hip.hip): a sub-tile of 64 / LPB boards per wavefront, lane = LPB *
// board + j; the LPB lanes of a board play the game alike and share the row work -- lane j writes bytes [64 j / LPB, 64 (j + 1) /
// LPB) of the board's mask row (mask_row_part) and drops channels j, j + LPB, ... of its observation row (obs_scatter_part).
// `plies` plies, the outputs of the last one stored (what gbl_collect's last slot holds), all three roles folded into one walk.
}  // extern "C"
template <int LPB>
static void rollout_sub(int8_t *state, int8_t *to_move, int8_t *done, int32_t *actions_out, int8_t *winner_out,
                        int8_t *reward_out, int8_t *mask_out, int8_t *obs_out, int64_t n, uint64_t seed, uint64_t env_base,
                        uint32_t ply0, uint32_t plies, int illegal_mode)
{
    constexpr int BPS = kTile / LPB, SH = LPB == 4 ? 2 : LPB == 2 ? 1 : 0;
    const int64_t nsub = (n + BPS - 1) / BPS;
    for (int64_t sub = 0; sub < nsub; ++sub) {
        const int64_t left = n - sub * BPS;
        const int rows = left < BPS ? (int)left : BPS;
        std::vector<uint32_t> is(BPS * kCells / 4 + 4, 0xDEADBEEFu), im(BPS * kActions / 4 + 4, 0xDEADBEEFu),
            io(BPS * kObs / 4 + 4, 0xDEADBEEFu);
        for (int lane = 0; lane < 64; ++lane) sub_in<kCells, BPS>(state + sub * (BPS * kCells), is.data(), lane, rows);
        Planes P[64];
        int MOVER[64];
        uint64_t LEGAL[64];
        uint32_t R[64][7];
        int MOVER0[64];
        for (int lane = 0; lane < 64; ++lane) {  // all lanes read their rows and movers before any lane patches or stores (lockstep)
            row_load<kCells>(is.data(), lane >> SH, R[lane]);
            R[lane][6] &= 0x00FFFFFFu;
            MOVER0[lane] = (lane >> SH) < rows ? (to_move[sub * BPS + (lane >> SH)] != 0) : 0;
        }
        for (int lane = 0; lane < 64; ++lane) {  // every lane of a board has read the same row and plays the same game
            const int bq = lane >> SH, j = lane & (LPB - 1);
            const bool valid = bq < rows;
            const int64_t b = sub * BPS + bq;
            Planes p = make_planes(R[lane]);
            p.nz = valid ? p.nz : 0u;
            int mover = MOVER0[lane];
            Ply y{0, 0, 0, false, false};
            int dn = 0, action = -1;
            uint64_t legal = legal54(p, mover);
            for (uint32_t k = 0; k < plies; ++k) {
                action = sample54(legal, seed, env_base + (uint64_t)b, ply0 + k);
                // (only lane 0 of a board patches the state image, as role 0's lanes write the same bytes)
                if (j == 0)
                    y = play_ply(p, ImageRow{reinterpret_cast<uint8_t *>(is.data()) + bq * kCells}, mover, legal, action, illegal_mode);
                else
                    y = play_ply(p, RegRowNone{}, mover, legal, action, illegal_mode);
                dn = y.terminal ? 1 : 0;
                if (y.terminal) {
                    p = Planes{0u, 0u, 0u};
                    mover = 0;
                    if (j == 0) ImageRow{reinterpret_cast<uint8_t *>(is.data()) + bq * kCells}.reset();
                }
                legal = legal54(p, mover);
            }
            P[lane] = p; MOVER[lane] = mover; LEGAL[lane] = legal;
            if (valid && j == 0) {
                to_move[b] = (int8_t)mover;
                done[b] = (int8_t)dn;
                if (actions_out) actions_out[b] = action;
                if (winner_out) winner_out[b] = (int8_t)y.winner;
                if (reward_out) { reward_out[2 * b] = (int8_t)y.r0; reward_out[2 * b + 1] = (int8_t)y.r1; }
            }
        }
        for (int lane = 0; lane < 64; ++lane) sub_out<kCells, kStorePlain, BPS>(state + sub * (BPS * kCells), is.data(), lane, rows);
        // the output images leave through the registers of sub_fetch / sub_store (full sub-tiles) or byte by byte (the ragged last one)
        auto image_out = [&](auto rowb, int8_t *dst, const uint32_t *img) {
            constexpr int ROWB = decltype(rowb)::value;
            if (rows == BPS) {
                for (int lane = 0; lane < 64; ++lane) {
                    SubVecs<sub_vectors<ROWB, BPS>()> v{};
                    sub_fetch<ROWB, BPS>(img, lane, v);
                    sub_store<ROWB, kStorePlain, BPS>(dst, v, lane);
                }
            } else {
                memcpy(dst, img, (size_t)rows * ROWB);
            }
        };
        if (mask_out) {
            for (int lane = 0; lane < 64; ++lane)
                mask_row_part<LPB>(reinterpret_cast<uint8_t *>(im.data()) + (lane >> SH) * kActions, LEGAL[lane], lane & (LPB - 1));
            image_out(std::integral_constant<int, kActions>{}, mask_out + sub * (BPS * kActions), im.data());
        }
        if (obs_out) {
            for (int lane = 0; lane < 64; ++lane) sub_obs_zero<BPS>(io.data(), lane);
            for (int lane = 0; lane < 64; ++lane)
                obs_scatter_part<LPB>(reinterpret_cast<uint8_t *>(io.data()) + (lane >> SH) * kObs, P[lane], MOVER[lane], lane & (LPB - 1));
            image_out(std::integral_constant<int, kObs>{}, obs_out + sub * (BPS * kObs), io.data());
        }
    }
}

extern "C" {
void emu_rollout_small(int8_t *state, int8_t *to_move, int8_t *done, int32_t *actions_out, int8_t *winner_out,
                       int8_t *reward_out, int8_t *mask_out, int8_t *obs_out, int64_t n, uint64_t seed, uint64_t env_base,
                       uint32_t ply0, uint32_t plies, int illegal_mode, int lpb)
{
    if (lpb == 4)
        rollout_sub<4>(state, to_move, done, actions_out, winner_out, reward_out, mask_out, obs_out, n, seed, env_base, ply0, plies, illegal_mode);
    else if (lpb == 2)
        rollout_sub<2>(state, to_move, done, actions_out, winner_out, reward_out, mask_out, obs_out, n, seed, env_base, ply0, plies, illegal_mode);
    else
        rollout_sub<1>(state, to_move, done, actions_out, winner_out, reward_out, mask_out, obs_out, n, seed, env_base, ply0, plies, illegal_mode);
}

// risky_from_have (threat squares) against the walk over the eight lines it replaced, for all 512 sets: mismatches
int emu_risky_mismatches(void)
{
    constexpr uint32_t L[8] = {0x007u, 0x038u, 0x1C0u, 0x049u, 0x092u, 0x124u, 0x111u, 0x054u};  // board.py:135-153
    int bad = 0;
    for (uint32_t T = 0; T < 512; ++T) {
        uint32_t risky = 0;
        bool full = false;
        for (int l = 0; l < 8; ++l) {
            const uint32_t miss = L[l] & ~T;
            full = full || miss == 0;
            if ((miss & (miss - 1)) == 0) risky |= miss;  // exactly one square missing: that square completes the line
        }
        bad += risky_from_have(T) != (full ? 0x1FFu : risky);
    }
    return bad;
}

void emu_sample(const int8_t *mask, int32_t *actions, int64_t n, uint64_t seed, uint64_t env_base, uint32_t ply)
{
    for_tiles(n, [&](TileCtx t) {
        Image<kActions> img;
        in_all<kActions>(mask + t.tile * (kTile * kActions), img.p(), t.rows);
        for (int l = 0; l < t.rows; ++l) {
            uint32_t d[14];
            row_load<kActions>(img.p(), l, d);
            int64_t b = t.tile * 64 + l;
            actions[b] = sample54(mask_bits(d), seed, env_base + (uint64_t)b, ply);
        }
    });
}

void emu_decode_obs(const int8_t *obs, int8_t *state, int8_t *to_move, int64_t n)
{
    for_tiles(n, [&](TileCtx t) {
        Image<kObs> img;
        in_all<kObs>(obs + t.tile * (kTile * kObs), img.p(), t.rows);
        uint32_t r[64][7];
        for (int l = 0; l < 64; ++l) {
            uint32_t d[30];
            row_load<kObs>(img.p(), l, d);
            int agent = decode_obs_row(d, r[l]);
            if (l < t.rows) to_move[t.tile * 64 + l] = (int8_t)agent;
        }
        Image<kCells> is;
        stage_all<kCells, 7>(is.p(), r);
        out_all<kCells>(state + t.tile * (kTile * kCells), is.p(), t.rows);
    });
}

// statistics of the pooled flow: pairs evaluated, pairs deferred to the exact evaluation, and pairs whose
// cheap evaluation differs from the exact one (a bug if ever non-zero)
static int64_t g_pairs = 0, g_deferred = 0, g_fast_mismatch = 0, g_held = 0, g_items = 0;
// The rule of greedy_root_rule.h against the exact evaluation, on every candidate it would settle.
// out: boards, candidates settled from the root, of them with a winning reply, placements sent to the exact evaluation
// (risky squares), MISMATCHES (summary or candidate-set bits differ from greedy_reply: must be 0)
void emu_greedy_root_rule(const int8_t *state, const int8_t *to_move, const int8_t *mask_in, int64_t n, int64_t *out)
{
    for (int k = 0; k < 5; ++k) out[k] = 0;
    for (int64_t b = 0; b < n; ++b) {
        uint32_t r[7] = {0, 0, 0, 0, 0, 0, 0};
        memcpy(r, state + b * kCells, kCells);
        const Planes p = make_planes(r);
        const int me = to_move[b] != 0;
        uint64_t mask = legal54(p, me);
        if (mask_in) {
            mask = 0;
            for (int a = 0; a < kActions; ++a) mask |= (uint64_t)(mask_in[b * kActions + a] != 0) << a;
        }
        const GreedyHead h = greedy_head(p, me, mask, 2);
        const GreedyRoot g = greedy_root(p, me);
        const uint64_t w0 = h.todo & ~h.dup, hand = w0 & greedy_from_hand(p, me), resolved = hand & ~spread9(g.risky);
        const GreedyHandSets hs = greedy_hand_sets(p, me, g.replies, h.legal_me);
        out[0]++;
        out[3] += __builtin_popcountll(hand & ~resolved);
        for (uint64_t it = resolved; it; it &= it - 1) {
            const uint32_t a = (uint32_t)__builtin_ctzll(it);
            const uint32_t want = greedy_reply(p, me, h.legal_me, a);
            const uint32_t got = greedy_hand_summary(p, me, g.replies, h.legal_me, a);
            const bool fl = (want & 1u) && ((h.legal_me >> ((want >> 1) & 63u)) & 1ull);
            out[1]++;
            out[2] += want & 1u;
            if (want != got || ((hs.threat >> a) & 1ull) != (want & 1u) || ((hs.second >> a) & 1ull) != ((want >> 7) & 1u) ||
                ((hs.block >> a) & 1ull) != ((want >> 8) & 1u) || ((hs.flegal >> a) & 1ull) != (fl ? 1u : 0u))
                out[4]++;
        }
    }
}

// The virtual-root rule (greedy_root_rule.h) against the exact evaluation, on every move of a placed piece it would settle.
// out: boards, pairs the kernel evaluates today, of them moves of placed pieces, virtual roots used (pieces with a settled
// candidate; at most `cap` per board, largest piece first), candidates they settle, candidates left to the evaluation, items
// they deal out, MISMATCHES (must be 0)
void emu_greedy_vroot_rule(const int8_t *state, const int8_t *to_move, const int8_t *mask_in, int64_t n, int cap, int64_t *out)
{
    for (int k = 0; k < 8; ++k) out[k] = 0;
    for (int64_t b = 0; b < n; ++b) {
        uint32_t r[7] = {0, 0, 0, 0, 0, 0, 0};
        memcpy(r, state + b * kCells, kCells);
        const Planes p = make_planes(r);
        const int me = to_move[b] != 0;
        uint64_t mask = legal54(p, me);
        if (mask_in) {
            mask = 0;
            for (int a = 0; a < kActions; ++a) mask |= (uint64_t)(mask_in[b * kActions + a] != 0) << a;
        }
        const GreedyHead h = greedy_head(p, me, mask, 2);
        const GreedyRootPlan plan = greedy_root_plan(h, p, me, greedy_root(p, me));
        const uint64_t placed_moves = plan.eval & ~greedy_from_hand(p, me);
        out[0]++;
        out[1] += __builtin_popcountll(plan.eval);
        out[2] += __builtin_popcountll(placed_moves);
        int used = 0;
        for (int pi = 5; pi >= 0; --pi) {
            const uint32_t cand = (uint32_t)(placed_moves >> (9 * pi)) & 0x1FFu;
            if (!cand) continue;
            const Planes v = greedy_lifted(p, me, (uint32_t)pi);
            const GreedyRoot g = greedy_root(v, me);
            const int nr = __builtin_popcountll(g.replies);
            const uint32_t settled = cand & ~g.risky;
            if (used >= cap || nr > kRootItems || !settled) {
                out[5] += __builtin_popcount(cand);
                continue;
            }
            ++used;
            out[3]++;
            out[4] += __builtin_popcount(settled);
            out[5] += __builtin_popcount(cand & g.risky);
            out[6] += nr;
            uint64_t und[kRootItems] = {0, 0, 0, 0, 0, 0};
            for (int j = 0; j < nr; ++j) und[j] = greedy_undefused(v, me, kth_bit64(g.replies, (uint32_t)j));
            const GreedyHandSets hs = greedy_hand_merge(g.replies, h.legal_me, und);
            for (uint32_t it = settled; it; it &= it - 1) {
                const uint32_t a = 9u * (uint32_t)pi + (uint32_t)__builtin_ctz(it);
                const uint32_t want = greedy_reply(p, me, h.legal_me, a);
                const uint32_t got = nr ? greedy_hand_lookup(g.replies, h.legal_me, und, a) : 0u;
                const bool fl = (want & 1u) && ((h.legal_me >> ((want >> 1) & 63u)) & 1ull);
                if (want != got || ((hs.threat >> a) & 1ull) != (want & 1u) || ((hs.second >> a) & 1ull) != ((want >> 7) & 1u) ||
                    ((hs.block >> a) & 1ull) != ((want >> 8) & 1u) || ((hs.flegal >> a) & 1ull) != (fl ? 1u : 0u))
                    out[7]++;
            }
        }
    }
}

void emu_greedy_stats(int64_t *out)
{
    out[0] = g_pairs;
    out[1] = g_deferred;
    out[2] = g_fast_mismatch;
    out[3] = g_held;
    out[4] = g_items;
}

// pooled != 0: the kernel's flow (candidate lists of a tile back to back, every pair evaluated by
// "lane" g % 64, threat / allwin sets, greedy_replay_sets); pooled == 0: greedy_decide per board.
void emu_greedy(const int8_t *state, const int8_t *to_move, const int8_t *mask_in, const int8_t *hist, int depth,
                int32_t *action_out, int8_t *cand_out, int8_t *fallback_out, int64_t n, int pooled, int8_t *hist_rw,
                int32_t *final_out, uint64_t seed, uint64_t env_base, uint32_t call)
{
    if (hist_rw) hist = hist_rw;  // gbl_greedy_act: history read from and appended to hist_rw
    for_tiles(n, [&](TileCtx t) {
        uint32_t r[64][7], dc[64][14];
        Planes P[64];
        int ME[64];
        uint32_t PREV[64];
        uint64_t MASK[64];
        GreedyHead H[64];
        load_rows(state, t, r);
        Image<kActions> im;
        if (mask_in) in_all<kActions>(mask_in + t.tile * (kTile * kActions), im.p(), t.rows);
        for (int l = 0; l < 64; ++l) {
            bool valid = l < t.rows;
            int64_t b = t.tile * 64 + l;
            Planes p = make_planes(r[l]);
            int me = valid ? (to_move[b] != 0) : 0;
            uint64_t mask;
            if (mask_in) {
                uint32_t d[14];
                row_load<kActions>(im.p(), l, d);
                mask = mask_bits(d);
            } else
                mask = legal54(p, me);
            if (!valid) mask = 0;
            uint32_t prev3 = 0x00FFFFFFu;
            if (hist && valid) {
                const int8_t *h = hist + (b * 2 + me) * 3;
                prev3 = (uint32_t)(uint8_t)h[0] | ((uint32_t)(uint8_t)h[1] << 8) | ((uint32_t)(uint8_t)h[2] << 16);
            }
            P[l] = p;
            ME[l] = me;
            PREV[l] = prev3;
            MASK[l] = mask;
            H[l] = greedy_head(p, me, mask, depth);
        }
        static uint16_t pair[64 * kActions], reply[64][kActions];
        uint64_t threat[64] = {0}, allwin[64] = {0}, second[64] = {0}, block[64] = {0}, flegal[64] = {0};
        // the kernel's flow (greedy_tile): the candidates split by greedy_root_plan -- placements from hand on non-risky
        // squares are settled from the root's replies (table rows written by "item lanes", merged by the owner), all
        // others are evaluated exactly
        GreedyRootPlan PLAN[64];
        static uint64_t undef[64][kRootItems];
        int total = 0;
        for (int l = 0; l < 64; ++l) {
            PLAN[l] = GreedyRootPlan{0ull, 0ull, 0ull};
            for (int j = 0; j < kRootItems; ++j) undef[l][j] = 0xDEADBEEFDEADBEEFull;  // rows nobody writes: poison (see the merge below)
            if (pooled && depth > 1 && MASK[l] != 0) PLAN[l] = greedy_root_plan(H[l], P[l], ME[l], greedy_root(P[l], ME[l]));
            for (uint64_t it = PLAN[l].eval; it; it &= it - 1) pair[total++] = (uint16_t)((l << 8) | __builtin_ctzll(it));
            const int nr = __builtin_popcountll(PLAN[l].items);
            for (int j = 0; j < nr; ++j) {  // the items (board, j)
                undef[l][j] = greedy_item_row(P[l], ME[l], H[l].legal_me, kth_bit64(PLAN[l].items, (uint32_t)j));  // (tagged, as the kernel's item lanes leave them)
                g_items++;
            }
        }
        auto record = [&](uint32_t o, uint32_t a, uint32_t sum) {
            if (sum & 1u) {
                reply[o][a] = (uint16_t)sum;
                threat[o] |= 1ull << a;
                if (sum & (1u << 7)) second[o] |= 1ull << a;
                if (sum & (1u << 8)) block[o] |= 1ull << a;
                if ((H[o].legal_me >> ((sum >> 1) & 63u)) & 1ull) flegal[o] |= 1ull << a;
            }
            if (sum >> 15) allwin[o] |= 1ull << a;
        };
        for (int g = 0; g < total; ++g) {
            const uint32_t o = pair[g] >> 8, a = pair[g] & 0xFFu;
            // the kernel's choice: the fast form of the pair evaluation (threat squares) where the board's greedy_nonplain set
            // says the pair is plain -- the set's bit checked against reply_is_plain on the moved board, the fast form against
            // the ordered line steps, on every pair --, the ordered form elsewhere
            const bool plain = !((greedy_nonplain(P[o], ME[o]) >> a) & 1ull);
            if (plain != greedy_pair_is_plain(P[o], ME[o], a)) g_fast_mismatch++;
            const uint32_t sum = plain ? greedy_reply<true>(P[o], ME[o], H[o].legal_me, a) : greedy_reply<false>(P[o], ME[o], H[o].legal_me, a);
            g_pairs++;
            if (!plain) g_deferred++;
            if (plain && sum != greedy_reply<false>(P[o], ME[o], H[o].legal_me, a)) g_fast_mismatch++;
            record(o, a, sum);
        }
        uint64_t UND[64][kRootItems];
        for (int l = 0; l < 64; ++l) {
            for (int j = 0; j < kRootItems; ++j) UND[l][j] = 0ull;
            if (!PLAN[l].items) continue;
            // exactly what the kernel does (gobblet_hip.hip, greedy_tile's tail): ALL kRootItems rows, `& resolved` only -- the rows
            // nobody wrote hold the poison above (the kernel: stale LDS), and only the `live` guards of greedy_hand_merge /
            // greedy_hand_lookup keep them out of the result: a regression there shows up as a mismatch below
            for (int j = 0; j < kRootItems; ++j) UND[l][j] = undef[l][j];
            const int nr = __builtin_popcountll(PLAN[l].items);
            const GreedyHandSets hs = greedy_hand_merge_tagged(nr, PLAN[l].resolved, UND[l]);
            {   // the tagged merge against the plain one (greedy_hand_merge on rows without tags, live rows only)
                uint64_t plain[kRootItems];
                for (int j = 0; j < kRootItems; ++j) plain[j] = (j < nr ? undef[l][j] & PLAN[l].resolved : 0xDEADBEEFDEADBEEFull);
                const GreedyHandSets hp = greedy_hand_merge(PLAN[l].items, H[l].legal_me, plain);
                if (hp.threat != hs.threat || hp.second != hs.second || hp.block != hs.block || hp.flegal != hs.flegal) g_fast_mismatch++;
            }
            threat[l] |= hs.threat; second[l] |= hs.second; block[l] |= hs.block; flegal[l] |= hs.flegal;
        }
        // THE RULE ITSELF, on every candidate it settles: the summary looked up in the table and the merged set bits must be
        // what the exact evaluation gives (a settled candidate of a board with no replies to deal out: summary 0)
        for (int l = 0; l < 64; ++l)
            for (uint64_t it = PLAN[l].resolved; it; it &= it - 1) {
                const uint32_t a = (uint32_t)__builtin_ctzll(it);
                const uint32_t want = greedy_reply(P[l], ME[l], H[l].legal_me, a);
                const uint32_t got = PLAN[l].items ? greedy_hand_lookup_tagged(__builtin_popcountll(PLAN[l].items), H[l].legal_me, UND[l], a) : 0u;
                const bool fl = (want & 1u) && ((H[l].legal_me >> ((want >> 1) & 63u)) & 1ull);
                g_held++;
                if (want != got || ((threat[l] >> a) & 1ull) != (want & 1u) || ((second[l] >> a) & 1ull) != ((want >> 7) & 1u) ||
                    ((block[l] >> a) & 1ull) != ((want >> 8) & 1u) || ((flegal[l] >> a) & 1ull) != (fl ? 1u : 0u) || ((allwin[l] >> a) & 1ull))
                    g_fast_mismatch++;
            }
        for (int l = 0; l < 64; ++l) {
            bool valid = l < t.rows;
            int64_t b = t.tile * 64 + l;
            GreedyResult g;
            if (pooled) {
                auto reply_of = [&](int a) {
                    const uint32_t twin = ((H[l].dup >> a) & 1ull) ? (uint32_t)a - 9u : (uint32_t)a;
                    return ((PLAN[l].resolved >> twin) & 1ull) ? greedy_hand_lookup_tagged(__builtin_popcountll(PLAN[l].items), H[l].legal_me, UND[l], twin)
                                                               : (uint32_t)reply[l][twin];
                };
                GreedyHead seq = H[l];  // the loop form and the closed form must agree on everything they leave behind
                greedy_replay_sets(seq, threat[l], allwin[l], reply_of);
                greedy_replay_closed(H[l], ReplySets{threat[l], allwin[l], second[l], block[l], flegal[l]}, reply_of);
                if (seq.chosen != H[l].chosen || seq.cands != H[l].cands || seq.ncands != H[l].ncands) g_fast_mismatch++;
                g = greedy_finish(H[l], PREV[l]);
            } else {
                g = greedy_decide(P[l], ME[l], MASK[l], depth, PREV[l]);
            }
            mask_row(g.cands, dc[l]);
            if (valid) {
                if (action_out) action_out[b] = g.fallback ? -1 : g.chosen;
                if (fallback_out) fallback_out[b] = g.fallback ? 1 : 0;
                if (hist_rw) {
                    int fin = g.fallback ? pick54(g.cands, draw32(seed, env_base + (uint64_t)b, call, kStreamGreedy)) : g.chosen;
                    final_out[b] = fin;
                    int8_t *hp = hist_rw + (b * 2 + ME[l]) * 3;
                    hp[0] = (int8_t)(PREV[l] >> 8);
                    hp[1] = (int8_t)(PREV[l] >> 16);
                    hp[2] = (int8_t)fin;
                }
            }
        }
        if (cand_out) {
            Image<kActions> ic;
            stage_all<kActions, 14>(ic.p(), dc);
            out_all<kActions>(cand_out + t.tile * (kTile * kActions), ic.p(), t.rows);
        }
    });
}

}  // extern "C"
