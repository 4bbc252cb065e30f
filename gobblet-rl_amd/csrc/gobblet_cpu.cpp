// gobblet_cpu.cpp -- the HOST flavour of the C-ABI (include/gobblet_hip.h): gbl_cpu_* with the signatures of the device entry
// points (the `stream` argument is ignored), for callers without a GPU -- BASELINE config 1 (one environment behind
// gobblet_v1.env() "on CPU"), a reference maintainer's own tests, and the CPU twin bench.py times beside the GPU (SURVEY.md 8b /
// 8d(ii)).  It is NOT a fallback: nothing in the package routes a GPU call here; a caller asks for device="cpu" explicitly.
//
// The game logic is the DEVICE code itself: this file compiles csrc/gobblet_device.h for the host (GBL_HOST_EMU selects the
// header's host paths; the handful of AMDGPU builtins it uses are given plain C++ meanings below) and walks the boards one by one
// -- row in, bit planes, the same lane functions the kernels call (legal54, play_ply, winner_of, obs_scatter_row, pick54 on the
// same Philox words, greedy_decide ...), row out.  So the twin is bit-identical to the HIP path by construction of the shared
// header and is parity-tested against the oracle like it (tests/test_cpu_twin.py).  It never includes, links or calls anything
// under oracle/.  Boards are dealt over std::threads (gbl_cpu_set_threads; default: the hardware's).
#define GBL_HOST_EMU
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <thread>
#include <vector>

#define __device__
#define __forceinline__ inline
struct uint4 {
    uint32_t x, y, z, w;
};

static inline uint32_t host_alignbyte(uint32_t hi, uint32_t lo, uint32_t n)
{
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> (8 * (n & 3u)));
}
static inline uint32_t host_udot4(uint32_t a, uint32_t b, uint32_t c, bool)
{
    for (int i = 0; i < 4; ++i) c += ((a >> (8 * i)) & 0xFFu) * ((b >> (8 * i)) & 0xFFu);
    return c;
}
static inline uint32_t host_umul24(uint32_t a, uint32_t b) { return (uint32_t)((uint64_t)(a & 0xFFFFFFu) * (b & 0xFFFFFFu)); }
static inline uint32_t host_umulhi(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }
#define __builtin_amdgcn_alignbyte host_alignbyte
#define __builtin_amdgcn_udot4 host_udot4
#define __umul24 host_umul24
#define __umulhi host_umulhi
#define __popc __builtin_popcount
#define __popcll __builtin_popcountll
#define __shfl_down(v, delta) (0u)  // (row_stage's neighbour fetch: the host flavour moves rows with memcpy and never stages)

#include "gobblet_device.h"

#include "../../include/gobblet_cpu.h"

using namespace gbl;

namespace {

thread_local char g_err[256] = "";
std::atomic<int> g_threads{0};

int fail(int code, const char *msg)
{
    snprintf(g_err, sizeof g_err, "%s", msg);
    return code;
}

int thread_count(int64_t n)
{
    int t = g_threads.load();
    if (t <= 0) t = (int)std::thread::hardware_concurrency();
    if (t <= 0) t = 1;
    const int64_t by_work = (n + 2047) / 2048;  // a thread is worth starting for a few thousand boards
    return (int)std::max<int64_t>(1, std::min<int64_t>(t, by_work));
}

// f(b0, b1) over [0, n) in contiguous ranges, one per thread
template <typename F>
void parallel_for(int64_t n, F f)
{
    const int t = thread_count(n);
    if (t == 1) {
        f((int64_t)0, n);
        return;
    }
    std::vector<std::thread> pool;
    const int64_t per = (n + t - 1) / t;
    for (int i = 0; i < t; ++i) {
        const int64_t b0 = i * per, b1 = std::min(n, b0 + per);
        if (b0 < b1) pool.emplace_back([=] { f(b0, b1); });
    }
    for (auto &th : pool) th.join();
}

// a board's 27 state bytes as the seven little-endian dwords the device code works on
inline void load_row(const int8_t *state, int64_t b, uint32_t (&r)[7])
{
    r[6] = 0;
    memcpy(r, state + b * kCells, kCells);
}
inline void store_row(int8_t *state, int64_t b, const uint32_t (&r)[7]) { memcpy(state + b * kCells, r, kCells); }

struct HostRow {  // play_ply's row: the 27 bytes where they live
    uint8_t *row;
    void apply(const MoveCells &m) const
    {
        row[m.had ? m.cold : m.cnew] = 0;
        row[m.cnew] = (uint8_t)m.val;
    }
    void reset() const { memset(row, 0, kCells); }
};

inline void write_mask(int8_t *dst, uint64_t m)
{
    uint32_t d[14];
    mask_row(m, d);
    memcpy(dst, d, kActions);
}

inline void write_obs(int8_t *dst, const Planes &p, int observer)
{
    memset(dst, 0, kObs);
    obs_scatter_row(reinterpret_cast<uint8_t *>(dst), p, observer);
}

inline uint64_t read_mask(const int8_t *src)
{
    uint64_t m = 0;
    for (int a = 0; a < kActions; ++a) m |= (uint64_t)(src[a] != 0) << a;
    return m;
}

struct Tally {
    unsigned long long plies = 0, games = 0, w1 = 0, w2 = 0;
};

void add_tally(int64_t *counters, const Tally &t)  // (stripe 0: the totals are sums over the stripes)
{
    if (!counters) return;
    auto *c = reinterpret_cast<std::atomic<unsigned long long> *>(counters);
    c[0] += t.plies; c[1] += t.games; c[2] += t.w1; c[3] += t.w2;
}

bool strides_ok(int64_t n, uint32_t plies, int64_t ply_stride, int64_t tile_stride)
{
    const int64_t tiles = (n + kTile - 1) / kTile;
    const bool aligned = ply_stride > 0 && tile_stride > 0 && !(ply_stride & 15) && !(tile_stride & 15);
    const bool time_major = tile_stride >= kTile && (plies == 1 || ply_stride >= (tiles - 1) * tile_stride + kTile);
    const bool tile_major = ply_stride >= kTile && (tiles == 1 || tile_stride >= ((int64_t)plies - 1) * ply_stride + kTile);
    return aligned && (time_major || tile_major);
}

inline int64_t cell_of(int64_t b, uint32_t t, int64_t ply_stride, int64_t tile_stride)
{
    return (int64_t)t * ply_stride + (b / kTile) * tile_stride + (b % kTile);
}

inline uint32_t hist_prev3(const int8_t *hist, int64_t b, int me)
{
    if (!hist) return 0x00FFFFFFu;
    const uint8_t *h = reinterpret_cast<const uint8_t *>(hist) + (b * 2 + me) * 3;
    return (uint32_t)h[0] | ((uint32_t)h[1] << 8) | ((uint32_t)h[2] << 16);
}

}  // namespace

#define GBL_CHECK_N(n)                                      \
    do {                                                    \
        if ((n) < 0) return fail(GBL_ERR_ARG, "n < 0");     \
        if ((n) == 0) return GBL_OK;                        \
    } while (0)
#define GBL_NEED(p, name)                                                 \
    do {                                                                  \
        if (!(p)) return fail(GBL_ERR_ARG, name " must not be NULL");     \
    } while (0)

extern "C" {

const char *gbl_cpu_last_error(void) { return g_err; }

int gbl_cpu_set_threads(int threads)  // 0 = the hardware's (host flavour only)
{
    if (threads < 0) return fail(GBL_ERR_ARG, "threads < 0");
    g_threads.store(threads);
    return GBL_OK;
}

int gbl_cpu_layout_info(int32_t out[6])
{
    if (!out) return fail(GBL_ERR_ARG, "out must not be NULL");
    out[0] = GBL_ABI_VERSION; out[1] = kCells; out[2] = kActions; out[3] = kObs; out[4] = kTile; out[5] = 1;  // (no alignment asked of host buffers)
    return GBL_OK;
}

int gbl_cpu_reset(int8_t *state, int8_t *to_move, int8_t *done, int8_t *winner, int64_t n, void *)
{
    GBL_CHECK_N(n);
    GBL_NEED(state, "state"); GBL_NEED(to_move, "to_move"); GBL_NEED(done, "done");
    memset(state, 0, (size_t)n * kCells);
    memset(to_move, 0, (size_t)n);
    memset(done, 0, (size_t)n);
    if (winner) memset(winner, 0, (size_t)n);
    return GBL_OK;
}

int gbl_cpu_legal_mask(const int8_t *state, const int8_t *to_move, int8_t *mask, int64_t n, void *)
{
    GBL_CHECK_N(n);
    GBL_NEED(state, "state"); GBL_NEED(to_move, "to_move"); GBL_NEED(mask, "mask");
    parallel_for(n, [=](int64_t b0, int64_t b1) {
        for (int64_t b = b0; b < b1; ++b) {
            uint32_t r[7];
            load_row(state, b, r);
            write_mask(mask + b * kActions, legal54(make_planes(r), to_move[b] != 0));
        }
    });
    return GBL_OK;
}

int gbl_cpu_is_legal(const int8_t *state, const int8_t *agent_index, const int32_t *actions, int8_t *out, int64_t n, void *)
{
    GBL_CHECK_N(n);
    GBL_NEED(state, "state"); GBL_NEED(agent_index, "agent_index"); GBL_NEED(actions, "actions"); GBL_NEED(out, "out");
    for (int64_t b = 0; b < n; ++b) {
        uint32_t r[7];
        load_row(state, b, r);
        const int a = actions[b];
        const uint64_t m = legal54(make_planes(r), agent_index[b] != 0);
        out[b] = ((uint32_t)a < (uint32_t)kActions && ((m >> (a & 63)) & 1ull)) ? 1 : 0;
    }
    return GBL_OK;
}

int gbl_cpu_play_turn(int8_t *state, const int8_t *agent_index, const int32_t *actions, int64_t n, void *)
{
    GBL_CHECK_N(n);
    GBL_NEED(state, "state"); GBL_NEED(agent_index, "agent_index"); GBL_NEED(actions, "actions");
    for (int64_t b = 0; b < n; ++b) {
        uint32_t r[7];
        load_row(state, b, r);
        Planes p = make_planes(r);
        const int a = actions[b], mover = agent_index[b] != 0;
        const uint64_t m = legal54(p, mover);
        if ((uint32_t)a < (uint32_t)kActions && ((m >> (a & 63)) & 1ull)) {
            apply_move(p, r, mover, (uint32_t)a);
            store_row(state, b, r);
        }
    }
    return GBL_OK;
}

int gbl_cpu_winner(const int8_t *state, int8_t *winner, int64_t n, void *)
{
    GBL_CHECK_N(n);
    GBL_NEED(state, "state"); GBL_NEED(winner, "winner");
    for (int64_t b = 0; b < n; ++b) {
        uint32_t r[7];
        load_row(state, b, r);
        winner[b] = (int8_t)winner_of(make_planes(r));
    }
    return GBL_OK;
}

int gbl_cpu_flatboard(const int8_t *state, int8_t *flat, int64_t n, void *)
{
    GBL_CHECK_N(n);
    GBL_NEED(state, "state"); GBL_NEED(flat, "flat");
    for (int64_t b = 0; b < n; ++b) {
        uint32_t r[7], d[3];
        load_row(state, b, r);
        flat_row(make_planes(r), r, d);
        memcpy(flat + b * 9, d, 9);
    }
    return GBL_OK;
}

int gbl_cpu_covered(const int8_t *state, int8_t *cov, int64_t n, void *)
{
    GBL_CHECK_N(n);
    GBL_NEED(state, "state"); GBL_NEED(cov, "cov");
    for (int64_t b = 0; b < n; ++b) {
        uint32_t r[7], d[7];
        load_row(state, b, r);
        covered_row(make_planes(r), d);
        memcpy(cov + b * kCells, d, kCells);
    }
    return GBL_OK;
}

int gbl_cpu_validate(const int8_t *state, int8_t *flags, int64_t n, void *)
{
    GBL_CHECK_N(n);
    GBL_NEED(state, "state"); GBL_NEED(flags, "flags");
    for (int64_t b = 0; b < n; ++b) {
        uint32_t r[7];
        load_row(state, b, r);
        flags[b] = (int8_t)validate_row(r);
    }
    return GBL_OK;
}

int gbl_cpu_observe(const int8_t *state, const int8_t *to_move, int agent_sel, int8_t *obs, int64_t n, void *)
{
    GBL_CHECK_N(n);
    GBL_NEED(state, "state"); GBL_NEED(obs, "obs");
    if (agent_sel < -1 || agent_sel > 1) return fail(GBL_ERR_ARG, "agent_sel must be -1, 0 or 1");
    if (agent_sel < 0) GBL_NEED(to_move, "to_move (agent_sel == -1)");
    parallel_for(n, [=](int64_t b0, int64_t b1) {
        for (int64_t b = b0; b < b1; ++b) {
            uint32_t r[7];
            load_row(state, b, r);
            const int who = agent_sel >= 0 ? agent_sel : to_move[b];
            write_obs(obs + b * kObs, make_planes(r), who != 0);
        }
    });
    return GBL_OK;
}

int gbl_cpu_board_eval(int8_t *state, const int8_t *agent_index, const int32_t *actions, int8_t *record_out, int64_t n, void *)
{
    GBL_CHECK_N(n);
    GBL_NEED(state, "state"); GBL_NEED(record_out, "record_out");
    if (actions) GBL_NEED(agent_index, "agent_index (with actions)");
    for (int64_t b = 0; b < n; ++b) {
        uint32_t r[7];
        load_row(state, b, r);
        Planes p = make_planes(r);
        if (actions) {  // board.py:118-132
            const int a = actions[b], mover = agent_index[b] != 0;
            const uint64_t m = legal54(p, mover);
            if ((uint32_t)a < (uint32_t)kActions && ((m >> (a & 63)) & 1ull)) apply_move(p, r, mover, (uint32_t)a);
            store_row(state, b, r);
        }
        int8_t *rec = record_out + b * GBL_REC_BYTES;
        memset(rec, 0, GBL_REC_BYTES);
        memcpy(rec + GBL_REC_SQUARES, r, kCells);
        rec[GBL_REC_WINNER] = (int8_t)winner_of(p);
        uint32_t f[3], c[7];
        flat_row(p, r, f);
        covered_row(p, c);
        memcpy(rec + GBL_REC_FLAT, f, 9);
        memcpy(rec + GBL_REC_COVERED, c, kCells);
        write_mask(rec + GBL_REC_MASK0, legal54(p, 0));
        write_mask(rec + GBL_REC_MASK1, legal54(p, 1));
        obs_scatter_row(reinterpret_cast<uint8_t *>(rec) + GBL_REC_OBS0, p, 0);
        obs_scatter_row(reinterpret_cast<uint8_t *>(rec) + GBL_REC_OBS1, p, 1);
    }
    return GBL_OK;
}

int gbl_cpu_step_ex(int8_t *state, int8_t *to_move, int8_t *done, const int32_t *actions, int8_t *winner_out, int8_t *reward_out,
                    int8_t *mask_out, int8_t *obs_out, int32_t *turn, int32_t *actions_out, int8_t *done_out, int8_t *to_move_out,
                    int8_t *status_out, int32_t *next_actions_out, uint64_t seed, uint64_t env_base, uint32_t ply,
                    const uint32_t *ply_dev, int64_t n, int illegal_mode, int auto_reset, void *)
{
    GBL_CHECK_N(n);
    GBL_NEED(state, "state"); GBL_NEED(to_move, "to_move"); GBL_NEED(done, "done"); GBL_NEED(actions, "actions");
    if (illegal_mode != GBL_ILLEGAL_NOOP && illegal_mode != GBL_ILLEGAL_TERMINATE)
        return fail(GBL_ERR_ARG, "illegal_mode must be GBL_ILLEGAL_NOOP or GBL_ILLEGAL_TERMINATE");
    auto_reset = auto_reset != 0;
    if (ply_dev) ply += *ply_dev;
    parallel_for(n, [=](int64_t b0, int64_t b1) {
        for (int64_t b = b0; b < b1; ++b) {
            uint32_t r[7];
            load_row(state, b, r);
            Planes p = make_planes(r);
            int mover = to_move[b] != 0;
            const int was_done = !auto_reset && done[b] != 0, action = actions[b];
            Ply y;
            int dn;
            step_lane(HostRow{reinterpret_cast<uint8_t *>(state) + b * kCells}, p, mover, was_done, action, illegal_mode, auto_reset, dn, y);
            to_move[b] = (int8_t)mover;
            done[b] = (int8_t)dn;
            if (winner_out) winner_out[b] = (int8_t)y.winner;
            if (reward_out) { reward_out[2 * b] = (int8_t)y.r0; reward_out[2 * b + 1] = (int8_t)y.r1; }
            if (turn) turn[b] = next_turn(turn[b], y, auto_reset);
            if (obs_out) write_obs(obs_out + b * kObs, p, mover);
            const uint64_t legal = next_mask(p, mover, dn, auto_reset);
            if (mask_out) write_mask(mask_out + b * kActions, legal);
            if (actions_out) actions_out[b] = action;
            if (done_out) done_out[b] = (int8_t)dn;
            if (to_move_out) to_move_out[b] = (int8_t)mover;
            if (status_out) status_out[b] = (int8_t)(was_done ? 0 : action_status(y.ok, action));  // (a frozen board consumes no action)
            if (next_actions_out) next_actions_out[b] = sample54(legal, seed, env_base + (uint64_t)b, ply);  // (may alias `actions`)
        }
    });
    return GBL_OK;
}

int gbl_cpu_step_into(int8_t *state, int8_t *to_move, int8_t *done, const int32_t *actions, int8_t *winner_out, int8_t *reward_out,
                      int8_t *mask_out, int8_t *obs_out, int32_t *turn, int32_t *actions_out, int8_t *done_out,
                      int8_t *to_move_out, int64_t n, int illegal_mode, int auto_reset, void *stream)
{
    return gbl_cpu_step_ex(state, to_move, done, actions, winner_out, reward_out, mask_out, obs_out, turn, actions_out, done_out,
                           to_move_out, nullptr, nullptr, 0, 0, 0, nullptr, n, illegal_mode, auto_reset, stream);
}

int gbl_cpu_step(int8_t *state, int8_t *to_move, int8_t *done, const int32_t *actions, int8_t *winner_out, int8_t *reward_out,
                 int8_t *mask_out, int8_t *obs_out, int32_t *turn, int64_t n, int illegal_mode, int auto_reset, void *stream)
{
    return gbl_cpu_step_ex(state, to_move, done, actions, winner_out, reward_out, mask_out, obs_out, turn, nullptr, nullptr, nullptr,
                           nullptr, nullptr, 0, 0, 0, nullptr, n, illegal_mode, auto_reset, stream);
}

int gbl_cpu_sample_at(const int8_t *mask, int32_t *actions, int64_t n, uint64_t seed, uint64_t env_base, uint32_t ply,
                      const uint32_t *ply_dev, void *)
{
    GBL_CHECK_N(n);
    GBL_NEED(mask, "mask"); GBL_NEED(actions, "actions");
    if (ply_dev) ply += *ply_dev;
    parallel_for(n, [=](int64_t b0, int64_t b1) {
        for (int64_t b = b0; b < b1; ++b) actions[b] = sample54(read_mask(mask + b * kActions), seed, env_base + (uint64_t)b, ply);
    });
    return GBL_OK;
}

int gbl_cpu_sample(const int8_t *mask, int32_t *actions, int64_t n, uint64_t seed, uint64_t env_base, uint32_t ply, void *stream)
{
    return gbl_cpu_sample_at(mask, actions, n, seed, env_base, ply, nullptr, stream);
}

int gbl_cpu_counter_add(uint32_t *counter, uint32_t by, void *)
{
    GBL_NEED(counter, "counter");
    *counter += by;
    return GBL_OK;
}

// the masked-random ply of one board (sample -> step -> auto-reset), shared by rollout / collect
static inline Ply random_ply(Planes &p, HostRow row, int &mover, uint64_t &legal, int &action, int given, bool use_given, uint64_t seed,
                             uint64_t id, uint32_t ply, int illegal_mode, int &dn)
{
    action = use_given ? given : sample54(legal, seed, id, ply);
    Ply y = play_ply(p, row, mover, legal, action, illegal_mode);
    dn = y.terminal ? 1 : 0;
    if (y.terminal) {  // raw_env.reset, gobblet.py:275-290
        p = Planes{0u, 0u, 0u};
        mover = 0;
        row.reset();
    }
    legal = legal54(p, mover);
    return y;
}

int gbl_cpu_rollout_at(int8_t *state, int8_t *to_move, int8_t *done, int32_t *actions_out, int8_t *winner_out, int8_t *reward_out,
                       int8_t *mask_out, int8_t *obs_out, int64_t n, uint64_t seed, uint64_t env_base, uint32_t ply0,
                       const uint32_t *ply_dev, uint32_t plies, int illegal_mode, int64_t *counters, int32_t *turn, void *)
{
    GBL_CHECK_N(n);
    GBL_NEED(state, "state"); GBL_NEED(to_move, "to_move"); GBL_NEED(done, "done");
    if (illegal_mode != GBL_ILLEGAL_NOOP && illegal_mode != GBL_ILLEGAL_TERMINATE)
        return fail(GBL_ERR_ARG, "illegal_mode must be GBL_ILLEGAL_NOOP or GBL_ILLEGAL_TERMINATE");
    if (plies == 0) return GBL_OK;
    if (ply_dev) ply0 += *ply_dev;
    parallel_for(n, [=](int64_t b0, int64_t b1) {
        Tally tl;
        for (int64_t b = b0; b < b1; ++b) {
            uint32_t r[7];
            load_row(state, b, r);
            Planes p = make_planes(r);
            const HostRow row{reinterpret_cast<uint8_t *>(state) + b * kCells};
            int mover = to_move[b] != 0, dn = 0, action = -1, tcount = 0;
            bool treset = false;
            uint64_t legal = legal54(p, mover);
            Ply y{0, 0, 0, false, false};
            for (uint32_t t = 0; t < plies; ++t) {
                y = random_ply(p, row, mover, legal, action, 0, false, seed, env_base + (uint64_t)b, ply0 + t, illegal_mode, dn);
                tcount = next_turn(tcount, y, 1);
                treset = treset || y.terminal;
                tl.games += y.terminal; tl.w1 += y.winner == 1; tl.w2 += y.winner == -1;
            }
            tl.plies += plies;
            to_move[b] = (int8_t)mover;
            done[b] = (int8_t)dn;
            if (actions_out) actions_out[b] = action;
            if (winner_out) winner_out[b] = (int8_t)y.winner;
            if (reward_out) { reward_out[2 * b] = (int8_t)y.r0; reward_out[2 * b + 1] = (int8_t)y.r1; }
            if (turn) turn[b] = treset ? tcount : turn[b] + tcount;
            if (obs_out) write_obs(obs_out + b * kObs, p, mover);
            if (mask_out) write_mask(mask_out + b * kActions, legal);
        }
        add_tally(counters, tl);
    });
    return GBL_OK;
}

int gbl_cpu_rollout(int8_t *state, int8_t *to_move, int8_t *done, int32_t *actions_out, int8_t *winner_out, int8_t *reward_out,
                    int8_t *mask_out, int8_t *obs_out, int64_t n, uint64_t seed, uint64_t env_base, uint32_t ply0, uint32_t plies,
                    int illegal_mode, int64_t *counters, int32_t *turn, void *stream)
{
    return gbl_cpu_rollout_at(state, to_move, done, actions_out, winner_out, reward_out, mask_out, obs_out, n, seed, env_base, ply0,
                              nullptr, plies, illegal_mode, counters, turn, stream);
}

int gbl_cpu_collect_from(int8_t *state, int8_t *to_move, int8_t *done, const int32_t *first_actions, int32_t *actions_traj,
                         int8_t *winner_traj, int8_t *reward_traj, int8_t *done_traj, int8_t *to_move_traj, int8_t *mask_traj,
                         int8_t *obs_traj, int64_t n, int64_t ply_stride, int64_t tile_stride, uint64_t seed, uint64_t env_base,
                         uint32_t ply0, const uint32_t *ply_dev, uint32_t plies, int illegal_mode, int64_t *counters, int32_t *turn,
                         void *stream)
{
    return gbl_cpu_collect_from_ex(state, to_move, done, first_actions, nullptr, actions_traj, winner_traj, reward_traj, done_traj,
                                   to_move_traj, mask_traj, obs_traj, n, ply_stride, tile_stride, seed, env_base, ply0, ply_dev, plies,
                                   illegal_mode, counters, turn, stream);
}

int gbl_cpu_collect_from_ex(int8_t *state, int8_t *to_move, int8_t *done, const int32_t *first_actions, int8_t *first_status,
                            int32_t *actions_traj, int8_t *winner_traj, int8_t *reward_traj, int8_t *done_traj, int8_t *to_move_traj,
                            int8_t *mask_traj, int8_t *obs_traj, int64_t n, int64_t ply_stride, int64_t tile_stride, uint64_t seed,
                            uint64_t env_base, uint32_t ply0, const uint32_t *ply_dev, uint32_t plies, int illegal_mode,
                            int64_t *counters, int32_t *turn, void *)
{
    GBL_CHECK_N(n);
    if (first_status && !first_actions) return fail(GBL_ERR_ARG, "first_status without first_actions");
    GBL_NEED(state, "state"); GBL_NEED(to_move, "to_move"); GBL_NEED(done, "done");
    if (illegal_mode != GBL_ILLEGAL_NOOP && illegal_mode != GBL_ILLEGAL_TERMINATE)
        return fail(GBL_ERR_ARG, "illegal_mode must be GBL_ILLEGAL_NOOP or GBL_ILLEGAL_TERMINATE");
    if (plies == 0) return GBL_OK;
    if (!strides_ok(n, plies, ply_stride, tile_stride))
        return fail(GBL_ERR_ARG, "ply_stride / tile_stride: multiples of 16 boards that keep the (ply, tile) cells apart");
    if (ply_dev) ply0 += *ply_dev;
    parallel_for(n, [=](int64_t b0, int64_t b1) {
        Tally tl;
        for (int64_t b = b0; b < b1; ++b) {
            uint32_t r[7];
            load_row(state, b, r);
            Planes p = make_planes(r);
            const HostRow row{reinterpret_cast<uint8_t *>(state) + b * kCells};
            int mover = to_move[b] != 0, dn = 0, action = -1, tcount = 0;
            bool treset = false;
            uint64_t legal = legal54(p, mover);
            for (uint32_t t = 0; t < plies; ++t) {
                const bool given = first_actions && t == 0;
                const Ply y = random_ply(p, row, mover, legal, action, given ? first_actions[b] : 0, given, seed, env_base + (uint64_t)b,
                                         ply0 + t, illegal_mode, dn);
                if (given && first_status) first_status[b] = (int8_t)action_status(y.ok, action);
                tcount = next_turn(tcount, y, 1);
                treset = treset || y.terminal;
                tl.games += y.terminal; tl.w1 += y.winner == 1; tl.w2 += y.winner == -1;
                const int64_t at = cell_of(b, t, ply_stride, tile_stride);
                if (actions_traj) actions_traj[at] = action;
                if (winner_traj) winner_traj[at] = (int8_t)y.winner;
                if (reward_traj) { reward_traj[2 * at] = (int8_t)y.r0; reward_traj[2 * at + 1] = (int8_t)y.r1; }
                if (done_traj) done_traj[at] = (int8_t)dn;
                if (to_move_traj) to_move_traj[at] = (int8_t)mover;
                if (obs_traj) write_obs(obs_traj + at * kObs, p, mover);
                if (mask_traj) write_mask(mask_traj + at * kActions, legal);
            }
            tl.plies += plies;
            to_move[b] = (int8_t)mover;
            done[b] = (int8_t)dn;
            if (turn) turn[b] = treset ? tcount : turn[b] + tcount;
        }
        add_tally(counters, tl);
    });
    return GBL_OK;
}

int gbl_cpu_collect(int8_t *state, int8_t *to_move, int8_t *done, int32_t *actions_traj, int8_t *winner_traj, int8_t *reward_traj,
                    int8_t *done_traj, int8_t *to_move_traj, int8_t *mask_traj, int8_t *obs_traj, int64_t n, int64_t ply_stride,
                    int64_t tile_stride, uint64_t seed, uint64_t env_base, uint32_t ply0, const uint32_t *ply_dev, uint32_t plies,
                    int illegal_mode, int64_t *counters, int32_t *turn, void *stream)
{
    return gbl_cpu_collect_from(state, to_move, done, nullptr, actions_traj, winner_traj, reward_traj, done_traj, to_move_traj, mask_traj,
                                obs_traj, n, ply_stride, tile_stride, seed, env_base, ply0, ply_dev, plies, illegal_mode, counters, turn,
                                stream);
}

int gbl_cpu_decode_obs(const int8_t *obs, int8_t *state, int8_t *to_move, int64_t n, void *)
{
    GBL_CHECK_N(n);
    GBL_NEED(obs, "obs"); GBL_NEED(state, "state"); GBL_NEED(to_move, "to_move");
    for (int64_t b = 0; b < n; ++b) {
        uint32_t d[30], r[7];
        d[29] = 0;
        memcpy(d, obs + b * kObs, kObs);
        to_move[b] = (int8_t)decode_obs_row(d, r);
        store_row(state, b, r);
    }
    return GBL_OK;
}

// one decision: chosen-before-fallback / candidate set / fallback flag (gbl_greedy), and with hist_rw the returned action and the
// history append (gbl_greedy_act)
static int greedy_run(const int8_t *state, const int8_t *to_move, const int8_t *mask, const int8_t *hist, int depth, int32_t *action_out,
                      int8_t *cand_out, int8_t *fallback_out, int64_t n, int8_t *hist_rw, int32_t *final_out, uint64_t seed,
                      uint64_t env_base, uint32_t call)
{
    if (hist_rw) hist = hist_rw;
    parallel_for(n, [=](int64_t b0, int64_t b1) {
        for (int64_t b = b0; b < b1; ++b) {
            uint32_t r[7];
            load_row(state, b, r);
            const Planes p = make_planes(r);
            const int me = to_move[b] != 0;
            const uint64_t m = mask ? read_mask(mask + b * kActions) : legal54(p, me);
            const uint32_t prev3 = hist_prev3(hist, b, me);
            const GreedyResult g = greedy_decide(p, me, m, depth, prev3);
            if (action_out) action_out[b] = g.fallback ? -1 : g.chosen;
            if (fallback_out) fallback_out[b] = g.fallback ? 1 : 0;
            if (cand_out) write_mask(cand_out + b * kActions, g.cands);
            if (hist_rw) {  // :211-217 with the library's sampler, then :219
                const int fin = g.fallback ? pick54(g.cands, draw32(seed, env_base + (uint64_t)b, call, kStreamGreedy)) : g.chosen;
                final_out[b] = fin;
                int8_t *hp = hist_rw + (b * 2 + me) * 3;
                hp[0] = (int8_t)(prev3 >> 8);
                hp[1] = (int8_t)(prev3 >> 16);
                hp[2] = (int8_t)fin;
            }
        }
    });
    return GBL_OK;
}

int gbl_cpu_greedy(const int8_t *state, const int8_t *to_move, const int8_t *mask, const int8_t *hist, int depth, int32_t *action_out,
                   int8_t *cand_mask_out, int8_t *fallback_out, int64_t n, void *)
{
    GBL_CHECK_N(n);
    GBL_NEED(state, "state"); GBL_NEED(to_move, "to_move"); GBL_NEED(action_out, "action_out");
    if (depth < 1 || depth > 3) return fail(GBL_ERR_ARG, "depth must be 1, 2 or 3");
    return greedy_run(state, to_move, mask, hist, depth, action_out, cand_mask_out, fallback_out, n, nullptr, nullptr, 0, 0, 0);
}

int gbl_cpu_greedy_act_at(const int8_t *state, const int8_t *to_move, const int8_t *mask, int8_t *hist, int depth, uint64_t seed,
                          uint64_t env_base, uint32_t call, const uint32_t *call_dev, int32_t *action_out, int32_t *chosen_out,
                          int8_t *cand_mask_out, int8_t *fallback_out, int64_t n, void *)
{
    GBL_CHECK_N(n);
    GBL_NEED(state, "state"); GBL_NEED(to_move, "to_move"); GBL_NEED(hist, "hist"); GBL_NEED(action_out, "action_out");
    if (depth < 1 || depth > 3) return fail(GBL_ERR_ARG, "depth must be 1, 2 or 3");
    if (call_dev) call += *call_dev;
    return greedy_run(state, to_move, mask, nullptr, depth, chosen_out, cand_mask_out, fallback_out, n, hist, action_out, seed, env_base,
                      call);
}

int gbl_cpu_greedy_act(const int8_t *state, const int8_t *to_move, const int8_t *mask, int8_t *hist, int depth, uint64_t seed,
                       uint64_t env_base, uint32_t call, int32_t *action_out, int32_t *chosen_out, int8_t *cand_mask_out,
                       int8_t *fallback_out, int64_t n, void *stream)
{
    return gbl_cpu_greedy_act_at(state, to_move, mask, hist, depth, seed, env_base, call, nullptr, action_out, chosen_out, cand_mask_out,
                                 fallback_out, n, stream);
}

int gbl_cpu_collect_policy(int8_t *state, int8_t *to_move, int8_t *done, int8_t *hist, int32_t *actions_traj, int8_t *winner_traj,
                           int8_t *reward_traj, int8_t *done_traj, int8_t *to_move_traj, int8_t *mask_traj, int8_t *obs_traj,
                           int32_t *chosen_traj, int8_t *how_traj, int8_t *cand_traj, int64_t n, int64_t ply_stride,
                           int64_t tile_stride, uint64_t seed, uint64_t env_base, uint32_t ply0, const uint32_t *ply_dev,
                           uint32_t plies, int policy0, int policy1, int opening_plies, int illegal_mode, int64_t *counters,
                           int32_t *turn, void *)
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
    if (!strides_ok(n, plies, ply_stride, tile_stride))
        return fail(GBL_ERR_ARG, "ply_stride / tile_stride: multiples of 16 boards that keep the (ply, tile) cells apart");
    if (ply_dev) ply0 += *ply_dev;
    parallel_for(n, [=](int64_t b0, int64_t b1) {
        Tally tl;
        for (int64_t b = b0; b < b1; ++b) {
            uint32_t r[7];
            load_row(state, b, r);
            Planes p = make_planes(r);
            const HostRow row{reinterpret_cast<uint8_t *>(state) + b * kCells};
            int mover = to_move[b] != 0, dn = 0, tabs = turn ? turn[b] : 0;
            uint32_t hp[2] = {hist_prev3(hist, b, 0), hist_prev3(hist, b, 1)};
            uint64_t legal = legal54(p, mover);
            for (uint32_t t = 0; t < plies; ++t) {
                const uint32_t ply = ply0 + t;
                const int pol = mover ? policy1 : policy0;
                const bool gre = pol > 0 && tabs >= opening_plies;
                GreedyResult g{-1, 0ull, false};
                int action;
                if (gre) {
                    g = greedy_decide(p, mover, legal, pol, hp[mover]);
                    action = g.fallback ? pick54(g.cands, draw32(seed, env_base + (uint64_t)b, ply, kStreamGreedy)) : g.chosen;
                    hp[mover] = (hp[mover] >> 8) | (((uint32_t)action & 0xFFu) << 16);  // :219
                } else {
                    action = sample54(legal, seed, env_base + (uint64_t)b, ply);
                }
                const Ply y = play_ply(p, row, mover, legal, action, illegal_mode);
                dn = y.terminal ? 1 : 0;
                if (y.terminal) {
                    p = Planes{0u, 0u, 0u};
                    mover = 0;
                    row.reset();
                }
                tabs = next_turn(tabs, y, 1);
                tl.games += y.terminal; tl.w1 += y.winner == 1; tl.w2 += y.winner == -1;
                legal = legal54(p, mover);
                const int64_t at = cell_of(b, t, ply_stride, tile_stride);
                if (actions_traj) actions_traj[at] = action;
                if (winner_traj) winner_traj[at] = (int8_t)y.winner;
                if (reward_traj) { reward_traj[2 * at] = (int8_t)y.r0; reward_traj[2 * at + 1] = (int8_t)y.r1; }
                if (done_traj) done_traj[at] = (int8_t)dn;
                if (to_move_traj) to_move_traj[at] = (int8_t)mover;
                if (chosen_traj) chosen_traj[at] = (gre && !g.fallback) ? g.chosen : -1;
                if (how_traj) how_traj[at] = (int8_t)(gre ? (g.fallback ? GBL_HOW_FALLBACK : GBL_HOW_GREEDY) : GBL_HOW_RANDOM);
                if (cand_traj) write_mask(cand_traj + at * kActions, gre ? g.cands : 0ull);
                if (obs_traj) write_obs(obs_traj + at * kObs, p, mover);
                if (mask_traj) write_mask(mask_traj + at * kActions, legal);
            }
            tl.plies += plies;
            to_move[b] = (int8_t)mover;
            done[b] = (int8_t)dn;
            if (turn) turn[b] = tabs;
            if (hist) {
                uint8_t *h = reinterpret_cast<uint8_t *>(hist) + b * 6;
                for (int a = 0; a < 2; ++a)
                    for (int k = 0; k < 3; ++k) h[3 * a + k] = (uint8_t)(hp[a] >> (8 * k));
            }
        }
        add_tally(counters, tl);
    });
    return GBL_OK;
}

}  // extern "C"
