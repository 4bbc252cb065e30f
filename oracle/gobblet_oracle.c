/*
 * gobblet_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, single-board, scalar restatement of the reference's game engine
 * (gobblet_rl/game/board.py), of raw_env.observe/_legal_moves/step
 * (gobblet_rl/game/gobblet.py) and of GreedyGobbletPolicy.compute_action
 * (gobblet_rl/game/greedy_policy.py).  It exists so that the HIP path can be
 * checked bit-for-bit on a GPU box where /root/reference does not exist.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library -- and only as the checker / the timed CPU baseline.
 * Nothing under gobblet-rl_amd/ imports, links or calls it.
 *
 * Parity pin: tests/test_oracle_golden.py checks every function here against
 * tests/golden/ (vectors produced by importing the reference's own board.py /
 * gobblet.py / greedy_policy.py in the build container -- see
 * tests/golden/make_golden.py) and against the upstream known-answer test
 * tests/test_manual_policy_collector.py (masks output0..output5, legal list
 * output6, board output8).
 *
 * The restatement follows the reference statement by statement (including
 * behaviour on states that legal play never reaches), and each function cites
 * the reference lines it restates.  State is int8[27]: squares[9*level+pos]
 * (board.py:6-33), values -6..6.
 */
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
#include <pthread.h>

#define GBO_CELLS 27
#define GBO_ACTIONS 54
#define GBO_OBS 117

static inline int iabs(int v) { return v < 0 ? -v : v; }

/* board.py:135-153 calculate_winners(): 3 "vertical" triples, 3 "horizontal"
 * triples, two diagonals -- in exactly this order (order matters, see
 * gbo_check_for_winner). */
static const int8_t GBO_LINES[8][3] = {
    {0, 1, 2}, {3, 4, 5}, {6, 7, 8}, {0, 3, 6},
    {1, 4, 7}, {2, 5, 8}, {0, 4, 8}, {2, 4, 6}};

/* board.py:63-79 action decoders */
static inline int gbo_pos(int action) { return action % 9; }
static inline int gbo_piece(int action) { return action / 9 + 1; }
static inline int gbo_piece_size(int action) { return (gbo_piece(action) + 1) / 2; }
static inline int gbo_index(int action) { return gbo_pos(action) + 9 * (gbo_piece_size(action) - 1); }

/* board.py:159-177 get_flatboard(): per square, amax(|column|), FIRST level
 * holding that magnitude, sign of that cell times the magnitude. */
void gbo_get_flatboard(const int8_t *s, int8_t *flat)
{
    for (int i = 0; i < 9; ++i) {
        int a0 = iabs(s[i]), a1 = iabs(s[9 + i]), a2 = iabs(s[18 + i]);
        int top = a0;
        if (a1 > top) top = a1;
        if (a2 > top) top = a2;
        int idx = (a0 == top) ? 0 : (a1 == top) ? 1 : 2; /* list.index: first match */
        int v = s[9 * idx + i];
        int sign = (v > 0) - (v < 0);
        flat[i] = (int8_t)(sign * top);
    }
}

/* board.py:203-220 check_covered() */
void gbo_check_covered(const int8_t *s, int8_t *cov)
{
    memset(cov, 0, GBO_CELLS);
    for (int i = 0; i < 9; ++i)
        if (s[i] != 0 && (s[9 + i] != 0 || s[18 + i] != 0)) cov[i] = 1;
    for (int i = 0; i < 9; ++i)
        if (s[9 + i] != 0 && s[18 + i] != 0) cov[9 + i] = 1;
    /* covered[2,:] = 0 */
}

/* board.py:82-115 is_legal().  Returns 1 / 0, or -1 where the reference
 * raises Exception("PIECE HAS BEEN USED TWICE") (board.py:94-95). */
int gbo_is_legal(const int8_t *s, int action, int agent_index)
{
    int pos = gbo_pos(action);
    int piece = gbo_piece(action);
    int piece_size = gbo_piece_size(action);
    int mult = (agent_index == 0) ? 1 : -1; /* board.py:86 */
    const int8_t *level = s + 9 * (piece_size - 1);
    int count = 0, loc = -1;
    for (int q = 0; q < 9; ++q)
        if (level[q] == piece * mult) {
            if (loc < 0) loc = q;
            ++count;
        }
    if (count > 0) {
        if (count > 1) return -1;
        int8_t cov[GBO_CELLS];
        gbo_check_covered(s, cov);
        if (cov[9 * (piece_size - 1) + loc] == 1) return 0;
    }
    int8_t flat[9];
    gbo_get_flatboard(s, flat);
    if (flat[pos] == 0) return 1;
    int existing_size = (iabs(flat[pos]) + 1) / 2;
    return piece_size > existing_size ? 1 : 0;
}

/* board.py:118-132 play_turn(): silent no-op when illegal. */
void gbo_play_turn(int8_t *s, int agent_index, int action)
{
    int piece = gbo_piece(action);
    if (agent_index == 1) piece = -piece; /* board.py:120-121 */
    int index = gbo_index(action);
    if (gbo_is_legal(s, action, agent_index) != 1) return;
    for (int k = 0; k < GBO_CELLS; ++k) /* np.where(squares == piece)[0][0] */
        if (s[k] == piece) {
            s[k] = 0;
            break;
        }
    s[index] = (int8_t)piece;
}

/* board.py:183-194 check_for_winner(): no early exit, the LAST matching line
 * decides, and within one line +1 is tested before -1. */
int gbo_check_for_winner(const int8_t *s)
{
    int8_t flat[9];
    gbo_get_flatboard(s, flat);
    int winner = 0;
    for (int l = 0; l < 8; ++l) {
        int a = flat[GBO_LINES[l][0]], b = flat[GBO_LINES[l][1]], c = flat[GBO_LINES[l][2]];
        if (a > 0 && b > 0 && c > 0) winner = 1;
        if (a < 0 && b < 0 && c < 0) winner = -1;
    }
    return winner;
}

/* board.py:196-201 */
int gbo_check_game_over(const int8_t *s)
{
    int w = gbo_check_for_winner(s);
    return (w == 1 || w == -1) ? 1 : 0;
}

/* board.py:50-60 get_action() */
int gbo_get_action(const int8_t *s, int pos, int piece_size, int agent_index)
{
    int piece1 = piece_size * 2 - 1, piece2 = piece_size * 2;
    int action1 = pos + 9 * (piece1 - 1), action2 = pos + 9 * (piece2 - 1);
    if (gbo_is_legal(s, action1, agent_index) == 1) return action1;
    if (gbo_is_legal(s, action2, agent_index) == 1) return action2;
    return -1;
}

/* gobblet.py:223-228 _legal_moves() + gobblet.py:211-213 mask fill */
void gbo_legal_mask(const int8_t *s, int agent_index, int8_t *mask)
{
    for (int a = 0; a < GBO_ACTIONS; ++a)
        mask[a] = (int8_t)(gbo_is_legal(s, a, agent_index) == 1);
}

/* gobblet.py:179-208 observe(): observation part only.
 * obs[r][c][ch] with cell 9*level + 3r + c ; int8[3][3][13]. */
void gbo_observation(const int8_t *s, int agent_index, int8_t *obs)
{
    int flip = (agent_index == 1) ? -1 : 1; /* gobblet.py:182-185 */
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            int8_t *o = obs + (3 * r + c) * 13;
            for (int i = 1; i <= 6; ++i) { /* gobblet.py:189-197 */
                int v = flip * s[9 * ((i - 1) / 2) + 3 * r + c];
                o[i - 1] = (int8_t)(v == i);
                o[6 + i - 1] = (int8_t)(v == -i);
            }
            o[12] = (int8_t)(agent_index == 1); /* gobblet.py:199-206 */
        }
}

/* gobblet.py:179-215 observe(agent): observation + action_mask; the mask is
 * the current mover's legal moves only when agent == agent_selection
 * (gobblet.py:209), else all zeros. */
void gbo_observe(const int8_t *s, int agent_index, int agent_selection, int8_t *obs, int8_t *mask)
{
    gbo_observation(s, agent_index, obs);
    if (agent_index == agent_selection)
        gbo_legal_mask(s, agent_selection, mask);
    else
        memset(mask, 0, GBO_ACTIONS);
}

/* ------------------------------------------------------------------------- */
/* raw_env.step bookkeeping, gobblet.py:231-271, for one board.              */

#define GBO_ILLEGAL_NOOP 0      /* raw_env: silent no-op, the turn still passes */
#define GBO_ILLEGAL_TERMINATE 1 /* env(): TerminateIllegalWrapper(illegal_reward=-1), gobblet.py:114 */

typedef struct {
    int8_t to_move; /* index of agent_selection */
    int8_t done;    /* terminations[*] (both agents terminate together, gobblet.py:263) */
    int8_t winner;  /* check_for_winner() after the step */
    int8_t reward[2];
} gbo_status_t;

/* One raw_env.step(action).  Returns the winner value evaluated after the
 * move.  `done` boards are left untouched (the reference routes them to
 * _was_dead_step, gobblet.py:232-236). */
int gbo_step(int8_t *s, int8_t *to_move, int8_t *done, int action, int illegal_mode, int8_t *reward2, int *stepped)
{
    /* *stepped (may be NULL): 1 iff raw_env.step ran, i.e. the reference would do `self.turn += 1` (gobblet.py:270) */
    reward2[0] = reward2[1] = 0;
    if (stepped) *stepped = 0;
    if (*done) return gbo_check_for_winner(s);
    int mover = *to_move;
    int in_range = (action >= 0 && action < GBO_ACTIONS);
    int legal = in_range ? (gbo_is_legal(s, action, mover) == 1) : 0;
    if (!legal && illegal_mode == GBO_ILLEGAL_TERMINATE) {
        /* gobblet.py:50-51 + TerminateIllegalWrapper: mover -1, other 0,
         * everyone terminated, board untouched, agent_selection unchanged. */
        reward2[mover] = -1;
        *done = 1;
        return 0;
    }
    if (stepped) *stepped = 1;
    if (legal) gbo_play_turn(s, mover, action); /* gobblet.py:244 */
    *to_move = (int8_t)(1 - mover);              /* gobblet.py:246,267 */
    int w = gbo_check_for_winner(s);             /* gobblet.py:248-249 */
    if (w == 1) {                                /* gobblet.py:253-256 */
        reward2[0] = 1;
        reward2[1] = -1;
    } else if (w == -1) {                        /* gobblet.py:257-260 */
        reward2[1] = 1;
        reward2[0] = -1;
    }
    if (w != 0) *done = 1;                       /* gobblet.py:263 */
    return w;
}

/* ------------------------------------------------------------------------- */
/* Counter-based RNG shared with the HIP path: Philox4x32-10 (Salmon et al.,
 * "Parallel random numbers: as easy as 1, 2, 3", SC'11).  Known answers in
 * tests/test_oracle_golden.py. */
static inline void philox_round(uint32_t c[4], const uint32_t k[2])
{
    uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k[0];
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k[1];
    uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

void gbo_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c[4] = {ctr[0], ctr[1], ctr[2], ctr[3]};
    uint32_t k[2] = {key[0], key[1]};
    for (int r = 0; r < 10; ++r) {
        if (r) { k[0] += 0x9E3779B9u; k[1] += 0xBB67AE85u; }
        philox_round(c, k);
    }
    memcpy(out, c, sizeof c);
}

/* The masked-uniform sampling rule ("masked-random actions" in the configs;
 * reference sites examples/example_basic.py:58-61,
 * random_admissible_policy_rllib.py:23-30: uniform over legal actions).
 * Draw r = philox(ctr=(env_lo, env_hi, ply >> 2, stream), key=(seed_lo, seed_hi))[ply & 3]
 * (one generator block serves four consecutive plies of a board),
 * k = (r * nlegal) >> 32, return the k-th legal action in ascending order.
 * Returns -1 when the mask is empty.
 * stream separates the consumers of one (seed, board) pair: 0 = the masked-random
 * actions of the environment / random policy, 1 = the greedy policy's fallback
 * draw -- so a greedy agent and a random opponent with the same seed never
 * consume the same generator word. */
int gbo_sample_action_stream(const int8_t *mask, uint64_t seed, uint64_t env_id, uint32_t ply, uint32_t stream)
{
    int n = 0;
    for (int a = 0; a < GBO_ACTIONS; ++a) n += (mask[a] != 0);
    if (n == 0) return -1;
    uint32_t ctr[4] = {(uint32_t)env_id, (uint32_t)(env_id >> 32), ply >> 2, stream};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t out[4];
    gbo_philox4x32_10(ctr, key, out);
    int k = (int)(((uint64_t)out[ply & 3u] * (uint64_t)n) >> 32);
    for (int a = 0; a < GBO_ACTIONS; ++a)
        if (mask[a] != 0 && k-- == 0) return a;
    return -1;
}

int gbo_sample_action(const int8_t *mask, uint64_t seed, uint64_t env_id, uint32_t ply)
{
    return gbo_sample_action_stream(mask, seed, env_id, ply, 0u);
}

/* ------------------------------------------------------------------------- */
/* Batched drivers (same arrays as the HIP C-ABI in include/gobblet_hip.h).   */

void gbo_batch_reset(int8_t *state, int8_t *to_move, int8_t *done, int8_t *winner, int64_t n)
{
    /* gobblet.py:275-290: new Board() (zeros), agent_selection = player_1 */
    memset(state, 0, (size_t)n * GBO_CELLS);
    memset(to_move, 0, (size_t)n);
    memset(done, 0, (size_t)n);
    if (winner) memset(winner, 0, (size_t)n);
}

void gbo_batch_legal_mask(const int8_t *state, const int8_t *to_move, int8_t *mask, int64_t n)
{
    for (int64_t b = 0; b < n; ++b)
        gbo_legal_mask(state + b * GBO_CELLS, to_move[b], mask + b * GBO_ACTIONS);
}

void gbo_batch_winner(const int8_t *state, int8_t *winner, int64_t n)
{
    for (int64_t b = 0; b < n; ++b) winner[b] = (int8_t)gbo_check_for_winner(state + b * GBO_CELLS);
}

void gbo_batch_flatboard(const int8_t *state, int8_t *flat, int64_t n)
{
    for (int64_t b = 0; b < n; ++b) gbo_get_flatboard(state + b * GBO_CELLS, flat + b * 9);
}

void gbo_batch_covered(const int8_t *state, int8_t *cov, int64_t n)
{
    for (int64_t b = 0; b < n; ++b) gbo_check_covered(state + b * GBO_CELLS, cov + b * GBO_CELLS);
}

/* agent_sel: -1 = observe from each board's current mover; 0/1 = that agent */
void gbo_batch_observe(const int8_t *state, const int8_t *to_move, int agent_sel, int8_t *obs, int64_t n)
{
    for (int64_t b = 0; b < n; ++b)
        gbo_observation(state + b * GBO_CELLS, agent_sel < 0 ? to_move[b] : agent_sel, obs + b * GBO_OBS);
}

/* Lockstep step over boards [b0, b1): raw_env.step + observe(next mover).
 * auto_reset: a board that terminates on this step reports winner / reward /
 * done=1 for the step and is then reset in place (zeros, player_1 to move);
 * the mask / obs written are those of the fresh board.  With auto_reset=0 a
 * terminated board stays frozen, its mask is all zeros (nobody is to move:
 * gobblet.py:209 gives the off-turn mask) and its obs is that of the agent
 * whose turn it would have been.  Any of winner/reward/mask/obs may be NULL. */
static void batch_step_range(int8_t *state, int8_t *to_move, int8_t *done, const int32_t *actions,
                             int8_t *winner, int8_t *reward, int8_t *mask, int8_t *obs,
                             int64_t b0, int64_t b1, int illegal_mode, int auto_reset, int32_t *turn)
{
    for (int64_t b = b0; b < b1; ++b) {
        int8_t *s = state + b * GBO_CELLS;
        int8_t r2[2];
        int stepped;
        int w = gbo_step(s, &to_move[b], &done[b], actions[b], illegal_mode, r2, &stepped);
        if (turn && stepped) turn[b] += 1; /* gobblet.py:270 */
        if (winner) winner[b] = (int8_t)w;
        if (reward) { reward[2 * b] = r2[0]; reward[2 * b + 1] = r2[1]; }
        if (auto_reset && done[b]) { /* done[b] stays 1: "episode ended on this step" */
            memset(s, 0, GBO_CELLS);
            to_move[b] = 0;
            if (turn) turn[b] = 0; /* gobblet.py:289 */
        }
        if (mask) {
            if (done[b] && !auto_reset) memset(mask + b * GBO_ACTIONS, 0, GBO_ACTIONS);
            else gbo_legal_mask(s, to_move[b], mask + b * GBO_ACTIONS);
        }
        if (obs) gbo_observation(s, to_move[b], obs + b * GBO_OBS);
    }
}

void gbo_batch_step(int8_t *state, int8_t *to_move, int8_t *done, const int32_t *actions,
                    int8_t *winner, int8_t *reward, int8_t *mask, int8_t *obs,
                    int64_t n, int illegal_mode, int auto_reset)
{
    if (auto_reset) /* flags raised last step belong to boards already reset */
        memset(done, 0, (size_t)n);
    batch_step_range(state, to_move, done, actions, winner, reward, mask, obs, 0, n, illegal_mode, auto_reset, NULL);
}

/* Status byte of externally supplied actions, taken of the position BEFORE the step (include/gobblet_hip.h GBL_STATUS_*):
 * bit 0 = not a legal move of the mover -- what Board.play_turn silently ignores (board.py:125-126) and what
 * TerminateIllegalWrapper punishes (gobblet.py:114); bit 1 = outside [0, 54), where the reference's env() asserts
 * (AssertOutOfBoundsWrapper, gobblet.py:110-117).  A frozen board (done, no auto-reset) consumes no action
 * (_was_dead_step, gobblet.py:232-236): 0. */
void gbo_batch_action_status(const int8_t *state, const int8_t *to_move, const int8_t *done, const int32_t *actions,
                             int8_t *status, int64_t n, int auto_reset)
{
    for (int64_t b = 0; b < n; ++b) {
        if (!auto_reset && done[b]) { status[b] = 0; continue; }
        int a = actions[b];
        int in_range = (a >= 0 && a < GBO_ACTIONS);
        int legal = in_range ? (gbo_is_legal(state + b * GBO_CELLS, a, to_move[b]) == 1) : 0;
        status[b] = (int8_t)((legal ? 0 : 1) | (in_range ? 0 : 2));
    }
}

/* masked-uniform sampler over a batch (separate-kernel form) */
void gbo_batch_sample(const int8_t *mask, int32_t *actions, int64_t n, uint64_t seed, uint64_t env_base, uint32_t ply)
{
    for (int64_t b = 0; b < n; ++b)
        actions[b] = gbo_sample_action(mask + b * GBO_ACTIONS, seed, env_base + (uint64_t)b, ply);
}

/* Fused masked-random rollout of `plies` lockstep plies with auto-reset:
 * per ply and board: mask(mover) -> sample(seed, env_base+b, ply0+t) -> step.
 * Counters (int64[4]): plies played, games finished, P1 wins, P2 wins.
 * mask/obs (may be NULL) receive the outputs of the LAST ply. */
typedef struct {
    int8_t *state, *to_move, *done, *winner, *reward, *mask, *obs;
    int32_t *actions, *turn;
    int64_t b0, b1;
    uint64_t seed, env_base;
    uint32_t ply0, plies;
    int illegal_mode;
    int64_t counters[4];
} rollout_job_t;

static void *rollout_worker(void *arg)
{
    rollout_job_t *j = (rollout_job_t *)arg;
    int8_t m[GBO_ACTIONS];
    for (int64_t b = j->b0; b < j->b1; ++b) {
        int8_t *s = j->state + b * GBO_CELLS;
        for (uint32_t t = 0; t < j->plies; ++t) {
            /* a flag raised on the previous ply belongs to a board already reset */
            j->done[b] = 0;
            gbo_legal_mask(s, j->to_move[b], m);
            int a = gbo_sample_action(m, j->seed, j->env_base + (uint64_t)b, j->ply0 + t);
            int8_t r2[2];
            int stepped;
            int w = gbo_step(s, &j->to_move[b], &j->done[b], a, j->illegal_mode, r2, &stepped);
            if (j->turn && stepped) j->turn[b] += 1;
            j->counters[0] += 1;
            if (j->actions) j->actions[b] = a;
            if (j->winner) j->winner[b] = (int8_t)w;
            if (j->reward) { j->reward[2 * b] = r2[0]; j->reward[2 * b + 1] = r2[1]; }
            if (j->done[b]) {
                j->counters[1] += 1;
                j->counters[2] += (w == 1);
                j->counters[3] += (w == -1);
                memset(s, 0, GBO_CELLS);
                j->to_move[b] = 0;
                if (j->turn) j->turn[b] = 0;
            }
        }
        if (j->mask) gbo_legal_mask(s, j->to_move[b], j->mask + b * GBO_ACTIONS);
        if (j->obs) gbo_observation(s, j->to_move[b], j->obs + b * GBO_OBS);
    }
    return NULL;
}

void gbo_batch_rollout(int8_t *state, int8_t *to_move, int8_t *done, int32_t *actions,
                       int8_t *winner, int8_t *reward, int8_t *mask, int8_t *obs,
                       int64_t n, uint64_t seed, uint64_t env_base, uint32_t ply0, uint32_t plies,
                       int illegal_mode, int threads, int64_t *counters, int32_t *turn)
{
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    rollout_job_t jobs[256];
    pthread_t tid[256];
    int64_t per = (n + threads - 1) / threads;
    int used = 0;
    for (int t = 0; t < threads; ++t) {
        int64_t b0 = (int64_t)t * per, b1 = b0 + per;
        if (b0 >= n) break;
        if (b1 > n) b1 = n;
        rollout_job_t *j = &jobs[used];
        memset(j, 0, sizeof *j);
        j->state = state; j->to_move = to_move; j->done = done; j->actions = actions;
        j->winner = winner; j->reward = reward; j->mask = mask; j->obs = obs; j->turn = turn;
        j->b0 = b0; j->b1 = b1; j->seed = seed; j->env_base = env_base;
        j->ply0 = ply0; j->plies = plies; j->illegal_mode = illegal_mode;
        if (threads == 1) rollout_worker(j);
        else pthread_create(&tid[used], NULL, rollout_worker, j);
        ++used;
    }
    for (int c = 0; c < 4; ++c) counters[c] = 0;
    for (int t = 0; t < used; ++t) {
        if (threads > 1) pthread_join(tid[t], NULL);
        for (int c = 0; c < 4; ++c) counters[c] += jobs[t].counters[c];
    }
}

/* Multi-threaded lockstep step (cpu_baseline leg of bench.py): same contract
 * as gbo_batch_step, boards split into contiguous shards, one per thread. */
typedef struct {
    int8_t *state, *to_move, *done, *winner, *reward, *mask, *obs;
    const int32_t *actions;
    int32_t *turn;
    int64_t b0, b1;
    int illegal_mode, auto_reset;
} step_job_t;

static void *step_worker(void *arg)
{
    step_job_t *j = (step_job_t *)arg;
    if (j->auto_reset) memset(j->done + j->b0, 0, (size_t)(j->b1 - j->b0));
    batch_step_range(j->state, j->to_move, j->done, j->actions, j->winner, j->reward, j->mask, j->obs,
                     j->b0, j->b1, j->illegal_mode, j->auto_reset, j->turn);
    return NULL;
}

void gbo_batch_step_mt(int8_t *state, int8_t *to_move, int8_t *done, const int32_t *actions,
                       int8_t *winner, int8_t *reward, int8_t *mask, int8_t *obs,
                       int64_t n, int illegal_mode, int auto_reset, int threads, int32_t *turn)
{
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    step_job_t jobs[256];
    pthread_t tid[256];
    int64_t per = (n + threads - 1) / threads;
    int used = 0;
    for (int t = 0; t < threads; ++t) {
        int64_t b0 = (int64_t)t * per, b1 = b0 + per;
        if (b0 >= n) break;
        if (b1 > n) b1 = n;
        step_job_t *j = &jobs[used];
        j->state = state; j->to_move = to_move; j->done = done; j->actions = actions;
        j->winner = winner; j->reward = reward; j->mask = mask; j->obs = obs; j->turn = turn;
        j->b0 = b0; j->b1 = b1; j->illegal_mode = illegal_mode; j->auto_reset = auto_reset;
        if (threads == 1) step_worker(j);
        else pthread_create(&tid[used], NULL, step_worker, j);
        ++used;
    }
    if (threads > 1)
        for (int t = 0; t < used; ++t) pthread_join(tid[t], NULL);
}

/* cpu_baseline leg of bench.py: one lockstep ply of the benchmark pipeline per
 * call -- sample an action from the mask buffer (what gbl_sample does), then
 * the fused step with auto-reset writing mask / obs (what gbl_step does) --
 * boards split into contiguous shards, one per thread. */
typedef struct {
    step_job_t st;
    int32_t *actions_rw;
    uint64_t seed, env_base;
    uint32_t ply;
} ply_job_t;

static void *ply_worker(void *arg)
{
    ply_job_t *j = (ply_job_t *)arg;
    for (int64_t b = j->st.b0; b < j->st.b1; ++b)
        j->actions_rw[b] = gbo_sample_action(j->st.mask + b * GBO_ACTIONS, j->seed, j->env_base + (uint64_t)b, j->ply);
    return step_worker(&j->st);
}

void gbo_batch_sample_step_mt(int8_t *state, int8_t *to_move, int8_t *done, int32_t *actions, int8_t *winner,
                              int8_t *reward, int8_t *mask, int8_t *obs, int64_t n, uint64_t seed, uint64_t env_base,
                              uint32_t ply, int illegal_mode, int threads)
{
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    ply_job_t jobs[256];
    pthread_t tid[256];
    int64_t per = (n + threads - 1) / threads;
    int used = 0;
    for (int t = 0; t < threads; ++t) {
        int64_t b0 = (int64_t)t * per, b1 = b0 + per;
        if (b0 >= n) break;
        if (b1 > n) b1 = n;
        ply_job_t *j = &jobs[used];
        j->st.state = state; j->st.to_move = to_move; j->st.done = done; j->st.actions = actions;
        j->st.winner = winner; j->st.reward = reward; j->st.mask = mask; j->st.obs = obs;
        j->st.b0 = b0; j->st.b1 = b1; j->st.illegal_mode = illegal_mode; j->st.auto_reset = 1; j->st.turn = NULL;
        j->actions_rw = actions; j->seed = seed; j->env_base = env_base; j->ply = ply;
        if (threads == 1) ply_worker(j);
        else pthread_create(&tid[used], NULL, ply_worker, j);
        ++used;
    }
    if (threads > 1)
        for (int t = 0; t < used; ++t) pthread_join(tid[t], NULL);
}

/* ------------------------------------------------------------------------- */
/* GreedyGobbletPolicy.compute_action, greedy_policy.py:38-221, depth 1, 2 or 3. */

/* greedy_policy.py:43-71: rebuild the signed 27-vector from an observation.
 * Returns agent_index (obs[...,12].max()). */
int gbo_greedy_decode_obs(const int8_t *obs, int8_t *squares)
{
    int agent_index = 0;
    for (int rc = 0; rc < 9; ++rc)
        if (obs[rc * 13 + 12] > agent_index) agent_index = obs[rc * 13 + 12];
    for (int k = 0; k < 3; ++k) /* i = 0,2,4 -> level k = i/2 */
        for (int rc = 0; rc < 9; ++rc) {
            int i = 2 * k;
            const int8_t *o = obs + rc * 13;
            int bp = (i + 1) * o[i] + (i + 2) * o[i + 1];         /* :44-49 */
            int bo = (i + 1) * o[6 + i] + (i + 2) * o[6 + i + 1]; /* :50-56 */
            int v = (bp > bo) ? bp : -bo;                         /* :57 */
            if (agent_index == 1) v = -v;                         /* :67-68 */
            squares[9 * k + rc] = (int8_t)v;
        }
    return agent_index;
}

/* Algorithmic work counters of gbo_greedy: legality tests and leaf
 * evaluations (play_turn + check_for_winner) performed BY THE CALLING THREAD (thread-local: as plain
 * globals every increment of a threaded run bounced one cache line between the cores, and sixteen
 * threads ran at the speed of one); read by the bench on a single-threaded sample. */
static __thread int64_t g_greedy_legality_tests = 0, g_greedy_leaves = 0;
void gbo_greedy_work(int64_t *out, int reset)
{
    out[0] = g_greedy_legality_tests;
    out[1] = g_greedy_leaves;
    if (reset) g_greedy_legality_tests = g_greedy_leaves = 0;
}

/* Result of one decision.
 *   chosen      : value of chosen_action just before the fallback test
 *                 (greedy_policy.py:211), -1 for None
 *   cand_mask   : membership of actions_depth1 at that point (int8[54])
 *   fallback    : 1 iff the reference would call np.random.choice(actions_depth1)
 * The returned action is `chosen` when fallback == 0; with fallback == 1 the
 * caller picks from cand_mask (the reference uses numpy's global RNG there). */
void gbo_greedy(const int8_t *squares, int agent_index, const int8_t *mask, int depth,
                const int8_t *prev3 /* last <=3 own actions, -1 padded */,
                int *chosen_out, int8_t *cand_mask, int *fallback_out)
{
    int opponent_index = 1 - agent_index;
    const int winner_values[2] = {1, -1}; /* :74 */
    int legal_actions[GBO_ACTIONS], n_legal = 0;
    for (int a = 0; a < GBO_ACTIONS; ++a)
        if (mask[a] != 0) legal_actions[n_legal++] = a; /* :76 */
    int8_t in_d1[GBO_ACTIONS]; /* actions_depth1 as a membership set (order = ascending) */
    int n_d1 = n_legal;
    memset(in_d1, 0, sizeof in_d1);
    for (int i = 0; i < n_legal; ++i) in_d1[legal_actions[i]] = 1; /* :77-79 */
    int chosen = -1;                                               /* :80 */

    int res_key[GBO_ACTIONS], res_val[GBO_ACTIONS], n_res = 0; /* results dict, insertion order */
    for (int i = 0; i < n_legal; ++i) {                        /* :84 */
        int action = legal_actions[i];
        ++g_greedy_legality_tests;
        if (gbo_is_legal(squares, action, agent_index) != 1) continue; /* :85 */
        ++g_greedy_leaves;
        int8_t d1[GBO_CELLS];
        memcpy(d1, squares, GBO_CELLS);
        gbo_play_turn(d1, agent_index, action); /* :86-88 */
        int r = gbo_check_for_winner(d1);       /* :91 */
        res_key[n_res] = action;
        res_val[n_res++] = r;
        if (r == winner_values[agent_index]) { /* :92-94 */
            chosen = action;
            break;
        } else if (r == winner_values[opponent_index]) { /* :95-101 */
            if (n_d1 > 1) {
                in_d1[action] = 0; /* present by construction */
                --n_d1;
            } else
                break;
        }
    }

    if (depth > 1) { /* :103 */
        for (int i = 0; i < n_res; ++i) {
            if (res_val[i] != 0) continue; /* :105 */
            int action = res_key[i];
            int8_t d1[GBO_CELLS];
            memcpy(d1, squares, GBO_CELLS);
            gbo_play_turn(d1, agent_index, action); /* :107-109 */
            int all_me = 1, none_opp = 1;           /* all() over results_depth2.values() */
            int r2_key[GBO_ACTIONS], r2_val[GBO_ACTIONS], n_r2 = 0; /* results_depth2, insertion order */
            for (int a2 = 0; a2 < GBO_ACTIONS; ++a2) {
                ++g_greedy_legality_tests;
                if (gbo_is_legal(d1, a2, opponent_index) != 1) continue; /* :112-116 */
                ++g_greedy_leaves;
                int8_t d2[GBO_CELLS];
                memcpy(d2, d1, GBO_CELLS);
                gbo_play_turn(d2, opponent_index, a2); /* :120-124 */
                int r2 = gbo_check_for_winner(d2);     /* :126 */
                r2_key[n_r2] = a2;
                r2_val[n_r2++] = r2;
                if (r2 != winner_values[agent_index]) all_me = 0;
                if (r2 == winner_values[opponent_index]) none_opp = 0;
                if (r2 == winner_values[opponent_index]) { /* :129-131 */
                    if (n_d1 > 1) {                        /* :132-134 */
                        if (in_d1[action]) {
                            in_d1[action] = 0;
                            --n_d1;
                        }
                    } else
                        break;                                                /* :135-136 */
                    if (gbo_is_legal(squares, a2, agent_index) == 1)          /* :141 */
                        if (chosen < 0) chosen = a2;                          /* :142-143 */
                }
            }
            if (all_me) { /* :146-151 */
                chosen = action;
                break;
            }
            if (none_opp) { /* :153-157 */
                chosen = action;
                if (depth == 3) { /* :160-208, restated as written: the moves are played by
                                   * agent_index, and the leaf replays `action`, not act_depth3 */
                    memcpy(d1, squares, GBO_CELLS);
                    gbo_play_turn(d1, agent_index, action); /* :161-163 */
                    for (int j = 0; j < n_r2; ++j) {
                        if (r2_val[j] != 0) continue; /* :166-168 */
                        int8_t d2[GBO_CELLS];
                        memcpy(d2, d1, GBO_CELLS);
                        gbo_play_turn(d2, agent_index, r2_key[j]); /* :169-173 */
                        int8_t in_d3[GBO_ACTIONS];
                        int n_d3 = 0;
                        for (int a3 = 0; a3 < GBO_ACTIONS; ++a3) { /* :175-182 */
                            ++g_greedy_legality_tests;
                            in_d3[a3] = (int8_t)(gbo_is_legal(d2, a3, agent_index) == 1);
                            n_d3 += in_d3[a3];
                        }
                        int8_t was_legal3[GBO_ACTIONS];
                        memcpy(was_legal3, in_d3, GBO_ACTIONS); /* legal_actions_depth3 vs actions_depth3 */
                        for (int a3 = 0; a3 < GBO_ACTIONS; ++a3) { /* :186 */
                            if (!was_legal3[a3]) continue;
                            ++g_greedy_leaves;
                            int8_t d3[GBO_CELLS];
                            memcpy(d3, d2, GBO_CELLS);
                            gbo_play_turn(d3, agent_index, action); /* :187-191 */
                            int r3 = gbo_check_for_winner(d3);      /* :194 */
                            if (r3 == winner_values[agent_index]) { /* :195-199 */
                                chosen = action;
                                break;
                            } else if (r3 == winner_values[opponent_index]) { /* :200-208 */
                                if (n_d3 > 1) {
                                    if (in_d3[action]) {
                                        in_d3[action] = 0;
                                        --n_d3;
                                    }
                                } else
                                    break;
                            }
                        }
                    }
                }
            }
        }
    }

    int fallback = 0; /* :211-217 */
    if (chosen < 0) fallback = 1;
    else
        for (int k = 0; k < 3; ++k)
            if (prev3 && prev3[k] >= 0 && prev3[k] == chosen) fallback = 1;
    *chosen_out = chosen;
    memcpy(cand_mask, in_d1, GBO_ACTIONS);
    *fallback_out = fallback;
}

/* Batched greedy over boards: derives mask from the state when mask == NULL.
 * hist: int8[n][2][3] last three actions per agent (-1 = none), may be NULL.
 * action_out: chosen action, or -1 where fallback_out == 1 and the caller
 * must draw from cand_mask. */
void gbo_batch_greedy(const int8_t *state, const int8_t *to_move, const int8_t *mask_in, const int8_t *hist,
                      int depth, int32_t *action_out, int8_t *cand_mask_out, int8_t *fallback_out, int64_t n)
{
    for (int64_t b = 0; b < n; ++b) {
        int8_t m[GBO_ACTIONS];
        const int8_t *mask = mask_in ? mask_in + b * GBO_ACTIONS : m;
        if (!mask_in) gbo_legal_mask(state + b * GBO_CELLS, to_move[b], m);
        int chosen, fb;
        int8_t cm[GBO_ACTIONS];
        gbo_greedy(state + b * GBO_CELLS, to_move[b], mask, depth,
                   hist ? hist + (b * 2 + to_move[b]) * 3 : NULL, &chosen, cm, &fb);
        action_out[b] = fb ? -1 : chosen;
        if (cand_mask_out) memcpy(cand_mask_out + b * GBO_ACTIONS, cm, GBO_ACTIONS);
        if (fallback_out) fallback_out[b] = (int8_t)fb;
    }
}

/* One policy step (the C-ABI's gbl_greedy_act): gbo_batch_greedy, the fallback draw of
 * greedy_policy.py:211-217 with gbo_sample_action over the candidate set keyed
 * (seed, env_base + b, call), and the history append of :219 (hist is updated in place). */
void gbo_batch_greedy_act(const int8_t *state, const int8_t *to_move, const int8_t *mask_in, int8_t *hist, int depth,
                          uint64_t seed, uint64_t env_base, uint32_t call, int32_t *action_out, int32_t *chosen_out,
                          int8_t *cand_mask_out, int8_t *fallback_out, int64_t n)
{
    for (int64_t b = 0; b < n; ++b) {
        int32_t chosen;
        int8_t cm[GBO_ACTIONS], fb;
        gbo_batch_greedy(state + b * GBO_CELLS, to_move + b, mask_in ? mask_in + b * GBO_ACTIONS : NULL,
                         hist + b * 6, depth, &chosen, cm, &fb, 1);
        int fin = fb ? gbo_sample_action_stream(cm, seed, env_base + (uint64_t)b, call, 1u) : chosen;
        action_out[b] = fin;
        if (chosen_out) chosen_out[b] = chosen;
        if (cand_mask_out) memcpy(cand_mask_out + b * GBO_ACTIONS, cm, GBO_ACTIONS);
        if (fallback_out) fallback_out[b] = fb;
        int8_t *h = hist + (b * 2 + (to_move[b] != 0)) * 3; /* prev_actions[agent_index].append(...), last three kept */
        h[0] = h[1];
        h[1] = h[2];
        h[2] = (int8_t)fin;
    }
}
