/*
 * gobblet_hip.h -- C-ABI of the MI355X (gfx950) batched Gobblet hot path.
 *
 * The reference (elliottower/gobblet-rl) is pure Python and has no FFI seam;
 * the seam this library sits behind is the `Board` object interface consumed
 * by gobblet_rl/game/gobblet.py, greedy_policy.py and manual_policy.py
 * (SURVEY.md section 8b).  Every entry point below is the lockstep, N-board
 * form of one reference function and cites it.  INTEGRATION.md shows the
 * ctypes binding a reference maintainer would add.
 *
 * Conventions
 *   - All pointers are DEVICE pointers owned by the caller (e.g. torch-ROCm
 *     tensors' data_ptr()).  The library keeps no state between calls, and the
 *     compute entry points allocate nothing and only enqueue on `stream`.  The
 *     exceptions are helpers, none of them on the step path:
 *     gbl_pinned_alloc / gbl_pinned_free and gbl_block_alloc / gbl_block_free
 *     allocate and free memory the CALLER then owns (the library keeps no
 *     reference), and gbl_placement_probe creates HIP events for its own
 *     duration and BLOCKS the host until its measurement has run.
 *   - Every buffer that holds per-board ROWS (state, mask, obs, flat, cov)
 *     must be 16-byte aligned at board 0 (hipMalloc / torch allocations are).
 *   - `stream` is a hipStream_t (NULL = default stream).  Calls only enqueue.
 *   - Return value: 0 = OK, negative = GBL_ERR_*; gbl_last_error() gives the
 *     thread-local message.  No C++ exception crosses the ABI.
 *   - Kernels never trap on bad data: an action outside [0,54) is an illegal
 *     action (handled per `illegal_mode`) and is flagged in the status byte of
 *     gbl_step_ex / gbl_collect_from_ex (GBL_STATUS_OUT_OF_RANGE).
 *
 * Data layout in HBM (env-major, int8)
 *   state   int8 [n][27]      Board.squares per board: squares[9*level + pos]
 *                             (board.py:6-33); 0 empty, +piece player_1,
 *                             -piece player_2, piece in 1..6.
 *   to_move int8 [n]          index of agent_selection (0 = player_1)
 *   done    int8 [n]          terminations (both agents at once, gobblet.py:263)
 *   winner  int8 [n]          check_for_winner(): -1 / 0 / +1
 *   reward  int8 [n][2]       rewards of (player_1, player_2) for this step
 *   mask    int8 [n][54]      action_mask of the agent to move
 *   obs     int8 [n][3][3][13] observation of the agent to move
 *   actions int32[n]
 *   turn    int32[n]          raw_env.turn (plies since reset), optional
 * Contract on `state`: a cell of level k holds 0 or +-(2k+1) or +-(2k+2) --
 * what legal play from reset can produce.  For such states every function is
 * bit-identical to the reference.  Outside it nothing is promised (gbl_validate
 * flags such boards); in particular gbl_collect on small batches keeps the
 * boards as bit planes and REBUILDS the state rows from them on return, so a
 * cell that held a value its level cannot hold comes back as the contract's
 * reading of it (sign and parity kept), where larger batches leave it as found.
 */
#ifndef GOBBLET_HIP_H
#define GOBBLET_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GBL_OK 0
#define GBL_ERR_ARG (-1)   /* null / negative / out-of-range argument */
#define GBL_ERR_ALIGN (-2) /* a row buffer is not 16-byte aligned */
#define GBL_ERR_HIP (-3)   /* a HIP runtime call failed; see gbl_last_error() */

#define GBL_CELLS 27
#define GBL_ACTIONS 54
#define GBL_OBS_BYTES 117

/* gbl_rollout tallies: int64[GBL_COUNTER_STRIPES][GBL_COUNTER_STRIDE]; words 0..3 of every
 * stripe hold {plies played, games finished, player_1 wins, player_2 wins}; a total is the
 * sum over stripes.  (Striped so that concurrently finishing wavefronts do not serialise on
 * one address; one stripe per 128-byte line.) */
#define GBL_COUNTER_STRIPES 64
#define GBL_COUNTER_STRIDE 16

/* illegal_mode */
#define GBL_ILLEGAL_NOOP 0      /* raw_env.step: silent no-op, the turn still passes (gobblet.py:244-246, board.py:125-126) */
#define GBL_ILLEGAL_TERMINATE 1 /* env(): TerminateIllegalWrapper(illegal_reward=-1) (gobblet.py:114, :50-51) */

/* Version / layout query: writes {abi_version, cells, actions, obs_bytes, tile_boards, row_alignment}.
 * abi_version 2 (round 6): + gbl_step_ex, gbl_collect_from_ex, GBL_STATUS_*; gbl_collect_variant reports GBL_COLLECT_TRIO as 4
 * (3, once GBL_COLLECT_SMALL, is retired).  Every entry point of version 1 keeps its signature and meaning. */
#define GBL_ABI_VERSION 2
int gbl_layout_info(int32_t out[6]);

/* Message of the last error on this thread ("" if none). */
const char *gbl_last_error(void);

/* raw_env.reset(), gobblet.py:275-290: new Board() (zeros, board.py:33),
 * agent_selection = player_1, terminations False.  winner may be NULL. */
int gbl_reset(int8_t *state, int8_t *to_move, int8_t *done, int8_t *winner, int64_t n, void *stream);

/* raw_env._legal_moves() + mask fill, gobblet.py:223-228,211-213
 * (54 x Board.is_legal, board.py:82-115) for agent to_move[b]. */
int gbl_legal_mask(const int8_t *state, const int8_t *to_move, int8_t *mask, int64_t n, void *stream);

/* Board.is_legal(action, agent_index), board.py:82-115, one action per board.
 * agent_index: device int8[n]; out: int8[n] (1 legal / 0 illegal or out of range). */
int gbl_is_legal(const int8_t *state, const int8_t *agent_index, const int32_t *actions, int8_t *out, int64_t n,
                 void *stream);

/* Board.play_turn(agent_index, action), board.py:118-132: in-place, silent no-op when illegal. */
int gbl_play_turn(int8_t *state, const int8_t *agent_index, const int32_t *actions, int64_t n, void *stream);

/* Board.check_for_winner(), board.py:183-194 (lines board.py:135-153; the last matching line decides). */
int gbl_winner(const int8_t *state, int8_t *winner, int64_t n, void *stream);

/* Board.get_flatboard(), board.py:159-177: flat int8[n][9], signed piece number of the top piece. */
int gbl_flatboard(const int8_t *state, int8_t *flat, int64_t n, void *stream);

/* Board.check_covered(), board.py:203-220: cov int8[n][27]. */
int gbl_covered(const int8_t *state, int8_t *cov, int64_t n, void *stream);

/* State-contract check for callers that assign `state` themselves (the reference lets callers
 * assign Board.squares: greedy_policy.py:71, manual_policy.py:60).  flags int8[n]: bit 0 = a cell
 * holds a value its level cannot hold; bit 1 = a piece number occurs twice, where the reference's
 * is_legal raises Exception("PIECE HAS BEEN USED TWICE") (board.py:94-95).  0 = board is valid. */
int gbl_validate(const int8_t *state, int8_t *flags, int64_t n, void *stream);

/* raw_env.observe(agent)["observation"], gobblet.py:179-208.
 * agent_sel = 0 / 1: observe every board as that agent; -1: as to_move[b]
 * (to_move may be NULL unless agent_sel == -1). */
int gbl_observe(const int8_t *state, const int8_t *to_move, int agent_sel, int8_t *obs, int64_t n, void *stream);

/* One lockstep raw_env.step(actions[b]) + observe(next mover) per board,
 * gobblet.py:231-271 + :179-215, fused.  In place on state / to_move / done.
 *   done[b] != 0 on entry: the board is frozen (reference: _was_dead_step,
 *     gobblet.py:232-236); its mask is written as zeros, obs as observed.
 *   auto_reset != 0: `done` on entry is ignored; a board that terminates on
 *     this step reports winner / reward / done = 1 and is reset in place
 *     (zeros, player_1 to move); mask / obs are those of the fresh board.
 *   turn (int32[n], in/out, may be NULL): raw_env.turn per board -- +1 whenever raw_env.step
 *     runs (also on an illegal no-op, gobblet.py:270), 0 after a reset (gobblet.py:289).
 *   winner_out / reward_out / mask_out / obs_out may each be NULL. */
int gbl_step(int8_t *state, int8_t *to_move, int8_t *done, const int32_t *actions, int8_t *winner_out,
             int8_t *reward_out, int8_t *mask_out, int8_t *obs_out, int32_t *turn, int64_t n, int illegal_mode,
             int auto_reset, void *stream);
/* gbl_step for a collector whose policy lives outside the library (the loop of the reference's Tianshou / RLlib
 * trainers: policy(obs, mask) -> env.step -> buffer.add): the same ply, with mask_out / obs_out / winner_out /
 * reward_out pointing at slot t of trajectory arrays, and in the same launch copies of the ply's other scalars into
 * that slot (each may be NULL): actions_out int32[n] = the actions played, done_out int8[n] = done after the ply,
 * to_move_out int8[n] = the agent to move next. */
int gbl_step_into(int8_t *state, int8_t *to_move, int8_t *done, const int32_t *actions, int8_t *winner_out,
                  int8_t *reward_out, int8_t *mask_out, int8_t *obs_out, int32_t *turn, int32_t *actions_out,
                  int8_t *done_out, int8_t *to_move_out, int64_t n, int illegal_mode, int auto_reset, void *stream);

/* gbl_step_into with two more optional outputs (each may be NULL; all NULL = gbl_step_into):
 *   status_out int8[n]: what became of actions[b] -- 0 = a legal move, played; GBL_STATUS_ILLEGAL = not a legal move of
 *     the mover (handled per illegal_mode: raw_env.step's silent no-op, gobblet.py:244-246 / board.py:125-126, or
 *     TerminateIllegalWrapper's -1, gobblet.py:114); GBL_STATUS_ILLEGAL | GBL_STATUS_OUT_OF_RANGE = outside [0, 54),
 *     where the reference's env() asserts (AssertOutOfBoundsWrapper, gobblet.py:110-117) -- the kernels never trap, a
 *     batched caller tells "illegal" from "garbage index" here.  A board that was frozen on entry consumes no action: 0.
 *   next_actions_out int32[n]: the NEXT mover's masked-uniform draw from the mask this very launch stores -- exactly
 *     gbl_sample_at(mask_out, next_actions_out, n, seed, env_base, ply, ply_dev) run behind the step (-1 where nobody
 *     is to move), without the sampler's launch and its 58 bytes per board of traffic: the random opponent's reply /
 *     an epsilon-greedy policy's exploration move of the reference's trainer loops (examples/example_basic.py:58-61,
 *     example_tianshou_DQN.py: MultiAgentPolicyManager([agent, RandomPolicy])).  May alias `actions`: a board's action
 *     is read before its next one is written, so one array can carry a masked-random game from launch to launch.
 *     mask_out may be NULL (the draw is taken from the same 54-bit set either way). */
#define GBL_STATUS_ILLEGAL 1
#define GBL_STATUS_OUT_OF_RANGE 2
int gbl_step_ex(int8_t *state, int8_t *to_move, int8_t *done, const int32_t *actions, int8_t *winner_out,
                int8_t *reward_out, int8_t *mask_out, int8_t *obs_out, int32_t *turn, int32_t *actions_out,
                int8_t *done_out, int8_t *to_move_out, int8_t *status_out, int32_t *next_actions_out, uint64_t seed,
                uint64_t env_base, uint32_t ply, const uint32_t *ply_dev, int64_t n, int illegal_mode, int auto_reset,
                void *stream);

/* Host memory that the kernels read and write directly (pinned and mapped into the device's address space), for
 * callers that want a result on the host without a separate copy -- the single-environment facade keeps its
 * board, action and 432-byte record there, so a ply crosses PCIe inside its one launch.  *host_ptr is the CPU's
 * view, *dev_ptr the pointer to hand to the gbl_* entry points (same memory).  The caller owns the block and
 * frees it with gbl_pinned_free(host_ptr); the library keeps no reference. */
int gbl_pinned_alloc(int64_t bytes, void **host_ptr, void **dev_ptr);
int gbl_pinned_free(void *host_ptr);

/* Everything the reference derives from a position, for n boards, in ONE launch (the single-environment
 * facade gobblet_v1.env() asks all of it once per ply): optionally Board.play_turn(agent_index[b], actions[b])
 * first (board.py:118-132; illegal or out-of-range: silent no-op; `state` is updated in place), then one
 * GBL_REC_BYTES-byte record per board of the resulting position:
 *   +GBL_REC_SQUARES int8[27]  Board.squares                     board.py:33
 *   +GBL_REC_WINNER  int8      check_for_winner()                board.py:183-194
 *   +GBL_REC_FLAT    int8[9]   get_flatboard()                   board.py:159-177
 *   +GBL_REC_COVERED int8[27]  check_covered()                   board.py:203-220
 *   +GBL_REC_MASK0 / +GBL_REC_MASK1  int8[54]  is_legal(a, agent 0 / 1), a = 0..53   board.py:82-115
 *   +GBL_REC_OBS0  / +GBL_REC_OBS1   int8[117] raw_env.observe planes of agent 0 / 1   gobblet.py:179-208
 * (every field 4-byte aligned, padding bytes zero).  actions == NULL: no move (agent_index is then unused and
 * may be NULL).  record_out: int8[n][GBL_REC_BYTES], 16-byte aligned. */
#define GBL_REC_BYTES 432
#define GBL_REC_SQUARES 0
#define GBL_REC_WINNER 28
#define GBL_REC_FLAT 32
#define GBL_REC_COVERED 44
#define GBL_REC_MASK0 72
#define GBL_REC_MASK1 128
#define GBL_REC_OBS0 184
#define GBL_REC_OBS1 304
int gbl_board_eval(int8_t *state, const int8_t *agent_index, const int32_t *actions, int8_t *record_out, int64_t n,
                   void *stream);

/* Masked-uniform action sampling -- the rule behind "masked-random actions"
 * (examples/example_basic.py:58-61, random_admissible_policy_rllib.py:23-30:
 * uniform over legal actions) -- with a counter-based RNG so CPU and GPU draw
 * the same action: r = word (ply & 3) of Philox4x32-10(ctr = (env_lo, env_hi,
 * ply >> 2, stream), key = (seed_lo, seed_hi)) -- one generator block serves four
 * consecutive plies of a board; k = (r * nlegal) >> 32; the k-th legal action
 * in ascending order (-1 if the mask is empty).  env id = env_base + b.
 * Counter word 3 separates the consumers of one (seed, board) pair: stream 0 here
 * and in gbl_rollout, stream 1 for the fallback draw of gbl_greedy_act. */
int gbl_sample(const int8_t *mask, int32_t *actions, int64_t n, uint64_t seed, uint64_t env_base, uint32_t ply,
               void *stream);

/* Fused masked-random rollout: `plies` lockstep plies in ONE
 * launch; per ply and board: legal mask -> gbl_sample rule with ply index
 * ply0 + t -> gbl_step with auto-reset.  The board stays in registers between
 * plies; state / to_move / done and the optional outputs (action, winner,
 * reward, mask, obs of the agent to move) are stored after the LAST ply.
 * plies = 1 is "sample + step" fused into one launch: every ply's outputs
 * are then materialised in HBM for a consumer, as with gbl_sample + gbl_step.
 * counters: device int64[GBL_COUNTER_STRIPES][GBL_COUNTER_STRIDE], 128-byte
 * aligned, atomically incremented (see above; may be NULL -- the atomics cost
 * a few microseconds per launch at 2^20 boards).  turn: as in gbl_step (may be
 * NULL).  actions_out / winner_out / reward_out / mask_out / obs_out may be NULL. */
int gbl_rollout(int8_t *state, int8_t *to_move, int8_t *done, int32_t *actions_out, int8_t *winner_out,
                int8_t *reward_out, int8_t *mask_out, int8_t *obs_out, int64_t n, uint64_t seed, uint64_t env_base,
                uint32_t ply0, uint32_t plies, int illegal_mode, int64_t *counters, int32_t *turn, void *stream);

/* GreedyGobbletPolicy.compute_action board decode, greedy_policy.py:43-71:
 * obs int8[n][3][3][13] -> state int8[n][27], to_move int8[n] (channel 12). */
int gbl_decode_obs(const int8_t *obs, int8_t *state, int8_t *to_move, int64_t n, void *stream);

/* GreedyGobbletPolicy.compute_action, greedy_policy.py:38-221, depth 1, 2 or
 * 3, for the agent to move on each board.  depth 3 returns the depth-2
 * decision: the only assignment in the reference's depth-3 block (:160-208)
 * is `chosen_action = action` (:197), which :157 has just made, and
 * actions_depth3 is local, so that block cannot change the result (pinned on
 * the reference itself by tests/golden/greedy_depth3.npz).
 *   mask     : legal mask handed to the policy (NULL = derive from state)
 *   hist     : int8[n][2][3] last three actions per agent, -1 = none (NULL = empty)
 *   action_out   int32[n] : chosen action; -1 where the reference falls back
 *                           to np.random.choice(actions_depth1) (:211-217)
 *   cand_mask_out int8[n][54] : membership of actions_depth1 at :211 (may be NULL)
 *   fallback_out  int8[n]     : 1 where the fallback fires (may be NULL) */
int gbl_greedy(const int8_t *state, const int8_t *to_move, const int8_t *mask, const int8_t *hist, int depth,
               int32_t *action_out, int8_t *cand_mask_out, int8_t *fallback_out, int64_t n, void *stream);

/* One whole policy step of GreedyGobbletPolicy.compute_action, greedy_policy.py:38-221, in one launch:
 * gbl_greedy, then where it reports the fallback (:211-217) the gbl_sample rule over the candidate set
 * with (seed, env_base + b, call) on generator stream 1 in place of numpy's global RNG (so a random opponent
 * sampling with the same seed never consumes the same word), then the history append of :219
 * (hist[b][agent to move] shifts left by one and takes the returned action).
 *   hist      : int8[n][2][3], read AND updated (required)
 *   action_out  int32[n] : the action the policy returns (-1 only if the candidate set is empty)
 *   chosen_out  int32[n] : gbl_greedy's action_out (chosen, or -1 where the fallback fired; may be NULL)
 *   cand_mask_out / fallback_out : as gbl_greedy (may be NULL) */
int gbl_greedy_act(const int8_t *state, const int8_t *to_move, const int8_t *mask, int8_t *hist, int depth,
                   uint64_t seed, uint64_t env_base, uint32_t call, int32_t *action_out, int32_t *chosen_out,
                   int8_t *cand_mask_out, int8_t *fallback_out, int64_t n, void *stream);

/* Graph-replay forms of the three entry points that draw random numbers.  seed / env_base / ply travel by value,
 * so a captured hipGraph would replay the same draws; here the index is  ply + *ply_dev  (call + *call_dev) with
 * the base in device memory, and gbl_counter_add -- one more node at the end of the captured sequence -- moves
 * it on, so every replay draws afresh.  A NULL pointer makes them the by-value forms. */
int gbl_sample_at(const int8_t *mask, int32_t *actions, int64_t n, uint64_t seed, uint64_t env_base, uint32_t ply,
                  const uint32_t *ply_dev, void *stream);
int gbl_rollout_at(int8_t *state, int8_t *to_move, int8_t *done, int32_t *actions_out, int8_t *winner_out,
                   int8_t *reward_out, int8_t *mask_out, int8_t *obs_out, int64_t n, uint64_t seed, uint64_t env_base,
                   uint32_t ply0, const uint32_t *ply_dev, uint32_t plies, int illegal_mode, int64_t *counters,
                   int32_t *turn, void *stream);
int gbl_greedy_act_at(const int8_t *state, const int8_t *to_move, const int8_t *mask, int8_t *hist, int depth,
                      uint64_t seed, uint64_t env_base, uint32_t call, const uint32_t *call_dev, int32_t *action_out,
                      int32_t *chosen_out, int8_t *cand_mask_out, int8_t *fallback_out, int64_t n, void *stream);

/* Trajectory collection (K plies per launch with EVERY ply materialised).  `plies` masked-random
 * plies with auto-reset in ONE launch; ply t (t = 0 .. plies-1) of board b leaves in element
 *     cell(t, b) = t * ply_stride + (b / 64) * tile_stride + b % 64
 * of the trajectory arrays exactly what gbl_rollout(plies = 1) with ply index ply0 + t leaves in element b of
 * its output arrays:
 *   actions_traj int32[cells]      the action played at ply t
 *   winner_traj  int8 [cells]      check_for_winner() after it        reward_traj int8[cells][2]
 *   done_traj    int8 [cells]      1 where the episode ended on ply t (the board was then reset)
 *   to_move_traj int8 [cells]      the agent to move next -- the one mask / obs of the element belong to
 *   mask_traj    int8 [cells][54]  obs_traj int8[cells][117]
 * (any of them may be NULL).  The two strides, in boards and multiples of 16, choose the layout:
 *   time-major  [plies][slot_boards]:  ply_stride = slot_boards >= n rounded up to 64, tile_stride = 64
 *                                      (element (t, b) at t * slot_boards + b: one array slice per ply);
 *   tile-major  [tiles][plies][64]:    ply_stride = 64, tile_stride = 64 * plies
 *                                      (each tile of 64 boards keeps its whole trajectory contiguous: every
 *                                      wavefront writes ONE sequential region per array).
 * state / to_move / done / turn / counters are read at entry and hold the position after the last ply on
 * return, as with gbl_rollout.  A consumer (replay buffer, trainer) reads the trajectory after the launch;
 * between plies nothing but the per-ply outputs touches HBM.
 * ply_dev: NULL, or the device-resident base of the ply index as in gbl_rollout_at. */
int gbl_collect(int8_t *state, int8_t *to_move, int8_t *done, int32_t *actions_traj, int8_t *winner_traj,
                int8_t *reward_traj, int8_t *done_traj, int8_t *to_move_traj, int8_t *mask_traj, int8_t *obs_traj,
                int64_t n, int64_t ply_stride, int64_t tile_stride, uint64_t seed, uint64_t env_base, uint32_t ply0,
                const uint32_t *ply_dev, uint32_t plies, int illegal_mode, int64_t *counters, int32_t *turn,
                void *stream);
/* Trajectory collection with a DEVICE-SIDE POLICY per side (the reference plays whole games with its greedy policy on
 * either or both sides: tutorials/GreedyAgent/tutorial_greedy.py:16-54 -- one GreedyGobbletPolicy object acting for both
 * agents, the first two plies of every game drawn at random -- and gobblet_rl/game/greedy_policy_tianshou.py:63-84,
 * greedy against a learner or a random agent).  gbl_collect with the mover's action chosen inside the launch:
 *   policy0 / policy1  how player_1 / player_2 decide: GBL_POLICY_RANDOM = the masked-uniform draw of gbl_sample
 *                      (generator stream 0, ply index ply0 + t); GBL_POLICY_GREEDY1 / 2 / 3 =
 *                      GreedyGobbletPolicy.compute_action at that depth (greedy_policy.py:38-221; 3 decides like 2, see
 *                      gbl_greedy) on the mover's legal mask and hist[b][mover], and where the reference falls back to
 *                      np.random.choice(actions_depth1) (:211-217) the gbl_sample rule over that candidate set on
 *                      generator stream 1 with the same ply index -- exactly gbl_greedy_act(call = ply0 + t) -- followed
 *                      by the history append of :219.
 *   opening_plies      a greedy side plays the first plies of every game (turn[b] < opening_plies) at random instead,
 *                      without touching its history (tutorial_greedy.py:34-41 uses 2); needs `turn`.  0: none.
 *   hist               int8[n][2][3], read at entry, holds the histories after the last ply on return (they survive a
 *                      game's end, like the reference's policy object); NULL = empty histories, nothing written back.
 * Ply t of board b leaves in cell(t, b) (see gbl_collect) what gbl_greedy_act / gbl_sample + gbl_step with auto-reset would:
 * the seven arrays of gbl_collect, and (each may be NULL)
 *   chosen_traj int32[cells]      gbl_greedy's action_out: the chosen action, -1 where the fallback fired or no greedy
 *                                 policy acted
 *   how_traj    int8 [cells]      GBL_HOW_RANDOM / GBL_HOW_GREEDY / GBL_HOW_FALLBACK: how the action was arrived at
 *   cand_traj   int8 [cells][54]  membership of actions_depth1 at :211 (zeros where no greedy policy acted)
 * Everything else (strides, state / to_move / done / turn / counters at entry and on return, ply_dev) as gbl_collect. */
#define GBL_POLICY_RANDOM 0
#define GBL_POLICY_GREEDY1 1
#define GBL_POLICY_GREEDY2 2
#define GBL_POLICY_GREEDY3 3
#define GBL_HOW_RANDOM 0
#define GBL_HOW_GREEDY 1
#define GBL_HOW_FALLBACK 2
int gbl_collect_policy(int8_t *state, int8_t *to_move, int8_t *done, int8_t *hist, int32_t *actions_traj,
                       int8_t *winner_traj, int8_t *reward_traj, int8_t *done_traj, int8_t *to_move_traj, int8_t *mask_traj,
                       int8_t *obs_traj, int32_t *chosen_traj, int8_t *how_traj, int8_t *cand_traj, int64_t n,
                       int64_t ply_stride, int64_t tile_stride, uint64_t seed, uint64_t env_base, uint32_t ply0,
                       const uint32_t *ply_dev, uint32_t plies, int policy0, int policy1, int opening_plies,
                       int illegal_mode, int64_t *counters, int32_t *turn, void *stream);

/* Which kernel a gbl_collect call of this shape runs (no launch; >= 0, or GBL_ERR_ARG): benchmarks and profiles label
 * their records with it instead of re-deriving the library's dispatch rule.
 *   GBL_COLLECT_STREAM  k_collect,  one wavefront per tile of 64 boards, trajectory rows stored non-temporally
 *   GBL_COLLECT_CACHED  k_collect with plain stores (does not exist in the product build: A/B builds only)
 *   GBL_COLLECT_PAIR    k_collect2, two wavefronts per tile (one plays, one stores): grids of up to 2560 tiles
 *   GBL_COLLECT_TRIO    k_collect3, three wavefronts per tile: one plays and hands every ply's position over (one barrier per
 *                       ply), one builds and stores the mask rows, one the observation rows
 *   GBL_COLLECT_GROUP32 k_collect5 (round 6): batches that do not fill the chip and have a mask trajectory -- groups of 32 boards; ONE
 *                       wavefront plays a group (two lanes per board) and leaves every ply's position, legal mask, action and results
 *                       in an LDS ring of 2 x 4 plies; a second wavefront builds the mask rows, stores the scalars and runs the
 *                       sampler's generator, two more the observation rows of 16 boards each; one rendezvous per four plies.
 *                       FULL up to 9 216 boards, MASK_ONLY up to 32 768
 *   GBL_COLLECT_ROLES(la, ko, merge) = 1000 + 100 la + 10 ko + merge:  k_collect_small<la, ko, merge> -- batches that do
 *                       not fill the chip, whose launch lasts as long as ONE wavefront's serial path: role wavefronts that
 *                       share nothing, each playing the whole game and materialising one share of the outputs.  A workgroup
 *                       is a group of 64 / la boards: one scalars wavefront (which also builds the mask rows when merge = 1),
 *                       unless merged one mask wavefront, both with la lanes per board, and ko observation wavefronts of
 *                       la * ko lanes per board over 1 / ko of the group each.  Round 4's small-batch kernel is (4, 1, 0).  Since
 *                       round 6 only launches WITHOUT a mask trajectory (up to 8 192 boards) run it; with one: GBL_COLLECT_GROUP32.
 * (abi_version 2 added GBL_COLLECT_GROUP32; a consumer that switches on the code should treat unknown values as "another kernel".) */
#define GBL_COLLECT_STREAM 0
#define GBL_COLLECT_CACHED 1
#define GBL_COLLECT_PAIR 2
#define GBL_COLLECT_TRIO 4
/* (3 was GBL_COLLECT_SMALL until ABI version 1's role kernel got its forms; the code is retired, not reused: a consumer built
 *  against that header never reads k_collect3 as the small-batch kernel.  The name stays as an alias of the form it stood for.) */
#define GBL_COLLECT_SMALL GBL_COLLECT_ROLES(4, 1, 0) /* deprecated */
#define GBL_COLLECT_GROUP32 5 /* k_collect5 (round 6): groups of 32 boards, one playing wavefront + row wavefronts behind a hand-over ring */
#define GBL_COLLECT_ROLES(la, ko, merge) (1000 + 100 * (la) + 10 * (ko) + (merge))
#define GBL_COLLECT_IS_ROLES(variant) ((variant) >= 1000)
int gbl_collect_variant(int64_t n, uint32_t plies, int with_mask, int with_obs);
/* gbl_collect whose FIRST ply plays caller-supplied actions (first_actions int32[n]; NULL = gbl_collect): the collector
 * step of a policy that lives outside the library against masked-random replies -- the loops of the reference's trainers
 * with a random opponent (gobblet_rl/examples/example_tianshou_DQN.py: MultiAgentPolicyManager([agent, RandomPolicy])) --
 * in ONE launch per decision of the external policy instead of one per ply: with plies = 2, slot 0 receives the given
 * action's ply (illegal or out-of-range actions per illegal_mode, as gbl_step) and slot 1 the sampled reply (ply index
 * ply0 + 1), both with auto-reset; the policy reads slot 1's observation and mask for its next decision.  plies = 1 is
 * gbl_step_into with auto-reset; plies > 2 lets the sampler play on.  Everything else as gbl_collect. */
int gbl_collect_from(int8_t *state, int8_t *to_move, int8_t *done, const int32_t *first_actions, int32_t *actions_traj,
                     int8_t *winner_traj, int8_t *reward_traj, int8_t *done_traj, int8_t *to_move_traj, int8_t *mask_traj,
                     int8_t *obs_traj, int64_t n, int64_t ply_stride, int64_t tile_stride, uint64_t seed,
                     uint64_t env_base, uint32_t ply0, const uint32_t *ply_dev, uint32_t plies, int illegal_mode,
                     int64_t *counters, int32_t *turn, void *stream);
/* gbl_collect_from with the status byte of the caller's actions (first_status int8[n], GBL_STATUS_* as in gbl_step_ex;
 * NULL = gbl_collect_from; needs first_actions). */
int gbl_collect_from_ex(int8_t *state, int8_t *to_move, int8_t *done, const int32_t *first_actions, int8_t *first_status,
                        int32_t *actions_traj, int8_t *winner_traj, int8_t *reward_traj, int8_t *done_traj,
                        int8_t *to_move_traj, int8_t *mask_traj, int8_t *obs_traj, int64_t n, int64_t ply_stride,
                        int64_t tile_stride, uint64_t seed, uint64_t env_base, uint32_t ply0, const uint32_t *ply_dev,
                        uint32_t plies, int illegal_mode, int64_t *counters, int32_t *turn, void *stream);
/* *counter += by, enqueued on the stream (device uint32). */
int gbl_counter_add(uint32_t *counter, uint32_t by, void *stream);

/* Placement helper (no counterpart in the reference: it concerns where the caller puts the two large trajectory
 * arrays of gbl_collect in device memory).  On MI355X two write streams that lie in the same third of the HBM
 * address space (a 96 GiB class -- by its size one of the three die groups of the 12-high stacks) do not overlap:
 * gbl_collect then takes the SUM of what its observation stream and its mask stream take alone (33 us per ply at 2^20
 * boards), against 27 us when the two arrays lie in different classes.  Physical addresses are not
 * visible to a process, so the property is measured: the probe replays gbl_collect's store pattern (64 x 117 bytes
 * into a, 64 x 54 bytes into b per wavefront and slot) with both streams, with a alone and with b alone, and reports
 * the three times in microseconds.  us_both close to us_a + us_b: the two buffers share a class; us_both about 0.8
 * of the sum: they do not.  slot_boards > 0 (a multiple of 128) and plies: the geometry of the time-major trajectory
 * the buffers will hold -- slot t of a at t * slot_boards * 117 bytes, of b at t * slot_boards * 54 -- so that the
 * probe pairs exactly the regions the kernel writes together (it matters when an array straddles two classes);
 * slot_boards = 0: four slots spread over the whole of the smaller buffer.  OVERWRITES both buffers with zeros; blocks
 * the host until the probe has run (not capturable into a graph); buffers of less than about 64 MiB are too small for
 * a meaningful answer. */
int gbl_placement_probe(void *a, int64_t a_bytes, void *b, int64_t b_bytes, int64_t slot_boards, int plies, float *us_both,
                        float *us_a, float *us_b, void *stream);

/* Device memory blocks for the placement search (and for C callers without an allocator of their own): one hipMalloc
 * / hipFree each, on the calling thread's current device, outside any caching allocator -- a block handed back is free
 * for every other user of the device at once, with no process-wide cache flush.  The caller owns a block until it
 * frees it.  gbl_block_alloc returns GBL_ERR_HIP (and leaves *dev_ptr alone) when the device cannot provide the block:
 * the search then stops with what it has.  gbl_device_memory: hipMemGetInfo (either pointer may be NULL); the search
 * caps what it holds by a fraction of the free bytes. */
int gbl_block_alloc(int64_t bytes, void **dev_ptr);
int gbl_block_free(void *dev_ptr);
int gbl_device_memory(int64_t *free_bytes, int64_t *total_bytes);

#ifdef __cplusplus
}
#endif
#endif /* GOBBLET_HIP_H */
