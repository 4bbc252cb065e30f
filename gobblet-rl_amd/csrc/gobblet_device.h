// gobblet_device.h -- device-side building blocks of the gfx950 Gobblet kernels.
//
// Execution shape: ONE wavefront (64 lanes) per workgroup, ONE board per lane,
// a TILE of 64 consecutive boards per wavefront.  Rows of a tile are contiguous
// in HBM (env-major int8 rows of 27 / 54 / 117 bytes), so a tile is moved with
// full-width 16-byte-per-lane loads/stores (1 KiB per wave instruction) and
// (un)packed to one-row-per-lane through LDS:
//
//   HBM tile --dwordx4--> LDS image --ds_read_b32 + v_alignbyte--> row in VGPRs
//   row in VGPRs --v_alignbyte + ds_write_b32--> LDS image --dwordx4--> HBM tile
//
// Row sizes are odd (27, 54, 117 bytes) so lane l's row starts at byte l*ROWB,
// which is dword-aligned only every 4th (2nd) lane; the byte shift is done in
// registers with v_alignbyte_b32 so that every LDS access is an aligned dword
// and every HBM access a full 16-byte vector.  The 117-byte observation rows are
// the exception on the way out: they are mostly zeros, so the wave zero-fills
// the image and each lane scatters its <= 21 one-bytes (obs_scatter).
//
// The game logic runs on three 27-bit planes per board (bit c = cell c of
// Board.squares, c = 9*level + pos):  nz (cell occupied), neg (player_2's
// piece), odd (piece number odd = first piece of its size).
#pragma once
#ifndef GBL_HOST_EMU  // tests/emu/ compiles this header for the host with shims for the few builtins used
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>

namespace gbl {

constexpr int kTile = 64;   // boards per wavefront
constexpr int kCells = 27;  // Board.squares, board.py:33
constexpr int kActions = 54;
constexpr int kObs = 117;   // 3*3*13, gobblet.py:145

__device__ __forceinline__ uint32_t alignbyte(uint32_t hi, uint32_t lo, uint32_t nbytes)
{
    // ({hi,lo} >> 8*nbytes)[31:0], nbytes in 0..3
    return __builtin_amdgcn_alignbyte(hi, lo, nbytes);
}

// Blocks are dealt round-robin over the 8 XCDs (block b and b+8 share one); give every XCD a
// CONTIGUOUS range of tiles so that the 128-byte lines that straddle two tiles are completed inside
// one L2 instead of being written partially by two.  Speed only; any placement is correct.
// Measured (fused kernel): -2.5 % time at 2^20 boards, where the working set lives in the Infinity
// Cache, but +7 % at 2^22, where eight far-apart write fronts cost more in HBM than the shared
// lines save -- so batches beyond 2^21 boards keep the identity map.
constexpr int64_t kXcdRemapMaxTiles = (int64_t)1 << 15;

__device__ __forceinline__ int64_t xcd_tile(uint32_t bid, int64_t ntiles)
{
    if (ntiles > kXcdRemapMaxTiles) return (int64_t)bid;
    int64_t chunk = (ntiles + 7) >> 3;
    return (int64_t)(bid & 7u) * chunk + (bid >> 3);
}

// ---- tile <-> LDS image ---------------------------------------------------------------
// g points at the tile's first byte (16-byte aligned); rows = valid boards in the tile.
// Full tiles move as 16-byte vectors, lane l handling vectors l, l+64, ... : every wave
// instruction covers 1 KiB of contiguous HBM.  All loads of a tile are issued before the first
// dependent LDS write (and all LDS reads before the first store) so they overlap.
// NT = store policy (kStorePlain / kStoreStream / kStoreStreamDrop, see store16).  It is a TEMPLATE parameter on purpose:
// as a runtime flag the `if (nt) nontemporal-store else store` pair is merged by the optimiser into
// one plain store and the hint is silently lost.
typedef uint32_t __attribute__((ext_vector_type(4))) vec4u;

// Store policies of a tile's 16-byte vectors.  kStoreStream = non-temporal hint (`nt`).  kStoreStreamDrop = `nt sc1`:
// the same, and the line does not stay in the XCD's L2 -- for write-once streams and outputs nothing on the GPU reads
// back (2-4 % over `nt` on the trajectory stream, 0.5-2 % on the one-ply kernels' observation stream, in-process A/B
// scripts/ab_inproc.py; `sc1`, `sc0 sc1`, `sc0` without `nt`: 22 % slower at 2^20 boards).  There is no builtin for that
// combination on a global store; the raw BUFFER store builtin takes the cache-policy bits (gfx940 encoding: nt = 2,
// sc1 = 16), so the tile is addressed through a buffer descriptor of exactly its size.  (An inline-assembly store was
// tried first and is WRONG: the compiler does not know it reads its data registers and overwrites them too early.)
constexpr int kStorePlain = 0, kStoreStream = 1, kStoreStreamDrop = 2;

template <int POLICY>
__device__ __forceinline__ void store16(uint4 *dst, const uint4 &v)
{
    static_assert(POLICY == kStorePlain || POLICY == kStoreStream, "kStoreStreamDrop goes through a buffer descriptor (tile_out)");
#ifndef GBL_HOST_EMU
    if (POLICY == kStoreStream) {
        vec4u t = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(t, reinterpret_cast<vec4u *>(dst));
        return;
    }
#endif
    *dst = v;
}

// `between` runs after the tile's global loads have been issued and before the first dependent LDS
// write: work that does not depend on the tile (e.g. the Philox draw) placed there overlaps the
// load latency instead of lengthening the wave's serial path.
#ifndef GBL_HOST_EMU
#define GBL_PIN4(a) "v"((a).x), "v"((a).y), "v"((a).z), "v"((a).w)
template <int N>
__device__ __forceinline__ void pin_loads(const uint4 (&v)[N])
{
    // ONE asm statement per (up to) four vectors: separate statements would let the scheduler slide a
    // later vector's load behind an earlier statement and serialise the round trips again
    if constexpr (N == 1) asm volatile("" ::GBL_PIN4(v[0]) : "memory");
    else if constexpr (N == 2) asm volatile("" ::GBL_PIN4(v[0]), GBL_PIN4(v[1]) : "memory");
    else if constexpr (N == 3) asm volatile("" ::GBL_PIN4(v[0]), GBL_PIN4(v[1]), GBL_PIN4(v[2]) : "memory");
    else {
        asm volatile("" ::GBL_PIN4(v[0]), GBL_PIN4(v[1]), GBL_PIN4(v[2]), GBL_PIN4(v[3]) : "memory");
        if constexpr (N > 4) {
            uint4 rest[N - 4];
#pragma unroll
            for (int i = 0; i < N - 4; ++i) rest[i] = v[i + 4];
            pin_loads<N - 4>(rest);
        }
    }
}
#endif

struct NoWork {
    __device__ __forceinline__ void operator()() const {}
};

template <int ROWB, typename Between = NoWork>
__device__ __forceinline__ void tile_in(const int8_t *__restrict__ g, uint32_t *lds, int lane, int rows,
                                        Between between = Between())
{
    constexpr int NV = kTile * ROWB / 16, FULL = NV / 64, REM = NV % 64;
    if (rows == kTile) {
        const uint4 *gv = reinterpret_cast<const uint4 *>(g);
        uint4 *lv = reinterpret_cast<uint4 *>(lds);
        uint4 v[FULL + 1];
#pragma unroll
        for (int i = 0; i < FULL; ++i) v[i] = gv[lane + 64 * i];
        if (REM) v[FULL] = gv[lane < REM ? lane + 64 * FULL : NV - 1];  // branch-free (a branch would pin the wait)
        between();
#ifndef GBL_HOST_EMU
        // Keep the LDS writes (and their s_waitcnt) behind `between`, and keep EVERY load of the tile up
        // here, issued back to back: naming all loaded dwords as operands stops the optimiser from sinking
        // the last vector's load into the `lane < REM` branch below (or splitting it), and from moving the
        // other loads behind this point -- either would cost a second, serial round trip to memory.
        pin_loads<FULL + 1>(v);
#endif
#pragma unroll
        for (int i = 0; i < FULL; ++i) lv[lane + 64 * i] = v[i];
        if (REM && lane < REM) lv[lane + 64 * FULL] = v[FULL];
    } else {
        // ragged last tile: the whole vectors of its rows in one round trip (clamped to the last whole one: nothing beyond the rows
        // is read), then the last few bytes -- byte by byte all the way it was up to 27 serial round trips per launch (round 5)
        const int bytes = rows * ROWB, nvec = bytes >> 4;
        const uint4 *gv = reinterpret_cast<const uint4 *>(g);
        uint4 *lv = reinterpret_cast<uint4 *>(lds);
        int8_t *lb = reinterpret_cast<int8_t *>(lds);
        uint4 v[FULL + 1];
        const int last = nvec > 0 ? nvec - 1 : 0;
#pragma unroll
        for (int i = 0; i <= FULL; ++i) v[i] = nvec > 0 ? gv[lane + 64 * i < nvec ? lane + 64 * i : last] : uint4{0u, 0u, 0u, 0u};
        const int ti = (nvec << 4) + lane;
        const int8_t tb = ti < bytes ? g[ti] : (int8_t)0;
        between();
#pragma unroll
        for (int i = 0; i <= FULL; ++i)
            if (lane + 64 * i < nvec) lv[lane + 64 * i] = v[i];
        if (ti < bytes) lb[ti] = tb;
    }
}

template <int ROWB, int NT = kStorePlain>
__device__ __forceinline__ void tile_out(int8_t *__restrict__ g, const uint32_t *lds, int lane, int rows)
{
    constexpr int NV = kTile * ROWB / 16, FULL = NV / 64, REM = NV % 64;
    if (rows == kTile) {
        uint4 *gv = reinterpret_cast<uint4 *>(g);
        const uint4 *lv = reinterpret_cast<const uint4 *>(lds);
        // (with cached stores and a 117-byte row the compiler keeps v[] in 144 B of scratch; the
        // observation stream is always stored non-temporally, where it does not -- measured 0.3 %
        // faster than forms that avoid the scratch in the unused variant)
        uint4 v[FULL + 1];
#pragma unroll
        for (int i = 0; i < FULL; ++i) v[i] = lv[lane + 64 * i];
        if (REM && lane < REM) v[FULL] = lv[lane + 64 * FULL];
#ifndef GBL_HOST_EMU
        if constexpr (NT == kStoreStreamDrop) {
            // raw buffer stores with cache policy nt | sc1 (see store16); the descriptor covers exactly this tile
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(g, 0, kTile * ROWB, 0x00020000);
#pragma unroll
            for (int i = 0; i < FULL; ++i) {
                vec4u t = {v[i].x, v[i].y, v[i].z, v[i].w};
                __builtin_amdgcn_raw_buffer_store_b128(t, rs, (lane + 64 * i) * 16, 0, 2 | 16);
            }
            if (REM && lane < REM) {
                vec4u t = {v[FULL].x, v[FULL].y, v[FULL].z, v[FULL].w};
                __builtin_amdgcn_raw_buffer_store_b128(t, rs, (lane + 64 * FULL) * 16, 0, 2 | 16);
            }
            return;
        }
#endif
        constexpr int P = NT == kStoreStreamDrop ? kStoreStream : NT;  // (host emulation: a plain copy either way)
#pragma unroll
        for (int i = 0; i < FULL; ++i) store16<P>(&gv[lane + 64 * i], v[i]);
        if (REM && lane < REM) store16<P>(&gv[lane + 64 * FULL], v[FULL]);
    } else {
        // the ragged last tile: the whole 16-byte vectors of its rows, then the last few bytes (byte by byte all the way it was up
        // to 115 round trips, ~3 us, and a launch lasts as long as its slowest tile: 65 599 boards ran 4.66 us per ply where 65 536
        // run 1.70 -- round 5).  A compact loop, NOT the unrolled fetch-then-store of the whole tiles: unrolled, the compiler
        // re-balanced k_collect around it (111 -> 93 VGPRs) and the 2^20-board launch lost 2 %.
        const int bytes = rows * ROWB, nvec = bytes >> 4;
        uint4 *gv = reinterpret_cast<uint4 *>(g);
        const uint4 *lv = reinterpret_cast<const uint4 *>(lds);
#pragma nounroll
        for (int i = lane; i < nvec; i += 64) gv[i] = lv[i];
        const int8_t *lb = reinterpret_cast<const int8_t *>(lds);
        const int i = (nvec << 4) + lane;
        if (i < bytes) g[i] = lb[i];
    }
}

// tile_out for kernels that are NOT bound by their stores (gbl_collect_policy: the greedy decision dominates): at most
// CHUNK vectors per lane in registers at a time instead of the whole tile (the observation tile alone is 8 vectors = 32
// VGPRs, which there would cost a wavefront of occupancy).  Same bytes, same store policy.
template <int ROWB, int NT, int CHUNK>
__device__ __forceinline__ void tile_out_narrow(int8_t *__restrict__ g, const uint32_t *lds, int lane, int rows)
{
    constexpr int NV = kTile * ROWB / 16;
    if (rows != kTile) {
        tile_out<ROWB, NT>(g, lds, lane, rows);  // (ragged last tile: byte granular)
        return;
    }
    const uint4 *lv = reinterpret_cast<const uint4 *>(lds);
#ifndef GBL_HOST_EMU
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(g, 0, kTile * ROWB, 0x00020000);
#endif
#pragma unroll
    for (int i0 = 0; i0 < NV; i0 += 64 * CHUNK) {
        uint4 v[CHUNK];
#pragma unroll
        for (int u = 0; u < CHUNK; ++u) {
            const int i = i0 + 64 * u + lane;
            v[u] = lv[i < NV ? i : NV - 1];
        }
#pragma unroll
        for (int u = 0; u < CHUNK; ++u) {
            const int i = i0 + 64 * u + lane;
            if (i0 + 64 * u < NV && i < NV) {
#ifndef GBL_HOST_EMU
                if constexpr (NT == kStoreStreamDrop) {
                    vec4u t = {v[u].x, v[u].y, v[u].z, v[u].w};
                    __builtin_amdgcn_raw_buffer_store_b128(t, rs, i * 16, 0, 2 | 16);
                } else
#endif
                    store16<NT == kStoreStreamDrop ? kStoreStream : NT>(reinterpret_cast<uint4 *>(g) + i, v[u]);
            }
        }
    }
}

// tile_out in two halves, for a wavefront that takes a FULL tile's image over from another one: tile_fetch copies
// the image into registers (after which the image may be rebuilt), tile_store sends them out.
template <int ROWB>
__device__ __forceinline__ void tile_fetch(const uint32_t *lds, int lane, uint4 (&v)[kTile * ROWB / 16 / 64 + 1])
{
    constexpr int NV = kTile * ROWB / 16, FULL = NV / 64, REM = NV % 64;
    const uint4 *lv = reinterpret_cast<const uint4 *>(lds);
#pragma unroll
    for (int i = 0; i < FULL; ++i) v[i] = lv[lane + 64 * i];
    v[FULL] = lv[REM && lane < REM ? lane + 64 * FULL : lane];
}

// `bytes`: what of the image leaves -- the whole tile, or the WHOLE vectors of a ragged last tile's rows (a multiple of 16; its
// last few bytes go out with sub_tail while the image still stands)
template <int ROWB, int NT = kStorePlain>
__device__ __forceinline__ void tile_store(int8_t *__restrict__ g, const uint4 (&v)[kTile * ROWB / 16 / 64 + 1], int lane,
                                           int bytes = kTile * ROWB)
{
    constexpr int NV = kTile * ROWB / 16, FULL = NV / 64, REM = NV % 64;
#ifndef GBL_HOST_EMU
    if constexpr (NT == kStoreStreamDrop) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(g, 0, bytes, 0x00020000);  // (lanes beyond it: dropped)
#pragma unroll
        for (int i = 0; i < FULL; ++i) {
            vec4u t = {v[i].x, v[i].y, v[i].z, v[i].w};
            __builtin_amdgcn_raw_buffer_store_b128(t, rs, (lane + 64 * i) * 16, 0, 2 | 16);
        }
        if (REM && lane < REM) {
            vec4u t = {v[FULL].x, v[FULL].y, v[FULL].z, v[FULL].w};
            __builtin_amdgcn_raw_buffer_store_b128(t, rs, (lane + 64 * FULL) * 16, 0, 2 | 16);
        }
        return;
    }
#endif
    constexpr int P = NT == kStoreStreamDrop ? kStoreStream : NT;
    uint4 *gv = reinterpret_cast<uint4 *>(g);
    const int nvec = bytes >> 4;
#pragma unroll
    for (int i = 0; i < FULL; ++i)
        if (lane + 64 * i < nvec) store16<P>(&gv[lane + 64 * i], v[i]);
    if (REM && lane < REM && lane + 64 * FULL < nvec) store16<P>(&gv[lane + 64 * FULL], v[FULL]);
}

// ---- sub-tiles: BPS boards per wavefront, LPB = 64 / BPS lanes per board (batches that do not fill the chip) -------------------
// A batch of a few thousand boards is a few dozen 64-board tiles: most CUs idle and a launch lasts as long as ONE wavefront's
// serial path.  The role kernels (k_collect_small) cut a tile in LPB parts: lane = LPB * board + j, the LPB lanes of a board
// compute the game (sample, move, winner, legal mask) REDUNDANTLY -- no cross-lane traffic at all -- and share the per-row work:
// lane j builds bytes [64 j / LPB, 64 (j + 1) / LPB) of the board's mask row and drops every LPB-th channel of its observation
// row, and a sub-tile's images are 1 / LPB of a tile's (16 x 117 B = 117 vectors: two store instructions instead of eight).
// LPB = 4: 16 boards per wavefront (up to 8 192 boards), 2: 32 boards (round 5), 1: a whole tile per wavefront.  Sub-tile s of an
// array of ROWB-byte rows starts at byte BPS s ROWB, a multiple of 16 like a tile.
constexpr int kSub = 16;

template <int ROWB, int BPS = kSub, typename Between = NoWork>
__device__ __forceinline__ void sub_in(const int8_t *__restrict__ g, uint32_t *lds, int lane, int rows, Between between = Between())
{
    constexpr int NV = BPS * ROWB / 16;
    static_assert(NV <= 128 && (BPS * ROWB) % 16 == 0, "at most two vectors per lane");
    if (rows == BPS) {
        const uint4 *gv = reinterpret_cast<const uint4 *>(g);
        uint4 *lv = reinterpret_cast<uint4 *>(lds);
        uint4 v[2];
        v[0] = gv[lane < NV ? lane : NV - 1];  // branch-free, like tile_in
        v[1] = gv[NV > 64 && lane + 64 < NV ? lane + 64 : NV - 1];
        between();
#ifndef GBL_HOST_EMU
        pin_loads<2>(v);
#endif
        if (lane < NV) lv[lane] = v[0];
        if (NV > 64 && lane + 64 < NV) lv[lane + 64] = v[1];
    } else {  // ragged last sub-tile: its whole vectors (clamped: nothing beyond the rows is read), then the last few bytes (tile_in)
        const int bytes = rows * ROWB, nvec = bytes >> 4;
        const uint4 *gv = reinterpret_cast<const uint4 *>(g);
        uint4 *lv = reinterpret_cast<uint4 *>(lds);
        int8_t *lb = reinterpret_cast<int8_t *>(lds);
        uint4 v[2];
        const int last = nvec > 0 ? nvec - 1 : 0;
#pragma unroll
        for (int i = 0; i < 2; ++i) v[i] = nvec > 0 ? gv[lane + 64 * i < nvec ? lane + 64 * i : last] : uint4{0u, 0u, 0u, 0u};
        const int ti = (nvec << 4) + lane;
        const int8_t tb = ti < bytes ? g[ti] : (int8_t)0;
        between();
#pragma unroll
        for (int i = 0; i < 2; ++i)
            if (lane + 64 * i < nvec) lv[lane + 64 * i] = v[i];
        if (ti < bytes) lb[ti] = tb;
    }
}

template <int ROWB, int NT, int BPS = kSub>
__device__ __forceinline__ void sub_out(int8_t *__restrict__ g, const uint32_t *lds, int lane, int rows)
{
    constexpr int NV = BPS * ROWB / 16;
    static_assert(NV <= 128 && (BPS * ROWB) % 16 == 0, "at most two vectors per lane");
    if (rows == BPS) {
        const uint4 *lv = reinterpret_cast<const uint4 *>(lds);
        uint4 v[2];
        v[0] = lv[lane < NV ? lane : NV - 1];
        if (NV > 64) v[1] = lv[lane + 64 < NV ? lane + 64 : NV - 1];
#ifndef GBL_HOST_EMU
        if constexpr (NT == kStoreStreamDrop) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(g, 0, BPS * ROWB, 0x00020000);
            if (lane < NV) {
                vec4u t = {v[0].x, v[0].y, v[0].z, v[0].w};
                __builtin_amdgcn_raw_buffer_store_b128(t, rs, lane * 16, 0, 2 | 16);
            }
            if (NV > 64 && lane + 64 < NV) {
                vec4u t = {v[1].x, v[1].y, v[1].z, v[1].w};
                __builtin_amdgcn_raw_buffer_store_b128(t, rs, (lane + 64) * 16, 0, 2 | 16);
            }
            return;
        }
#endif
        constexpr int P = NT == kStoreStreamDrop ? kStoreStream : NT;
        uint4 *gv = reinterpret_cast<uint4 *>(g);
        if (lane < NV) store16<P>(&gv[lane], v[0]);
        if (NV > 64 && lane + 64 < NV) store16<P>(&gv[lane + 64], v[1]);
    } else {
        const int bytes = rows * ROWB;
        const int8_t *lb = reinterpret_cast<const int8_t *>(lds);
        for (int i = lane; i < bytes; i += 64) g[i] = lb[i];
    }
}

// A sub-tile's output image (mask / observation rows of BPS boards: up to 468 vectors) on its way out: fetched from LDS into
// registers at the end of one ply, stored behind the next ply's sample and move.  Named members, not an array: an array handed
// around by reference was "promoted" to LDS by the compiler (round 4).
template <int N>
struct SubVecs {
    uint4 head;
    SubVecs<N - 1> tail;
    template <int I>
    __device__ __forceinline__ uint4 &at()
    {
        if constexpr (I == 0) return head;
        else return tail.template at<I - 1>();
    }
};
template <>
struct SubVecs<0> {
};

template <int ROWB, int BPS>
constexpr int sub_vectors() { return (BPS * ROWB / 16 + 63) / 64; }

template <int ROWB, int BPS, int I = 0>
__device__ __forceinline__ void sub_fetch(const uint32_t *lds, int lane, SubVecs<sub_vectors<ROWB, BPS>()> &v)
{
    constexpr int NV = BPS * ROWB / 16, N = sub_vectors<ROWB, BPS>();
    if constexpr (I < N) {
        const uint4 *lv = reinterpret_cast<const uint4 *>(lds);
        const int i = lane + 64 * I;
        v.template at<I>() = lv[64 * I + 63 < NV ? i : (i < NV ? i : NV - 1)];
        sub_fetch<ROWB, BPS, I + 1>(lds, lane, v);
    }
}

// `bytes`: what of the image leaves -- BPS * ROWB for a whole sub-tile; the WHOLE 16-byte vectors of a ragged one's rows (a multiple
// of 16; its last few bytes go out with sub_tail, so that a ragged sub-tile takes the same deferred register path as a whole one:
// the launch of a latency-bound batch lasts as long as its slowest wavefront, and the ragged one used to copy its rows LDS -> HBM
// vector by vector inside the ply -- 16 447 boards 1.00 us per ply where 16 384 take 0.74 and 20 480 take 0.84).
template <int ROWB, int NT, int BPS, int I = 0>
__device__ __forceinline__ void sub_store(int8_t *__restrict__ g, SubVecs<sub_vectors<ROWB, BPS>()> &v, int lane, int bytes = BPS * ROWB)
{
    constexpr int N = sub_vectors<ROWB, BPS>();
    if constexpr (I < N) {
        const int i = lane + 64 * I;
        const uint4 &x = v.template at<I>();
#ifndef GBL_HOST_EMU
        if constexpr (NT == kStoreStreamDrop) {
            // (no predicate: the descriptor covers exactly the bytes that leave, and the hardware drops a lane whose offset lies
            //  beyond it -- the lanes past the last vector)
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(g, 0, bytes, 0x00020000);
            vec4u t = {x.x, x.y, x.z, x.w};
            __builtin_amdgcn_raw_buffer_store_b128(t, rs, i * 16, 0, 2 | 16);
        } else
#endif
        {
            constexpr int P = NT == kStoreStreamDrop ? kStoreStream : NT;
            if (16 * i + 16 <= bytes) store16<P>(reinterpret_cast<uint4 *>(g) + i, x);
        }
        sub_store<ROWB, NT, BPS, I + 1>(g, v, lane, bytes);
    }
}

// the last bytes of a ragged sub-tile's rows, behind its whole vectors (< 16 bytes: one byte per lane, from the LDS image)
__device__ __forceinline__ void sub_tail(int8_t *__restrict__ g, const uint32_t *lds, int lane, int bytes)
{
    const int i = (bytes & ~15) + lane;
    if (i < bytes) g[i] = reinterpret_cast<const int8_t *>(lds)[i];
}

// zero image of a sub-tile's BPS observation rows
template <int BPS = kSub>
__device__ __forceinline__ void sub_obs_zero(uint32_t *img, int lane)
{
    constexpr int NV = BPS * kObs / 16;
    uint4 *lv = reinterpret_cast<uint4 *>(img);
    const uint4 z = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < (NV + 63) / 64; ++i) lv[(64 * i + 63 < NV || lane + 64 * i < NV) ? lane + 64 * i : NV - 1] = z;  // (no branch)
}

// Orders this wave's LDS accesses across lanes.  A workgroup is ONE wavefront, whose LDS
// instructions execute in issue order, so no s_barrier and no vmcnt drain is needed (a
// __syncthreads() would also wait for every outstanding global store); the fence only keeps the
// compiler from moving LDS accesses across this point.
__device__ __forceinline__ void wave_lds_fence()
{
#ifndef GBL_HOST_EMU
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#endif
}

// Workgroup barrier for kernels whose wavefronts share LDS only: waits for the caller's LDS traffic (lgkmcnt) but NOT for
// its outstanding global stores (vmcnt), which __syncthreads() would -- a wavefront's trajectory stores stay in flight
// across the rendezvous.
__device__ __forceinline__ void pair_barrier()
{
#ifndef GBL_HOST_EMU
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}

// ---- LDS image <-> one row per lane ---------------------------------------------------
// Lane l's row = image bytes [l*ROWB, (l+1)*ROWB).  r[] holds the row as
// little-endian dwords; bytes past ROWB in the last dword are unspecified.
// The image needs 4 readable bytes of slack after the tile.
template <int ROWB>
__device__ __forceinline__ void row_load(const uint32_t *lds, int lane, uint32_t (&r)[(ROWB + 3) / 4])
{
    constexpr int NW = (ROWB + 3) / 4;
    uint32_t byteoff = (uint32_t)lane * ROWB;
    uint32_t q0 = byteoff >> 2, sh = byteoff & 3u;
    uint32_t x[NW + 1];
#pragma unroll
    for (int k = 0; k <= NW; ++k) x[k] = lds[q0 + k];
#pragma unroll
    for (int k = 0; k < NW; ++k) r[k] = alignbyte(x[k + 1], x[k], sh);
}

// Each lane owns the image dwords whose FIRST byte lies in its row; the last
// owned dword may end with up to 3 bytes of the next lane's row, fetched with
// one DPP/shuffle of that lane's first dword.  64*ROWB is a multiple of 4, so
// lane 63's last dword ends exactly at the tile end.
template <int ROWB>
__device__ __forceinline__ void row_stage(uint32_t *lds, int lane, const uint32_t (&d)[(ROWB + 3) / 4])
{
    constexpr int NW = (ROWB + 3) / 4;
    constexpr int TAIL = ROWB % 4;
    static_assert(TAIL != 0, "dword-multiple rows need no shifting");
    uint32_t byteoff = (uint32_t)lane * ROWB;
    uint32_t h = (4u - (byteoff & 3u)) & 3u;  // leading bytes that belong to the previous lane's last dword
    uint32_t qfirst = (byteoff + 3u) >> 2;
    uint32_t cnt = (ROWB - h + 3u) >> 2;      // NW or NW-1
    uint32_t nd0 = __shfl_down(d[0], 1);
    uint32_t e_last = (d[NW - 1] & ((1u << (8 * TAIL)) - 1u)) | (nd0 << (8 * TAIL));
    uint32_t e_over = nd0 >> (8 * (4 - TAIL));
#pragma unroll
    for (int k = 0; k < NW; ++k) {
        uint32_t lo = (k == NW - 1) ? e_last : d[k];
        uint32_t hi = (k + 1 == NW - 1) ? e_last : (k + 1 == NW) ? e_over : d[k + 1];
        uint32_t w = alignbyte(hi, lo, h);
        if (k < NW - 1 || cnt == NW) lds[qfirst + k] = w;
    }
}

// ---- bit planes ------------------------------------------------------------------------
struct Planes {
    uint32_t nz, neg, odd;  // 27 bits each; neg / odd are only meaningful where nz is set
};

// r[0..6] = the 27 state bytes (r[6] byte 3 ignored).  Exact per-byte tests in
// SWAR, then v_dot4_u32_u8 with weights (1,2,4,8 | 16,32,64,128) gathers one
// flag bit per byte of two dwords into 8 contiguous bits.
__device__ __forceinline__ Planes make_planes(const uint32_t (&r)[7])
{
    Planes p{0u, 0u, 0u};
#pragma unroll
    for (int j = 0; j < 7; j += 2) {
        uint32_t anz = 0, ang = 0, aod = 0;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (j + u >= 7) break;
            uint32_t x = r[j + u];
            if (j + u == 6) x &= 0x00FFFFFFu;
            uint32_t w = u ? 0x80402010u : 0x08040201u;
            uint32_t t = (((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u;  // 0x80 where byte != 0
            anz = __builtin_amdgcn_udot4(t, w, anz, false);
            ang = __builtin_amdgcn_udot4(x & 0x80808080u, w, ang, false);
            aod = __builtin_amdgcn_udot4(x & 0x01010101u, w, aod, false);
        }
        p.nz |= (anz >> 7) << (4 * j);
        p.neg |= (ang >> 7) << (4 * j);
        p.odd |= aod << (4 * j);
    }
    return p;
}

// Board.check_for_winner, board.py:183-194, over the tops of get_flatboard
// (board.py:159-177).  The reference walks the 8 lines in a fixed order with no
// early exit, so the matching line with the HIGHEST index decides; one line
// cannot match both colours, so comparing the two 8-bit match masks as
// integers picks that line's colour.
//
// line_matches: the 8-bit mask of the lines (bit l = line l of board.py:135-153) on which every top piece is `colour`'s
// (0 = player_1).  side = the colour's cells (nz & ~neg / nz & neg), passed in so that a caller whose lanes split the two colours
// selects it with one operation.
__device__ __forceinline__ uint32_t line_matches_of(uint32_t side, uint32_t nz)
{
    uint32_t o1 = (nz >> 9) & 0x1FFu, o2 = (nz >> 18) & 0x1FFu;
    uint32_t t = ((side >> 18) & 0x1FFu) | (~o2 & (((side >> 9) & 0x1FFu) | (~o1 & (side & 0x1FFu))));  // the colour's tops
    // board.py:135-153: (0,1,2) (3,4,5) (6,7,8) (0,3,6) (1,4,7) (2,5,8) (0,4,8) (2,4,6).
    // Three lines per word in 10-bit fields (bit 9 = guard): adding 0x1FF to "squares of the line the
    // side lacks" carries into the guard iff something is missing.  Lines (0,3,6) / (1,4,7) / (2,5) share
    // a word so that shifting the guards by 9 / 8 / 7 lays the eight match bits out in line order.
    constexpr uint32_t L[8] = {0x007u, 0x038u, 0x1C0u, 0x049u, 0x092u, 0x124u, 0x111u, 0x054u};
    constexpr uint32_t LOW3 = 0x00100401u, G3 = LOW3 << 9, F3 = LOW3 * 0x1FFu;
    constexpr uint32_t WA = L[0] | (L[3] << 10) | (L[6] << 20), WB = L[1] | (L[4] << 10) | (L[7] << 20);
    constexpr uint32_t WC = L[2] | (L[5] << 10), NONE = 1u << 20;  // third field of WC: always "missing"
    uint32_t n = ~(t | (t << 10) | (t << 20));
    return ((G3 & ~((WA & n) + F3)) >> 9) | ((G3 & ~((WB & n) + F3)) >> 8) | ((G3 & ~(((WC & n) | NONE) + F3)) >> 7);
}

__device__ __forceinline__ int winner_from_matches(uint32_t m1, uint32_t m2) { return m2 > m1 ? -1 : (m1 > m2 ? 1 : 0); }

__device__ __forceinline__ int winner_of(const Planes &p)
{
    return winner_from_matches(line_matches_of(p.nz & ~p.neg, p.nz), line_matches_of(p.nz & p.neg, p.nz));
}

// winner_of for kernels whose lanes come in PAIRS that hold the same board (the role kernels with two or four lanes per board):
// the even lane of a pair walks player_1's lines, the odd lane player_2's, and one DPP quad permutation ([1, 0, 3, 2]) hands
// each the other's match mask -- half the line arithmetic per lane (~28 instead of ~48 instructions on the ply's serial chain).
// j = the lane's index among its board's lanes (only its parity is used).  The host emulation, whose lanes run one after the
// other, computes both masks itself: the same function of the same board.
__device__ __forceinline__ int winner_of_pair(const Planes &p, int j)
{
#ifndef GBL_HOST_EMU
    const uint32_t flip = (j & 1) ? 0u : ~0u;                                     // (a per-lane constant)
    const uint32_t mine = line_matches_of(p.nz & (p.neg ^ flip), p.nz);           // colour j & 1
    const uint32_t theirs = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mine, 0xB1, 0xf, 0xf, false);  // quad_perm [1, 0, 3, 2]
    const int w = mine > theirs ? 1 : (theirs > mine ? -1 : 0);                   // seen from colour j & 1
    return (j & 1) ? -w : w;
#else
    (void)j;
    return winner_of(p);
#endif
}

// 54-bit legal mask of agent `mover` (0 / 1): raw_env._legal_moves,
// gobblet.py:223-228 = 54 x Board.is_legal, board.py:82-115.
//   mask[9*piece + pos] = piece not covered (board.py:90-102, check_covered :203-220)
//                         and size(piece) > size(top at pos) (board.py:106-115)
__device__ __forceinline__ uint64_t legal54(const Planes &p, int mover)
{
    uint32_t o0 = p.nz & 0x1FFu, o1 = (p.nz >> 9) & 0x1FFu, o2 = (p.nz >> 18) & 0x1FFu;
    uint32_t cov = (o0 & (o1 | o2)) | ((o1 & o2) << 9);  // covered cells (levels 0,1)
    uint32_t mine = mover ? (p.nz & p.neg) : (p.nz & ~p.neg);
    uint32_t bl = mine & cov;
    uint32_t blo = bl & p.odd, ble = bl & ~p.odd;
    uint32_t ok0 = ~(o0 | o1 | o2) & 0x1FFu, ok1 = ~(o1 | o2) & 0x1FFu, ok2 = ~o2 & 0x1FFu;
    uint32_t g0 = (blo & 0x1FFu) ? 0u : ok0;      // piece 1
    uint32_t g1 = (ble & 0x1FFu) ? 0u : ok0;      // piece 2
    uint32_t g2 = (blo & 0x3FE00u) ? 0u : ok1;    // piece 3
    uint32_t g3 = (ble & 0x3FE00u) ? 0u : ok1;    // piece 4
    uint32_t lo = g0 | (g1 << 9) | (g2 << 18) | (g3 << 27);
    uint32_t hi = (g3 >> 5) | (ok2 << 4) | (ok2 << 13);  // pieces 5, 6 are never covered
    return ((uint64_t)hi << 32) | lo;
}

// Board.play_turn(mover, a), board.py:118-132, given that `a` is legal: clear the cell holding the
// piece (if placed), write it at 9*level + pos.  The planes are updated here; the two cells of the
// 27-byte row that change are returned for whoever holds the row.
struct MoveCells {
    uint32_t cold, cnew, val;  // cell to clear (only if `had`), cell to write, signed piece number as a byte
    bool had;                  // the piece was on the board already
};

__device__ __forceinline__ MoveCells move_planes(Planes &p, int mover, uint32_t a)
{
    // a / 9 for a < 54 through a 24-bit multiply (full rate; the compiler turns __umul24 back into the quarter-rate v_mul_lo_u32
    // wherever it cannot bound `a`, hence the instruction by name)
#ifndef GBL_HOST_EMU
    uint32_t a57;
    asm("v_mul_u32_u24 %0, %1, 57" : "=v"(a57) : "v"(a));
    uint32_t pi = a57 >> 9;
#else
    uint32_t pi = (a * 57u) >> 9;
#endif
    uint32_t q = a - 9u * pi;
    uint32_t k = pi >> 1;
    uint32_t first = (~pi) & 1u;   // piece number odd
    uint32_t mine = mover ? (p.nz & p.neg) : (p.nz & ~p.neg);
    uint32_t ploc = mine & (first ? p.odd : ~p.odd) & (0x1FFu << (9u * k));
    uint32_t cnew = 9u * k + q, bit = 1u << cnew;
    p.nz = (p.nz & ~ploc) | bit;
    p.neg = (p.neg & ~ploc & ~bit) | (mover ? bit : 0u);
    p.odd = (p.odd & ~ploc & ~bit) | (first ? bit : 0u);
    MoveCells m;
    m.had = ploc != 0;
    m.cold = m.had ? (uint32_t)__builtin_ctz(ploc) : 0u;
    m.cnew = cnew;
    m.val = (mover ? (0u - (pi + 1u)) : (pi + 1u)) & 0xFFu;
    return m;
}

// No row: a wavefront that plays the game without writing the state back (the role kernels' row wavefronts) patches nothing.
struct NoRow {
    __device__ __forceinline__ void apply(const MoveCells &) const {}
    __device__ __forceinline__ void reset() const {}
};

// The row in 7 registers (board-level kernels: the row is re-staged into a fresh image anyway).
struct RegRow {
    uint32_t (&r)[7];
    __device__ __forceinline__ void apply(const MoveCells &m)
    {
        uint32_t cold = m.had ? m.cold : 63u * 4u;
        uint32_t mold = ~(0xFFu << (8u * (cold & 3u)));
        uint32_t mnew = ~(0xFFu << (8u * (m.cnew & 3u)));
        uint32_t vnew = m.val << (8u * (m.cnew & 3u));
#pragma unroll
        for (uint32_t j = 0; j < 7; ++j) {
            uint32_t x = r[j];
            x = (j == (cold >> 2)) ? (x & mold) : x;
            x = (j == (m.cnew >> 2)) ? ((x & mnew) | vnew) : x;
            r[j] = x;
        }
    }
    __device__ __forceinline__ void reset()
    {
#pragma unroll
        for (int j = 0; j < 7; ++j) r[j] = 0;
    }
};

// The row where the tile load put it, in the LDS image (step kernels: the image goes back out as it
// is, so a move costs two byte stores and a reset one 27-byte clear instead of rebuilding and
// re-staging 7 dwords).  Rows are 27 bytes apart: the clear relies on unaligned LDS stores (gfx950).
struct ImageRow {
    uint8_t *row;
    __device__ __forceinline__ void apply(const MoveCells &m) const
    {
        // (no branch: a piece that comes from the hand clears the cell it is about to fill -- LDS stores of a lane land in order)
        row[m.had ? m.cold : m.cnew] = 0;
        row[m.cnew] = (uint8_t)m.val;
    }
    __device__ __forceinline__ void reset() const { __builtin_memset(row, 0, kCells); }
};

__device__ __forceinline__ void apply_move(Planes &p, uint32_t (&r)[7], int mover, uint32_t a)
{
    RegRow{r}.apply(move_planes(p, mover, a));
}

// The 27 state bytes of a board from its planes (the inverse of make_planes on contract states: a level-k cell holds 0 or
// +-(2k+1), +-(2k+2)): a kernel that plays many plies on the planes alone writes the state back ONCE, from here, instead of
// patching a byte row every ply (k_collect5).  r[6]'s top byte is 0.
__device__ __forceinline__ void planes_to_row(const Planes &p, uint32_t (&r)[7])
{
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        uint32_t w = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = 4 * j + u;
            if (c < kCells) {
                const int k = c / 9;
                const uint32_t nz = (p.nz >> c) & 1u, ng = (p.neg >> c) & nz, od = (p.odd >> c) & nz;
                const uint32_t v = nz * (uint32_t)(2 * k + 2) - od;   // 2k+2, or 2k+1 where the piece number is odd; 0 where empty
                w |= (ng ? (0u - v) & 0xFFu : v) << (8 * u);
            }
        }
        r[j] = w;
    }
}

// ---- row encoders ----------------------------------------------------------------------
// 54 mask bits -> 54 bytes of 0/1 (14 dwords; bytes 54,55 are zero)
__device__ __forceinline__ void mask_row(uint64_t m, uint32_t (&d)[14])
{
    uint32_t lo = (uint32_t)m, hi = (uint32_t)(m >> 32);
#pragma unroll
    for (int j = 0; j < 14; ++j) {
        uint32_t nib = ((j < 8 ? lo : hi) >> (4 * (j & 7))) & 0xFu;
        d[j] = __umul24(nib, 0x00204081u) & 0x01010101u;  // bit i -> byte i
    }
}

// raw_env.observe, gobblet.py:179-208: obs[pos][ch], int8[9][13]:
//   ch 0..5 : cell(level ch/2, pos) == +(ch+1) seen from the observer (own pieces)
//   ch 6..11: cell(level (ch-6)/2, pos) == -(ch-5)                     (opponent's)
//   ch 12   : the observer's agent index
// written SPARSELY into a tile's LDS image: an observation has at most 12
// ones among channels 0..11 (one per piece on the board) plus the 9 bytes of channel 12, so instead
// of composing 30 dwords per board (~2 VALU per byte) the wave zero-fills the image with
// 16-byte stores and every lane then drops single bytes at  lane*117 + 13*pos + ch  -- one
// predicated ds_write_b8 per piece.  Relies on the state contract (a piece number occurs at most
// once), like everything else.  Call obs_image_zero, fence, obs_scatter, fence, tile_out.
__device__ __forceinline__ void obs_image_zero(uint32_t *img, int lane)
{
    constexpr int NV = kTile * kObs / 16, FULL = NV / 64, REM = NV % 64;
    uint4 *lv = reinterpret_cast<uint4 *>(img);
    const uint4 z = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < FULL; ++i) lv[lane + 64 * i] = z;
    if (REM && lane < REM) lv[lane + 64 * FULL] = z;
}

__device__ __forceinline__ void obs_scatter_row(uint8_t *row, const Planes &p, int observer);

__device__ __forceinline__ void obs_scatter(uint32_t *img, int lane, const Planes &p, int observer)
{
    obs_scatter_row(reinterpret_cast<uint8_t *>(img) + lane * kObs, p, observer);
}

// row: the board's 117 observation bytes, zero on entry.  Straight-line code: a piece that is not on the board writes a 0 to square
// 8 of its own channel, which no other piece can set, and channel 12 is written whoever observes (round 5: the predicated form --
// one EXEC round trip per piece -- cost the latency-bound kernels 10-15 % of a ply; the HBM-bound ones do not care, +-1-2 %).
__device__ __forceinline__ void obs_scatter_row(uint8_t *row, const Planes &p, int observer)
{
    uint32_t pos = p.nz & ~p.neg, ngv = p.nz & p.neg;
    uint32_t own = observer ? ngv : pos, opp = observer ? pos : ngv;
    uint32_t X[4] = {own & p.odd, own & ~p.odd, opp & p.odd, opp & ~p.odd};  // A B C D
#pragma unroll
    for (int ch = 0; ch < 12; ++ch) {
        int k = (ch % 6) / 2;
        uint32_t grp = (X[(ch < 6 ? 0 : 2) + (ch & 1)] >> (9 * k)) & 0x1FFu;
        row[13 * __builtin_ctz(grp | 0x100u) + ch] = (uint8_t)(grp ? 1u : 0u);
    }
#pragma unroll
    for (int q = 0; q < 9; ++q) row[13 * q + 12] = (uint8_t)observer;
}

// Lane j (of the LPB of a board) drops channels j, j + LPB, ... and the channel-12 bytes of squares j, j + LPB, ... of the board's
// observation row (zero on entry): obs_scatter_row dealt over the board's lanes.
template <int LPB = 4>
__device__ __forceinline__ void obs_scatter_part(uint8_t *row, const Planes &p, int observer, int j)
{
    static_assert(LPB == 1 || LPB == 2 || LPB == 4, "lanes per board");
    if constexpr (LPB == 1) {
        (void)j;
        obs_scatter_row(row, p, observer);
    } else {
        // STRAIGHT-LINE code (the role kernels' wavefronts run alone on their SIMDs: a predicated region costs them its EXEC
        // round trips, not its instructions): a piece that is not on the board writes a 0 to square 8 of its own channel, which
        // no other piece can set; a channel-12 byte past square 8 is redirected to the lane's first square (the same value again).
        const uint32_t pos = p.nz & ~p.neg, ngv = p.nz & p.neg;
        const uint32_t own = observer ? ngv : pos, opp = observer ? pos : ngv;
        // channel ch = j + LPB i: side = ch / 6, level = (ch % 6) / 2, parity = ch & 1 = j & 1 (LPB is even)
        const uint32_t oddsel = (j & 1) ? ~p.odd : p.odd;
        const uint32_t ownsel = own & oddsel, oppsel = opp & oddsel;
#pragma unroll
        for (int i = 0; i < 12 / LPB; ++i) {
            const int ch = j + LPB * i;                  // 0..11
            const uint32_t side = ch >= 6 ? oppsel : ownsel;
            const int k = (ch >= 6 ? ch - 6 : ch) >> 1;
            const uint32_t grp = (side >> (9 * k)) & 0x1FFu;
            row[13 * __builtin_ctz(grp | 0x100u) + ch] = (uint8_t)(grp ? 1u : 0u);
        }
#pragma unroll
        for (int i = 0; i < (9 + LPB - 1) / LPB; ++i) {
            const int q = j + LPB * i;
            row[13 * (q < 9 ? q : j) + 12] = (uint8_t)observer;
        }
    }
}

__device__ __forceinline__ void obs_scatter_quad(uint8_t *row, const Planes &p, int observer, int j)
{
    obs_scatter_part<4>(row, p, observer, j);
}

// Lane j writes S = 64 / LPB bytes of the board's 54-byte mask row at `row` -- bytes [S j, S j + S), the last lane the LAST S
// bytes of the row (it overlaps its neighbour's share with the same values: no branch, see obs_scatter_part); one lane per board:
// all 54.  Rows are 2-byte aligned: unaligned LDS stores, like ImageRow::reset.
template <int LPB = 4>
__device__ __forceinline__ void mask_row_part(uint8_t *row, uint64_t m, int j)
{
    static_assert(LPB == 1 || LPB == 2 || LPB == 4, "lanes per board");
    constexpr int S = 64 / LPB, NW = S / 4;  // bytes per lane, dwords per lane
    const int first = LPB == 1 ? 0 : (S * j < kActions - S ? S * j : kActions - S);  // the lane's first byte (= bit of m)
    const uint64_t part = LPB == 1 ? m : (m >> first);
    uint32_t d[NW];
#pragma unroll
    for (int k = 0; k < NW; ++k) {
        const uint32_t nib = (uint32_t)(part >> (4 * k)) & 0xFu;
        d[k] = __umul24(nib, 0x00204081u) & 0x01010101u;  // bit i -> byte i
    }
    if constexpr (LPB == 1) __builtin_memcpy(row, d, kActions);
    else __builtin_memcpy(row + first, d, S);
}

__device__ __forceinline__ void mask_row_quad(uint8_t *row, uint64_t m, int j) { mask_row_part<4>(row, m, j); }

// Board.get_flatboard, board.py:159-177: signed piece number of the top piece per
// square, as a 9-byte row in 3 dwords.
__device__ __forceinline__ void flat_row(const Planes &p, const uint32_t (&r)[7], uint32_t (&d)[3])
{
    uint32_t o1 = (p.nz >> 9) & 0x1FFu, o2 = (p.nz >> 18) & 0x1FFu;
    d[0] = d[1] = d[2] = 0;
#pragma unroll
    for (int q = 0; q < 9; ++q) {
        uint32_t b0 = (r[q >> 2] >> (8 * (q & 3))) & 0xFFu;
        uint32_t b1 = (r[(9 + q) >> 2] >> (8 * ((9 + q) & 3))) & 0xFFu;
        uint32_t b2 = (r[(18 + q) >> 2] >> (8 * ((18 + q) & 3))) & 0xFFu;
        uint32_t v = ((o2 >> q) & 1u) ? b2 : (((o1 >> q) & 1u) ? b1 : b0);
        d[q >> 2] |= v << (8 * (q & 3));
    }
}

// Board.check_covered, board.py:203-220: 27 bytes of 0/1 in 7 dwords.
__device__ __forceinline__ void covered_row(const Planes &p, uint32_t (&d)[7])
{
    uint32_t o0 = p.nz & 0x1FFu, o1 = (p.nz >> 9) & 0x1FFu, o2 = (p.nz >> 18) & 0x1FFu;
    uint32_t cov = (o0 & (o1 | o2)) | ((o1 & o2) << 9);
#pragma unroll
    for (int j = 0; j < 7; ++j) d[j] = __umul24((cov >> (4 * j)) & 0xFu, 0x00204081u) & 0x01010101u;
}

// State-contract check (not on the hot path): bit 0 = some cell of level k holds a value other than
// 0, +-(2k+1), +-(2k+2); bit 1 = some piece number occurs twice -- the condition on which the
// reference raises Exception("PIECE HAS BEEN USED TWICE") (board.py:94-95).
__device__ __forceinline__ uint32_t validate_row(const uint32_t (&r)[7])
{
    uint32_t bad = 0;
#pragma unroll
    for (int c = 0; c < kCells; ++c) {
        int v = (int)(int8_t)((r[c >> 2] >> (8 * (c & 3))) & 0xFFu);
        int a = v < 0 ? -v : v, k = c / 9;
        bad |= (v != 0 && a != 2 * k + 1 && a != 2 * k + 2) ? 1u : 0u;
    }
    Planes p = make_planes(r);
    uint32_t pos = p.nz & ~p.neg, ngv = p.nz & p.neg;
    uint32_t X[4] = {pos & p.odd, pos & ~p.odd, ngv & p.odd, ngv & ~p.odd};
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int k = 0; k < 3; ++k) bad |= (__popc((X[g] >> (9 * k)) & 0x1FFu) > 1) ? 2u : 0u;
    return bad;
}

// ---- Philox4x32-10 (Salmon et al., SC'11); known answers are checked in tests/ ---------------
struct Draw4 {
    uint32_t w[4];
};

__device__ __forceinline__ Draw4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                               uint32_t k1)
{
#pragma unroll
    for (int rnd = 0; rnd < 10; ++rnd) {
        // one 32 x 32 -> 64 multiply per product (v_mad_u64_u32: 2.9 cycles of SIMD time at full occupancy, the
        // same as v_mul_hi_u32 or v_mul_lo_u32 alone -- scripts/microbench/valu_rates.hip) instead of two
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c0 = n0; c1 = (uint32_t)p1; c2 = n2; c3 = (uint32_t)p0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return Draw4{{c0, c1, c2, c3}};
}

// index of the k-th (0-based) set bit of w; k < popcount(w)
__device__ __forceinline__ uint32_t kth_bit32(uint32_t w, uint32_t k)
{
    uint32_t pos = 0;
#pragma unroll
    for (int s = 16; s >= 1; s >>= 1) {
        uint32_t c = __popc(w & ((1u << s) - 1u));
        bool up = k >= c;
        k = up ? k - c : k;
        w = up ? (w >> s) : w;
        pos += up ? s : 0;
    }
    return pos;
}

// masked-uniform draw (see gbl_sample in include/gobblet_hip.h), in two halves: the 32-bit draw
// depends only on (seed, board id, ply) -- a kernel can compute it while its tile is still in
// flight -- and the pick needs the mask.  One Philox block serves four consecutive plies of a board
// (word ply & 3 of the block with counter ply >> 2): a kernel that plays several plies per launch
// runs the generator once per four.
// `stream` (counter word 3) separates the consumers of one (seed, board) pair: kStreamEnv = the masked-random
// actions of gbl_sample / gbl_rollout, kStreamGreedy = the greedy policy's fallback draw (gbl_greedy_act).
constexpr uint32_t kStreamEnv = 0u, kStreamGreedy = 1u;

__device__ __forceinline__ Draw4 draw_block(uint64_t seed, uint64_t env_id, uint32_t ply, uint32_t stream = kStreamEnv)
{
    return philox4x32_10((uint32_t)env_id, (uint32_t)(env_id >> 32), ply >> 2, stream, (uint32_t)seed,
                         (uint32_t)(seed >> 32));
}

__device__ __forceinline__ uint32_t draw_word(const Draw4 &d, uint32_t ply)
{
    uint32_t lo = (ply & 1u) ? d.w[1] : d.w[0], hi = (ply & 1u) ? d.w[3] : d.w[2];
    return (ply & 2u) ? hi : lo;
}

// which word of the block draw_word picks: ply & 3 = 0, 1, 2, 3 -> w[0], w[1], w[2], w[3]
__device__ __forceinline__ uint32_t draw_word_index(uint32_t ply) { return ply & 3u; }

__device__ __forceinline__ uint32_t draw32(uint64_t seed, uint64_t env_id, uint32_t ply, uint32_t stream = kStreamEnv)
{
    return draw_word(draw_block(seed, env_id, ply, stream), ply);
}

// legal54 of the empty board, for either mover: every piece on every cell (what a reset leaves; the kernels that live on a
// lone wavefront's latency take the legal mask of the moved position BESIDE the winner test and swap this in on a reset)
constexpr uint64_t kLegalEmpty = (1ull << kActions) - 1ull;

__device__ __forceinline__ int pick54(uint64_t m, uint32_t r)
{
    uint32_t lo = (uint32_t)m, hi = (uint32_t)(m >> 32);
    uint32_t nlo = __popc(lo), n = nlo + __popc(hi);
    uint32_t k = __umulhi(r, n);
    bool low = k < nlo;  // pick the word first: one bit search instead of two
    int a = (int)kth_bit32(low ? lo : hi, low ? k : k - nlo) + (low ? 0 : 32);
    return n ? a : -1;
}

__device__ __forceinline__ int sample54(uint64_t m, uint64_t seed, uint64_t env_id, uint32_t ply)
{
    return pick54(m, draw32(seed, env_id, ply));
}

// ---- one ply of one board: raw_env.step bookkeeping, gobblet.py:231-271 -------------------
constexpr int kIllegalNoop = 0;       // GBL_ILLEGAL_NOOP
constexpr int kIllegalTerminate = 1;  // GBL_ILLEGAL_TERMINATE

struct Ply {
    int winner;     // check_for_winner() after the move
    int r0, r1;     // rewards of player_1 / player_2 for this step
    bool terminal;  // the episode ended on this step
    bool stepped;   // raw_env.step ran, i.e. the reference did `self.turn += 1` (gobblet.py:270)
    bool ok;        // the action was a legal move of the mover and was played (false: illegal or outside [0, 54) -- action_status)
};

// The status byte of an externally supplied action (gbl_step_ex / gbl_collect_from_ex; include/gobblet_hip.h GBL_STATUS_*):
// bit 0 = not a legal move of the mover (the reference's silent no-op, board.py:125-126, or TerminateIllegalWrapper's case),
// bit 1 = outside [0, 54) (where the reference's env() asserts: AssertOutOfBoundsWrapper, gobblet.py:110-117).
__device__ __forceinline__ int action_status(bool ok, int action)
{
    return (ok ? 0 : 1) | ((uint32_t)action < (uint32_t)kActions ? 0 : 2);
}

// ... of an action against the mover's legal mask (the test play_ply makes)
__device__ __forceinline__ int action_status_of(uint64_t legal, int action)
{
    return action_status((uint32_t)action < (uint32_t)kActions && ((legal >> (action & 63)) & 1ull), action);
}

// TRUSTED: the action comes from pick54 on `legal` itself (the masked-random sampler inside a T-plies-per-launch kernel): it is
// legal by construction, or -1 where a board has no legal move at all (none in the game: only outside the state contract) -- the
// legality test and the illegal-action branch leave the ply's serial chain.  PAIR > 0: winner_of_pair with j = pair_j.
template <bool TRUSTED = false, bool PAIR = false, typename Row>
__device__ __forceinline__ Ply play_ply(Planes &p, Row row, int &mover, uint64_t legal, int action, int illegal_mode, int pair_j = 0)
{
    Ply y{0, 0, 0, false, false};
    bool ok = TRUSTED ? action >= 0 : ((uint32_t)action < (uint32_t)kActions && ((legal >> (action & 63)) & 1ull));
    y.ok = ok;
    if (!TRUSTED && !ok && illegal_mode == kIllegalTerminate) {
        // gobblet.py:50-51, :114: mover -1, the other 0, everyone terminated, board untouched
        y.r0 = mover ? 0 : -1;
        y.r1 = mover ? -1 : 0;
        y.terminal = true;
        return y;
    }
    y.stepped = true;
    // gobblet.py:244 (illegal: silent no-op, board.py:125-126).  (Straight-line for the role kernel's row-less wavefronts -- the
    // planes masked instead of the EXEC round trip: no effect, profiles/r05/ab_legal_ahead.txt)
    if (ok) row.apply(move_planes(p, mover, (uint32_t)action));
    mover ^= 1;                                          // gobblet.py:246,267
    y.winner = PAIR ? winner_of_pair(p, pair_j) : winner_of(p);  // gobblet.py:248-249
    y.r0 = y.winner;                                     // gobblet.py:253-260
    y.r1 = -y.winner;
    y.terminal = y.winner != 0;                          // gobblet.py:263
    return y;
}

// Lane body of gbl_step: raw_env.step + the state the next observe() is taken from.
// In: row (RegRow / ImageRow) / p (its planes) / mover / was_done / action.  Out: row, p (after the
// step), mover, dn, y.
template <typename Row>
__device__ __forceinline__ void step_lane(Row row, Planes &p, int &mover, int was_done, int action, int illegal_mode,
                                          int auto_reset, int &dn, Ply &y)
{
    y = Ply{0, 0, 0, false, false};
    dn = was_done;
    if (was_done) {
        y.ok = true;              // (no action is consumed)
        y.winner = winner_of(p);  // frozen board (reference: _was_dead_step, gobblet.py:232-236): standing result
    } else {
        y = play_ply(p, row, mover, legal54(p, mover), action, illegal_mode);
        dn = y.terminal ? 1 : 0;
        if (y.terminal && auto_reset) {  // raw_env.reset, gobblet.py:275-290
            p = Planes{0u, 0u, 0u};
            mover = 0;
            row.reset();
        }
    }
}

// raw_env.turn of one board after a ply (gobblet.py:270 `self.turn += 1`, :289 reset to 0)
__device__ __forceinline__ int next_turn(int turn, const Ply &y, int auto_reset)
{
    turn += y.stepped ? 1 : 0;
    return (y.terminal && auto_reset) ? 0 : turn;
}

// gobblet.py:209: the mask belongs to the agent to move; a frozen board has nobody to move
__device__ __forceinline__ uint64_t next_mask(const Planes &p, int mover, int dn, int auto_reset)
{
    return (dn && !auto_reset) ? 0ull : legal54(p, mover);
}

// 54 mask bytes (14 dwords, row_load order) -> 54 bits
__device__ __forceinline__ uint64_t mask_bits(const uint32_t (&d)[14])
{
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int j = 0; j < 14; j += 2) {
        uint32_t acc = 0;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            uint32_t x = d[j + u];
            if (j + u == 13) x &= 0x0000FFFFu;
            uint32_t t = (((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u;
            acc = __builtin_amdgcn_udot4(t, u ? 0x80402010u : 0x08040201u, acc, false);
        }
        uint32_t bits8 = acc >> 7;
        if (j < 8) lo |= bits8 << (4 * j);
        else hi |= bits8 << (4 * (j - 8));
    }
    return ((uint64_t)hi << 32) | lo;
}

// GreedyGobbletPolicy board decode, greedy_policy.py:43-71: 117 obs bytes -> 27 state bytes + agent index
__device__ __forceinline__ int decode_obs_row(const uint32_t (&d)[30], uint32_t (&r)[7])
{
    auto ob = [&](int idx) -> int { return (int)(int8_t)((d[idx >> 2] >> (8 * (idx & 3))) & 0xFFu); };
    int agent = 0;
#pragma unroll
    for (int q = 0; q < 9; ++q) agent = ob(13 * q + 12) > agent ? ob(13 * q + 12) : agent;  // obs[...,12].max()
#pragma unroll
    for (int j = 0; j < 7; ++j) r[j] = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            int i = 2 * k;
            int bp = (i + 1) * ob(13 * q + i) + (i + 2) * ob(13 * q + i + 1);          // :44-49
            int bo = (i + 1) * ob(13 * q + 6 + i) + (i + 2) * ob(13 * q + 6 + i + 1);  // :50-56
            int v = bp > bo ? bp : -bo;                                                // :57
            if (agent == 1) v = -v;                                                    // :67-68
            int c = 9 * k + q;
            r[c >> 2] |= ((uint32_t)v & 0xFFu) << (8 * (c & 3));
        }
    return agent;
}

// ---- GreedyGobbletPolicy.compute_action, greedy_policy.py:38-221 (depth 1 / 2 / 3) ------------
// planes-only form of apply_move (no byte row to maintain)
__device__ __forceinline__ Planes moved(const Planes &p0, int mover, uint32_t a)
{
    Planes p = p0;
    uint32_t pi = (a * 57u) >> 9, q = a - 9u * pi, k = pi >> 1, first = (~pi) & 1u;
    uint32_t mine = mover ? (p.nz & p.neg) : (p.nz & ~p.neg);
    uint32_t ploc = mine & (first ? p.odd : ~p.odd) & (0x1FFu << (9u * k));
    uint32_t bit = 1u << (9u * k + q);
    p.nz = (p.nz & ~ploc) | bit;
    p.neg = (p.neg & ~ploc & ~bit) | (mover ? bit : 0u);
    p.odd = (p.odd & ~ploc & ~bit) | (first ? bit : 0u);
    return p;
}

struct GreedyResult {
    int chosen;      // chosen_action just before the fallback test (:211), -1 = None
    uint64_t cands;  // actions_depth1 as a 54-bit set
    bool fallback;   // the reference would call np.random.choice(actions_depth1)
};

// Outcome of EVERY move of `mover` on board p at once: bit a of `win` / `lose` is set iff
// check_for_winner() after play_turn(mover, a) is the mover's / the other side's value
// (board.py:118-132 then :183-194).  Only meaningful for legal a -- the caller masks with legal54.
//
// Bit-parallel over the 9 destinations of a piece, and SWAR over three pieces at a time (10-bit fields
// of a 32-bit word, bit 9 of each field a guard): lifting a piece from its square s exposes what
// lies beneath (tops after the lift: Tm, To); dropping it on q sets the mover's top at q and clears
// the other side's.  Line l is then complete for the mover iff q supplies its only missing square
// (or nothing is missing), and stays complete for the other side iff it was complete and q is not
// on it.  The reference lets the LAST matching line decide, so lines are resolved from index 7 down,
// each destination taking the first verdict it meets.
//
// QUIET: the caller guarantees that neither side holds a complete line on p (check_for_winner() == 0).
// A lift never adds a top piece of the mover, so no line can then be complete for the mover before the
// drop, and the "nothing missing" term is dropped.
//
template <bool QUIET = false>
__device__ __forceinline__ void outcomes54(const Planes &p, int mover, uint64_t &win, uint64_t &lose)
{
    uint32_t mine = mover ? (p.nz & p.neg) : (p.nz & ~p.neg);
    uint32_t othr = mover ? (p.nz & ~p.neg) : (p.nz & p.neg);
    uint32_t m0 = mine & 0x1FFu, m1 = (mine >> 9) & 0x1FFu, m2 = (mine >> 18) & 0x1FFu;
    uint32_t t0 = othr & 0x1FFu, t1 = (othr >> 9) & 0x1FFu, t2 = (othr >> 18) & 0x1FFu;
    uint32_t o1 = m1 | t1, o2 = m2 | t2;
    uint32_t Tm = m2 | (~o2 & (m1 | (~o1 & m0)));  // squares whose top piece is the mover's
    uint32_t To = t2 | (~o2 & (t1 | (~o1 & t0)));  // ... the other side's
    // owner of the highest piece strictly below level k (what a lift from level k exposes)
    uint32_t um[3] = {0u, m0, m1 | (~o1 & m0)};
    uint32_t uo[3] = {0u, t0, t1 | (~o1 & t0)};
    constexpr uint32_t L[8] = {0x007u, 0x038u, 0x1C0u, 0x049u, 0x092u, 0x124u, 0x111u, 0x054u};  // board.py:135-153
    constexpr uint32_t LOW3 = 0x00100401u;   // bit 0 of each field
    constexpr uint32_t G3 = LOW3 << 9;       // guard bit of each field
    constexpr uint32_t F3 = LOW3 * 0x1FFu;   // the 9 board bits of each field
    uint32_t TmR = Tm | (Tm << 10) | (Tm << 20), ToR = To | (To << 10) | (To << 20);
    win = 0;
    lose = 0;
#pragma unroll
    for (int w = 0; w < 2; ++w) {  // word w: pieces 3w, 3w+1, 3w+2
        uint32_t lost = 0, gain = 0;
#pragma unroll
        for (int f = 0; f < 3; ++f) {
            int pi = 3 * w + f, k = pi >> 1;
            uint32_t src = ((mine & ((pi & 1) ? ~p.odd : p.odd)) >> (9 * k)) & 0x1FFu;  // where the piece stands (0: in hand)
            lost |= (src & ~um[k]) << (10 * f);  // the mover's top that leaves with the lift
            gain |= (src & uo[k]) << (10 * f);   // a piece of the other side the lift exposes
        }
        uint32_t nTm = ~(TmR & ~lost), nTo = ~(ToR | gain);  // complements of the tops after the lift
        uint32_t W = 0, Z = 0, open = F3;
#pragma unroll
        for (int l = 7; l >= 0; --l) {
            const uint32_t Lr = L[l] | (L[l] << 10) | (L[l] << 20);
            uint32_t miss = Lr & nTm;                         // squares of line l the mover lacks
            uint32_t multi = miss & ((miss | G3) - LOW3);     // != 0 in a field: two or more missing
            uint32_t mg = (multi + F3) & G3, mm = mg - (mg >> 9);
            uint32_t need = miss & ~mm;                       // destinations that complete line l
            if (!QUIET) {
                uint32_t zg = G3 & ~(miss + F3);              // fields with nothing missing: every destination
                need |= zg - (zg >> 9);
            }
            uint32_t kg = G3 & ~((Lr & nTo) + F3);            // fields where the other side holds line l
            uint32_t keep = (kg - (kg >> 9)) & ~Lr;           // ... and keeps it: destinations off the line
            uint32_t wv = need & open;
            W |= wv;
            open &= ~wv;
            uint32_t xv = keep & open;
            Z |= xv;
            open &= ~xv;
        }
#pragma unroll
        for (int f = 0; f < 3; ++f) {
            win |= (uint64_t)((W >> (10 * f)) & 0x1FFu) << (9 * (3 * w + f));
            lose |= (uint64_t)((Z >> (10 * f)) & 0x1FFu) << (9 * (3 * w + f));
        }
    }
}

// true iff the predicate holds on ANY active lane of the wavefront (host emulation: lanes run one by one)
__device__ __forceinline__ bool wave_any(bool x)
{
#ifndef GBL_HOST_EMU
    return __any(x) != 0;
#else
    return x;
#endif
}

// outcomes54 of a ROOT position: the policy is asked about positions on which nobody holds a line (the game would be over), and
// then the QUIET form does (80 instructions less); a wavefront with a board that does hold one takes the general form.
__device__ __forceinline__ void outcomes54_root(const Planes &p, int mover, uint64_t &win, uint64_t &lose)
{
    if (wave_any(winner_of(p) != 0))
        outcomes54<false>(p, mover, win, lose);
    else
        outcomes54<true>(p, mover, win, lose);
}

// ---- the depth-2 pair evaluation's FAST form ---------------------------------------------------------------------------------
// On a position where nobody holds a line (a depth-1 result with value 0), a reply of `mover` can leave the OTHER side a
// complete line only by lifting a piece that lies directly on one of theirs, on the third square of a line they otherwise hold
// (their tops before the lift hold no line, and a lift adds exactly the square it exposes).  reply_is_plain() says that no TOP
// piece of the mover stands like that; then every line complete after a reply is the mover's, the order in which the
// reference walks the lines is irrelevant, nobody but the mover can win (`lose` = 0), and the mover's winning moves are the
// THREAT SQUARES of its tops after the lift -- squares whose two partners on some line it tops (wins54_plain: per direction
// two shifts of the tops onto the square and a mask, 12 terms, SWAR over three pieces, instead of outcomes54's eight ordered
// line steps: 119 instead of 280 instructions).  The test is conservative in one respect only: a line with two exposable squares
// counts, though one reply lifts one piece.  tests/emu checks the fast form against outcomes54 on every pair it is used for.
__device__ __forceinline__ bool reply_is_plain(const Planes &p, int mover)
{
    uint32_t mine = mover ? (p.nz & p.neg) : (p.nz & ~p.neg);
    uint32_t othr = mover ? (p.nz & ~p.neg) : (p.nz & p.neg);
    uint32_t m1 = (mine >> 9) & 0x1FFu, m2 = (mine >> 18) & 0x1FFu;
    uint32_t t0 = othr & 0x1FFu, t1 = (othr >> 9) & 0x1FFu, t2 = (othr >> 18) & 0x1FFu;
    uint32_t o1 = m1 | t1, o2 = m2 | t2;
    uint32_t To = t2 | (~o2 & (t1 | (~o1 & t0)));                    // the other side's tops
    uint32_t X = (m2 & (t1 | (~o1 & t0))) | (m1 & ~o2 & t0);         // theirs directly below a TOP piece of the mover
    uint32_t have = To | X;
    uint32_t nh = ~(have | (have << 10) | (have << 20));
    constexpr uint32_t LOW3 = 0x00100401u, G3 = LOW3 << 9, F3 = LOW3 * 0x1FFu;
    constexpr uint32_t LA = 0x007u | (0x038u << 10) | (0x1C0u << 20), LB = 0x049u | (0x092u << 10) | (0x124u << 20);
    constexpr uint32_t LC = 0x111u | (0x054u << 10) | (0x1FFu << 20);  // third field: a "line" nobody can hold
    // guard bit of a field stays clear iff none of the line's squares is missing from `have`
    uint32_t full = (G3 & ~((LA & nh) + F3)) | (G3 & ~((LB & nh) + F3)) | (G3 & ~(((LC & nh) | (1u << 20)) + F3));
    return full == 0;
}

// THREAT SQUARES, three boards at a time (10-bit fields of a word, bit 9 of each unused): the squares whose two partners on
// some line are both in T.  The partners are shifted onto the square (rows: neighbours at distance 1, columns 3, diagonal 4,
// anti-diagonal 2); the masks pick the squares for which that pair of shifts IS the line, which also keeps every term inside
// its field.
__device__ __forceinline__ uint32_t threat3(uint32_t T)
{
    constexpr uint32_t LOW3 = 0x00100401u;
    constexpr uint32_t M0 = 0x049u * LOW3, M1 = 0x092u * LOW3, M2 = 0x124u * LOW3;  // column 0 / 1 / 2 of the board, per field
    constexpr uint32_t R0 = 0x007u * LOW3, R1 = 0x038u * LOW3, R2 = 0x1C0u * LOW3;  // row 0 / 1 / 2
    constexpr uint32_t B0 = 0x001u * LOW3, B2 = 0x004u * LOW3, B4 = 0x010u * LOW3, B6 = 0x040u * LOW3, B8 = 0x100u * LOW3;
    return ((T >> 1) & (T >> 2) & M0) | ((T << 1) & (T >> 1) & M1) | ((T << 1) & (T << 2) & M2)    // (0,1,2) (3,4,5) (6,7,8)
         | ((T >> 3) & (T >> 6) & R0) | ((T << 3) & (T >> 3) & R1) | ((T << 3) & (T << 6) & R2)    // (0,3,6) (1,4,7) (2,5,8)
         | ((T >> 4) & (T >> 8) & B0) | ((T << 4) & (T >> 4) & B4) | ((T << 4) & (T << 8) & B8)    // (0,4,8)
         | ((T >> 2) & (T >> 4) & B2) | ((T << 2) & (T >> 2) & B4) | ((T << 2) & (T << 4) & B6);   // (2,4,6)
}

// the mover's winning moves on a quiet position whose replies are plain (see above); bit a as in outcomes54's `win`
__device__ __forceinline__ uint64_t wins54_plain(const Planes &p, int mover)
{
    uint32_t mine = mover ? (p.nz & p.neg) : (p.nz & ~p.neg);
    uint32_t othr = mover ? (p.nz & ~p.neg) : (p.nz & p.neg);
    uint32_t m0 = mine & 0x1FFu, m1 = (mine >> 9) & 0x1FFu, m2 = (mine >> 18) & 0x1FFu;
    uint32_t t1 = (othr >> 9) & 0x1FFu, t2 = (othr >> 18) & 0x1FFu;
    uint32_t o1 = m1 | t1, o2 = m2 | t2;
    uint32_t Tm = m2 | (~o2 & (m1 | (~o1 & m0)));  // squares whose top piece is the mover's
    uint32_t um[3] = {0u, m0, m1 | (~o1 & m0)};    // the mover's highest piece strictly below level k
    uint32_t TmR = Tm | (Tm << 10) | (Tm << 20);
    uint64_t win = 0;
#pragma unroll
    for (int w = 0; w < 2; ++w) {  // word w: pieces 3w, 3w+1, 3w+2 in 10-bit fields
        uint32_t lost = 0;
#pragma unroll
        for (int f = 0; f < 3; ++f) {
            int pi = 3 * w + f, k = pi >> 1;
            uint32_t src = ((mine & ((pi & 1) ? ~p.odd : p.odd)) >> (9 * k)) & 0x1FFu;  // where the piece stands (0: in hand)
            lost |= (src & ~um[k]) << (10 * f);  // the mover's top that leaves with the lift
        }
        const uint32_t W = threat3(TmR & ~lost);
#pragma unroll
        for (int f = 0; f < 3; ++f) win |= (uint64_t)((W >> (10 * f)) & 0x1FFu) << (9 * (3 * w + f));
    }
    return win;
}

// reply_is_plain(moved(p, me, a), 1 - me) for ALL moves a of `me` at once: bit a SET = NOT plain, the pair (p, a) takes the
// ordered form of the evaluation.  With H = our tops and the squares where a TOP piece of the opponent lies directly on one of
// ours -- the set reply_is_plain looks for a full line in -- a move of our piece i from s (nowhere: in hand) to q changes H in
// two squares only: q joins it (our piece is the top there now, whatever lay there), and s stays iff what the lift leaves on
// top of s is ours, or the opponent's lying directly on a piece of ours (the set C below); every other stack is as it was.
// So per piece B_i = (H - s) + (C & s), and the pair is not plain iff B_i + q holds a full line: iff B_i does already, or q is
// a threat square of B_i -- the threat squares of six 9-bit sets in two words.  One wavefront that idles through the depth-1
// walk does this for a whole tile; the work lists are then split by it.  tests/emu checks every candidate's bit against
// reply_is_plain on the moved board.
__device__ __forceinline__ uint64_t greedy_nonplain(const Planes &p, int me)
{
    const uint32_t mine = me ? (p.nz & p.neg) : (p.nz & ~p.neg);
    const uint32_t othr = me ? (p.nz & ~p.neg) : (p.nz & p.neg);
    const uint32_t m0 = mine & 0x1FFu, m1 = (mine >> 9) & 0x1FFu, m2 = (mine >> 18) & 0x1FFu;
    const uint32_t t0 = othr & 0x1FFu, t1 = (othr >> 9) & 0x1FFu, t2 = (othr >> 18) & 0x1FFu;
    const uint32_t o1 = m1 | t1, o2 = m2 | t2;
    const uint32_t below2 = m1 | (~o1 & m0);                      // our piece is the highest one strictly below level 2
    const uint32_t Tm = m2 | (~o2 & below2);                      // our tops
    const uint32_t X = (t2 & below2) | (t1 & ~o2 & m0);           // ours directly below a TOP piece of the opponent
    const uint32_t H = Tm | X;
    const uint32_t C = (m2 & (below2 | (t1 & m0))) | (m1 & ~o2 & m0);  // squares of H that stay when our top piece leaves
    const uint32_t HR = H | (H << 10) | (H << 20), CR = C | (C << 10) | (C << 20);
    constexpr uint32_t LOW3 = 0x00100401u, G3 = LOW3 << 9, F3 = LOW3 * 0x1FFu;
    uint64_t np = 0;
#pragma unroll
    for (int w = 0; w < 2; ++w) {
        uint32_t src = 0;
#pragma unroll
        for (int f = 0; f < 3; ++f) {
            const int pi = 3 * w + f, k = pi >> 1;
            src |= (((mine & ((pi & 1) ? ~p.odd : p.odd)) >> (9 * k)) & 0x1FFu) << (10 * f);  // where the piece stands
        }
        const uint32_t B = (HR & ~src) | (CR & src);
        const uint32_t th = threat3(B);
        const uint32_t fg = (((th & B) + F3) & G3);  // a full line inside B itself: guard bit of the field
        const uint32_t N = th | (fg - (fg >> 9));
#pragma unroll
        for (int f = 0; f < 3; ++f) np |= (uint64_t)((N >> (10 * f)) & 0x1FFu) << (9 * (3 * w + f));
    }
    return np;
}

__device__ __forceinline__ uint64_t below_eq(int b) { return (2ull << b) - 1ull; }  // bits 0..b, b < 63

// One decision for the agent `me` on board p, in four per-board pieces so that a kernel can spread
// the expensive one (greedy_reply) over the lanes of a wavefront.  `mask` is the legal mask handed to
// the policy (greedy_policy.py:76), prev3 the agent's last three actions packed one per byte (0xFF =
// none).  The reference's control flow -- the depth-1 loop (:84-101), the depth-2 loop with its
// order-dependent pruning (:103-157), the fallback test (:211-214) -- is replayed in order over
// outcome bit-sets, so that no per-leaf loop remains: per depth-1 candidate one moved(), one
// legal54() and one outcomes54() give every opponent reply's result.
struct GreedyHead {
    uint64_t cands;     // actions_depth1 as a 54-bit set, :77-79
    uint64_t todo;      // depth-1 results with value 0, in insertion (= ascending) order: the depth-2 loop's list
    uint64_t legal_me;  // board.is_legal(agent_index, a) on the root position, :85 and :141
    uint64_t dup;       // candidates in todo whose reply summary equals that of candidate a - 9 (see greedy_head)
    int ncands, chosen;
};

// depth 1, :84-101
__device__ __forceinline__ GreedyHead greedy_head(const Planes &p, int me, uint64_t mask, int depth)
{
    GreedyHead h;
    h.cands = mask;
    h.ncands = __popcll(mask);
    h.chosen = -1;
    h.legal_me = legal54(p, me);
    uint64_t tried = mask & h.legal_me;  // actions the depth-1 loop evaluates, ascending
    uint64_t win1, lose1;
    outcomes54_root(p, me, win1, lose1);
    win1 &= tried;
    lose1 &= tried;
    // The walk over the decisive results in order (:92-101), in CLOSED FORM (round 5: the loop ran as long as the busiest lane's
    // count of decisive moves, on the owners' path of the greedy kernels' B phase).  Let fw be the first winning move.  Every
    // losing move before it takes itself off actions_depth1 while more than one action is left (:95-99) -- and since the losing
    // moves are candidates themselves, the list can only run down to one if EVERY candidate loses (no win before the last one):
    // then the walk stops at the last of them (:100-101) with that one left.  Otherwise all losing moves before fw go, and the
    // walk ends at fw (chosen, :92-94) or runs through.  Everything before the stop is in `results`.
    const uint64_t before_fw = win1 ? (1ull << __builtin_ctzll(win1)) - 1ull : ~0ull;
    const uint64_t lb = lose1 & before_fw;                    // the losing moves the walk meets
    const int m = __popcll(lb);
    const bool all_lose = m > 0 && m == h.ncands;             // (then there is no win among the tried moves either)
    const int last = all_lose ? 63 - __builtin_clzll(lb) : 0;
    h.cands = all_lose ? (1ull << last) : (h.cands & ~lb);
    h.ncands = all_lose ? 1 : h.ncands - m;
    h.chosen = win1 ? __builtin_ctzll(win1) : -1;
    uint64_t seen = win1 ? (tried & (before_fw | (before_fw + 1ull))) : all_lose ? (tried & below_eq(last)) : tried;
    // :103; depth 3 adds :160-208, whose only assignment repeats :157 -- no effect
    h.todo = depth > 1 ? (seen & ~win1 & ~lose1) : 0ull;
    // The two pieces of a size are interchangeable: while both are still in hand, placing piece 2k+1 on q
    // gives the opponent exactly the position that placing piece 2k there does (the planes differ in one
    // `odd` bit of OUR piece, which none of the opponent's legality / outcome terms reads), so candidate
    // a + 9 has the reply summary of candidate a.  27 % of the candidates on the masked-random mix.
    uint32_t mine = me ? (p.nz & p.neg) : (p.nz & ~p.neg);
    h.dup = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        uint64_t both = (h.todo >> (18 * k)) & (h.todo >> (18 * k + 9)) & 0x1FFull;
        if (((mine >> (9 * k)) & 0x1FFu) == 0) h.dup |= both << (18 * k + 9);
    }
    return h;
}

// all placements of `me`'s pieces that are still in hand (54-bit set)
__device__ __forceinline__ uint64_t greedy_from_hand(const Planes &p, int me)
{
    const uint32_t mine = me ? (p.nz & p.neg) : (p.nz & ~p.neg);
    uint64_t from_hand = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const uint32_t lvl = (mine >> (9 * k)) & 0x1FFu, oddp = (p.odd >> (9 * k)) & 0x1FFu;
        if ((lvl & oddp) == 0) from_hand |= 0x1FFull << (18 * k);        // piece 2k
        if ((lvl & ~oddp) == 0) from_hand |= 0x1FFull << (18 * k + 9);   // piece 2k + 1
    }
    return from_hand;
}

// have = a 9-bit set of squares: the squares that complete a line inside it (exactly one square of the line missing) -- or every
// square, if it holds a complete line already.  As threat squares (round 4: ~35 instead of the eight-line walk's ~60 instructions, on
// the longest wavefront of the greedy kernels' B phase): a square both of whose partners on some line are in `have` is a threat
// square; one that is itself in `have` means a full line.  tests/emu checks all 512 sets against the line walk.
__device__ __forceinline__ uint32_t risky_from_have(uint32_t have)
{
    const uint32_t th = threat3(have & 0x1FFu) & 0x1FFu;  // (one field; what the shifts push into the fields above is masked off)
    return (th & have) ? 0x1FFu : th;
}

// The squares on which a placement of ours could let a LIFT by the opponent hand us a line (then the opponent's replies
// are not just "the root's winning moves, minus the defused ones", see below): with "have" = our tops and our pieces
// directly under the opponent's, the squares that complete a line inside have -- or every square, if have holds a line already.
__device__ __forceinline__ uint32_t greedy_risky_squares(const Planes &p, int me)
{
    const uint32_t mine = me ? (p.nz & p.neg) : (p.nz & ~p.neg);
    const uint32_t othr = me ? (p.nz & ~p.neg) : (p.nz & p.neg);
    const uint32_t t0 = mine & 0x1FFu, t1 = (mine >> 9) & 0x1FFu, t2 = (mine >> 18) & 0x1FFu;
    const uint32_t m1 = (othr >> 9) & 0x1FFu, m2 = (othr >> 18) & 0x1FFu;
    const uint32_t o1 = m1 | t1, o2 = m2 | t2;
    const uint32_t To = t2 | (~o2 & (t1 | (~o1 & t0)));       // our tops
    const uint32_t X = (m1 & t0) | (m2 & (t1 | (~o1 & t0)));  // ours directly below a piece of the opponent's
    return risky_from_have((To | X) & 0x1FFu);
}

// ---- placements from hand, settled from the ROOT position alone (round 3) ------------------------------------------------
// Let R be the OPPONENT's winning moves on the root, were it to move.  After a placement `a` of ours from hand on q every
// winning reply is in R: a reply that wins after `a` is legal on the root too (we only covered q) and finds there the same
// tops except that q is what it was instead of ours -- which only helps the opponent -- or, if it gobbles our new piece,
// exactly the same tops.  And a2 = (piece pj, square q2) in R stops winning ("is defused") exactly if
//   q is where pj stands (we gobbled it: it cannot move), or
//   q = q2 and our piece is at least as large as pj (the reply is no longer legal), or
//   q != q2 lies on EVERY line the opponent holds after a2 (each of them now has our piece on it),
// provided no lift by the opponent can hand US a line, which with "have" = our tops and our pieces directly under the
// opponent's is the case iff q does not complete a line inside have (greedy_risky_squares).  So for the
// placements from hand on non-risky squares the whole summary -- first / second winning reply, the first one we could
// play ourselves -- follows from R and a few set operations per member of R, with no depth-2 evaluation at all: two
// thirds of the (board, candidate) pairs of the masked-random mix.  Placements on risky squares are evaluated.  (R empty
// -- half of the boards -- means: none of these placements has a winning reply.)  Pinned against the exact evaluation on every candidate it settles: tests/emu (emu_greedy_root_rule).
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
    uint64_t ow, ol;
    outcomes54_root(p, 1 - me, ow, ol);  // (the root itself need not be free of lines)
    return GreedyRoot{ow & legal54(p, 1 - me), greedy_risky_squares(p, me)};
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

// The rule in the shape the kernel uses it: the members of R are dealt out as ITEMS (board, j), j = the member's rank in
// R, to whichever lane is free; an item's lane leaves undef[j] = greedy_undefused(R's j-th member) & resolved in the
// board's table, and the board's owner merges the table in ascending order -- the reference's reply order -- into the
// candidate sets of greedy_replay_closed.  Boards with more than kRootItems members (rare) do not use the rule.
constexpr int kRootItems = 6;

struct GreedyHandSets {
    uint64_t threat, second, block, flegal;
};

__device__ __forceinline__ GreedyHandSets greedy_hand_merge(uint64_t replies, uint64_t legal_me, const uint64_t (&undef)[kRootItems])
{
    GreedyHandSets s{0ull, 0ull, 0ull, 0ull};
    uint64_t it = replies;
#pragma unroll
    for (int j = 0; j < kRootItems; ++j) {
        const bool live = it != 0;
        const uint32_t a2 = live ? (uint32_t)__builtin_ctzll(it) : 0u;
        const uint64_t u = live ? undef[j] : 0ull;
        const bool ours = live && ((legal_me >> a2) & 1ull);
        s.second |= u & s.threat;
        s.flegal |= ours ? (u & ~s.threat) : 0ull;  // the FIRST winning reply is a legal move of ours
        s.block |= ours ? u : 0ull;
        s.threat |= u;
        it &= it - 1;
    }
    return s;
}

// the 16-bit summary (see greedy_reply) of ONE settled placement `a`, from the board's table
__device__ __forceinline__ uint32_t greedy_hand_lookup(uint64_t replies, uint64_t legal_me, const uint64_t (&undef)[kRootItems], uint32_t a)
{
    uint64_t ow = 0, it = replies;
#pragma unroll
    for (int j = 0; j < kRootItems; ++j) {
        const bool live = it != 0;
        const uint32_t a2 = live ? (uint32_t)__builtin_ctzll(it) : 0u;
        if (live && ((undef[j] >> a) & 1ull)) ow |= 1ull << a2;
        it &= it - 1;
    }
    const uint64_t block = ow & legal_me;
    uint32_t s = ow ? 1u : 0u;
    s |= (ow ? (uint32_t)__builtin_ctzll(ow) : 0u) << 1;
    s |= (ow & (ow - 1)) ? 1u << 7 : 0u;
    s |= block ? 1u << 8 : 0u;
    s |= (block ? (uint32_t)__builtin_ctzll(block) : 0u) << 9;
    return s;
}

// The same two in the form the kernel's owners run them (round 4): an item's lane TAGS its table row -- bits 0-53 the set
// greedy_undefused gives, bits 56-61 the member a2 itself, bit 63 "a2 is a legal move of ours on the root" -- so that the owner's
// merge needs neither the k-th member of R (a 64-bit find-first-set and clear per step) nor a 64-bit shift of its legal set:
// the rows are in rank order, row j is live iff j < |R|, and everything else is in the row.  ~14 instead of ~30 instructions per
// step of a loop that runs on ONE wavefront per SIMD.  tests/emu runs the kernel's walk through these and checks every settled
// candidate against the exact evaluation, as before.
__device__ __forceinline__ uint64_t greedy_item_row(const Planes &p, int me, uint64_t legal_me, uint32_t a2)
{
    return greedy_undefused(p, me, a2) | ((uint64_t)a2 << 56) | (((legal_me >> a2) & 1ull) << 63);
}

__device__ __forceinline__ GreedyHandSets greedy_hand_merge_tagged(int n, uint64_t resolved, const uint64_t (&rows)[kRootItems])
{
    GreedyHandSets s{0ull, 0ull, 0ull, 0ull};
#pragma unroll
    for (int j = 0; j < kRootItems; ++j) {
        const bool live = j < n;
        const uint64_t u = live ? rows[j] & resolved : 0ull;     // (resolved has no bit above 53: the tags go with the mask)
        const bool ours = live && (int64_t)rows[j] < 0;
        s.second |= u & s.threat;
        s.flegal |= ours ? (u & ~s.threat) : 0ull;  // the FIRST winning reply is a legal move of ours
        s.block |= ours ? u : 0ull;
        s.threat |= u;
    }
    return s;
}

__device__ __forceinline__ uint32_t greedy_hand_lookup_tagged(int n, uint64_t legal_me, const uint64_t (&rows)[kRootItems], uint32_t a)
{
    uint64_t ow = 0;
#pragma unroll
    for (int j = 0; j < kRootItems; ++j)
        if (j < n && ((rows[j] >> a) & 1ull)) ow |= 1ull << ((uint32_t)(rows[j] >> 56) & 63u);
    const uint64_t block = ow & legal_me;
    uint32_t s = ow ? 1u : 0u;
    s |= (ow ? (uint32_t)__builtin_ctzll(ow) : 0u) << 1;
    s |= (ow & (ow - 1)) ? 1u << 7 : 0u;
    s |= block ? 1u << 8 : 0u;
    s |= (block ? (uint32_t)__builtin_ctzll(block) : 0u) << 9;
    return s;
}

// Which candidates are evaluated (exactly, in the pooled round), which are settled from the root, and the root's
// replies if they are to be dealt out as items.
struct GreedyRootPlan {
    uint64_t eval, resolved, items;
};

__device__ __forceinline__ GreedyRootPlan greedy_root_plan(const GreedyHead &h, const Planes &p, int me, const GreedyRoot &g)
{
    const uint64_t w0 = h.todo & ~h.dup;
    const bool rule = __popcll(g.replies) <= kRootItems;
    const uint64_t resolved = rule ? (w0 & greedy_from_hand(p, me) & ~spread9(g.risky)) : 0ull;
    return GreedyRootPlan{w0 & ~resolved, resolved, resolved ? g.replies : 0ull};
}

// What the depth-2 loop needs to know about candidate `a` (:107-126), packed in 16 bits:
//   bit 0      the opponent has a winning reply            (ow != 0)
//   bits 1-6   the first winning reply f
//   bit 7      there is a second one
//   bit 8      some winning reply is a legal move of ours on the root position (:141)
//   bits 9-14  the first such reply
//   bit 15     every reply wins the game for us            (all(), :146-149; implies bit 0 clear)
// PLAIN: the caller has checked greedy_pair_is_plain(p, me, a) -- the fast form (threat squares; nobody but the opponent
// can win, so "every reply wins for us" holds only if the opponent cannot move at all); otherwise outcomes54's ordered line steps.
__device__ __forceinline__ bool greedy_pair_is_plain(const Planes &p, int me, uint32_t a)
{
    return reply_is_plain(moved(p, me, a), 1 - me);
}

template <bool PLAIN = false>
__device__ __forceinline__ uint32_t greedy_reply(const Planes &p, int me, uint64_t legal_me, uint32_t a)
{
    const int opp = 1 - me;
    Planes d1 = moved(p, me, a);         // :107-109
    uint64_t legal2 = legal54(d1, opp);  // :112-116
    uint64_t ow, mw;                     // the opponent wins / we win after reply a2, :120-126
    if (PLAIN) {
        ow = wins54_plain(d1, opp);
        mw = 0;
    } else {
        outcomes54<true>(d1, opp, ow, mw);  // (a is a depth-1 result with value 0: nobody holds a line on d1)
    }
    ow &= legal2;
    mw &= legal2;
    uint64_t block = ow & legal_me;
    uint32_t s = ow ? 1u : 0u;
    s |= (ow ? (uint32_t)__builtin_ctzll(ow) : 0u) << 1;
    s |= (ow & (ow - 1)) ? 1u << 7 : 0u;
    s |= block ? 1u << 8 : 0u;
    s |= (block ? (uint32_t)__builtin_ctzll(block) : 0u) << 9;
    s |= (legal2 & ~mw) == 0 ? 1u << 15 : 0u;
    return s;
}

// :129-143 for a candidate a whose summary s has bit 0 set.  none_yet: `chosen_action is None` (:142) can
// still hold as far as the candidates WITHOUT a winning reply are concerned (h.chosen tracks the rest).
__device__ __forceinline__ void greedy_threat(GreedyHead &h, int a, uint32_t s, bool none_yet)
{
    int f = (int)((s >> 1) & 63u);
    if (h.ncands > 1) {
        if ((h.cands >> a) & 1ull) {
            h.cands &= ~(1ull << a);
            --h.ncands;
        }
        if (h.ncands > 1 || !(s & (1u << 7))) {  // no break: every opponent win is looked at
            if (none_yet && h.chosen < 0 && (s & (1u << 8))) h.chosen = (int)((s >> 9) & 63u);
        } else {                                  // break at the second opponent win
            if (none_yet && h.chosen < 0 && ((h.legal_me >> f) & 1ull)) h.chosen = f;
        }
    }                                             // else: break at the first opponent win
}

// One iteration of the depth-2 loop (:129-157) for candidate a with summary s; true = `break` (:151).
// With a winning reply f among the evaluated replies neither all() can hold (f is evaluated in
// every early-break variant), so only the no-winning-reply case reaches :146-157.
__device__ __forceinline__ bool greedy_replay(GreedyHead &h, int a, uint32_t s)
{
    if (s & 1u) {
        greedy_threat(h, a, s, true);
        return false;
    }
    h.chosen = a;              // all(... != their win), :153-157 -- or all(... == our win), :146-151
    return (s >> 15) & 1u;
}

// The whole depth-2 loop from two candidate sets -- threat: summaries with bit 0, allwin: with bit 15 --
// looking up summaries only for the (few) candidates in `threat`.  A candidate without a winning reply
// assigns chosen_action unconditionally (:157, :150), so the last of them before the :151 break is
// the result if there is one, and from the first of them on `chosen_action is None` (:142) is false.
template <typename ReplyOf>
__device__ __forceinline__ void greedy_replay_sets(GreedyHead &h, uint64_t threat, uint64_t allwin, ReplyOf reply_of)
{
    uint64_t todo = h.todo;
    threat |= (threat << 9) & h.dup;  // twin placements share their partner's summary (reply_of must serve them too)
    allwin |= (allwin << 9) & h.dup;
    allwin &= todo;
    if (allwin) todo &= below_eq(__builtin_ctzll(allwin));  // :151: nothing after the break is looked at
    const uint64_t calm = todo & ~threat;
    const uint64_t before_calm = calm ? (1ull << __builtin_ctzll(calm)) - 1ull : ~0ull;
    for (uint64_t it = threat & todo; it; it &= it - 1) {
        const int a = __builtin_ctzll(it);
        greedy_threat(h, a, reply_of(a), (before_calm >> a) & 1ull);
    }
    if (calm) h.chosen = 63 - __builtin_clzll(calm);
}

// position of the k-th (0-based) set bit of a 54-bit set; k < popcount(m)
__device__ __forceinline__ uint32_t kth_bit64(uint64_t m, uint32_t k)
{
    const uint32_t lo = (uint32_t)m, hi = (uint32_t)(m >> 32), nlo = __popc(lo);
    const bool low = k < nlo;
    return kth_bit32(low ? lo : hi, low ? k : k - nlo) + (low ? 0u : 32u);
}

// The depth-2 loop (:103-157) in CLOSED FORM -- no per-candidate iteration.  What the loop does with a candidate
// that has a winning reply (greedy_threat) depends on the candidate only through three bits of its summary and on
// how many such candidates came before it:
//   * each of them takes itself off actions_depth1 while more than one action is left (:129-134), so the j-th of
//     them (j = 0, 1, ...) is handled iff j < n0 - 1 (n0 = len(actions_depth1) on entry), and those removed are the
//     first min(count, n0 - 1) of the set;
//   * the reply loop breaks at the SECOND opponent win only when the removal left a single action (:135-143), i.e.
//     for j = n0 - 2 only; every earlier one looks at all opponent wins and proposes the first that we could play
//     ourselves (summary bit 8), the one at j = n0 - 2 with a second winning reply (bit 7) proposes the FIRST winning
//     reply if that is a legal move of ours;
//   * a proposal is taken only while chosen_action is None (:142): before the first candidate without a winning
//     reply, and only the first proposal counts.
// So the loop's effect is a handful of set operations on per-board candidate sets, which the evaluating lanes
// accumulate: threat (summary bit 0), allwin (bit 15), second (bit 7), block (bit 8), flegal (first winning reply is
// a legal move of ours) -- plus ONE summary lookup, for the candidate whose proposal is taken.
struct ReplySets {
    uint64_t threat, allwin, second, block, flegal;
};

template <typename ReplyOf>
__device__ __forceinline__ void greedy_replay_closed(GreedyHead &h, ReplySets r, ReplyOf reply_of)
{
    uint64_t todo = h.todo;
    // twin placements share their partner's summary (reply_of must serve them too)
    r.threat |= (r.threat << 9) & h.dup;
    r.allwin |= (r.allwin << 9) & h.dup;
    r.second |= (r.second << 9) & h.dup;
    r.block |= (r.block << 9) & h.dup;
    r.flegal |= (r.flegal << 9) & h.dup;
    r.allwin &= todo;
    if (r.allwin) todo &= below_eq(__builtin_ctzll(r.allwin));  // :151: nothing after the break is looked at
    const uint64_t calm = todo & ~r.threat;
    const uint64_t before_calm = calm ? (1ull << __builtin_ctzll(calm)) - 1ull : ~0ull;
    const uint64_t th = r.threat & todo & h.cands;  // (every candidate of todo is still in actions_depth1)
    const int m = __popcll(th), n0 = h.ncands;
    if (m > 0 && n0 > 1) {
        // A: the first n0 - 2 of th (they never break early); e: the one after them, if any
        uint64_t A = th, e = 0;
        if (n0 - 2 < m) {
            e = 1ull << kth_bit64(th, (uint32_t)(n0 - 2));
            A = th & (e - 1ull);
        }
        const uint64_t e_prop = (e & r.second) ? (e & r.flegal) : (e & r.block);
        const uint64_t prop = before_calm & ((A & r.block) | e_prop);
        if (h.chosen < 0 && prop) {
            const int a = __builtin_ctzll(prop);
            const uint32_t s = reply_of(a);
            const bool early = ((e >> a) & 1ull) && ((r.second >> a) & 1ull);  // broke at the second opponent win
            h.chosen = early ? (int)((s >> 1) & 63u) : (int)((s >> 9) & 63u);
        }
        const uint64_t removed = A | e;
        h.cands &= ~removed;
        h.ncands = n0 - __popcll(removed);
    }
    if (calm) h.chosen = 63 - __builtin_clzll(calm);
}

// :211-214
__device__ __forceinline__ GreedyResult greedy_finish(const GreedyHead &h, uint32_t prev3)
{
    uint32_t c = (uint32_t)h.chosen;
    bool fb = h.chosen < 0 || (prev3 & 0xFFu) == c || ((prev3 >> 8) & 0xFFu) == c || ((prev3 >> 16) & 0xFFu) == c;
    return GreedyResult{h.chosen, h.cands, fb};
}

// the four pieces in sequence on one board
__device__ __forceinline__ GreedyResult greedy_decide(const Planes &p, int me, uint64_t mask, int depth, uint32_t prev3)
{
    GreedyHead h = greedy_head(p, me, mask, depth);
    for (uint64_t it = h.todo; it;) {
        int a = __builtin_ctzll(it);
        it &= it - 1;
        const uint32_t twin = ((h.dup >> a) & 1ull) ? (uint32_t)a - 9u : (uint32_t)a;  // same summary, see greedy_head
        if (greedy_replay(h, a, greedy_reply(p, me, h.legal_me, twin))) break;
    }
    return greedy_finish(h, prev3);
}

}  // namespace gbl
