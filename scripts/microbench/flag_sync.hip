// What a RESIDENT step server would pay per ply instead of a launch boundary (round 4): an external policy needs
// ALL boards' outputs of a ply before it can decide and the server ALL actions before it can play, i.e. two device-wide
// dependencies per ply carried by flags in device memory instead of by the command processor.  Measured here, with bounded
// polls (every wait gives up after 20 ms of the 100 MHz clock and the program says so):
//   1. flag round trip between two resident kernels on two streams (one lane each): store flag (agent scope) -> poll -> store
//      ack -> poll;  per hop = half of it;
//   2. a device-wide rendezvous of W resident wavefronts (8 per CU = the 2 048 tiles of a 131 072-board shard, and 1 per CU):
//      every wavefront adds to ONE counter (agent-scope atomic, after an agent-scope release as a server publishing rows would
//      need) and polls it until all have arrived;  time per round, rounds back to back.
//   hipcc --offload-arch=gfx950 -O3 -o flag_sync flag_sync.hip && ./flag_sync
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
constexpr unsigned long long kTimeoutTicks = 2000000ull;  // 20 ms of the 100 MHz clock

__device__ __forceinline__ bool wait_for(const uint32_t *p, uint32_t want, unsigned long long t0)
{
    for (;;) {
        if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) return true;
        if (__builtin_amdgcn_s_memrealtime() - t0 > kTimeoutTicks) return false;
        __builtin_amdgcn_s_sleep(1);
    }
}

__global__ void k_ping(uint32_t *flag, uint32_t *ack, uint32_t n, unsigned long long *ticks, uint32_t *failed)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (uint32_t i = 1; i <= n; ++i) {
        __hip_atomic_store(flag, i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!wait_for(ack, i, __builtin_amdgcn_s_memrealtime())) { *failed = i; break; }
    }
    *ticks = __builtin_amdgcn_s_memrealtime() - t0;
}

__global__ void k_pong(uint32_t *flag, uint32_t *ack, uint32_t n, uint32_t *failed)
{
    for (uint32_t i = 1; i <= n; ++i) {
        if (!wait_for(flag, i, __builtin_amdgcn_s_memrealtime())) { *failed = i; break; }
        __hip_atomic_store(ack, i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// W wavefronts (one per workgroup), `rounds` rendezvous back to back; RELEASE: an agent-scope release fence before the arrive
template <bool RELEASE>
__global__ __launch_bounds__(64) void k_rendezvous(uint32_t *counter, uint32_t waves, uint32_t rounds, unsigned long long *ticks,
                                                   uint32_t *failed, uint32_t *payload)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (uint32_t r = 1; r <= rounds; ++r) {
        if (RELEASE) {
            payload[(size_t)blockIdx.x * 64 + threadIdx.x] = r;  // something to publish: 256 B per wavefront
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        bool ok = true;
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = wait_for(counter, r * waves, __builtin_amdgcn_s_memrealtime());
        }
        ok = __builtin_amdgcn_readfirstlane((int)ok) != 0;
        if (!ok) { if (threadIdx.x == 0) *failed = r; break; }
        if (RELEASE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *ticks = __builtin_amdgcn_s_memrealtime() - t0;
}

int main()
{
    uint32_t *d, *payload;
    unsigned long long *ticks;
    CHECK(hipMalloc(&d, 4096));
    CHECK(hipMalloc(&ticks, 64));
    CHECK(hipMalloc(&payload, (size_t)2048 * 256));
    hipStream_t sa, sb;
    CHECK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    uint32_t h[1024];
    unsigned long long ht;
    for (int rep = 0; rep < 3; ++rep) {
        const uint32_t n = 2000;
        CHECK(hipMemset(d, 0, 4096));
        k_pong<<<1, 1, 0, sb>>>(d + 0, d + 64, n, d + 128);
        k_ping<<<1, 1, 0, sa>>>(d + 0, d + 64, n, ticks, d + 192);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(h, d, 4096, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(&ht, ticks, 8, hipMemcpyDeviceToHost));
        if (h[128] || h[192]) printf("flag round trip: TIMED OUT at %u / %u (the two kernels did not run side by side?)\n", h[128], h[192]);
        else printf("flag round trip between two resident kernels: %.2f us (%u round trips; one hop = half)\n", ht * 0.01 / n, n);
    }
    const uint32_t shapes[2] = {2048, 256};
    for (uint32_t waves : shapes)
        for (int rel = 0; rel < 2; ++rel)
            for (int rep = 0; rep < 2; ++rep) {
                const uint32_t rounds = 200;
                CHECK(hipMemset(d, 0, 4096));
                if (rel) k_rendezvous<true><<<waves, 64, 0, sa>>>(d, waves, rounds, ticks, d + 64, payload);
                else k_rendezvous<false><<<waves, 64, 0, sa>>>(d, waves, rounds, ticks, d + 64, payload);
                CHECK(hipDeviceSynchronize());
                CHECK(hipMemcpy(h, d, 4096, hipMemcpyDeviceToHost));
                CHECK(hipMemcpy(&ht, ticks, 8, hipMemcpyDeviceToHost));
                if (h[64]) printf("rendezvous of %u wavefronts: TIMED OUT in round %u\n", waves, h[64]);
                else printf("rendezvous of %4u resident wavefronts (%s): %.2f us per round (%u rounds back to back)\n", waves,
                            rel ? "256 B published per wavefront: agent release before, acquire after" : "counter only", ht * 0.01 / rounds, rounds);
            }
    return 0;
}
