// Does a wave64 VALU instruction cost less when only lanes 0-31 are active (EXEC's upper half zero)?  CDNA4 SIMDs are 32 lanes wide
// and issue a wave64 instruction in two passes; if the second pass were skipped for an empty half, owner-only phases of the greedy
// kernels (one wavefront per SIMD) and the small-batch chain could be dealt over twice as many half-filled wavefronts.
//   hipcc --offload-arch=gfx950 -O3 -o half_exec half_exec.hip && ./half_exec
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

template <int CHAINS>
__global__ __launch_bounds__(256) void k(uint32_t *out, unsigned long long *cycles, int active_lanes, int iters)
{
    const int lane = threadIdx.x & 63;
    uint32_t x[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x * 2654435761u + c;
    unsigned long long t0 = 0, t1 = 0;
    if (lane < active_lanes) {  // (uniform per wavefront when active_lanes is 64; the lower half only when 32)
        t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int c = 0; c < CHAINS; ++c) x[c] = (x[c] ^ (x[c] >> 3)) + 0x9E3779B9u;  // xor-with-shifted (v_lshrrev + v_xor) + add: 3 simple instructions
        }
        t1 = __builtin_amdgcn_s_memtime();
    }
    uint32_t acc = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) acc ^= x[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (lane == 0) cycles[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int CHAINS>
void run(int waves_per_simd, int active)
{
    const int blocks = 256, threads = 256 * waves_per_simd > 1024 ? 1024 : 256 * waves_per_simd;  // 4 SIMDs x waves_per_simd
    const int iters = 200;
    uint32_t *out; unsigned long long *cyc, h[256 * 16];
    hipMalloc(&out, blocks * threads * 4); hipMalloc(&cyc, sizeof h);
    k<CHAINS><<<blocks, threads>>>(out, cyc, active, iters);
    k<CHAINS><<<blocks, threads>>>(out, cyc, active, iters);
    hipDeviceSynchronize();
    hipMemcpy(h, cyc, sizeof(unsigned long long) * blocks * (threads / 64), hipMemcpyDeviceToHost);
    double s = 0; int n = blocks * (threads / 64);
    for (int i = 0; i < n; ++i) s += (double)h[i];
    const double instr = (double)iters * 16 * CHAINS * 3;
    printf("%d chain(s), %d wavefront(s) per SIMD, %2d active lanes: %.2f cycles per instruction and wavefront (s_memtime)\n", CHAINS,
           threads / 256, active, s / n / instr);
    hipFree(out); hipFree(cyc);
}

int main()
{
    for (int a : {64, 32, 16}) { run<1>(1, a); run<8>(1, a); }  // one wavefront per SIMD: the owner-only phases' situation
    return 0;
}
