#include <hip/hip_runtime.h>
#include <stdint.h>
__device__ __forceinline__ uint32_t wave_incl_scan_add(uint32_t x)
{
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);
    return x;
}
__global__ void k(uint32_t* out, const uint32_t* in) { out[threadIdx.x] = wave_incl_scan_add(in[threadIdx.x]); }
int main() {
    uint32_t h[64], r[64], *di, *dout;
    for (int i = 0; i < 64; ++i) h[i] = (i * 7 + 3) % 11;
    hipMalloc(&di, 256); hipMalloc(&dout, 256);
    hipMemcpy(di, h, 256, hipMemcpyHostToDevice);
    k<<<1, 64>>>(dout, di);
    hipMemcpy(r, dout, 256, hipMemcpyDeviceToHost);
    uint32_t s = 0; int bad = 0;
    for (int i = 0; i < 64; ++i) { s += h[i]; if (r[i] != s) { ++bad; printf("lane %d: %u != %u\n", i, r[i], s); } }
    printf("dpp scan: %s\n", bad ? "WRONG" : "ok");
    return bad != 0;
}
