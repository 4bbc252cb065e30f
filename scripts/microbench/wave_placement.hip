// wave_placement.hip -- where do the wavefronts of a 4-wavefront workgroup land?  Every wavefront records HW_REG_HW_ID
// (SIMD, CU, SE) and XCC_ID; the host prints, for the workgroups that shared a CU, which SIMD each wavefront sat on.
//   hipcc --offload-arch=gfx950 -O2 -o wave_placement wave_placement.hip && ./wave_placement [workgroups] [lds_bytes] [wavefronts per workgroup]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <map>
#include <vector>

__global__ __launch_bounds__(1024) void k(unsigned *out, int spin, int waves)
{
    extern __shared__ unsigned lds[];
    unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));    // HW_REG_HW_ID
    unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));  // HW_REG_XCC_ID
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned acc = threadIdx.x;
    for (int i = 0; i < spin; ++i) acc = acc * 1664525u + 1013904223u;  // keep the workgroups resident together
    lds[threadIdx.x] = acc;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        unsigned *o = out + (blockIdx.x * waves + (threadIdx.x >> 6)) * 4;
        o[0] = hw; o[1] = xcc; o[2] = (unsigned)t0; o[3] = lds[(threadIdx.x + 1) & 63];
    }
}

int main(int argc, char **argv)
{
    int wgs = argc > 1 ? atoi(argv[1]) : 1024, ldsb = argc > 2 ? atoi(argv[2]) : 28672, waves = argc > 3 ? atoi(argv[3]) : 4;
    unsigned *d;
    hipMalloc(&d, wgs * waves * 4 * sizeof(unsigned));
    hipLaunchKernelGGL(k, dim3(wgs), dim3(64 * waves), ldsb, 0, d, 20000, waves);
    std::vector<unsigned> h(wgs * waves * 4);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    // gfx9 HW_ID: wave_id [3:0], simd_id [5:4], pipe_id [7:6], cu_id [11:8], sh_id [12], se_id [15:13] (gfx950: se [14:13])
    std::map<unsigned, std::vector<int>> cu;
    for (int b = 0; b < wgs; ++b) {
        unsigned hw = h[(b * waves) * 4], xcc = h[(b * waves) * 4 + 1] & 0xF;
        cu[(xcc << 16) | (hw & 0xFF00)].push_back(b);
    }
    int shown = 0, same = 0, total = 0;
    for (auto &kv : cu) {
        if (shown < 6) {
            printf("xcc %u cu-key %04x:", kv.first >> 16, kv.first & 0xFFFF);
            for (int b : kv.second) {
                printf("  wg %4d simd", b);
                for (int w = 0; w < waves; ++w) printf(" %u", (h[(b * waves + w) * 4] >> 4) & 3);
            }
            printf("\n");
            ++shown;
        }
        for (int b : kv.second) { ++total; same += ((h[(b * waves) * 4] >> 4) & 3) == ((h[(kv.second[0] * waves) * 4] >> 4) & 3); }
    }
    printf("CUs used %zu; workgroups whose wavefront 0 sits on the same SIMD as that of the first workgroup of their CU: %d of %d\n",
           cu.size(), same, total);
    return 0;
}
