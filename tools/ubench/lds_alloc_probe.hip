// lds_alloc_probe.hip -- what does HW_REG_LDS_ALLOC hold on gfx950, and is LDS_BASE / LDS_SIZE a usable "which workgroup of this CU am I" id?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <set>
#include <vector>
__global__ void probe(unsigned *out, int spin)
{
    extern __shared__ double lds[];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < spin) {}
    if ((threadIdx.x & 63) == 0) {
        unsigned *o = out + (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
        o[0] = __builtin_amdgcn_s_getreg(63494);      // HW_REG_LDS_ALLOC (6), 32 bits
        o[1] = __builtin_amdgcn_s_getreg(63492);      // HW_REG_HW_ID (4)
        o[2] = __builtin_amdgcn_s_getreg(63508);      // HW_REG_XCC_ID (20)
        o[3] = (unsigned)lds[threadIdx.x];
    }
}
int main()
{
    for (int ldskb : {52, 18}) {
        const int wgs = ldskb == 52 ? 768 : 1024;
        unsigned *d;
        hipMalloc(&d, sizeof(unsigned) * wgs * 16);
        hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, ldskb * 1024);
        hipLaunchKernelGGL(probe, dim3(wgs), dim3(256), ldskb * 1024, 0, d, 2000000);
        hipDeviceSynchronize();
        std::vector<unsigned> h(wgs * 16);
        hipMemcpy(h.data(), d, sizeof(unsigned) * wgs * 16, hipMemcpyDeviceToHost);
        std::map<unsigned, std::set<unsigned>> per_cu;     // (xcc, se, sh, cu) -> set of LDS_ALLOC values of wave 0
        for (int b = 0; b < wgs; ++b) {
            const unsigned alloc = h[b * 16], hw = h[b * 16 + 1], xcc = h[b * 16 + 2] & 15;
            const unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            per_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu].insert(alloc);
            if (b < 6) printf("lds %d KB wg %d: LDS_ALLOC %08x  HW_ID %08x (wave %u simd %u cu %u sh %u se %u) xcc %u; waves' alloc %08x %08x %08x slots %u %u %u %u simd %u %u %u %u\n", ldskb, b, alloc, hw,
                              hw & 15, (hw >> 4) & 3, cu, sh, se, xcc, h[b * 16 + 4], h[b * 16 + 8], h[b * 16 + 12], hw & 15, h[b*16+5] & 15, h[b*16+9] & 15, h[b*16+13] & 15,
                              (hw >> 4) & 3, (h[b*16+5] >> 4) & 3, (h[b*16+9] >> 4) & 3, (h[b*16+13] >> 4) & 3);
        }
        printf("lds %d KB: %zu distinct CUs; workgroups per CU and their LDS_ALLOC values:\n", ldskb, per_cu.size());
        int shown = 0;
        for (auto &kv : per_cu) {
            if (shown++ < 4) {
                printf("  cu key %05x:", kv.first);
                for (unsigned a : kv.second) printf(" %08x", a);
                printf("\n");
            }
        }
        hipFree(d);
    }
    return 0;
}
