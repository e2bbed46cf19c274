// Where do the waves of small workgroups land?  Workgroups of W waves (W = 2, 3, 4), each wave holding R vector registers
// (so that R decides how many fit a SIMD), spin for a while and record (XCC, SE, CU, SIMD) from HW_ID.  Printed per W:
// the number of distinct (XCC, SE, CU) seen, and the histogram of workgroups' SIMD sets and of waves per SIMD.
//   hipcc -O2 --offload-arch=gfx950 wave_placement.hip -o wave_placement
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#include <string>

template <int W>
__global__ __launch_bounds__(64 * W, W) void probe(unsigned *out, long long spin)
{
    // ~160 live registers: a dependent chain through an array the compiler cannot shrink
    double a[76];
#pragma unroll
    for (int i = 0; i < 76; ++i) a[i] = threadIdx.x * 1e-3 + i;
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) {
#pragma unroll
        for (int i = 0; i < 76; ++i) a[i] = a[i] * 1.0000001 + a[(i + 1) % 76];
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 76; ++i) s += a[i];
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * W + (threadIdx.x >> 6);
        out[3 * w] = hw;
        out[3 * w + 1] = xcc;
        out[3 * w + 2] = (unsigned)(t0 & 0xffffffffu) + (s == 12345.678 ? 1 : 0);
    }
}

template <int W>
void run(int blocks)
{
    unsigned *d;
    hipMalloc(&d, sizeof(unsigned) * 3 * blocks * W);
    hipLaunchKernelGGL(probe<W>, dim3(blocks), dim3(64 * W), 0, 0, d, 2000000LL);   // 100 MHz clock: 20 ms
    hipDeviceSynchronize();
    std::vector<unsigned> h(3 * blocks * W);
    hipMemcpy(h.data(), d, sizeof(unsigned) * h.size(), hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> per_cu;       // (xcc, se, cu) -> waves per SIMD among the FIRST round
    std::map<std::string, int> sets;
    unsigned tmin = 0xffffffffu;
    for (int b = 0; b < blocks * W; ++b) tmin = std::min(tmin, h[3 * b + 2]);
    int first_round = 0;
    for (int b = 0; b < blocks; ++b) {
        std::string key;
        bool early = true;
        for (int w = 0; w < W; ++w) {
            const unsigned hw = h[3 * (b * W + w)], xcc = h[3 * (b * W + w) + 1] & 0xf;
            const unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xf, se = (hw >> 13) & 7;
            if (h[3 * (b * W + w) + 2] - tmin > 500000u) early = false;    // started more than 5 ms after the first wave
            key += char('0' + simd);
            if (early) {
                auto &v = per_cu[(xcc << 16) | (se << 8) | cu];
                v.resize(4);
                v[simd]++;
            }
        }
        first_round += early;
        sets[key]++;
    }
    printf("W=%d blocks=%d: %zu CUs seen, %d workgroups started within 5 ms (%.2f per CU)\n", W, blocks, per_cu.size(), first_round,
           (double)first_round / per_cu.size());
    printf("  SIMDs of a workgroup's waves (wave 0,1,..): ");
    for (auto &kv : sets) printf("%s x%d  ", kv.first.c_str(), kv.second);
    printf("\n  waves per SIMD in the first round, histogram over CUs: ");
    std::map<std::string, int> hist;
    for (auto &kv : per_cu) {
        char buf[32];
        snprintf(buf, sizeof buf, "%d/%d/%d/%d", kv.second[0], kv.second[1], kv.second[2], kv.second[3]);
        hist[buf]++;
    }
    for (auto &kv : hist) printf("%s x%d  ", kv.first.c_str(), kv.second);
    printf("\n");
    hipFree(d);
}

int main()
{
    run<2>(1024);
    run<3>(1024);
    run<4>(1024);
    run<3>(768);
    return 0;
}
