// What a vector-memory instruction BEHIND A BRANCH costs inside a loop that prefetches (DESIGN.md section 4.4,
// chain_tile_split_kernel).  Every wave walks `iters` slices: the load of slice i + DEPTH goes into a ring of register buffers
// (loop unrolled by the ring size), ~`work` dependent FP64 FMAs stand for the slice's products, and ONE result per slice is
// stored -- by lane 0 behind `if (lane == 0)` (variant 0), or by all 64 lanes unconditionally (variant 1: same address, same
// value).  At the join behind the branch the compiler's s_waitcnt pass has to assume the path that issued nothing: the next
// wait for a prefetched buffer is emitted as if the store had not been issued, i.e. it also waits for the YOUNGER loads of the
// ring -- the prefetch distance is lost.  Printed: shader cycles per slice for both variants, ring depths 2 .. 4, and the
// s_waitcnt vmcnt operands the compiler chose in the loop (from the disassembly: `llvm-objdump -d`).
//   hipcc -O3 --offload-arch=gfx950 branch_wait.hip -o branch_wait
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int DEPTH, int VARIANT>
__global__ __launch_bounds__(64) void walk(const double2 *__restrict__ src, double *__restrict__ out, long long *__restrict__ cycles,
                                            int iters, int work)
{
    constexpr int R = DEPTH + 1;                       // ring buffers
    const int lane = threadIdx.x;
    const size_t row = (size_t)blockIdx.x * iters;
    const double2 *__restrict__ p = src + row * 256;   // 4 KB per slice and wave, as a 16 x 16 ComplexF64 tile
    double *__restrict__ o = out + row;
    double2 buf[R][4];
    auto load = [&](double2(&b)[4], int i) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            b[r] = p[(size_t)i * 256 + r * 64 + lane];
    };
#pragma unroll
    for (int j = 0; j < DEPTH; ++j)
        load(buf[j], j < iters ? j : iters - 1);
    double acc = lane * 1e-3;
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i += R) {
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const int nxt = i + j + DEPTH;
            load(buf[(j + DEPTH) % R], nxt < iters ? nxt : iters - 1);         // (clamped: the load itself sits behind no branch)
            double a = acc;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                a = fma(buf[j][r].x, 1e-9, a) + buf[j][r].y * 1e-9;
            for (int w = 0; w < work; ++w)             // the slice's products: a dependent chain
                a = fma(a, 1.0000001, 1e-12);
            acc = a;
            if (VARIANT == 0) {
                if (lane == 0)
                    o[i + j] = acc;
            } else {
                o[i + j] = acc;                        // all lanes: the same 8 bytes
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    if (lane == 0)
        cycles[blockIdx.x] = t1 - t0;
}

template <int DEPTH, int VARIANT>
static double run(const double2 *src, double *out, long long *cyc, int waves, int iters, int work)
{
    walk<DEPTH, VARIANT><<<waves, 64>>>(src, out, cyc, iters, work);
    (void)hipDeviceSynchronize();
    walk<DEPTH, VARIANT><<<waves, 64>>>(src, out, cyc, iters, work);
    (void)hipDeviceSynchronize();
    std::vector<long long> h(waves);
    (void)hipMemcpy(h.data(), cyc, sizeof(long long) * waves, hipMemcpyDeviceToHost);
    double s = 0;
    for (long long v : h) s += (double)v;
    return s / waves / iters;
}

int main(int argc, char **argv)
{
    const int waves = argc > 1 ? atoi(argv[1]) : 2048, iters = 480, work = argc > 2 ? atoi(argv[2]) : 400;
    double2 *src;
    double *out;
    long long *cyc;
    (void)hipMalloc(&src, sizeof(double2) * 256 * (size_t)waves * iters);
    (void)hipMalloc(&out, sizeof(double) * (size_t)waves * iters);
    (void)hipMalloc(&cyc, sizeof(long long) * waves);
    (void)hipMemset(src, 0, sizeof(double2) * 256 * (size_t)waves * iters);
    printf("%d waves x %d slices of 4 KB, %d dependent FMAs per slice (~%d cycles alone)\n", waves, iters, work, 8 * work);
    printf("ring depth 1: store behind `if (lane == 0)` %8.0f cycles per slice   all lanes store %8.0f\n",
           run<1, 0>(src, out, cyc, waves, iters, work), run<1, 1>(src, out, cyc, waves, iters, work));
    printf("ring depth 2: store behind `if (lane == 0)` %8.0f cycles per slice   all lanes store %8.0f\n",
           run<2, 0>(src, out, cyc, waves, iters, work), run<2, 1>(src, out, cyc, waves, iters, work));
    printf("ring depth 3: store behind `if (lane == 0)` %8.0f cycles per slice   all lanes store %8.0f\n",
           run<3, 0>(src, out, cyc, waves, iters, work), run<3, 1>(src, out, cyc, waves, iters, work));
    return 0;
}
