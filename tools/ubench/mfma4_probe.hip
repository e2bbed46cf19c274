// mfma4_probe.hip -- determines the lane layouts of v_mfma_f64_4x4x4_4b_f64 on gfx950 by brute force:
// D_b = A_b * B_b for 4 blocks; which lane holds A_b[i][k], B_b[k][j], D_b[i][j]?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>

__global__ void probe(const double *a, const double *b, double *d)
{
    const int l = threadIdx.x;
    d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, 0, 0, 0);
}

int main()
{
    double ha[64], hb[64], hd[64], *da, *db, *dd;
    srand(1);
    for (int i = 0; i < 64; ++i) { ha[i] = rand() % 17 - 8; hb[i] = rand() % 13 - 6; }
    hipMalloc(&da, 512); hipMalloc(&db, 512); hipMalloc(&dd, 512);
    hipMemcpy(da, ha, 512, hipMemcpyHostToDevice); hipMemcpy(db, hb, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dd);
    hipMemcpy(hd, dd, 512, hipMemcpyDeviceToHost);
    // lane = 16*x + 4*y + z with (x,y,z) a permutation of the three indices
    const int perms[6][3] = {{0,1,2},{0,2,1},{1,0,2},{1,2,0},{2,0,1},{2,1,0}};
    auto lane_of = [&](const int *p, int u0, int u1, int u2) { int v[3] = {u0, u1, u2}; return 16 * v[p[0]] + 4 * v[p[1]] + v[p[2]]; };
    int found = 0;
    for (int pa = 0; pa < 6; ++pa) for (int pb = 0; pb < 6; ++pb) for (int pd = 0; pd < 6; ++pd) {
        bool ok = true;
        for (int blk = 0; blk < 4 && ok; ++blk) for (int i = 0; i < 4 && ok; ++i) for (int j = 0; j < 4 && ok; ++j) {
            double s = 0;
            for (int k = 0; k < 4; ++k) s += ha[lane_of(perms[pa], blk, i, k)] * hb[lane_of(perms[pb], blk, k, j)];
            if (fabs(s - hd[lane_of(perms[pd], blk, i, j)]) > 1e-9) ok = false;
        }
        if (ok) { ++found; printf("MATCH A(b,i,k) perm %d%d%d  B(b,k,j) perm %d%d%d  D(b,i,j) perm %d%d%d\n",
                                  perms[pa][0], perms[pa][1], perms[pa][2], perms[pb][0], perms[pb][1], perms[pb][2],
                                  perms[pd][0], perms[pd][1], perms[pd][2]); }
    }
    printf("matches: %d  (perm xyz means lane = 16*idx[x] + 4*idx[y] + idx[z], idx = (b, row-ish, col-ish) as listed)\n", found);
    for (int l = 0; l < 64; ++l) printf("%g%c", hd[l], (l % 16 == 15) ? '\n' : ' ');
    return 0;
}
