// sweep_coop.hip -- n = 17..32, unitary flow, time axis in chunks, FEW units (single problems, small ensembles): the
// dependent product chains of sweep_tile.hip with FOUR waves per product.  gfx950.
//
// A single 32 x 32 problem of 2000 slices is 3 S + 2 sqrt(N / S) dependent 32 x 32 products (S slices per chunk: chunk
// product, two-level scan, backward sweep) and nothing else to overlap them with: one wave takes 8 tile products x 12
// matrix-core instructions = 6 144 cycles per product plus a layout conversion, 3.4 us, while 3 of 4 SIMDs idle.  Here
// wave (I, J) of a 256-thread workgroup owns tile (I, J) of every matrix.  All products of these chains have the form
// Z^T W on D-layout registers (a D-layout dump is its own transpose as an A operand, sweep_tile.hip), so the operands a
// wave lacks are PLAIN copies of other waves' registers: every wave writes its tiles of Z and W to LDS as they are
// (conflict-free 16-byte parts), one barrier, reads Z(0, I), Z(1, I), W(1 - I, J), one more barrier, and runs 2 tile
// products (24 matrix-core instructions): ~1 us per product.
//   coop_chunk_product_kernel     = chunk_product_kernel<2>            Q_c = P_hi-1 ... P_lo
//   coop_scan_group_kernel        = chunk_scan_group_kernel<2>         suffix products inside a group of chunks
//   coop_chain_unitary_kernel     = chain_tile_unitary_kernel<2, 0, false, true> in chunk mode (UnitaryGate, sparse controls)
//   coop_scan_kernel              = chunk_scan_kernel<2, 0, false>     the serial scan over the groups, X_N, M_N (UnitaryGate)
// Same arithmetic per element as the one-wave kernels (tprod's three-product complex multiplication, Kt = 0 first), so
// results are bitwise theirs.
#include <cstdlib>

#include "grape_kernels.hpp"
#include "tile.hpp"
#include "done_signal.hpp"

namespace grape {

namespace {

struct CTile {                                    // one 16 x 16 tile in D layout: 4 registers of (re, im) per lane
    d4 re, im;
};

constexpr int kCoopTile = 4096;                   // bytes of a plain tile image: 4 parts x 64 lanes x 16 B
constexpr int kCoopMatrix = 4 * kCoopTile;

// part p of a tile at  tile + 1024 p + 16 lane : (re0 re1) (re2 re3) (im0 im1) (im2 im3)
GRAPE_DEV void coop_write(char *img, int tile, int lane, const CTile &t)
{
    typedef double d2x __attribute__((ext_vector_type(2)));
    d2x *p = reinterpret_cast<d2x *>(img + tile * kCoopTile + 16 * lane);
    p[0] = (d2x){t.re[0], t.re[1]};
    p[64] = (d2x){t.re[2], t.re[3]};
    p[128] = (d2x){t.im[0], t.im[1]};
    p[192] = (d2x){t.im[2], t.im[3]};
}

GRAPE_DEV void coop_read(CTile &t, const char *img, int tile, int lane)
{
    typedef double d2x __attribute__((ext_vector_type(2)));
    const d2x *p = reinterpret_cast<const d2x *>(img + tile * kCoopTile + 16 * lane);
    const d2x a = p[0], b = p[64], c = p[128], d = p[192];
    t.re = (d4){a[0], a[1], b[0], b[1]};
    t.im = (d4){c[0], c[1], d[0], d[1]};
}

// out(I, J) = sum_Kt op(Z(Kt, I))^T op(W(Kt, J)): tprod's arithmetic for one output tile
template <bool CZ, bool CW>
GRAPE_DEV void coop_tile_prod(CTile &out, const CTile &z0, const CTile &z1, const CTile &w0, const CTile &w1)
{
    d4 t1 = {0, 0, 0, 0}, t2 = {0, 0, 0, 0};
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        t1 = __builtin_amdgcn_mfma_f64_16x16x4f64(z0.re[kb], w0.re[kb], t1, 0, 0, 0);
        t2 = __builtin_amdgcn_mfma_f64_16x16x4f64(z0.im[kb], w0.im[kb], t2, 0, 0, 0);
    }
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        t1 = __builtin_amdgcn_mfma_f64_16x16x4f64(z1.re[kb], w1.re[kb], t1, 0, 0, 0);
        t2 = __builtin_amdgcn_mfma_f64_16x16x4f64(z1.im[kb], w1.im[kb], t2, 0, 0, 0);
    }
    d4 t3;
    if (CZ != CW) {                               // Zi Wi enters with the opposite sign
        out.re = t1 + t2;
        t3 = t2 - t1;
    } else {
        out.re = t1 - t2;
        t3 = -(t1 + t2);
    }
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        const double zs = CZ ? z0.re[kb] - z0.im[kb] : z0.re[kb] + z0.im[kb];
        const double ws = CW ? w0.re[kb] - w0.im[kb] : w0.re[kb] + w0.im[kb];
        t3 = __builtin_amdgcn_mfma_f64_16x16x4f64(zs, ws, t3, 0, 0, 0);
    }
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        const double zs = CZ ? z1.re[kb] - z1.im[kb] : z1.re[kb] + z1.im[kb];
        const double ws = CW ? w1.re[kb] - w1.im[kb] : w1.re[kb] + w1.im[kb];
        t3 = __builtin_amdgcn_mfma_f64_16x16x4f64(zs, ws, t3, 0, 0, 0);
    }
    out.im = t3;
}

// out = op(Z)^T op(W) for the workgroup's four waves; z, w: this wave's tiles (I, J).  s_z, s_w: two 16 KB images.
// new_z / new_w: the image of that factor has to be (re)written (false: it still holds this factor from the last call).
// Every wave reads all four operand tiles back (its own two included) and adds Kt = 0 first, as tprod does: the results
// are bitwise those of the one-wave kernels.
template <bool CZ, bool CW>
GRAPE_DEV void coop_tn_ordered(CTile &out, const CTile &z, const CTile &w, char *s_z, char *s_w, int I, int J, int lane,
                               bool new_z, bool new_w)
{
    if (new_z)
        coop_write(s_z, 2 * I + J, lane, z);
    if (new_w)
        coop_write(s_w, 2 * I + J, lane, w);
    __syncthreads();
    CTile z0, z1, w0, w1;
    coop_read(z0, s_z, I, lane);                                   // Z(0, I)
    coop_read(z1, s_z, 2 + I, lane);                               // Z(1, I)
    coop_read(w0, s_w, J, lane);                                   // W(0, J)
    coop_read(w1, s_w, 2 + J, lane);                               // W(1, J)
    __syncthreads();
    coop_tile_prod<CZ, CW>(out, z0, z1, w0, w1);
}

// the scans' step: out = Z^T W and, from the same two barriers, wt = tile (I, J) of W^T (W's tiles also go to the padded
// images img, read back with tile.hpp's transposing pattern) -- a separate transpose would cost three more barriers per step
GRAPE_DEV void coop_tn_and_transpose(CTile &out, CTile &wt, const CTile &z, const CTile &w, char *s_z, char *s_w, double2 *img,
                                     int I, int J, int lane)
{
    const int rho = lane & 15, q = lane >> 4;
    const int wr = 17 * (lane >> 4) + (lane & 15), rd = 68 * (rho >> 2) + 17 * (rho & 3) + q;
    coop_write(s_z, 2 * I + J, lane, z);
    coop_write(s_w, 2 * I + J, lane, w);
    double2 *mine = img + (2 * I + J) * kTileImage;
#pragma unroll
    for (int r = 0; r < 4; ++r)
        mine[68 * r + wr] = make_double2(w.re[r], w.im[r]);
    __syncthreads();
    CTile z0, z1, w0, w1;
    coop_read(z0, s_z, I, lane);
    coop_read(z1, s_z, 2 + I, lane);
    coop_read(w0, s_w, J, lane);
    coop_read(w1, s_w, 2 + J, lane);
    const double2 *other = img + (2 * J + I) * kTileImage;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        const double2 x = other[rd + 4 * kb];
        wt.re[kb] = x.x;
        wt.im[kb] = x.y;
    }
    __syncthreads();
    coop_tile_prod<false, false>(out, z0, z1, w0, w1);
}

GRAPE_DEV void coop_load(CTile &t, const double2 *__restrict__ dump, int tile, int lane)
{
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const double2 v = dump[(tile * 4 + r) * 64 + lane];
        t.re[r] = v.x;
        t.im[r] = v.y;
    }
}

GRAPE_DEV void coop_store(double2 *__restrict__ dump, int tile, int lane, const CTile &t)
{
#pragma unroll
    for (int r = 0; r < 4; ++r)
        dump[(tile * 4 + r) * 64 + lane] = make_double2(t.re[r], t.im[r]);
}

GRAPE_DEV void coop_identity(CTile &t, int I, int J, int lane)
{
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        t.re[r] = (I == J && 4 * r + (lane >> 4) == (lane & 15)) ? 1.0 : 0.0;
        t.im[r] = 0.0;
    }
}

// tile (I, J) of V^T: every wave writes its tile of V into a padded image (tile.hpp), wave (I, J) reads tile (J, I) back
// with the transposing pattern.  img: 4 x kTileImage double2 (may alias the product images: barriers on both sides)
GRAPE_DEV void coop_transpose(CTile &vt, const CTile &v, double2 *img, int I, int J, int lane)
{
    const int rho = lane & 15, q = lane >> 4;
    const int wr = 17 * (lane >> 4) + (lane & 15), rd = 68 * (rho >> 2) + 17 * (rho & 3) + q;
    __syncthreads();
    double2 *mine = img + (2 * I + J) * kTileImage;
#pragma unroll
    for (int r = 0; r < 4; ++r)
        mine[68 * r + wr] = make_double2(v.re[r], v.im[r]);
    __syncthreads();
    const double2 *other = img + (2 * J + I) * kTileImage;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        const double2 x = other[rd + 4 * kb];
        vt.re[kb] = x.x;
        vt.im[kb] = x.y;
    }
    __syncthreads();
}

constexpr size_t kCoopLds = 2 * kCoopMatrix > 4 * kTileImage * 16 ? 2 * kCoopMatrix : 4 * kTileImage * 16;

}  // namespace

__global__ __launch_bounds__(256) void coop_chunk_product_kernel(const TileParams p)
{
    constexpr int TSZ = 1024;
    extern __shared__ double2 s_coop[];
    char *s_z = reinterpret_cast<char *>(s_coop), *s_w = s_z + kCoopMatrix;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, I = wave >> 1, J = wave & 1, tile = 2 * I + J;
    const int k = blockIdx.x, c = blockIdx.z, N = p.N, C = p.tp_chunks;
    const size_t kw = (size_t)blockIdx.y * p.E + k;
    const double2 *__restrict__ Pk = p.props + kw * N * TSZ;
    const int t_lo = c * p.tp_S, t_hi = min(N, t_lo + p.tp_S);
    CTile V, Y, Pm, Pn;
    coop_identity(V, I, J, lane);
    coop_load(Pm, Pk + (size_t)(t_hi - 1) * TSZ, tile, lane);
    for (int t = t_hi - 1; t >= t_lo; --t) {                       // V <- P_t^T V: ends as (P_hi-1 ... P_lo)^T
        coop_load(Pn, Pk + (size_t)max(t - 1, 0) * TSZ, tile, lane);
        coop_tn_ordered<false, false>(Y, Pm, V, s_z, s_w, I, J, lane, true, true);
        V = Y;
        Pm = Pn;
    }
    if (p.tp_qt)
        coop_store(p.tp_qt + (kw * C + c) * TSZ, tile, lane, V);   // Q_c^T
    coop_transpose(Y, V, s_coop, I, J, lane);
    coop_store(p.tp_q + (kw * C + c) * TSZ, tile, lane, Y);
}

__global__ __launch_bounds__(256) void coop_scan_group_kernel(const TileParams p)
{
    constexpr int TSZ = 1024;
    extern __shared__ double2 s_coop[];
    char *s_z = reinterpret_cast<char *>(s_coop), *s_w = s_z + kCoopMatrix;
    double2 *s_t = s_coop + 2 * kCoopMatrix / 16;                  // transposes: their own images (they overlap the next product)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, I = wave >> 1, J = wave & 1, tile = 2 * I + J;
    const int k = blockIdx.x, j = blockIdx.z, C = p.tp_chunks, G = p.tp_groups;
    const size_t kw = (size_t)blockIdx.y * p.E + k;
    const double2 *__restrict__ Qk = p.tp_q + kw * C * TSZ;
    double2 *__restrict__ Rk = p.tp_r + kw * C * TSZ;
    const int c_lo = j * p.tp_gsize, c_hi = min(C, c_lo + p.tp_gsize);
    CTile V, Y, T, Q, Qn;
    coop_identity(V, I, J, lane);
    coop_load(Q, Qk + (size_t)(c_hi - 1) * TSZ, tile, lane);
    for (int c = c_hi - 1; c >= c_lo; --c) {
        coop_load(Qn, Qk + (size_t)max(c - 1, 0) * TSZ, tile, lane);
        coop_tn_and_transpose(Y, T, Q, V, s_z, s_w, s_t, I, J, lane);
        coop_store(Rk + (size_t)c * TSZ, tile, lane, T);           // product of the chunks after c inside the group
        V = Y;
        Q = Qn;
    }
    coop_transpose(Y, V, s_t, I, J, lane);
    coop_store(p.tp_a + ((kw + (size_t)gridDim.y * p.E) * G + j) * TSZ, tile, lane, Y);   // the group's product, behind the A_j block
}

// chunk_scan_kernel<2, SAND, false>: the serial scan over the chunks (or groups), then X_N = T Xi [T'] and
// M_N = X_N Xt' [- Xt' X_N], z = tr(X_N' Xt) (sandwich)
template <int SAND>
__global__ __launch_bounds__(256) void coop_scan_kernel(const TileParams p)
{
    constexpr int TSZ = 1024;
    extern __shared__ double2 s_coop[];
    char *s_z = reinterpret_cast<char *>(s_coop), *s_w = s_z + kCoopMatrix;
    double2 *s_t = s_coop + 2 * kCoopMatrix / 16;                  // padded images: 4 tiles, and 4 more for the last product
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, I = wave >> 1, J = wave & 1, tile = 2 * I + J;
    const int k = blockIdx.x, K = p.K, C = p.tp_chunks;
    const size_t kw = (size_t)blockIdx.y * p.E + k;
    const double2 *__restrict__ ops = p.ops + (size_t)k * (2 * K + 3) * TSZ;
    const double2 *__restrict__ Qk = p.tp_q + kw * C * TSZ;
    double2 *__restrict__ Rk = p.tp_r + kw * C * TSZ;
    CTile V, Y, T, Q, Qn;
    coop_identity(V, I, J, lane);                                  // V = R_c^T, from the last chunk down
    coop_load(Q, Qk + (size_t)(C - 1) * TSZ, tile, lane);
    for (int c = C - 1; c >= 0; --c) {
        coop_load(Qn, Qk + (size_t)max(c - 1, 0) * TSZ, tile, lane);
        coop_tn_and_transpose(Y, T, Q, V, s_z, s_w, s_t, I, J, lane);                        // R_{c-1}^T = Q_c^T R_c^T; T = R_c
        coop_store(Rk + (size_t)c * TSZ, tile, lane, T);           // R_c
        V = Y;
        Q = Qn;
    }
    // V = T^T: X_N = T Xi, M_N = X_N Xt'
    CTile X, L, M;
    {
        CTile Xi;
        coop_load(Xi, ops + (size_t)(1 + 2 * K) * TSZ, tile, lane);
        coop_tn_ordered<false, false>(X, V, Xi, s_z, s_w, I, J, lane, true, true);          // (T^T)^T Xi
    }
    const int rho = lane & 15, q = lane >> 4;
    const int wr = 17 * (lane >> 4) + (lane & 15), rd = 68 * (rho >> 2) + 17 * (rho & 3) + q;
    double2 *imgX = s_t, *imgL = s_t + 4 * kTileImage;
    if (SAND) {                                                    // X_N = (T Xi) T' = (T Xi) conj(T^T): A-layout of T Xi, V as it is
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r)
            imgX[tile * kTileImage + 68 * r + wr] = make_double2(X.re[r], X.im[r]);
        coop_write(s_w, tile, lane, V);
        __syncthreads();
        CTile x0, x1, v0, v1;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            const double2 a = imgX[(2 * I) * kTileImage + rd + 4 * kb], b = imgX[(2 * I + 1) * kTileImage + rd + 4 * kb];
            x0.re[kb] = a.x; x0.im[kb] = a.y;
            x1.re[kb] = b.x; x1.im[kb] = b.y;
        }
        coop_read(v0, s_w, J, lane);
        coop_read(v1, s_w, 2 + J, lane);
        __syncthreads();
        coop_tile_prod<false, true>(Y, x0, x1, v0, v1);
        X = Y;
    }
    coop_load(L, ops + (size_t)(2 + 2 * K) * TSZ, tile, lane);     // L_N = Xt
    {
        // M(I, J) = sum_Kt X(I, Kt) conj(L(J, Kt))^T: both factors as A-layout operands = transposing reads of their images
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            imgX[tile * kTileImage + 68 * r + wr] = make_double2(X.re[r], X.im[r]);
            imgL[tile * kTileImage + 68 * r + wr] = make_double2(L.re[r], L.im[r]);
        }
        __syncthreads();
        CTile x0, x1, l0, l1;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            const double2 a = imgX[(2 * I) * kTileImage + rd + 4 * kb], b = imgX[(2 * I + 1) * kTileImage + rd + 4 * kb];
            const double2 c = imgL[(2 * J) * kTileImage + rd + 4 * kb], d = imgL[(2 * J + 1) * kTileImage + rd + 4 * kb];
            x0.re[kb] = a.x; x0.im[kb] = a.y;
            x1.re[kb] = b.x; x1.im[kb] = b.y;
            l0.re[kb] = c.x; l0.im[kb] = c.y;
            l1.re[kb] = d.x; l1.im[kb] = d.y;
        }
        coop_tile_prod<false, true>(M, x0, x1, l0, l1);
    }
    if (SAND) {
        coop_tn_ordered<true, false>(Y, L, X, s_z, s_w, I, J, lane, true, true);           // L' X
        M.re -= Y.re;
        M.im -= Y.im;
        double zz[2] = {0.0, 0.0};                                 // tr(X' L): own tile, then the four waves
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            zz[0] = fma(X.re[r], L.re[r], zz[0]);
            zz[0] = fma(X.im[r], L.im[r], zz[0]);
            zz[1] = fma(X.re[r], L.im[r], zz[1]);
            zz[1] = fma(-X.im[r], L.re[r], zz[1]);
        }
        wave_sum_n(zz);
        double *s_red = reinterpret_cast<double *>(s_coop);
        __syncthreads();
        if (lane == 0) {
            s_red[2 * wave] = zz[0];
            s_red[2 * wave + 1] = zz[1];
        }
        __syncthreads();
        if (wave == 0) {
            const double zr = (s_red[0] + s_red[2]) + (s_red[4] + s_red[6]), zi = (s_red[1] + s_red[3]) + (s_red[5] + s_red[7]);
            p.tp_z[(kw * 64 + lane) * 2] = zr;
            p.tp_z[(kw * 64 + lane) * 2 + 1] = zi;
        }
    }
    coop_store(p.tp_m + kw * TSZ, tile, lane, M);
}

// chain_tile_unitary_kernel<2, SAND, false, true> in chunk mode: sparse control operators, grid.z = chunk
template <int SAND>
__global__ __launch_bounds__(256) void coop_chain_unitary_kernel(const TileParams p)
{
    constexpr int TSZ = 1024, NT = 2, MS = 16 * NT + 1;
    extern __shared__ double2 s_coop[];
    char *s_z = reinterpret_cast<char *>(s_coop), *s_w = s_z + kCoopMatrix;
    double2 *s_coef = s_coop + 2 * kCoopMatrix / 16;
    double2 *s_M = s_coef + (size_t)p.K * p.sp_nz;
    int *s_addr = reinterpret_cast<int *>(s_M + 16 * NT * MS);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, I = wave >> 1, J = wave & 1, tile = 2 * I + J;
    const int k = blockIdx.x, K = p.K, N = p.N, C = p.tp_chunks;
    {
        const double2 *__restrict__ gc = p.sp_coef + (size_t)k * K * p.sp_nz;
        const int32_t *__restrict__ ga = p.sp_addr + (size_t)k * K * p.sp_nz;
        for (int i = threadIdx.x; i < K * p.sp_nz; i += 256) {
            s_coef[i] = gc[i];
            s_addr[i] = ga[i];
        }
    }
    const size_t kw = (size_t)blockIdx.y * p.E + k;
    const double2 *__restrict__ Pk = p.props + kw * N * TSZ;
    double *__restrict__ out = p.member_out + ((size_t)blockIdx.y * p.E_members + k) * ((size_t)K * N + 1);
    const int t_lo = (int)blockIdx.z * p.tp_S, t_hi = min(N, t_lo + p.tp_S);
    CTile M, Y, Pm, Pn, Pnn;
    coop_load(M, p.tp_m + kw * TSZ, tile, lane);
    if (p.tp_groups) {                                             // two-level scan: R = A_group R_local
        coop_load(Pm, p.tp_a + (kw * p.tp_groups + blockIdx.z / p.tp_gsize) * TSZ, tile, lane);
        coop_tn_ordered<false, true>(Y, M, Pm, s_z, s_w, I, J, lane, true, true);
        coop_tn_ordered<false, false>(M, Y, Pm, s_z, s_w, I, J, lane, true, false);       // A' M_N A
    }
    coop_load(Pm, p.tp_r + (kw * C + blockIdx.z) * TSZ, tile, lane);
    coop_tn_ordered<false, true>(Y, M, Pm, s_z, s_w, I, J, lane, true, true);              // (R' M)^T
    coop_tn_ordered<false, false>(M, Y, Pm, s_z, s_w, I, J, lane, true, false);            // R' M R
    const double gs = SAND ? -p.dt : (p.variant == 0 ? -2.0 * p.dt : 2.0 * p.dt);
    // z = conj(tr M), the same for every t: through the image of M the traces read anyway
    auto image_of_M = [&]() {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            s_M[(16 * I + 4 * r + (lane >> 4)) * MS + 16 * J + (lane & 15)] = make_double2(M.re[r], M.im[r]);
        __syncthreads();
    };
    double zr = 0.0, zi = 0.0;
    if (SAND) {                                                    // tr(X' L), the same for every t, from the scan kernel
        zr = p.tp_z[(kw * 64 + lane) * 2];
        zi = p.tp_z[(kw * 64 + lane) * 2 + 1];
    } else
        image_of_M();
    if (!SAND && wave == 0) {
        double zz[2] = {0.0, 0.0};
        if (lane < 32) {
            const double2 d = s_M[lane * MS + lane];
            zz[0] = d.x;
            zz[1] = d.y;
        }
        wave_sum_n(zz);
        zr = zz[0];
        zi = -zz[1];
    }
    coop_load(Pm, Pk + (size_t)(t_hi - 1) * TSZ, tile, lane);
    coop_load(Pn, Pk + (size_t)max(t_hi - 2, 0) * TSZ, tile, lane);
    for (int t = t_hi - 1; t >= t_lo; --t) {
        coop_load(Pnn, Pk + (size_t)max(t - 2, 0) * TSZ, tile, lane);                      // two slices in flight
        coop_tn_ordered<false, true>(Y, M, Pm, s_z, s_w, I, J, lane, true, true);          // (P' M)^T
        coop_tn_ordered<false, false>(M, Y, Pm, s_z, s_w, I, J, lane, true, false);        // P' M P
        image_of_M();
        if (wave == 0) {                                           // the list part of sparse_traces (sweep_tile.hip)
            for (int c0 = 0; c0 < K; c0 += 8) {
                double q16[16];
#pragma unroll
                for (int cc = 0; cc < 8; ++cc) {
                    q16[cc] = 0.0;
                    q16[8 + cc] = 0.0;
                    if (c0 + cc < K) {
                        const int nz = p.sp_nz;
                        const double2 cf = s_coef[(c0 + cc) * nz + lane];
                        const double2 mv = s_M[s_addr[(c0 + cc) * nz + lane]];
                        const double pr = cf.x * mv.x - cf.y * mv.y, pi = cf.x * mv.y + cf.y * mv.x;
                        q16[cc] = SAND ? pi : fma(pr, zi, pi * zr);
                        for (int e = lane + 64; e < nz; e += 64) {
                            const double2 cf2 = s_coef[(c0 + cc) * nz + e];
                            const double2 mv2 = s_M[s_addr[(c0 + cc) * nz + e]];
                            const double pr2 = cf2.x * mv2.x - cf2.y * mv2.y, pi2 = cf2.x * mv2.y + cf2.y * mv2.x;
                            q16[cc] += SAND ? pi2 : fma(pr2, zi, pi2 * zr);
                        }
                    }
                }
                const double tot = reduce_scatter16(q16);
                const int c = c0 + (lane >> 2);
                if ((lane & 3) == 0 && lane < 32 && c < K) {
                    out[(size_t)t * K + c] = gs * tot;
                    fold_store(p, (size_t)t * K + c, gs * tot);
                }
            }
            if (t == N - 1 && lane == 0) {
                double Fk;
                if (SAND) {
                    const double inv = 1.0 / (double)p.n, ar = zr * inv, ai = zi * inv;
                    Fk = 1.0 - (ar * ar + ai * ai);
                } else {
                    Fk = zr * zr - zi * zi;
                }
                out[(size_t)K * N] = Fk;
                fold_store(p, (size_t)K * N, Fk);
            }
        }
        Pm = Pn;
        Pn = Pnn;
    }
    if (wave == 0 && lane == 0)                                    // (wave 0 made every store of this workgroup)
        fold_publish(p);
}

// do the one-wave kernels of this launch leave most SIMDs idle?  (then four waves per product pay)
bool coop_applies(const TileParams &p, int sandwich, bool keepl)
{
    static const bool off = std::getenv("GRAPE_NO_COOP") != nullptr;
    if (off || tile_count(p.n) != 2 || !p.unitary || keepl || p.tp_chunks < 2)
        return false;
    const long waves = (long)p.E * p.n_x * p.tp_chunks, cus = p.cus > 0 ? p.cus : 256;
    static const char *lim = std::getenv("GRAPE_COOP_MAX");        // tuning: largest number of (unit, chunk) pairs
    (void)sandwich;
    return waves <= (lim ? std::atol(lim) : 3 * cus);              // measured: 668 pairs (4 units) 0.418 -> 0.350 ms, 1024 pairs (8 units) 0.603 -> 0.615
}

size_t coop_chain_lds(const TileParams &p)
{
    return 2 * kCoopMatrix + sizeof(double2) * ((size_t)p.K * p.sp_nz + 16 * 2 * 33) + sizeof(int32_t) * (size_t)p.K * p.sp_nz;
}

hipError_t launch_coop_chunk_product(const TileParams &q, hipStream_t stream)
{
    GRAPE_LAUNCH(coop_chunk_product_kernel, dim3(q.E, q.n_x, q.tp_chunks), dim3(256), kCoopLds, stream, q);
    return hipGetLastError();
}

hipError_t launch_coop_scan_group(const TileParams &q, hipStream_t stream)
{
    GRAPE_LAUNCH(coop_scan_group_kernel, dim3(q.E, q.n_x, q.tp_groups), dim3(256),
                       2 * kCoopMatrix + 4 * kTileImage * 16, stream, q);
    return hipGetLastError();
}

hipError_t launch_coop_scan(int sandwich, const TileParams &q, hipStream_t stream)
{
    const size_t lds = 2 * kCoopMatrix + 8 * kTileImage * 16;
    if (sandwich) GRAPE_LAUNCH(coop_scan_kernel<1>, dim3(q.E, q.n_x), dim3(256), lds, stream, q);
    else          GRAPE_LAUNCH(coop_scan_kernel<0>, dim3(q.E, q.n_x), dim3(256), lds, stream, q);
    return hipGetLastError();
}

hipError_t launch_coop_chain_unitary(int sandwich, const TileParams &q, hipStream_t stream)
{
    const size_t lds = coop_chain_lds(q);
    auto kern = sandwich ? coop_chain_unitary_kernel<1> : coop_chain_unitary_kernel<0>;
    if (lds > 64 * 1024) {
        hipError_t e = ensure_dynamic_lds((const void *)kern, lds);
        if (e != hipSuccess)
            return e;
    }
    GRAPE_LAUNCH_AS("coop_chain_unitary_kernel", kern, dim3(q.E, q.n_x, q.tp_chunks), dim3(256), lds, stream, q);
    return hipGetLastError();
}

}  // namespace grape
