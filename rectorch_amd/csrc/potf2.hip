// potf2.hip -- the leaf of the EASE solver's recursive Cholesky (ease.hip): one 128x128 diagonal block.
#include "rtx_dgemm.h"

// One 128x128 diagonal block: W = inv(chol(Akk)) written to the diagonal blocks of W (lower) and W^T (upper).
// 256 threads as a 16 x 16 grid; thread (tr, tc) keeps the 8 x 8 elements (tr + 16 i, tc + 16 jx) of the block in
// REGISTERS for the whole kernel.  Every step publishes one column (and, for the inverse, one row) through a
// double-buffered 1-KB LDS line, so a step is one barrier + <= 17 independent LDS reads + <= 36 register FMAs per thread:
//   phase 1  right-looking Cholesky with UNSCALED columns (a_rc keeps a_rc^(j); l_rc = a_rc / sqrt(a_cc))
//   phase 2  columns scaled to L
//   phase 3  in-place inverse: step j folds row j of W into the rows below; a_rc, c <= j, is then the unnormalised
//            row r of W (it replaces the consumed column of L); W[r][c] = a_rc / l_rr is applied on the way out
// The loops over the 16-wide column blocks are unrolled (register indices and the live sub-blocks are compile-time),
// the 16 columns inside a block are a run-time loop.  (A first version with 1024 threads x 16 elements took 208 us:
// its 16 waves issued 17 LDS reads per 16 FMAs and the LDS pipe, not the FMA pipe, set the pace.)
__device__ __forceinline__ double rtx_rcp_f64(double x)
{
    double y = __builtin_amdgcn_rcp(x);
    y = y * (2.0 - x * y);
    return y * (2.0 - x * y);
}

// phase 1, columns 16 JB .. 16 JB + 15.  false: the block is not positive definite
template <int JB>
__device__ __forceinline__ bool potf2_chol_block(double (&a)[8][8], double (*colbuf)[128], double* dinv, int tid, int tr, int tc)
{
#pragma nounroll
    for (int jc = 0; jc < 16; ++jc) {
        const int j = JB * 16 + jc;
        double* cb = colbuf[j & 1];
        if (tc == jc) {
#pragma unroll
            for (int i = JB; i < 8; ++i) {
                const int r = tr + 16 * i;
                if (r >= j) cb[r] = a[i][JB];
            }
        }
        __syncthreads();
        const double ajj = cb[j];
        if (!(ajj > 0.0)) return false;   // uniform: every thread reads the same value
        if (tid == 0) dinv[j] = 1.0 / sqrt(ajj);
        const double rinv = rtx_rcp_f64(ajj);
        double mi[8], cv[8];
#pragma unroll
        for (int i = JB; i < 8; ++i) mi[i] = cb[tr + 16 * i] * rinv;
#pragma unroll
        for (int jx = JB; jx < 8; ++jx) cv[jx] = cb[tc + 16 * jx];
#pragma unroll
        for (int i = JB; i < 8; ++i)
#pragma unroll
            for (int jx = JB; jx <= i; ++jx) {
                // element (r, c) = (tr + 16 i, tc + 16 jx): updated iff j < c <= r
                const bool right = (jx > JB) || (tc > jc);
                const bool lower = (i > jx) || (tc <= tr);
                if (right && lower) a[i][jx] -= mi[i] * cv[jx];
            }
    }
    return true;
}

// phase 3, columns 16 JB .. 16 JB + 15
template <int JB>
__device__ __forceinline__ void potf2_inv_block(double (&a)[8][8], const double (&dr)[8], double (*colbuf)[128], double (*rowbuf)[128], int tr,
                                                int tc)
{
#pragma nounroll
    for (int jc = 0; jc < 16; ++jc) {
        const int j = JB * 16 + jc;
        double* cb = colbuf[j & 1];
        double* rb = rowbuf[j & 1];
        if (tc == jc) {   // column j of L below the diagonal
#pragma unroll
            for (int i = JB; i < 8; ++i) {
                const int r = tr + 16 * i;
                if (r > j) cb[r] = a[i][JB];
            }
        }
        if (tr == jc) {   // row j (= tr + 16 JB) of W, normalised
#pragma unroll
            for (int jx = 0; jx <= JB; ++jx) {
                const int c = tc + 16 * jx;
                if (c < j) rb[c] = a[JB][jx] * dr[JB];
                else if (c == j) rb[c] = dr[JB];
            }
        }
        __syncthreads();
        double wv[8];
#pragma unroll
        for (int jx = 0; jx <= JB; ++jx) wv[jx] = rb[tc + 16 * jx];
#pragma unroll
        for (int i = JB; i < 8; ++i) {
            const int r = tr + 16 * i;
            if (r > j) {
                const double lrj = cb[r];
#pragma unroll
                for (int jx = 0; jx <= JB; ++jx) {
                    const int c = tc + 16 * jx;
                    if (c < j) a[i][jx] -= lrj * wv[jx];
                    else if (c == j) a[i][jx] = -lrj * wv[jx];
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_potf2_inv(const double* Akk, long ld, double* Wkk, double* WTkk, long ldw, int* status)
{
    extern __shared__ __attribute__((aligned(16))) double t[];   // [128][129] staging tile
    __shared__ double colbuf[2][128], rowbuf[2][128], dinv[128];
    const int tid = threadIdx.x, tr = tid >> 4, tc = tid & 15;
    for (int e = tid; e < 128 * 128; e += 256) t[(e >> 7) * 129 + (e & 127)] = Akk[(size_t)(e >> 7) * ld + (e & 127)];
    __syncthreads();
    double a[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int jx = 0; jx < 8; ++jx) a[i][jx] = t[(tr + 16 * i) * 129 + tc + 16 * jx];

    bool ok = potf2_chol_block<0>(a, colbuf, dinv, tid, tr, tc);
    ok = ok && potf2_chol_block<1>(a, colbuf, dinv, tid, tr, tc);
    ok = ok && potf2_chol_block<2>(a, colbuf, dinv, tid, tr, tc);
    ok = ok && potf2_chol_block<3>(a, colbuf, dinv, tid, tr, tc);
    ok = ok && potf2_chol_block<4>(a, colbuf, dinv, tid, tr, tc);
    ok = ok && potf2_chol_block<5>(a, colbuf, dinv, tid, tr, tc);
    ok = ok && potf2_chol_block<6>(a, colbuf, dinv, tid, tr, tc);
    ok = ok && potf2_chol_block<7>(a, colbuf, dinv, tid, tr, tc);
    if (!ok) {
        if (tid == 0) *status = 1;
        return;
    }
    __syncthreads();
    // ---- phase 2: l_rc = a_rc / sqrt(a_cc) for c < r
#pragma unroll
    for (int jx = 0; jx < 8; ++jx) {
        const double dc = dinv[tc + 16 * jx];
#pragma unroll
        for (int i = jx; i < 8; ++i)
            if ((i > jx) || (tc < tr)) a[i][jx] *= dc;
    }
    double dr[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) dr[i] = dinv[tr + 16 * i];
    potf2_inv_block<0>(a, dr, colbuf, rowbuf, tr, tc);
    potf2_inv_block<1>(a, dr, colbuf, rowbuf, tr, tc);
    potf2_inv_block<2>(a, dr, colbuf, rowbuf, tr, tc);
    potf2_inv_block<3>(a, dr, colbuf, rowbuf, tr, tc);
    potf2_inv_block<4>(a, dr, colbuf, rowbuf, tr, tc);
    potf2_inv_block<5>(a, dr, colbuf, rowbuf, tr, tc);
    potf2_inv_block<6>(a, dr, colbuf, rowbuf, tr, tc);
    potf2_inv_block<7>(a, dr, colbuf, rowbuf, tr, tc);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int jx = 0; jx < 8; ++jx) {
            const int r = tr + 16 * i, c = tc + 16 * jx;
            t[r * 129 + c] = (c < r) ? a[i][jx] * dr[i] : (c == r ? dr[i] : 0.0);
        }
    __syncthreads();
    for (int e = tid; e < 128 * 128; e += 256) {
        const int i = e >> 7, j = e & 127;
        Wkk[(size_t)i * ldw + j] = t[i * 129 + j];
        WTkk[(size_t)i * ldw + j] = t[j * 129 + i];
    }
}


// ---------------------------------------------------------------------------------------------------------------------------
// Round 3: the BLOCKED leaf.  The kernel above walks 256 dependent steps of one barrier + one LDS round trip each (0.5 us per
// step, 128 us per leaf, 158 leaves = 20 ms of the 200-ms fit).  Here the 128x128 block lives in LDS and is processed in four
// 32-column panels:
//   potrf  per panel  (a) ONE WAVE factors the 32x32 diagonal block and inverts its factor entirely in registers: lane i holds
//                         row i (32 doubles), the pivot column travels by v_readlane (uniform operands, no LDS, no barrier):
//                         496 readlane pairs + FMAs, then 496 LDS broadcast reads + FMAs for W_pp = inv(L_pp), which replaces
//                         L_pp in the block and goes, transposed with explicit zeros, to a 32x32 side buffer
//                     (b) panel  L21 = A21 W_pp^T       as a small matrix product over all 256 threads
//                     (c) update A22 -= L21 L21^T       16 x 16 threads x (6 x 6 .. 2 x 2) register tiles
//   trtri  in place, block columns right to left: W21 = -W22 (L21 W_pp)  (W22 is already the inverse of the trailing block)
// ~30 barriers instead of 256.
__device__ __forceinline__ double rl64(double v, int lane)
{
    const long long u = __builtin_bit_cast(long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(u & 0xffffffffll), lane);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(u >> 32), lane);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

#define PB_LD 129
#define PB_WLD 33
// 1 / sqrt(d): v_rsq_f64 + two Newton steps (full double precision; a sqrt + a division would be ~150 dependent cycles per column)
__device__ __forceinline__ double rtx_rsqrt_f64(double d)
{
    double y = __builtin_amdgcn_rsq(d);
    y = y * (1.5 - 0.5 * d * y * y);
    return y * (1.5 - 0.5 * d * y * y);
}

// potrf of panel P (columns 32 P .. 32 P + 31): compile-time sizes -- with the row-group count as a run-time value the
// `if (ii < ni)` guards became a branch (and an s_waitcnt) per element: 150 us per leaf instead of 114
template <int P>
__device__ __forceinline__ bool pb_potrf_panel(double* M, double* Wt, int* bad, int tid, int ty, int tx, int lane, int wave)
{
    constexpr int c0 = 32 * P, r0 = c0 + 32, NI = (128 - r0) / 16;   // rows [r0, 128) lie below the diagonal block: NI groups of 16
    if (wave == 0) {
        const int i = lane & 31;   // (both halves of the wave compute the same rows: the readlanes name lanes < 32)
        double a[32], x[32], rinv[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) a[k] = (k <= i) ? M[(c0 + i) * PB_LD + c0 + k] : 0.0;
        bool ok = true;
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const double d = rl64(a[j], j);            // a_jj after the updates of columns < j (uniform)
            if (!(d > 0.0)) ok = false;
            const double r = rtx_rsqrt_f64(d);
            rinv[j] = r;
            const double lj = a[j] * r;                // l_ij (0 above the diagonal, sqrt(d) on it)
            a[j] = lj;
#pragma unroll
            for (int k = j + 1; k < 32; ++k) a[k] -= lj * rl64(lj, k);   // only lanes i >= k hold a live a[k]; the others are never read
        }
        // L_pp goes to LDS first: the inverse below reads its elements from there (the same address in every lane: a broadcast
        // read) -- taking them from the owning lane by readlane kept ~500 uniform values alive and spilled a thousand SGPRs
        if (lane < 32) {
#pragma unroll
            for (int k = 0; k < 32; ++k)
                if (k <= i) M[(c0 + i) * PB_LD + c0 + k] = a[k];
            if (!ok && i == 0) *bad = 1;
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the wave's own LDS writes have landed (one wave: no s_barrier needed)
        // W_pp = inv(L_pp): lane j solves L x = e_j (x_i for i >= j); four partial sums cut the dependent FMA chain
#pragma unroll
        for (int ii = 0; ii < 32; ++ii) {
            double acc[4] = {(ii == i) ? 1.0 : 0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int k = 0; k < ii; ++k) acc[k & 3] -= M[(c0 + ii) * PB_LD + c0 + k] * x[k];
            x[ii] = (ii >= i) ? ((acc[0] + acc[1]) + (acc[2] + acc[3])) * rinv[ii] : 0.0;
        }
        if (lane < 32) {
#pragma unroll
            for (int k = 0; k < 32; ++k) Wt[i * PB_WLD + k] = x[k];   // Wt[c][k] = W_pp[k][c] (0 for k < c): row i = column i of W_pp
        }
    }
    __syncthreads();
    if (*bad) return false;   // uniform
    // W_pp itself (lower, zeros above) replaces L_pp in the diagonal block: nothing below reads L_pp any more
    for (int e = tid; e < 32 * 32; e += 256) {
        const int k = e >> 5, c = e & 31;
        M[(c0 + k) * PB_LD + c0 + c] = Wt[c * PB_WLD + k];
    }
    if constexpr (NI > 0) {
        // (b) panel: L21[r][c] = sum_k A21[r][k] W_pp[c][k] = sum_k A21[r][k] Wt[k][c]   (fixed trip count: Wt carries the zeros)
        double out[NI][2];
#pragma unroll
        for (int ii = 0; ii < NI; ++ii) out[ii][0] = out[ii][1] = 0.0;
#pragma unroll 8
        for (int k = 0; k < 32; ++k) {
            const double w0 = Wt[k * PB_WLD + tx], w1 = Wt[k * PB_WLD + tx + 16];   // W_pp[c][k], zero for k > c
#pragma unroll
            for (int ii = 0; ii < NI; ++ii) {
                const double av = M[(r0 + ty + 16 * ii) * PB_LD + c0 + k];
                out[ii][0] += av * w0;
                out[ii][1] += av * w1;
            }
        }
        __syncthreads();
#pragma unroll
        for (int ii = 0; ii < NI; ++ii) {
            M[(r0 + ty + 16 * ii) * PB_LD + c0 + tx] = out[ii][0];
            M[(r0 + ty + 16 * ii) * PB_LD + c0 + tx + 16] = out[ii][1];
        }
        __syncthreads();
        // (c) update of the trailing lower triangle: A22[r][c] -= sum_k L21[r][k] L21[c][k]
        double acc[NI][NI];
#pragma unroll
        for (int ii = 0; ii < NI; ++ii)
#pragma unroll
            for (int jj = 0; jj < NI; ++jj) acc[ii][jj] = 0.0;
#pragma unroll 4
        for (int k = 0; k < 32; ++k) {
            double rv[NI], cv[NI];
#pragma unroll
            for (int ii = 0; ii < NI; ++ii) {
                rv[ii] = M[(r0 + ty + 16 * ii) * PB_LD + c0 + k];
                cv[ii] = M[(r0 + tx + 16 * ii) * PB_LD + c0 + k];
            }
#pragma unroll
            for (int ii = 0; ii < NI; ++ii)
#pragma unroll
                for (int jj = 0; jj <= ii; ++jj) acc[ii][jj] += rv[ii] * cv[jj];
        }
        // (nobody reads A22 during this phase and nobody else writes these elements: no barrier before the write-back)
#pragma unroll
        for (int ii = 0; ii < NI; ++ii)
#pragma unroll
            for (int jj = 0; jj <= ii; ++jj) {
                const int r = r0 + ty + 16 * ii, c = r0 + tx + 16 * jj;
                if (c <= r) M[r * PB_LD + c] -= acc[ii][jj];
            }
    }
    __syncthreads();
    return true;
}

// trtri, block column P: W21 = -W22 (L21 W_pp); W22 = the inverse of the trailing block (lower, zeros above: fixed trip counts)
template <int P>
__device__ __forceinline__ void pb_trtri_col(double* M, int ty, int tx)
{
    constexpr int c0 = 32 * P, r0 = c0 + 32, NI = (128 - r0) / 16;
    double out[NI][2];
#pragma unroll
    for (int ii = 0; ii < NI; ++ii) out[ii][0] = out[ii][1] = 0.0;
    // Y = W22 X : Y[r][c] = sum_{k >= r0} W[r][k] L[k][c0 + c]
#pragma unroll 4
    for (int k = r0; k < 128; ++k) {
        const double x0 = M[k * PB_LD + c0 + tx], x1 = M[k * PB_LD + c0 + tx + 16];
#pragma unroll
        for (int ii = 0; ii < NI; ++ii) {
            const double w = M[(r0 + ty + 16 * ii) * PB_LD + k];
            out[ii][0] += w * x0;
            out[ii][1] += w * x1;
        }
    }
    __syncthreads();
#pragma unroll
    for (int ii = 0; ii < NI; ++ii) {
        M[(r0 + ty + 16 * ii) * PB_LD + c0 + tx] = out[ii][0];
        M[(r0 + ty + 16 * ii) * PB_LD + c0 + tx + 16] = out[ii][1];
    }
    __syncthreads();
    // W21 = -Y W_pp : Z[r][c] = -sum_k Y[r][k] W_pp[k][c]   (W_pp lower: zeros for k < c)
#pragma unroll
    for (int ii = 0; ii < NI; ++ii) out[ii][0] = out[ii][1] = 0.0;
#pragma unroll 8
    for (int k = 0; k < 32; ++k) {
        const double w0 = M[(c0 + k) * PB_LD + c0 + tx], w1 = M[(c0 + k) * PB_LD + c0 + tx + 16];
#pragma unroll
        for (int ii = 0; ii < NI; ++ii) {
            const double y = M[(r0 + ty + 16 * ii) * PB_LD + c0 + k];
            out[ii][0] -= y * w0;
            out[ii][1] -= y * w1;
        }
    }
    __syncthreads();
#pragma unroll
    for (int ii = 0; ii < NI; ++ii) {
        M[(r0 + ty + 16 * ii) * PB_LD + c0 + tx] = out[ii][0];
        M[(r0 + ty + 16 * ii) * PB_LD + c0 + tx + 16] = out[ii][1];
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void k_potf2_inv_blk(const double* Akk, long ld, double* Wkk, double* WTkk, long ldw, int* status, unsigned long long* stamps)
{
    int n_stamp = 0;
#define PB_STAMP() do { if (stamps && threadIdx.x == 0) stamps[n_stamp++] = __builtin_amdgcn_s_memrealtime(); } while (0)
    PB_STAMP();
    extern __shared__ __attribute__((aligned(16))) double M[];   // [128][PB_LD]: lower = A -> L -> W, upper = 0; then Wt [32][PB_WLD]
    double* Wt = M + 128 * PB_LD;                                 // W_pp^T of the current panel as a FULL 32x32 matrix (zeros below its diagonal)
    __shared__ int bad;
    const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < 128 * 128; e += 256) {
        const int i = e >> 7, j = e & 127;
        M[i * PB_LD + j] = (j <= i) ? Akk[(size_t)i * ld + j] : 0.0;
    }
    if (tid == 0) bad = 0;
    __syncthreads();
    PB_STAMP();   // block loaded
    bool ok = pb_potrf_panel<0>(M, Wt, &bad, tid, ty, tx, lane, wave);
    PB_STAMP();
    ok = ok && pb_potrf_panel<1>(M, Wt, &bad, tid, ty, tx, lane, wave);
    PB_STAMP();
    ok = ok && pb_potrf_panel<2>(M, Wt, &bad, tid, ty, tx, lane, wave);
    PB_STAMP();
    ok = ok && pb_potrf_panel<3>(M, Wt, &bad, tid, ty, tx, lane, wave);
    if (!ok) {   // uniform
        if (tid == 0) *status = 1;
        return;
    }
    PB_STAMP();   // potrf done: the diagonal blocks hold W_pp, the blocks below them L
    pb_trtri_col<2>(M, ty, tx);
    pb_trtri_col<1>(M, ty, tx);
    pb_trtri_col<0>(M, ty, tx);
    PB_STAMP();   // trtri done
    for (int e = tid; e < 128 * 128; e += 256) {
        const int i = e >> 7, j = e & 127;
        Wkk[(size_t)i * ldw + j] = (j <= i) ? M[i * PB_LD + j] : 0.0;
        WTkk[(size_t)i * ldw + j] = (i <= j) ? M[j * PB_LD + i] : 0.0;
    }
    __syncthreads();
    PB_STAMP();   // stored
#undef PB_STAMP
}

static int g_potf2_blocked = 1;   // measurement knob (tests/native/test_potf2.cpp): 0 = the column-by-column leaf of round 1
void rtx_potf2_set_blocked(int on) { g_potf2_blocked = on; }
static unsigned long long* g_potf2_stamps = nullptr;   // device buffer of >= 16 entries: 100-MHz clock stamps of the phases (measurement)
void rtx_potf2_set_stamps(unsigned long long* dev) { g_potf2_stamps = dev; }

int rtx_potf2_inv_launch(const double* Akk, long ld, double* Wkk, double* WTkk, long ldw, int* status, hipStream_t stream)
{
    static bool configured = false;
    if (!configured) {
        RTX_HIP(hipFuncSetAttribute((const void*)k_potf2_inv, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 129 * 8));
        configured = true;
    }
    if (g_potf2_blocked) {
        static bool configured_blk = false;
        if (!configured_blk) {
            RTX_HIP(hipFuncSetAttribute((const void*)k_potf2_inv_blk, hipFuncAttributeMaxDynamicSharedMemorySize, (128 * PB_LD + 32 * PB_WLD) * 8));
            configured_blk = true;
        }
        hipLaunchKernelGGL(k_potf2_inv_blk, dim3(1), dim3(256), (128 * PB_LD + 32 * PB_WLD) * 8, stream, Akk, ld, Wkk, WTkk, ldw, status, g_potf2_stamps);
        RTX_HIP(hipGetLastError());
        return RTX_OK;
    }
    hipLaunchKernelGGL(k_potf2_inv, dim3(1), dim3(256), 128 * 129 * 8, stream, Akk, ld, Wkk, WTkk, ldw, status);
    RTX_HIP(hipGetLastError());
    return RTX_OK;
}
