// svae.hip -- Sequential VAE (BASELINE.json configs[4]) on MI355X: one user sequence per optimizer step (the reference's
// semantics), or a pack of users per step.
//
// Reference: rectorch/nets.py:624-693 (SVAE_net: Embedding -> 1-layer GRU (batch_first) -> VAE head that ALWAYS samples
// (VAE_net._reparameterize, nets.py:316-319) -> tanh MLP decoder), rectorch/models.py:1609-1635 (SVAE: loss =
// sum_t NLL_t / #target-ones + beta * mean_t KL_t; Adam with weight_decay 5e-3; predict = last time step, -inf at the
// items of the input sequence).
//
// The reference trains one user (T ~ 100 time steps) per Adam step, so every contraction is a [T, small] x [small, *]
// product: the step is bound by launch latency and by the strictly sequential GRU recurrence, not by MFMA throughput.
// Design for that regime (all float32, so parity with the reference is ~1e-6):
//   * one generic strided GEMM kernel (64x64 tile on the exact-float32 MFMA, bias / tanh / tanh'-mask epilogues) serves every
//     forward, backward-data and weight-gradient product -- operands are read in place with strides, no padded copies;
//   * the GRU recurrence (forward and BPTT) runs as ONE persistent workgroup of 1024 threads per direction: h_t lives
//     in LDS, W_hh streams from L2 (it is re-read every step), the input projections x_t W_ih^T for all t are one GEMM
//     before the loop, and the weight gradients are two GEMMs over all t after it;
//   * Adam over the 5 + 2(n_enc + n_dec) tensors is the one fused multi-tensor kernel of the Mult-VAE path (k_adam).
// rtx_svae_train_pack (round 2) takes several users per optimizer step: concatenated rows, one recurrence workgroup per user
// side by side, [sum T, .] products on the float32 MFMA, the recurrences with W_hh resident in registers + LDS -- 440 -> 806 users/s
// per user, 21 600 users/s with packs of 128 at the ml-1m shape (SVAE_Sampler(pack=N)).
// Round 3: both recurrences with the mat-vec split over K inside the wave (k_sv_gru_fwd_ks 1.43 us per time step, 3.6 in round 2;
// k_sv_gru_bwd_ks 1.63, was 2.06) -- 1 175-1 196 users/s per user, 26 900 with packs of 64.
#include "../../include/rectorch_hip.h"
#include "rtx_kernels.h"

#include <math.h>
#include <string.h>
#include <sched.h>
#include <algorithm>
#include <chrono>
#include <vector>

#define SV_MAX_LAYERS RTX_MAX_LAYERS
#define SV_GRU_KR 80   // weights of a half-row the forward recurrence keeps in registers
#define SV_GRU_KRB 88  // weights of a row chunk the backward recurrence keeps in registers

struct SvLayer {
    int in = 0, out = 0;
    bool tanh_act = false;
    float* A = nullptr;     // [Tmax][out] activation (post-tanh, or raw for the linear layers)
    float* D = nullptr;     // [Tmax][out] gradient w.r.t. the pre-activation
};

struct rtx_svae {
    rtx_svae_cfg cfg;
    int I = 0, E = 0, R = 0, Z = 0, NL = 0, n_enc = 0, Tmax = 0;
    std::vector<SvLayer> L;          // encoder layers then decoder layers
    int n_tensors = 0;
    std::vector<float*> params, grads, m, v;
    bool bound = false, can_train = false;
    // activations / scratch (device)
    float *X = nullptr, *GI = nullptr, *Hout = nullptr, *Hprev = nullptr, *Gr = nullptr, *Gz = nullptr, *Gn = nullptr, *Ghn = nullptr;
    float *mu = nullptr, *lv = nullptr, *eps = nullptr, *zl = nullptr, *dz = nullptr;
    float *dH = nullptr, *dGI = nullptr, *dGH = nullptr, *dX = nullptr;
    float *row_loss = nullptr, *kl_rows = nullptr;
    float* part = nullptr;           // split-K partial sums
    float* part2 = nullptr;          // ... of the GEMMs on the side stream
    hipStream_t side = nullptr;      // weight-gradient GEMMs of the MLPs: they overlap the single-workgroup GRU backward
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_fork2 = nullptr, ev_join2 = nullptr;
    float* WhhT = nullptr;           // [R][3R] transposed recurrent weights (refreshed per forward)
    size_t gru_fwd_lds = 0;          // > 0: the weight-resident forward recurrence runs, with this much dynamic LDS
    size_t gru_rows_lds = 0;         // > 0: ... its 512-thread whole-row form (round 3), preferred when it fits
    int opt_gemm_bf16 = 0;           // rtx_svae_set_option "gemm_bf16": bf16 operands (f32 accumulate) in every k_sv_gemm product
    size_t gru_bwd_ks_lds = 0;       // > 0: the backward recurrence in the same K-sliced layout
    size_t gru_ks_lds = 0;           // > 0: ... its K-sliced form (round 3), for 8 SL >= R and 64 NR >= 3R
    int opt_gru_rows = 1;            // measurement knob (RTX_SVAE_GRU_ROWS=0 in the environment at create time)
    int gru_kh = 0;                  //      K of the first half of a row
    size_t gru_bwd_lds = 0;          // > 0: the weight-resident backward recurrence runs
    int gru_nc = 0, gru_rp = 0;      //      its row chunks and rows per chunk
    size_t part_elems = 0;
    std::vector<void*> allocs;
    // rtx_svae_loss_mailbox: {loss bits, ticket} in coherent host memory, stored by k_sv_final_loss -- train_batch returns THIS step's
    // loss (reference models.py:835) as soon as the forward half of the step has run, without draining the stream behind it
    uint32_t* loss_mailbox = nullptr;
    uint32_t loss_ticket = 0;
};

// ------------------------------------------------------------------------------------------------ kernels
// C[m][n] (ldc) = epi( alpha * sum_k A(m,k) * B(n,k) (+ C if accumulate) ), A(m,k) = A[m*sam + k*sak], B(n,k) = B[n*sbn + k*sbk]
enum { SV_EPI_NONE = 0, SV_EPI_BIAS = 1, SV_EPI_BIAS_TANH = 2, SV_EPI_TANH_GRAD = 3 };
struct SvGemm {
    const float* A; long sam, sak;
    const float* B; long sbn, sbk;
    float* C; long ldc;
    int M, N, K;
    float alpha;
    int epi;
    const float* bias;   // [N]        (SV_EPI_BIAS*)
    const float* Q;      // [M][ldq]   (SV_EPI_TANH_GRAD: C = acc * (1 - Q^2))
    long ldq;
    int kchunk;          // split-K: workgroup z handles k in [z * kchunk, (z + 1) * kchunk) and stores its raw partial
    float* part;         //          sums to part[z][M][N]; k_sv_splitk_reduce then applies the epilogue (0 / NULL = off)
};

typedef __attribute__((ext_vector_type(16))) float sv_f32x16;
typedef __attribute__((ext_vector_type(4))) float sv_f32x4;
// LDS read with a compile-time byte offset (the GEMM kernels' idiom: an asm read is issued where it is written)
#define sv_lds_rd128(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF))

// 64 x 64 tile per workgroup, 4 waves, each a 32 x 32 block on the exact-float32 MFMA (v_mfma_f32_32x32x2_f32: lane l feeds
// A[l & 31][k = l >> 5] and B[k = l >> 5][l & 31], products and sums in float32).  Round 1 computed the tile with scalar FMAs
// (8 LDS floats per 16 FMAs per thread: LDS-bound); with several users packed per step these products are [sum T, .] GEMMs of
// tens of GFLOP and the matrix pipes carry them.  The operands are still read in place with two strides.
// BF = true (rtx_svae_set_option "gemm_bf16", the dtype BASELINE.json configs[4] names): the operands are rounded to bf16
// (round-to-nearest-even) on their way into LDS, stored row-major [m][16 k] so that a lane's eight k values are ONE 16-byte read,
// and a chunk is ONE v_mfma_f32_32x32x16_bf16 per wave instead of eight float32 MFMAs; sums, outputs, master weights stay float32.
typedef __attribute__((ext_vector_type(8))) __bf16 sv_bf16x8;
template <bool BF>
__global__ __launch_bounds__(256) void k_sv_gemm(const SvGemm g)
{
    // (K chunks of 32 instead of 16 -- twice the MFMA time per chunk over the next chunk's load latency -- measured slower per
    //  user, 1 164 vs 1 194 users/s: the per-user products are a handful of workgroups each, bound by their first loads and the launch)
    constexpr int KC = 16, NQ = KC / 4;
    __shared__ float sA[BF ? 1 : KC][65], sB[BF ? 1 : KC][65];
    __shared__ __attribute__((aligned(16))) bf16_t hA[BF ? 64 : 1][24], hB[BF ? 64 : 1][24];   // rows of 48 bytes: 16 k + padding
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    const int li = lane & 31, lk = lane >> 5;
    sv_f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const int kbeg = g.kchunk ? blockIdx.z * g.kchunk : 0;
    const int kend = g.kchunk ? min(g.K, kbeg + g.kchunk) : g.K;
    // 64 x KC elements per operand and K chunk, NQ per thread; the faster-varying thread index follows the unit-stride axis.
    // The next chunk's values are requested (into registers) before this chunk's MFMAs, so their latency runs under them.
    int amm[NQ], akk[NQ], bnn[NQ], bkk[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int e = tid + q * 256;
        if (g.sak == 1) { akk[q] = e % KC; amm[q] = e / KC; } else { amm[q] = e & 63; akk[q] = e >> 6; }
        if (g.sbk == 1) { bkk[q] = e % KC; bnn[q] = e / KC; } else { bnn[q] = e & 63; bkk[q] = e >> 6; }
    }
    float ra[NQ], rb[NQ];
    auto fetch = [&](int k0) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int m = m0 + amm[q], k = k0 + akk[q];
            ra[q] = (m < g.M && k < kend) ? g.A[(size_t)m * g.sam + (size_t)k * g.sak] : 0.f;
            const int n = n0 + bnn[q], kb = k0 + bkk[q];
            rb[q] = (n < g.N && kb < kend) ? g.B[(size_t)n * g.sbn + (size_t)kb * g.sbk] : 0.f;
        }
    };
    if (kbeg < kend) fetch(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += KC) {
        if constexpr (BF) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) { hA[amm[q]][akk[q]] = f32_to_bf16(ra[q]); hB[bnn[q]][bkk[q]] = f32_to_bf16(rb[q]); }
        } else {
#pragma unroll
            for (int q = 0; q < NQ; ++q) { sA[akk[q]][amm[q]] = ra[q]; sB[bkk[q]][bnn[q]] = rb[q]; }
        }
        __syncthreads();
        if (k0 + KC < kend) fetch(k0 + KC);
        if constexpr (BF) {
            // 32x32x16: lane l feeds row l & 31, k = 8 (l >> 5) .. + 7 of both operands
            const sv_bf16x8 a = *(const sv_bf16x8*)&hA[wm + li][lk * 8];
            const sv_bf16x8 b = *(const sv_bf16x8*)&hB[wn + li][lk * 8];
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        } else {
#pragma unroll
            for (int kk = 0; kk < KC; kk += 2)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sA[kk + lk][wm + li], sB[kk + lk][wn + li], acc, 0, 0, 0);
        }
        __syncthreads();
    }
    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    // What the epilogue reads is requested ONCE, up front, from clamped addresses (the bias of this lane's column; for the tanh'
    // mask all 16 values of Q): read per element under the row guard they were 16 dependent L2 round trips -- a load and a
    // vmcnt(0) per element -- about 3 us per launch of a kernel that takes 12.
    const int n = n0 + wn + li;
    if (n >= g.N) return;
    const int mb = m0 + wm + 4 * lk;
    const bool raw = g.kchunk != 0;
    float bv = 0.f;
    if (!raw && (g.epi == SV_EPI_BIAS || g.epi == SV_EPI_BIAS_TANH)) bv = g.bias[n];
    float qv[16];
    if (!raw && g.epi == SV_EPI_TANH_GRAD) {
#pragma unroll
        for (int e = 0; e < 16; ++e) qv[e] = g.Q[(size_t)min(mb + (e & 3) + 8 * (e >> 2), g.M - 1) * g.ldq + n];
    } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) qv[e] = 0.f;
    }
    float* out = raw ? g.part + (size_t)blockIdx.z * g.M * g.N : g.C;
    const size_t ldo = raw ? (size_t)g.N : (size_t)g.ldc;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int m = mb + (e & 3) + 8 * (e >> 2);
        float v = g.alpha * acc[e];
        if (!raw) {
            v += bv;
            if (g.epi == SV_EPI_BIAS_TANH) v = tanhf(v);
            v *= (1.f - qv[e] * qv[e]);
        }
        if (m < g.M) out[(size_t)m * ldo + n] = v;
    }
}

__global__ __launch_bounds__(256) void k_sv_splitk_reduce(const SvGemm g, int splits)
{
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)g.M * g.N) return;
    const int m = (int)(idx / g.N), n = (int)(idx % g.N);
    float v = 0.f;
    for (int z = 0; z < splits; ++z) v += g.part[(size_t)z * g.M * g.N + idx];
    if (g.epi == SV_EPI_BIAS || g.epi == SV_EPI_BIAS_TANH) v += g.bias[n];
    if (g.epi == SV_EPI_BIAS_TANH) v = tanhf(v);
    if (g.epi == SV_EPI_TANH_GRAD) { const float q = g.Q[(size_t)m * g.ldq + n]; v *= (1.f - q * q); }
    g.C[(size_t)m * g.ldc + n] = v;
}

// out[n] = sum_t D[t][n]  (bias gradients): a workgroup sums 32 time steps of 256 columns; the row blocks meet through
// atomicAdd on the zeroed output
__global__ __launch_bounds__(256) void k_sv_colsum(const float* D, long ld, int T, int N, float* out)
{
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const int t0 = blockIdx.y * 32, t1 = min(T, t0 + 32);
    float s = 0.f;
    for (int t = t0; t < t1; ++t) s += D[(size_t)t * ld + n];
    atomicAdd(out + n, s);
}

// the same in ONE pass for short inputs (one user: T ~ 150): a workgroup owns 64 columns, its four waves take every fourth row and
// meet in LDS -- no zero fill of the output, no atomics (round 3: the per-user step issued six 5-us memsets for these sums)
__global__ __launch_bounds__(256) void k_sv_colsum1(const float* D, long ld, int T, int N, float* out)
{
    __shared__ float part[4][64];
    const int c = threadIdx.x & 63, rg = threadIdx.x >> 6, n = blockIdx.x * 64 + c;
    float s0 = 0.f, s1 = 0.f;
    if (n < N) {
        int t = rg;
        for (; t + 4 < T; t += 8) { s0 += D[(size_t)t * ld + n]; s1 += D[(size_t)(t + 4) * ld + n]; }
        if (t < T) s0 += D[(size_t)t * ld + n];
    }
    part[rg][c] = s0 + s1;
    __syncthreads();
    if (rg == 0 && n < N) out[n] = (part[0][c] + part[1][c]) + (part[2][c] + part[3][c]);
}

__global__ __launch_bounds__(256) void k_sv_embed(const int32_t* items, int T, int E, const float* emb, float* X)
{
    const int t = blockIdx.x;
    const float* src = emb + (size_t)items[t] * E;
    for (int e = threadIdx.x; e < E; e += 256) X[(size_t)t * E + e] = src[e];
}

__global__ __launch_bounds__(256) void k_sv_embed_grad(const int32_t* items, int T, int E, const float* dX, float* demb)
{
    const int t = blockIdx.x;
    float* dst = demb + (size_t)items[t] * E;
    for (int e = threadIdx.x; e < E; e += 256) atomicAdd(dst + e, dX[(size_t)t * E + e]);   // an item may repeat in a sequence
}

// out[c][r] = in[r][c]  (W_hh -> W_hh^T once per sequence for the forward recurrence)
__global__ __launch_bounds__(256) void k_sv_transpose(const float* __restrict__ in, int rows, int cols, float* __restrict__ out)
{
    __shared__ float tile[64][65];
    const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64, tid = threadIdx.x;
    for (int e = tid; e < 4096; e += 256) {
        const int rr = e >> 6, cc = e & 63;
        tile[rr][cc] = (r0 + rr < rows && c0 + cc < cols) ? in[(size_t)(r0 + rr) * cols + c0 + cc] : 0.f;
    }
    __syncthreads();
    for (int e = tid; e < 4096; e += 256) {
        const int cc = e >> 6, rr = e & 63;
        if (r0 + rr < rows && c0 + cc < cols) out[(size_t)(c0 + cc) * rows + r0 + rr] = tile[rr][cc];
    }
}

__device__ __forceinline__ float sv_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }
// the recurrence's gate phase is one dependent chain per thread (200 of the 512 threads, nothing to hide it behind): libm's expf /
// tanhf / IEEE division make it ~1 000 cycles per time step.  v_exp_f32 / v_rcp_f32 (1 ulp each; the argument's scaling by log2 e
// adds |x| * 6e-8 relative) keep sigmoid within 3e-7 and tanh within 6e-7 absolute of libm's.
__device__ __forceinline__ float sv_sigmoid_fast(float x) { return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }
__device__ __forceinline__ float sv_tanh_fast(float x) { return 2.f * sv_sigmoid_fast(2.f * x) - 1.f; }
template <int CTRL> __device__ __forceinline__ float sv_dpp(float v)   // v of the lane DPP control CTRL points at (all rows, all banks)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}

// block-wide reductions for 256-thread blocks; red must hold >= 4 floats; result broadcast to all threads
__device__ __forceinline__ float block_sum(float v, float* red)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ float block_max(float v, float* red)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// GRU forward, torch.nn.GRU gate order r | z | n (weight_hh_l0 [3R][R], bias_hh_l0 [3R]); GI = x W_ih^T + b_ih [T][3R].
//   r = sig(gi_r + W_hr h + b_hr), z = sig(gi_z + W_hz h + b_hz), n = tanh(gi_n + r * (W_hn h + b_hn)), h' = (1-z) n + z h
// One persistent workgroup of 1024 threads; the recurrence is a chain of T dependent mat-vecs, so what matters is the
// latency of ONE step.  Thread `row` streams its own row of W_hh (all loads independent: the only dependent chain is the
// FMA accumulation) against h broadcast from LDS -- no cross-lane reduction, one barrier per phase.  The weights are read
// from a transposed copy made once per sequence, so that a wave's load is 256 contiguous bytes.
// Measured per step at R = 200: 27 us with one wave per row + shuffle reduction (37 dependent L2 round trips), 8.7 us in
// this form.  Splitting the hidden units over 2 workgroups with W_hh resident in registers and an exchange of h through
// L2 per step was SLOWER (10.8 us): device-scope release/acquire between compute units costs microseconds on a
// multi-XCD part, more than re-reading 480 KB from L2.
__global__ __launch_bounds__(1024) void k_sv_gru_fwd(const float* __restrict__ GI, const float* __restrict__ WhhT, const float* __restrict__ bhh,
                                                     const int32_t* __restrict__ seq_ptr, int T_one, int R,
                                                     float* __restrict__ Hout /* [T][R]: h after step t */,
                                                     float* __restrict__ Hprev /* [T][R]: h before step t (0 at a sequence start) */,
                                                     float* __restrict__ Gr, float* __restrict__ Gz, float* __restrict__ Gn, float* __restrict__ Ghn)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];   // h [Rp] | gh [3R]   (Rp = R rounded up to 4)
    const int Rp = (R + 3) & ~3;
    float* h = sm;
    float* gh = sm + Rp;
    const int tid = threadIdx.x;
    // packed sequences: workgroup b owns rows [seq_ptr[b], seq_ptr[b + 1]) of every [T][.] buffer; one sequence: all T_one rows
    const int t0 = seq_ptr ? seq_ptr[blockIdx.x] : 0;
    const int T = seq_ptr ? seq_ptr[blockIdx.x + 1] - t0 : T_one;
    GI += (size_t)t0 * 3 * R;
    Hout += (size_t)t0 * R; Hprev += (size_t)t0 * R;
    Gr += (size_t)t0 * R; Gz += (size_t)t0 * R; Gn += (size_t)t0 * R; Ghn += (size_t)t0 * R;
    for (int j = tid; j < Rp; j += 1024) h[j] = 0.f;
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        // this step's input projections are independent of h: fetch them before the mat-vec, not after its barrier
        const float* gi = GI + (size_t)t * 3 * R;
        float gir = 0.f, giz = 0.f, gin = 0.f;
        if (tid < R) { gir = gi[tid]; giz = gi[R + tid]; gin = gi[2 * R + tid]; }
        for (int row = tid; row < 3 * R; row += 1024) {
            // WhhT is [R][3R]: consecutive lanes (rows) read consecutive addresses -- 4 cache lines per wave load instead
            // of the 64 a row-major W_hh costs when every lane walks its own row
            const float* w = WhhT + row;
            const size_t ld = (size_t)3 * R;
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
            int k = 0;
            for (; k + 8 <= R; k += 8) {
                const float w0 = w[(k + 0) * ld], w1 = w[(k + 1) * ld], w2 = w[(k + 2) * ld], w3 = w[(k + 3) * ld];
                const float w4 = w[(k + 4) * ld], w5 = w[(k + 5) * ld], w6 = w[(k + 6) * ld], w7 = w[(k + 7) * ld];
                s0 += w0 * h[k] + w4 * h[k + 4];
                s1 += w1 * h[k + 1] + w5 * h[k + 5];
                s2 += w2 * h[k + 2] + w6 * h[k + 6];
                s3 += w3 * h[k + 3] + w7 * h[k + 7];
            }
            for (; k < R; ++k) s0 += w[k * ld] * h[k];
            gh[row] = (s0 + s1) + (s2 + s3) + bhh[row];
        }
        __syncthreads();
        for (int j = tid; j < R; j += 1024) {   // R <= 1024: one pass, j == tid
            const float r = sv_sigmoid(gir + gh[j]);
            const float z = sv_sigmoid(giz + gh[R + j]);
            const float hn = gh[2 * R + j];
            const float n = tanhf(gin + r * hn);
            const float hp = h[j];
            const float hv = (1.f - z) * n + z * hp;
            Gr[(size_t)t * R + j] = r; Gz[(size_t)t * R + j] = z; Gn[(size_t)t * R + j] = n; Ghn[(size_t)t * R + j] = hn;
            Hprev[(size_t)t * R + j] = hp;
            Hout[(size_t)t * R + j] = hv;
            h[j] = hv;   // element j is read and written by this thread only; the mat-vec above is behind the barrier
        }
        __syncthreads();
    }
}

// The same recurrence with ALL of W_hh resident in the compute unit.  The streaming kernel above re-reads W_hh (480 KB at
// R = 200) through one CU's 64-B/clk L1 path on every time step: 3.3 us of its 5.2 us.  Here the 3R rows are cut in two halves
// of K (6R half-rows; thread t owns half-row t): the first KR = 80 weights of a half-row live in the thread's REGISTERS for the
// whole sequence, the rest of it in LDS, and the half-rows beyond the 1024th live in LDS entirely (their owners are threads
// 0 .. 6R - 1025, which sum a second half-row per step) -- 1024 x 80 registers + 150 KB of LDS hold the 120 000 weights of
// R = 200 exactly.  A time step reads nothing from outside the CU but its own input projection, requested one step ahead.  The two
// halves of a row meet in LDS behind the barrier the gates need anyway.  LDS images are [chunk of 4 k][owner][4]: consecutive
// lanes read consecutive 16 bytes.  The barriers are raw s_barriers behind an LDS-only wait.
// Measured 3.6 us per step (5.2 streaming).  What holds it there is the register file: 80 weights + the step's working set do not
// fit the 128 VGPRs of a 1024-thread workgroup, hipcc keeps ~17 weights in scratch and reloads them one by one every step.  Tried:
// whole rows in registers (round 1: spills in the 200-FMA loop); 48 registers + 64 LDS + 88 streamed weights per row (4.0 us: 230
// scalar LDS reads per wave); the thread-less half-rows spread over all threads as fragments (4.6 us: more spills); a (unit,
// K-chunk) split with 88 / 64 registers per thread like the backward kernel's (4.2 us / 288 B of spills: the gate arithmetic's
// working set comes on top).  The backward kernel (no transcendental gate phase) fits and runs at 2.2 us.
__host__ __device__ inline int sv_gru_fwd_hs(int R, int Kh, int KR)
{
    const int CA = Kh > KR ? (Kh - KR + 3) / 4 : 0, CB = (Kh + 3) / 4;
    const int reach = Kh + (KR + 4 * CA > 4 * CB ? KR + 4 * CA : 4 * CB);
    return ((reach > R ? reach : R) + 7) & ~3;
}

template <int KR>
__global__ __launch_bounds__(1024) void k_sv_gru_fwd_all(const float* __restrict__ GI, const float* __restrict__ Whh, const float* __restrict__ bhh,
                                                         const int32_t* __restrict__ seq_ptr, int T_one, int R, int Kh /* K of the first half, % 4 == 0 */,
                                                         float* __restrict__ Hout, float* __restrict__ Hprev, float* __restrict__ Gr,
                                                         float* __restrict__ Gz, float* __restrict__ Gn, float* __restrict__ Ghn)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];   // h [HS] | gp [6R] | wlA [CA][1024][4] | wlB [CB][NE][4]
    const int R3 = 3 * R, HR = 2 * R3;
    const int NE = HR > 1024 ? HR - 1024 : 0;                     // half-rows without a thread of their own
    const int KLA = Kh > KR ? Kh - KR : 0, CA = (KLA + 3) / 4, CB = (Kh + 3) / 4;
    // h is read up to KR + 4 CA (or 4 CB) floats past a half's start whatever R is (the weights there are zero, but zero times
    // an uninitialised LDS word may be NaN): the vector's area covers every such read and is zeroed once
    const int HS = sv_gru_fwd_hs(R, Kh, KR);
    float* h = sm;
    float* gp = sm + HS;
    float* wlA = gp + ((HR + 3) & ~3);
    float* wlB = wlA + (size_t)CA * 1024 * 4;
    const int tid = threadIdx.x;
    const int t0 = seq_ptr ? seq_ptr[blockIdx.x] : 0;
    const int T = seq_ptr ? seq_ptr[blockIdx.x + 1] - t0 : T_one;
    GI += (size_t)t0 * R3;
    Hout += (size_t)t0 * R; Hprev += (size_t)t0 * R;
    Gr += (size_t)t0 * R; Gz += (size_t)t0 * R; Gn += (size_t)t0 * R; Ghn += (size_t)t0 * R;
    // half-row hr = half * 3R + row covers k in [half * Kh, half ? R : Kh)
    const bool own = tid < HR;
    const int half = own ? tid / R3 : 0, row = own ? tid % R3 : 0;
    const int k0 = half * Kh, len = own ? (half ? R - Kh : Kh) : 0;
    const int last = R3 * R - 1;
    float wr[KR];
    {
        // unconditional loads (index clamped to the matrix), then the tail beyond the half-row is zeroed
        const int base = row * R + k0;
#pragma unroll
        for (int q = 0; q < KR; ++q) {
            const float v = Whh[min(base + q, last)];
            wr[q] = q < len ? v : 0.f;
        }
        for (int q = 0; q < CA * 4; ++q) {
            const float v = Whh[min(base + KR + q, last)];
            wlA[((size_t)(q >> 2) * 1024 + tid) * 4 + (q & 3)] = (KR + q < len) ? v : 0.f;
        }
    }
    const bool extra = tid < NE;
    const int xrow = extra ? (1024 + tid) % R3 : 0;               // half-row 1024 + tid: always a second half (3R <= 1024)
    const int xlen = extra ? R - Kh : 0;
    if (extra)
        for (int q = 0; q < CB * 4; ++q) {
            const float v = Whh[min(xrow * R + Kh + q, last)];
            wlB[((size_t)(q >> 2) * NE + tid) * 4 + (q & 3)] = q < xlen ? v : 0.f;
        }
    for (int j = tid; j < HS; j += 1024) h[j] = 0.f;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const float4* hv0 = (const float4*)(h + k0);
    const float4* hvx = (const float4*)(h + Kh);
    float gir = 0.f, giz = 0.f, gin = 0.f;
    if (tid < R && T > 0) { gir = GI[tid]; giz = GI[R + tid]; gin = GI[2 * R + tid]; }
    for (int t = 0; t < T; ++t) {
        float nir = 0.f, niz = 0.f, nin = 0.f;
        if (tid < R && t + 1 < T) {
            const float* gi = GI + (size_t)(t + 1) * R3;
            nir = gi[tid]; niz = gi[R + tid]; nin = gi[2 * R + tid];
        }
        {
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
            for (int q = 0; q < KR; q += 4) {
                const float4 x = hv0[q >> 2];
                s0 += wr[q] * x.x; s1 += wr[q + 1] * x.y; s2 += wr[q + 2] * x.z; s3 += wr[q + 3] * x.w;
            }
#pragma nounroll
            for (int c = 0; c < CA; ++c) {
                const float4 w = *(const float4*)(wlA + ((size_t)c * 1024 + tid) * 4);
                const float4 x = hv0[(KR >> 2) + c];
                s0 += w.x * x.x; s1 += w.y * x.y; s2 += w.z * x.z; s3 += w.w * x.w;
            }
            if (own) gp[tid] = (s0 + s1) + (s2 + s3);
            if (extra) {
                float e0 = 0.f, e1 = 0.f, e2 = 0.f, e3 = 0.f;
#pragma nounroll
                for (int c = 0; c < CB; ++c) {
                    const float4 w = *(const float4*)(wlB + ((size_t)c * NE + tid) * 4);
                    const float4 x = hvx[c];
                    e0 += w.x * x.x; e1 += w.y * x.y; e2 += w.z * x.z; e3 += w.w * x.w;
                }
                gp[1024 + tid] = (e0 + e1) + (e2 + e3);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (tid < R) {
            const int j = tid;
            const float ghr = gp[j] + gp[R3 + j] + bhh[j];
            const float ghz = gp[R + j] + gp[R3 + R + j] + bhh[R + j];
            const float hn = gp[2 * R + j] + gp[R3 + 2 * R + j] + bhh[2 * R + j];
            const float r = sv_sigmoid(gir + ghr);
            const float z = sv_sigmoid(giz + ghz);
            const float n = tanhf(gin + r * hn);
            const float hp = h[j];
            const float hv = (1.f - z) * n + z * hp;
            Gr[(size_t)t * R + j] = r; Gz[(size_t)t * R + j] = z; Gn[(size_t)t * R + j] = n; Ghn[(size_t)t * R + j] = hn;
            Hprev[(size_t)t * R + j] = hp;
            Hout[(size_t)t * R + j] = hv;
            h[j] = hv;
        }
        gir = nir; giz = niz; gin = nin;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
}

// Round 3: the same weight-resident recurrence on 512 threads (8 waves, two per SIMD: 256 registers per lane).  The 1024-thread
// kernel above keeps 80 weights per thread and has 128 registers to do it in: hipcc spills ~17 of them and reloads them every
// step (3.6 us per step).  Here a thread owns a WHOLE row of W_hh: its first KR = 160 weights in registers, the rest of the row
// in LDS ([chunk of 4 k][512][4]); rows beyond the 512th live in LDS entirely and are summed as two half-rows each by the first
// 2 NE threads (R = 200: 88 rows, 176 threads; as four quarter-rows on 352 threads the first waves' mat-vec is shorter but the
// others' longer -- the LDS pipe is shared -- and the step is 5 % slower).  512 x 160 registers + 82 KB + 70 KB of LDS hold the 120 000 weights of R = 200;
// the hidden state is read as broadcast float4s.  No spills (207 VGPRs), one mat-vec phase and one gate phase per step as before.
// Measured (tools/svae_stamps.py, shader clock): 6 300 cycles = 2.6 us per step -- mat-vec 4 700 (the LDS pipe: every thread
// reads all of h, 50 float4, plus its LDS-resident weights), barrier 180, gate phase 1 040, barrier + loop 340 -- against 3.6 us
// for the 1024-thread kernel: 806 -> 893 users/s with one user per optimizer step.  The first version of this kernel carried an
// `if (q < nq)` inside the unrolled loop and took 10 200 cycles: a guard per float4 makes every read its own basic block.
#define SV_GRU_KR2 160
__host__ __device__ inline void sv_gru_rows_shape(int R, int KR, int* NE, int* Kh, int* CA, int* CB, int* HS)
{
    const int R3 = 3 * R;
    *NE = R3 > 512 ? R3 - 512 : 0;
    *Kh = (((R + 1) / 2) + 3) & ~3;                  // extra rows: first half k < Kh, second half the rest
    *CA = R > KR ? (R - KR + 3) / 4 : 0;
    *CB = (*Kh + 3) / 4;
    const int reach_own = KR + 4 * *CA, reach_x = *Kh + 4 * *CB;   // the register-resident prefix is summed unconditionally (zero weights past R)
    const int reach = reach_own > reach_x ? reach_own : reach_x;
    *HS = ((reach > R ? reach : R) + 7) & ~3;
}
__host__ __device__ inline size_t sv_gru_rows_lds(int R, int KR)
{
    int NE, Kh, CA, CB, HS;
    sv_gru_rows_shape(R, KR, &NE, &Kh, &CA, &CB, &HS);
    return sizeof(float) * ((size_t)HS + ((3 * R + 3) & ~3) + ((2 * NE + 3) & ~3) + (size_t)CA * 512 * 4 + (size_t)CB * 2 * NE * 4);
}

__device__ unsigned long long* g_sv_stamps = nullptr;   // measurement: shader-clock stamps of the first steps of the forward recurrence
template <int KR>
__global__ __launch_bounds__(512) void k_sv_gru_fwd_rows(const float* __restrict__ GI, const float* __restrict__ Whh, const float* __restrict__ bhh,
                                                         const int32_t* __restrict__ seq_ptr, int T_one, int R, float* __restrict__ Hout,
                                                         float* __restrict__ Hprev, float* __restrict__ Gr, float* __restrict__ Gz,
                                                         float* __restrict__ Gn, float* __restrict__ Ghn)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];   // h [HS] | gp [3R] | gx [2 NE] | wlA [CA][512][4] | wlB [CB][2 NE][4]
    int NE, Kh, CA, CB, HS;
    sv_gru_rows_shape(R, KR, &NE, &Kh, &CA, &CB, &HS);
    const int R3 = 3 * R, NX = 2 * NE;
    float* h = sm;
    float* gp = sm + HS;
    float* gx = gp + ((R3 + 3) & ~3);
    float* wlA = gx + ((NX + 3) & ~3);
    float* wlB = wlA + (size_t)CA * 512 * 4;
    const int tid = threadIdx.x;
    const int t0 = seq_ptr ? seq_ptr[blockIdx.x] : 0;
    const int T = seq_ptr ? seq_ptr[blockIdx.x + 1] - t0 : T_one;
    GI += (size_t)t0 * R3;
    Hout += (size_t)t0 * R; Hprev += (size_t)t0 * R;
    Gr += (size_t)t0 * R; Gz += (size_t)t0 * R; Gn += (size_t)t0 * R; Ghn += (size_t)t0 * R;
    const bool own = tid < R3;                     // (R3 < 512: the upper threads idle through the mat-vec)
    const int row = own ? tid : 0;
    const int last = R3 * R - 1;
    float wr[KR];
    {
        const int base = row * R;
#pragma unroll
        for (int q = 0; q < KR; ++q) {
            const float v = Whh[min(base + q, last)];
            wr[q] = (own && q < R) ? v : 0.f;
        }
        for (int q = 0; q < CA * 4; ++q) {
            const float v = Whh[min(base + KR + q, last)];
            wlA[((size_t)(q >> 2) * 512 + tid) * 4 + (q & 3)] = (own && KR + q < R) ? v : 0.f;
        }
    }
    const bool extra = tid < NX;
    const int xrow = extra ? 512 + (tid >> 1) : 0, xk0 = (tid & 1) ? Kh : 0, xlen = extra ? ((tid & 1) ? R - Kh : Kh) : 0;
    if (extra)
        for (int q = 0; q < CB * 4; ++q) {
            const float v = Whh[min(xrow * R + xk0 + q, last)];
            wlB[((size_t)(q >> 2) * NX + tid) * 4 + (q & 3)] = q < xlen ? v : 0.f;
        }
    for (int j = tid; j < HS; j += 512) h[j] = 0.f;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const float4* hv = (const float4*)h;
    const float4* hvx = (const float4*)(h + xk0);
    constexpr int NQ = KR / 4;
    float gir = 0.f, giz = 0.f, gin = 0.f;
    if (tid < R && T > 0) { gir = GI[tid]; giz = GI[R + tid]; gin = GI[2 * R + tid]; }
    unsigned long long* stamps = (tid == 0 && blockIdx.x == 0) ? g_sv_stamps : nullptr;
    for (int t = 0; t < T; ++t) {
        if (stamps && t >= 8 && t < 12) stamps[(t - 8) * 4 + 0] = __builtin_readcyclecounter();
        float nir = 0.f, niz = 0.f, nin = 0.f;
        if (tid < R && t + 1 < T) {
            const float* gi = GI + (size_t)(t + 1) * R3;
            nir = gi[tid]; niz = gi[R + tid]; nin = gi[2 * R + tid];
        }
        {
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
            // no condition inside the unrolled loop: a (uniform) guard per float4 turned every read into its own basic block with its
            // own s_waitcnt -- 8 200 cycles for this phase instead of ~2 000 (tools/svae_stamps.py)
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const float4 x = hv[q];
                s0 += wr[4 * q] * x.x; s1 += wr[4 * q + 1] * x.y; s2 += wr[4 * q + 2] * x.z; s3 += wr[4 * q + 3] * x.w;
            }
#pragma unroll 5
            for (int c = 0; c < CA; ++c) {
                const float4 w = *(const float4*)(wlA + ((size_t)c * 512 + tid) * 4);
                const float4 x = hv[NQ + c];
                s0 += w.x * x.x; s1 += w.y * x.y; s2 += w.z * x.z; s3 += w.w * x.w;
            }
            if (own) gp[tid] = (s0 + s1) + (s2 + s3);
            if (extra) {
                float e0 = 0.f, e1 = 0.f, e2 = 0.f, e3 = 0.f;
#pragma unroll 5
                for (int c = 0; c < CB; ++c) {
                    const float4 w = *(const float4*)(wlB + ((size_t)c * NX + tid) * 4);
                    const float4 x = hvx[c];
                    e0 += w.x * x.x; e1 += w.y * x.y; e2 += w.z * x.z; e3 += w.w * x.w;
                }
                gx[tid] = (e0 + e1) + (e2 + e3);
            }
        }
        if (stamps && t >= 8 && t < 12) stamps[(t - 8) * 4 + 1] = __builtin_readcyclecounter();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (stamps && t >= 8 && t < 12) stamps[(t - 8) * 4 + 2] = __builtin_readcyclecounter();
        if (tid < R) {
            const int j = tid;
            auto G = [&](int i) { return (i < 512 ? gp[i] : gx[2 * (i - 512)] + gx[2 * (i - 512) + 1]) + bhh[i]; };
            const float ghr = G(j), ghz = G(R + j), hn = G(2 * R + j);
            const float r = sv_sigmoid(gir + ghr);
            const float z = sv_sigmoid(giz + ghz);
            const float n = tanhf(gin + r * hn);
            const float hp = h[j];
            const float hvv = (1.f - z) * n + z * hp;
            Gr[(size_t)t * R + j] = r; Gz[(size_t)t * R + j] = z; Gn[(size_t)t * R + j] = n; Ghn[(size_t)t * R + j] = hn;
            Hprev[(size_t)t * R + j] = hp;
            Hout[(size_t)t * R + j] = hvv;
            h[j] = hvv;
        }
        if (stamps && t >= 8 && t < 12) stamps[(t - 8) * 4 + 3] = __builtin_readcyclecounter();
        gir = nir; giz = niz; gin = nin;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
}

// Round 3, second form: the mat-vec split over K inside the wave.  The whole-row kernel above is bound by the LDS pipe: every
// thread reads ALL of h (50 broadcast float4) plus its LDS-resident weights, ~520 LDS wave-instructions per step = the 4 700
// cycles of its mat-vec phase.  Here lane = (row group g = lane >> 3, K slice s = lane & 7): a thread owns NR rows (g + 64 i of
// the wave's... 64 row slots x NR covers 3R) times ONE slice of SL = R / 8 columns, so it reads only its slice of h (7 float4,
// all requested up front) and keeps the same ~250 weights (KG in registers, the rest in LDS as before).  The 8 slices of a row
// meet through three DPP adds per accumulator (quad_perm xor 1, xor 2, row_half_mirror: no LDS), then the 8 lanes of a group
// write one or two of the NR sums each.  ~180 LDS wave-instructions per step instead of ~520; the VALU work grows from 200 to
// ~280 operations per thread (the DPP reduction).  h lives slice-major ([8][SLP]) so a slice is 16-byte aligned.
template <int SL, int NR, int KG>
__global__ __launch_bounds__(512) void k_sv_gru_fwd_ks(const float* __restrict__ GI, const float* __restrict__ Whh, const float* __restrict__ bhh,
                                                       const int32_t* __restrict__ seq_ptr, int T_one, int R, float* __restrict__ Hout,
                                                       float* __restrict__ Hprev, float* __restrict__ Gr, float* __restrict__ Gz,
                                                       float* __restrict__ Gn, float* __restrict__ Ghn)
{
    constexpr int SLP = (SL + 3) & ~3, NQ = SLP / 4, NWT = NR * SL, CL = (NWT - KG + 3) / 4, GPN = 64 * NR;
    static_assert(KG % 4 == 0 && KG <= NWT, "register-resident weights: whole float4 groups");
    extern __shared__ __attribute__((aligned(16))) float sm[];   // h2 [8][SLP] | gp [64 NR] | wl [CL][512][4]
    float* h2 = sm;
    float* gp = sm + 8 * SLP;
    float* wl = gp + GPN;
    const int R3 = 3 * R;
    const int tid = threadIdx.x, lane = tid & 63, s = lane & 7, slot = (tid >> 6) * 8 + (lane >> 3);
    const int t0 = seq_ptr ? seq_ptr[blockIdx.x] : 0;
    const int T = seq_ptr ? seq_ptr[blockIdx.x + 1] - t0 : T_one;
    GI += (size_t)t0 * R3;
    Hout += (size_t)t0 * R; Hprev += (size_t)t0 * R;
    Gr += (size_t)t0 * R; Gz += (size_t)t0 * R; Gn += (size_t)t0 * R; Ghn += (size_t)t0 * R;
    // weight q = i * SL + kk of this thread is W_hh[slot + 64 i][s * SL + kk] (zero outside the matrix)
    const int last = R3 * R - 1;
    auto wload = [&](int q) {
        const int i = q / SL, kk = q % SL, row = slot + 64 * i, k = s * SL + kk;
        // every load unconditional (clamped address), the mask a factor: a select lets hipcc sink the load into a branch of its own,
        // and 250 such branches serialise the kernel's start
        const float v = Whh[min(row * R + k, last)];
        return v * ((row < R3 && k < R && q < NWT) ? 1.f : 0.f);
    };
    float wr[KG];
#pragma unroll
    for (int q = 0; q < KG; ++q) wr[q] = wload(q);
#pragma unroll 8
    for (int q = 0; q < CL * 4; ++q) wl[((size_t)(q >> 2) * 512 + tid) * 4 + (q & 3)] = wload(KG + q);
    for (int j = tid; j < 8 * SLP; j += 512) h2[j] = 0.f;
    for (int j = tid; j < GPN; j += 512) gp[j] = 0.f;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const float4* hs = (const float4*)(h2 + s * SLP);
    const int hj = (tid / SL) * SLP + (tid % SL);      // where h[tid] lives (gate-phase threads: tid < R)
    float gir = 0.f, giz = 0.f, gin = 0.f;
    if (tid < R && T > 0) { gir = GI[tid]; giz = GI[R + tid]; gin = GI[2 * R + tid]; }
    // the hidden-side biases of this thread's three gates, once (read inside the loop they are three global loads and a vmcnt(0)
    // -- an L2 round trip -- in every step's gate phase)
    const int bj = min(tid, R - 1);
    const float bh_r = bhh[bj], bh_z = bhh[R + bj], bh_n = bhh[2 * R + bj];
    unsigned long long* stamps = (tid == 0 && blockIdx.x == 0) ? g_sv_stamps : nullptr;
    for (int t = 0; t < T; ++t) {
        if (stamps && t >= 8 && t < 12) stamps[(t - 8) * 4 + 0] = __builtin_readcyclecounter();
        float nir = 0.f, niz = 0.f, nin = 0.f;
        if (tid < R && t + 1 < T) {
            const unsigned o = (unsigned)(t + 1) * (unsigned)R3 + (unsigned)tid;
            nir = GI[o]; niz = GI[o + (unsigned)R]; nin = GI[o + 2u * (unsigned)R];
        }
        {
            float4 hq[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) hq[q] = hs[q];
            float acc[NR];
#pragma unroll
            for (int i = 0; i < NR; ++i) acc[i] = 0.f;
            auto hval = [&](int kk) { const float4 v = hq[kk >> 2]; return (kk & 3) == 0 ? v.x : (kk & 3) == 1 ? v.y : (kk & 3) == 2 ? v.z : v.w; };
#pragma unroll
            for (int q = 0; q < KG; ++q) acc[q / SL] += wr[q] * hval(q % SL);
#pragma unroll
            for (int c = 0; c < CL; ++c) {
                const float4 w = *(const float4*)(wl + ((size_t)c * 512 + tid) * 4);
                const float we[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int q = KG + 4 * c + e;
                    if (q < NWT) acc[q / SL] += we[e] * hval(q % SL);
                }
            }
            // the 8 K slices of a row sit in 8 neighbouring lanes
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                acc[i] += sv_dpp<0xB1>(acc[i]);    // quad_perm [1,0,3,2]
                acc[i] += sv_dpp<0x4E>(acc[i]);    // quad_perm [2,3,0,1]
                acc[i] += sv_dpp<0x141>(acc[i]);   // row_half_mirror: the other quad of the 8
            }
            // lane s of the group writes sums s and s + 8
            float v0 = acc[0], v1 = acc[NR > 8 ? 8 : 0];
#pragma unroll
            for (int i = 1; i < 8 && i < NR; ++i) v0 = (s == i) ? acc[i] : v0;
#pragma unroll
            for (int i = 9; i < NR; ++i) v1 = (s == i - 8) ? acc[i] : v1;
            if (s < NR) gp[slot + 64 * s] = v0;
            if (s + 8 < NR) gp[slot + 64 * (s + 8)] = v1;
        }
        if (stamps && t >= 8 && t < 12) stamps[(t - 8) * 4 + 1] = __builtin_readcyclecounter();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (stamps && t >= 8 && t < 12) stamps[(t - 8) * 4 + 2] = __builtin_readcyclecounter();
        if (tid < R) {
            const int j = tid;
            const float ghr = gp[j] + bh_r, ghz = gp[R + j] + bh_z, hn = gp[2 * R + j] + bh_n;
            const float r = sv_sigmoid_fast(gir + ghr);
            const float z = sv_sigmoid_fast(giz + ghz);
            const float n = sv_tanh_fast(gin + r * hn);
            const float hp = h2[hj];
            const float hvv = (1.f - z) * n + z * hp;
            const unsigned o = (unsigned)t * (unsigned)R + (unsigned)j;   // 32-bit offsets: scalar base + one VGPR per store
            Gr[o] = r; Gz[o] = z; Gn[o] = n; Ghn[o] = hn;
            Hprev[o] = hp;
            Hout[o] = hvv;
            h2[hj] = hvv;
        }
        if (stamps && t >= 8 && t < 12) stamps[(t - 8) * 4 + 3] = __builtin_readcyclecounter();
        gir = nir; giz = niz; gin = nin;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
}
#define SV_KS_SL 25
#define SV_KS_NR 10
#define SV_KS_KG 180
static size_t sv_gru_ks_lds() { return sizeof(float) * (8 * ((SV_KS_SL + 3) & ~3) + 64 * SV_KS_NR + (size_t)((SV_KS_NR * SV_KS_SL - SV_KS_KG + 3) / 4) * 512 * 4); }

// Backward recurrence in the K-sliced layout of k_sv_gru_fwd_ks (same thread -> weights map: rows slot + 64 i, columns of slice s):
// dh_{t-1}[k] += sum_i W_hh[i][k] dgh[i] -- a thread multiplies its 10 x 25 weights with its 10 values of dgh (LDS, slot-major: three
// float4) into 25 column sums, the 8 row groups of the wave meet by a reduce-scatter over lanes (permlane32_swap, permlane16_swap,
// row_ror:8 -- two values per instruction, ~50 operations for the 25 sums; lane group g ends with columns 8 m + bitrev3(g)), the 8
// waves through part [8][R] in LDS.  Two barriers per step (k_sv_gru_bwd_all: three), ~240 LDS wave-instructions (~650).
__device__ __forceinline__ void sv_permlane32_swap(float& x, float& y)   // rows 2,3 of x <-> rows 0,1 of y
{
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
}
__device__ __forceinline__ void sv_permlane16_swap(float& x, float& y)   // odd rows of x <-> even rows of y
{
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
}
template <int SL, int NR, int KG>
__global__ __launch_bounds__(512) void k_sv_gru_bwd_ks(const float* __restrict__ dHout, const float* __restrict__ Whh, const int32_t* __restrict__ seq_ptr,
                                                       int T_one, int R, const float* __restrict__ Hprev, const float* __restrict__ Gr,
                                                       const float* __restrict__ Gz, const float* __restrict__ Gn, const float* __restrict__ Ghn,
                                                       float* __restrict__ dGI, float* __restrict__ dGH)
{
    constexpr int GS = (NR + 3) & ~3, NWT = NR * SL, CL = (NWT - KG + 3) / 4, PS = 8 * SL;
    constexpr int N32 = (SL + 1) / 2, N16 = (N32 + 1) / 2, N8 = (N16 + 1) / 2;
    static_assert(KG % 4 == 0 && KG <= NWT, "register-resident weights: whole float4 groups");
    extern __shared__ __attribute__((aligned(16))) float sm[];   // dg2 [64][GS] | part [8][PS] | wl [CL][512][4]
    float* dg2 = sm;
    float* part = sm + 64 * GS;
    float* wl = part + 8 * PS;
    const int R3 = 3 * R;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, s = lane & 7, slot = wv * 8 + (lane >> 3);
    const int t0 = seq_ptr ? seq_ptr[blockIdx.x] : 0;
    const int T = seq_ptr ? seq_ptr[blockIdx.x + 1] - t0 : T_one;
    dHout += (size_t)t0 * R; Hprev += (size_t)t0 * R;
    Gr += (size_t)t0 * R; Gz += (size_t)t0 * R; Gn += (size_t)t0 * R; Ghn += (size_t)t0 * R;
    dGI += (size_t)t0 * R3; dGH += (size_t)t0 * R3;
    const int last = R3 * R - 1;
    auto wload = [&](int q) {
        const int i = q / SL, kk = q % SL, row = slot + 64 * i, k = s * SL + kk;
        const float v = Whh[min(row * R + k, last)];
        return v * ((row < R3 && k < R && q < NWT) ? 1.f : 0.f);
    };
    float wr[KG];
#pragma unroll
    for (int q = 0; q < KG; ++q) wr[q] = wload(q);
#pragma unroll 8
    for (int q = 0; q < CL * 4; ++q) wl[((size_t)(q >> 2) * 512 + tid) * 4 + (q & 3)] = wload(KG + q);
    for (int j = tid; j < 64 * GS + 8 * PS; j += 512) sm[j] = 0.f;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // gate threads: column gj = tid for tid < R; the threads beyond shadow column R - 1 (same loads, same arithmetic, same stores to
    // the same places) so that the step has NO divergent block around its global loads and stores: behind an `if (tid < R)` hipcc
    // cannot count the outstanding stores on both paths and waits vmcnt(0) -- a store round trip per time step
    const int gj = min(tid, R - 1);
    // where this gate thread's three rows of dgh live: row i -> [(i & 63)][i >> 6]
    const int p0 = (gj & 63) * GS + (gj >> 6), p1 = ((R + gj) & 63) * GS + ((R + gj) >> 6), p2 = ((2 * R + gj) & 63) * GS + ((2 * R + gj) >> 6);
    const bool b3 = (lane & 8) != 0;
    const int rev = ((lane >> 5) & 1) | (((lane >> 4) & 1) << 1) | (((lane >> 3) & 1) << 2);   // 4 b3 + 2 b4 + b5
    float dh = 0.f;
    float vd = 0.f, vr = 0.f, vz = 0.f, vn = 0.f, vhn = 0.f, vhp = 0.f;
    if (T > 0) {
        const size_t o = (size_t)(T - 1) * R + gj;
        vd = dHout[o]; vr = Gr[o]; vz = Gz[o]; vn = Gn[o]; vhn = Ghn[o]; vhp = Hprev[o];
    }
    const float4* gv = (const float4*)(dg2 + slot * GS);
    unsigned long long* stamps = (tid == 0 && blockIdx.x == 0 && g_sv_stamps) ? g_sv_stamps + 16 : nullptr;   // entries 16..31: this kernel
    for (int t = T - 1; t >= 0; --t) {
        const bool stamp = stamps && t >= T - 12 && t < T - 8;
        if (stamp) stamps[(T - 9 - t) * 4 + 0] = __builtin_readcyclecounter();
        {
            const int j = gj;
            const float d = dh + vd;
            const float dn = d * (1.f - vz);
            const float dzp = d * (vhp - vn) * vz * (1.f - vz);
            const float dnp = dn * (1.f - vn * vn);
            const float drp = dnp * vhn * vr * (1.f - vr);
            const unsigned o = (unsigned)t * (unsigned)R3 + (unsigned)j, oz = o + (unsigned)R, on = o + 2u * (unsigned)R;
            dGI[o] = drp; dGI[oz] = dzp; dGI[on] = dnp;
            dGH[o] = drp; dGH[oz] = dzp; dGH[on] = dnp * vr;
            dg2[p0] = drp; dg2[p1] = dzp; dg2[p2] = dnp * vr;
            dh = d * vz;   // the direct path h_{t-1} -> h_t; the path through the gates is added below
        }
        // the saved gate values of step t - 1, requested BEHIND this step's use of its own (and unconditionally, clamped): requested
        // in front of it, under a condition, hipcc waits vmcnt(0) before the gate arithmetic -- for the loads it has just issued,
        // a full L2 round trip per time step (1 200 cycles in this phase instead of ~300)
        float nd, nr, nz, nn, nhn, nhp;
        {
            const unsigned o = (unsigned)max(t - 1, 0) * (unsigned)R + (unsigned)gj;   // 32-bit offsets: scalar base + one VGPR
            nd = dHout[o]; nr = Gr[o]; nz = Gz[o]; nn = Gn[o]; nhn = Ghn[o]; nhp = Hprev[o];
        }
        if (stamp) stamps[(T - 9 - t) * 4 + 1] = __builtin_readcyclecounter();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (stamp) stamps[(T - 9 - t) * 4 + 2] = __builtin_readcyclecounter();
        {
            float4 gq[GS / 4];
#pragma unroll
            for (int q = 0; q < GS / 4; ++q) gq[q] = gv[q];
            auto gval = [&](int i) { const float4 v = gq[i >> 2]; return (i & 3) == 0 ? v.x : (i & 3) == 1 ? v.y : (i & 3) == 2 ? v.z : v.w; };
            float acc[SL];
#pragma unroll
            for (int kk = 0; kk < SL; ++kk) acc[kk] = 0.f;
#pragma unroll
            for (int q = 0; q < KG; ++q) acc[q % SL] += wr[q] * gval(q / SL);
#pragma unroll
            for (int c = 0; c < CL; ++c) {
                const float4 w = *(const float4*)(wl + ((size_t)c * 512 + tid) * 4);
                const float we[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int q = KG + 4 * c + e;
                    if (q < NWT) acc[q % SL] += we[e] * gval(q / SL);
                }
            }
            // reduce-scatter over the 8 row groups of the wave (lane bits 5, 4, 3)
            float u[N32];
#pragma unroll
            for (int p = 0; p < N32; ++p) {
                float x = acc[2 * p], y = acc[2 * p + 1 < SL ? 2 * p + 1 : 2 * p];
                sv_permlane32_swap(x, y);
                u[p] = x + y;          // lanes 0-31: column 2p summed over bit 5, lanes 32-63: column 2p + 1
            }
            float v[N16];
#pragma unroll
            for (int n = 0; n < N16; ++n) {
                float x = u[2 * n], y = u[2 * n + 1 < N32 ? 2 * n + 1 : 2 * n];
                sv_permlane16_swap(x, y);
                v[n] = x + y;          // even rows: u[2n] summed over bit 4, odd rows: u[2n + 1]
            }
#pragma unroll
            for (int m = 0; m < N8; ++m) {
                const float pz = v[2 * m], qz = v[2 * m + 1 < N16 ? 2 * m + 1 : 2 * m];
                const float keep = b3 ? qz : pz, send = b3 ? pz : qz;
                const float f = keep + sv_dpp<0x128>(send);   // row_ror:8
                const int idx = 8 * m + rev;
                if (idx < SL) part[wv * PS + s * SL + idx] = f;
            }
        }
        if (stamp) stamps[(T - 9 - t) * 4 + 3] = __builtin_readcyclecounter();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        {
            float a = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) a += part[w * PS + gj];
            dh += a;
        }
        asm volatile("" : "+v"(nd), "+v"(nr), "+v"(nz), "+v"(nn), "+v"(nhn), "+v"(nhp));   // they landed a mat-vec ago: the wait belongs HERE
        vd = nd; vr = nr; vz = nz; vn = nn; vhn = nhn; vhp = nhp;
    }
}

extern "C" void rtxdbg_svae_set_stamps(unsigned long long* dev)   // measurement hook (tools/svae_stamps.py): 32 device entries (forward 0..15, backward 16..31); not part of the ABI
{
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_sv_stamps), &dev, sizeof(dev));
}

// GRU backward through time.  dHout[t] = gradient w.r.t. the GRU output at step t.  Writes the gate pre-activation
// gradients dGI [T][3R] (input side) and dGH [T][3R] (hidden side; differs in the n block by the factor r).
// dh_{t-1} += W_hh^T dgh: thread (column k, row chunk c) sums W_hh[i][k] dgh[i] over its chunk of rows -- consecutive
// lanes read consecutive k (coalesced rows), every load is independent -- and the chunks meet in LDS (5.5 us per step).
__global__ __launch_bounds__(1024) void k_sv_gru_bwd(const float* __restrict__ dHout, const float* __restrict__ Whh,
                                                     const int32_t* __restrict__ seq_ptr, int T_one, int R,
                                                     const float* __restrict__ Hprev, const float* __restrict__ Gr, const float* __restrict__ Gz,
                                                     const float* __restrict__ Gn, const float* __restrict__ Ghn, float* __restrict__ dGI,
                                                     float* __restrict__ dGH)
{
    const int t0 = seq_ptr ? seq_ptr[blockIdx.x] : 0;
    const int T = seq_ptr ? seq_ptr[blockIdx.x + 1] - t0 : T_one;
    dHout += (size_t)t0 * R; Hprev += (size_t)t0 * R;
    Gr += (size_t)t0 * R; Gz += (size_t)t0 * R; Gn += (size_t)t0 * R; Ghn += (size_t)t0 * R;
    dGI += (size_t)t0 * 3 * R; dGH += (size_t)t0 * 3 * R;
    extern __shared__ __attribute__((aligned(16))) float sm[];   // dh [R] | dgh [3R] | part [NC][R]
    float* dh = sm;
    float* dgh = sm + R;
    float* part = sm + 4 * R;
    const int tid = threadIdx.x;
    const int NC = max(1, min(16, 1024 / R));     // row chunks: as many as the workgroup has threads for
    const int rows_per = (3 * R + NC - 1) / NC;
    for (int j = tid; j < R; j += 1024) dh[j] = 0.f;
    __syncthreads();
    for (int t = T - 1; t >= 0; --t) {
        for (int j = tid; j < R; j += 1024) {
            const float d = dh[j] + dHout[(size_t)t * R + j];
            const float r = Gr[(size_t)t * R + j], z = Gz[(size_t)t * R + j], n = Gn[(size_t)t * R + j], hn = Ghn[(size_t)t * R + j];
            const float hp = Hprev[(size_t)t * R + j];
            const float dn = d * (1.f - z);
            const float dzp = d * (hp - n) * z * (1.f - z);
            const float dnp = dn * (1.f - n * n);
            const float drp = dnp * hn * r * (1.f - r);
            float* gi = dGI + (size_t)t * 3 * R;
            float* gh = dGH + (size_t)t * 3 * R;
            gi[j] = drp; gi[R + j] = dzp; gi[2 * R + j] = dnp;
            gh[j] = drp; gh[R + j] = dzp; gh[2 * R + j] = dnp * r;
            dgh[j] = drp; dgh[R + j] = dzp; dgh[2 * R + j] = dnp * r;
            dh[j] = d * z;   // the direct path h_{t-1} -> h_t; the path through the gates is added below
        }
        __syncthreads();
        for (int e = tid; e < NC * R; e += 1024) {
            const int c = e / R, k = e - c * R;
            const int i0 = c * rows_per, i1 = min(3 * R, i0 + rows_per);
            const float* w = Whh + (size_t)i0 * R + k;
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
            int i = i0;
            for (; i + 4 <= i1; i += 4) {
                a0 += w[0] * dgh[i];
                a1 += w[R] * dgh[i + 1];
                a2 += w[2 * (size_t)R] * dgh[i + 2];
                a3 += w[3 * (size_t)R] * dgh[i + 3];
                w += 4 * (size_t)R;
            }
            for (; i < i1; ++i) { a0 += w[0] * dgh[i]; w += R; }
            part[c * R + k] = (a0 + a1) + (a2 + a3);
        }
        __syncthreads();
        for (int j = tid; j < R; j += 1024) {
            float s = dh[j];
            for (int c = 0; c < NC; ++c) s += part[c * R + j];
            dh[j] = s;
        }
        __syncthreads();
    }
}

// Backward recurrence with ALL of W_hh resident (see k_sv_gru_fwd_all): thread (row chunk c, column k) keeps the first KRB
// weights W[c * RP + i][k] of its chunk in registers and the rest in LDS ([chunk of 4 i][thread][4]); a time step reads from
// outside the CU only its six saved gate values, requested one step ahead.  Raw s_barriers behind LDS-only waits.
template <int KRB>
__global__ __launch_bounds__(1024) void k_sv_gru_bwd_all(const float* __restrict__ dHout, const float* __restrict__ Whh,
                                                         const int32_t* __restrict__ seq_ptr, int T_one, int R, int NC, int RP /* rows per chunk, % 4 == 0 */,
                                                         const float* __restrict__ Hprev, const float* __restrict__ Gr, const float* __restrict__ Gz,
                                                         const float* __restrict__ Gn, const float* __restrict__ Ghn, float* __restrict__ dGI,
                                                         float* __restrict__ dGH)
{
    const int t0 = seq_ptr ? seq_ptr[blockIdx.x] : 0;
    const int T = seq_ptr ? seq_ptr[blockIdx.x + 1] - t0 : T_one;
    const int R3 = 3 * R;
    dHout += (size_t)t0 * R; Hprev += (size_t)t0 * R;
    Gr += (size_t)t0 * R; Gz += (size_t)t0 * R; Gn += (size_t)t0 * R; Ghn += (size_t)t0 * R;
    dGI += (size_t)t0 * R3; dGH += (size_t)t0 * R3;
    extern __shared__ __attribute__((aligned(16))) float sm[];   // dh [Rp] | dgh [NC * RP + KRB + 4] | part [NC * R] | wl [CL][1024][4]
    const int Rp = (R + 3) & ~3;
    // dgh is read up to KRB + 4 CL floats past a chunk's start whatever R is (zero weights there, but zero times an uninitialised
    // LDS word may be NaN): its area covers every such read and is zeroed once
    const int DGS = NC * RP + KRB + 4;
    float* dh = sm;
    float* dgh = sm + Rp;
    float* part = dgh + DGS;
    float* wl = part + ((NC * R + 3) & ~3);
    const int tid = threadIdx.x;
    const bool own = tid < NC * R;
    const int c = own ? tid / R : 0, k = own ? tid - c * R : 0;
    const int i0 = c * RP;
    const int CL = RP > KRB ? (RP - KRB + 3) / 4 : 0;
    float wr[KRB];
    {
        const int last = R3 * R - 1;
#pragma unroll
        for (int i = 0; i < KRB; ++i) {
            const float v = Whh[min((i0 + i) * R + k, last)];
            wr[i] = (own && i < RP && i0 + i < R3) ? v : 0.f;
        }
        for (int i = 0; i < CL * 4; ++i) {
            const float v = Whh[min((i0 + KRB + i) * R + k, last)];
            wl[((size_t)(i >> 2) * 1024 + tid) * 4 + (i & 3)] = (own && KRB + i < RP && i0 + KRB + i < R3) ? v : 0.f;
        }
    }
    for (int j = tid; j < Rp; j += 1024) dh[j] = 0.f;
    for (int j = tid; j < DGS; j += 1024) dgh[j] = 0.f;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const bool gate = tid < R;
    float vd = 0.f, vr = 0.f, vz = 0.f, vn = 0.f, vhn = 0.f, vhp = 0.f;
    if (gate && T > 0) {
        const size_t o = (size_t)(T - 1) * R + tid;
        vd = dHout[o]; vr = Gr[o]; vz = Gz[o]; vn = Gn[o]; vhn = Ghn[o]; vhp = Hprev[o];
    }
    const float4* dv = (const float4*)(dgh + i0);
    for (int t = T - 1; t >= 0; --t) {
        float nd = 0.f, nr = 0.f, nz = 0.f, nn = 0.f, nhn = 0.f, nhp = 0.f;
        if (gate && t > 0) {
            const size_t o = (size_t)(t - 1) * R + tid;
            nd = dHout[o]; nr = Gr[o]; nz = Gz[o]; nn = Gn[o]; nhn = Ghn[o]; nhp = Hprev[o];
        }
        if (gate) {
            const int j = tid;
            const float d = dh[j] + vd;
            const float dn = d * (1.f - vz);
            const float dzp = d * (vhp - vn) * vz * (1.f - vz);
            const float dnp = dn * (1.f - vn * vn);
            const float drp = dnp * vhn * vr * (1.f - vr);
            float* gi = dGI + (size_t)t * R3;
            float* gh = dGH + (size_t)t * R3;
            gi[j] = drp; gi[R + j] = dzp; gi[2 * R + j] = dnp;
            gh[j] = drp; gh[R + j] = dzp; gh[2 * R + j] = dnp * vr;
            dgh[j] = drp; dgh[R + j] = dzp; dgh[2 * R + j] = dnp * vr;
            dh[j] = d * vz;   // the direct path h_{t-1} -> h_t; the path through the gates is added below
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        {
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
            for (int i = 0; i < KRB; i += 4) {
                const float4 x = dv[i >> 2];
                a0 += wr[i] * x.x; a1 += wr[i + 1] * x.y; a2 += wr[i + 2] * x.z; a3 += wr[i + 3] * x.w;
            }
#pragma nounroll   // (unrolling by 4 / 2 spills 8 / 3 registers at the 128-VGPR cap of a 1024-thread workgroup: 896 -> 799 users/s)
            for (int q = 0; q < CL; ++q) {
                const float4 w = *(const float4*)(wl + ((size_t)q * 1024 + tid) * 4);
                const float4 x = dv[(KRB >> 2) + q];
                a0 += w.x * x.x; a1 += w.y * x.y; a2 += w.z * x.z; a3 += w.w * x.w;
            }
            if (own) part[tid] = (a0 + a1) + (a2 + a3);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (gate) {
            float sacc = dh[tid];
            for (int q = 0; q < NC; ++q) sacc += part[q * R + tid];
            dh[tid] = sacc;
        }
        vd = nd; vr = nr; vz = nz; vn = nn; vhn = nhn; vhp = nhp;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
}

// encoder head: out [T][2Z] = mu | logvar; z = mu + eps * exp(logvar / 2) with eps injected or Philox (always sampled)
__global__ __launch_bounds__(256) void k_sv_reparam(const float* out, int T, int Z, const float* eps_in, uint64_t seed, uint64_t offset,
                                                    float* mu, float* lv, float* eps_out, float* z)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= T * Z) return;
    const int t = idx / Z, j = idx % Z;
    const float m = out[(size_t)t * 2 * Z + j], l = out[(size_t)t * 2 * Z + Z + j];
    const float e = eps_in ? eps_in[idx] : rtx_normal(seed, offset, (uint64_t)idx);
    mu[idx] = m; lv[idx] = l; eps_out[idx] = e;
    z[idx] = m + e * expf(0.5f * l);
}

// gradient w.r.t. the encoder head output [T][2Z] from dz and the KL term  beta * mean_t(-0.5 sum_j (1 + lv - mu^2 - e^lv));
// kl_rows[t] = -0.5 sum_j(...)
__global__ __launch_bounds__(256) void k_sv_reparam_bwd(const float* dz, const float* mu, const float* lv, const float* eps, int T, int Z,
                                                        float beta_over_T, const float* __restrict__ kl_scale /* per row, or NULL */, float* dout,
                                                        float* kl_rows)
{
    __shared__ float red[4];
    const int t = blockIdx.x;
    if (kl_scale) beta_over_T = kl_scale[t];   // packed users: beta / (T_user * n_users)
    float kl = 0.f;
    for (int j = threadIdx.x; j < Z; j += 256) {
        const int idx = t * Z + j;
        const float m = mu[idx], l = lv[idx], el = expf(l), sd = expf(0.5f * l);
        kl += -0.5f * (1.f + l - m * m - el);
        if (dout) {
            const float d = dz[idx];
            dout[(size_t)t * 2 * Z + j] = d + beta_over_T * m;
            dout[(size_t)t * 2 * Z + Z + j] = d * eps[idx] * 0.5f * sd + beta_over_T * 0.5f * (el - 1.f);
        }
    }
    kl = block_sum(kl, red);
    if (threadIdx.x == 0) kl_rows[t] = kl;
}

// per time step: log-softmax NLL against the target row (CSR, or dense [T][I]) and the logits gradient
//   nll_t = -sum_i y_ti (x_ti - lse_t);  dlogits = (s_t softmax - y) * inv_d
__global__ __launch_bounds__(256) void k_sv_loss(const float* logits, int T, int I, const int64_t* tptr, const int32_t* tidx,
                                                 const float* ydense, float inv_d, const float* __restrict__ nll_scale /* per row, or NULL */,
                                                 float* dlogits, float* row_loss)
{
    __shared__ float red[4];
    const int t = blockIdx.x, tid = threadIdx.x;
    if (nll_scale) inv_d = nll_scale[t];   // packed users: 1 / (d_user * n_users)
    const float* x = logits + (size_t)t * I;
    float mx = -INFINITY;
    for (int i = tid; i < I; i += 256) mx = fmaxf(mx, x[i]);
    mx = block_max(mx, red);
    float se = 0.f;
    for (int i = tid; i < I; i += 256) se += expf(x[i] - mx);
    se = block_sum(se, red);
    const float lse = mx + logf(se);
    float s = 0.f, dot = 0.f;
    if (ydense) {
        const float* y = ydense + (size_t)t * I;
        for (int i = tid; i < I; i += 256) { s += y[i]; dot += y[i] * x[i]; }
    } else {
        for (int64_t k = tptr[t] + tid; k < tptr[t + 1]; k += 256) { s += 1.f; dot += x[tidx[k]]; }
    }
    s = block_sum(s, red);
    dot = block_sum(dot, red);
    if (tid == 0) row_loss[t] = s * lse - dot;
    if (dlogits) {
        float* d = dlogits + (size_t)t * I;
        if (ydense) {
            const float* y = ydense + (size_t)t * I;
            for (int i = tid; i < I; i += 256) d[i] = (s * expf(x[i] - lse) - y[i]) * inv_d;
        } else {
            for (int i = tid; i < I; i += 256) d[i] = s * expf(x[i] - lse) * inv_d;
            __syncthreads();
            for (int64_t k = tptr[t] + tid; k < tptr[t + 1]; k += 256) d[tidx[k]] -= inv_d;   // indices are distinct within a row
        }
    }
}

__global__ __launch_bounds__(256) void k_sv_final_loss(const float* row_loss, const float* kl_rows, int T, float inv_d, float beta_over_T,
                                                       const float* __restrict__ nll_scale, const float* __restrict__ kl_scale, float* loss_out,
                                                       float* loss_accum, uint32_t* mailbox, uint32_t seq)
{
    __shared__ float red[4];
    float a = 0.f, b = 0.f;
    if (nll_scale) {   // packed users: every row carries its user's factors
        for (int t = threadIdx.x; t < T; t += 256) { a += row_loss[t] * nll_scale[t]; b += kl_rows[t] * kl_scale[t]; }
        inv_d = 1.f; beta_over_T = 1.f;
    } else {
        for (int t = threadIdx.x; t < T; t += 256) { a += row_loss[t]; b += kl_rows[t]; }
    }
    a = block_sum(a, red);
    b = block_sum(b, red);
    if (threadIdx.x == 0) {
        const float l = a * inv_d + beta_over_T * b;
        if (loss_out) loss_out[0] = l;
        if (loss_accum) loss_accum[0] += l;
        if (mailbox) {   // (system scope: the host spins on the ticket -- rtx_svae_wait_loss)
            __hip_atomic_store(mailbox, __float_as_uint(l), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(mailbox + 1, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

__global__ __launch_bounds__(256) void k_sv_mask_items(const int32_t* items, int T, float* row)
{
    for (int t = threadIdx.x; t < T; t += 256) row[items[t]] = -INFINITY;
}

// ------------------------------------------------------------------------------------------------ host side
static int sv_alloc(rtx_svae* s, float** p, size_t n)
{
    hipError_t rc = hipMalloc((void**)p, sizeof(float) * (n ? n : 4));
    if (rc != hipSuccess) {
        rtx_set_error("svae: hipMalloc(%zu floats) failed: %s", n, hipGetErrorString(rc));
        return RTX_ENOMEM;
    }
    s->allocs.push_back(*p);
    return RTX_OK;
}

static int sv_gemm(rtx_svae* s, hipStream_t st, const float* A, long sam, long sak, const float* B, long sbn, long sbk, float* C, long ldc, int M,
                   int N, int K, int epi = SV_EPI_NONE, const float* bias = nullptr, const float* Q = nullptr, long ldq = 0, int lane = 0)
{
    if (M <= 0 || N <= 0) return RTX_OK;
    SvGemm g = {A, sam, sak, B, sbn, sbk, C, ldc, M, N, K, 1.f, epi, bias, Q, ldq, 0, nullptr};
    const int tiles = ((N + 63) / 64) * ((M + 63) / 64);
    // few output tiles and a long K (the [T, hidden] = [T, n_items] x [n_items, hidden] backward-data product): split K
    // so that a few hundred workgroups share the reduction instead of a handful walking it end to end
    // (with packed users K = sum T reaches tens of thousands in the weight-gradient products while their outputs are a few
    // hundred tiles: the [n_items, 150] decoder gradient ran 162 workgroups for 1 ms until it was split as well)
    int splits = 1;
    if (tiles < 512 && K >= 512) {
        splits = std::min((K + 255) / 256, std::max(1, 1024 / tiles));
        if ((size_t)splits * M * N > s->part_elems) splits = (int)(s->part_elems / ((size_t)M * N));
    }
    if (splits > 1) {
        g.kchunk = ((K + splits - 1) / splits + 15) / 16 * 16;
        splits = (K + g.kchunk - 1) / g.kchunk;
        g.part = lane ? s->part2 : s->part;   // the side stream sums into its own buffer
        if (s->opt_gemm_bf16) hipLaunchKernelGGL(k_sv_gemm<true>, dim3((N + 63) / 64, (M + 63) / 64, splits), dim3(256), 0, st, g);
        else hipLaunchKernelGGL(k_sv_gemm<false>, dim3((N + 63) / 64, (M + 63) / 64, splits), dim3(256), 0, st, g);
        hipLaunchKernelGGL(k_sv_splitk_reduce, dim3((unsigned)(((long)M * N + 255) / 256)), dim3(256), 0, st, g, splits);
    } else {
        if (s->opt_gemm_bf16) hipLaunchKernelGGL(k_sv_gemm<true>, dim3((N + 63) / 64, (M + 63) / 64), dim3(256), 0, st, g);
        else hipLaunchKernelGGL(k_sv_gemm<false>, dim3((N + 63) / 64, (M + 63) / 64), dim3(256), 0, st, g);
    }
    RTX_HIP(hipGetLastError());
    return RTX_OK;
}

static int sv_colsum(hipStream_t st, const float* D, long ld, int T, int N, float* out)
{
    if (T <= 2048) {
        hipLaunchKernelGGL(k_sv_colsum1, dim3((N + 63) / 64), dim3(256), 0, st, D, ld, T, N, out);
        RTX_HIP(hipGetLastError());
        return RTX_OK;
    }
    RTX_HIP(hipMemsetAsync(out, 0, sizeof(float) * N, st));
    hipLaunchKernelGGL(k_sv_colsum, dim3((N + 255) / 256, (T + 31) / 32), dim3(256), 0, st, D, ld, T, N, out);
    RTX_HIP(hipGetLastError());
    return RTX_OK;
}

// parameter order = SVAE_net.parameters(): enc W,b ..., dec W,b ..., item_embed.weight, gru.weight_ih_l0, weight_hh_l0,
// bias_ih_l0, bias_hh_l0 (VAE_net.__init__ registers the MLPs before SVAE_net adds the embedding and the GRU)
enum { SV_T_EMB = 0, SV_T_WIH = 1, SV_T_WHH = 2, SV_T_BIH = 3, SV_T_BHH = 4 };
static int sv_tail(const rtx_svae* s, int which) { return 2 * s->NL + which; }

static void sv_shape(const rtx_svae* s, int t, int* rows, int* cols)
{
    if (t < 2 * s->NL) {
        const SvLayer& l = s->L[t / 2];
        *rows = l.out;
        *cols = (t & 1) ? 1 : l.in;
        return;
    }
    switch (t - 2 * s->NL) {
    case SV_T_EMB: *rows = s->I; *cols = s->E; break;
    case SV_T_WIH: *rows = 3 * s->R; *cols = s->E; break;
    case SV_T_WHH: *rows = 3 * s->R; *cols = s->R; break;
    default: *rows = 3 * s->R; *cols = 1; break;
    }
}

// embedding -> GRU -> encoder -> (sampled) z -> decoder; logits of all T steps land in L.back().A
// `seq_ptr` (device, n_seq + 1 entries) cuts the T rows into independent sequences; NULL = one sequence
static int sv_forward(rtx_svae* s, const int32_t* items, int T, const int32_t* seq_ptr, int n_seq, const float* eps_in, uint64_t seed, uint64_t offset,
                      hipStream_t st)
{
    const int E = s->E, R = s->R, Z = s->Z;
    hipLaunchKernelGGL(k_sv_embed, dim3(T), dim3(256), 0, st, items, T, E, s->params[sv_tail(s, SV_T_EMB)], s->X);
    RTX_TRY(sv_gemm(s, st, s->X, E, 1, s->params[sv_tail(s, SV_T_WIH)], E, 1, s->GI, 3 * R, T, 3 * R, E, SV_EPI_BIAS,
                    s->params[sv_tail(s, SV_T_BIH)]));
    if (s->gru_ks_lds > 0) {   // all of W_hh resident, the mat-vec split over K inside the wave
        hipLaunchKernelGGL((k_sv_gru_fwd_ks<SV_KS_SL, SV_KS_NR, SV_KS_KG>), dim3(seq_ptr ? n_seq : 1), dim3(512), s->gru_ks_lds, st, s->GI,
                           s->params[sv_tail(s, SV_T_WHH)], s->params[sv_tail(s, SV_T_BHH)], seq_ptr, T, R, s->Hout, s->Hprev, s->Gr, s->Gz, s->Gn, s->Ghn);
    } else if (s->gru_rows_lds > 0 && s->opt_gru_rows) {   // all of W_hh resident: whole rows on 512 threads (no spills)
        hipLaunchKernelGGL(k_sv_gru_fwd_rows<SV_GRU_KR2>, dim3(seq_ptr ? n_seq : 1), dim3(512), s->gru_rows_lds, st, s->GI, s->params[sv_tail(s, SV_T_WHH)],
                           s->params[sv_tail(s, SV_T_BHH)], seq_ptr, T, R, s->Hout, s->Hprev, s->Gr, s->Gz, s->Gn, s->Ghn);
    } else if (s->gru_fwd_lds > 0) {   // all of W_hh resident in registers + LDS
        hipLaunchKernelGGL(k_sv_gru_fwd_all<SV_GRU_KR>, dim3(seq_ptr ? n_seq : 1), dim3(1024), s->gru_fwd_lds, st, s->GI, s->params[sv_tail(s, SV_T_WHH)],
                           s->params[sv_tail(s, SV_T_BHH)], seq_ptr, T, R, s->gru_kh, s->Hout, s->Hprev, s->Gr, s->Gz, s->Gn, s->Ghn);
    } else {
        hipLaunchKernelGGL(k_sv_transpose, dim3((R + 63) / 64, (3 * R + 63) / 64), dim3(256), 0, st, s->params[sv_tail(s, SV_T_WHH)], 3 * R, R,
                           s->WhhT);
        hipLaunchKernelGGL(k_sv_gru_fwd, dim3(seq_ptr ? n_seq : 1), dim3(1024), sizeof(float) * (4 * R + 4), st, s->GI, s->WhhT,
                           s->params[sv_tail(s, SV_T_BHH)], seq_ptr, T, R, s->Hout, s->Hprev, s->Gr, s->Gz, s->Gn, s->Ghn);
    }
    const float* in = s->Hout;   // rnn_out[t] = h after step t
    long ld_in = R;
    for (int li = 0; li < s->NL; ++li) {
        SvLayer& l = s->L[li];
        RTX_TRY(sv_gemm(s, st, in, ld_in, 1, s->params[2 * li], l.in, 1, l.A, l.out, T, l.out, l.in, l.tanh_act ? SV_EPI_BIAS_TANH : SV_EPI_BIAS,
                        s->params[2 * li + 1]));
        in = l.A;
        ld_in = l.out;
        if (li == s->n_enc - 1) {
            hipLaunchKernelGGL(k_sv_reparam, dim3((T * Z + 255) / 256), dim3(256), 0, st, l.A, T, Z, eps_in, seed, offset, s->mu, s->lv, s->eps,
                               s->zl);
            in = s->zl;
            ld_in = Z;
        }
    }
    RTX_HIP(hipGetLastError());
    return RTX_OK;
}

static int sv_check(const rtx_svae* s, const int32_t* items, int T, bool train)
{
    RTX_CHECK(s, RTX_EINVAL, "svae: NULL handle");
    RTX_CHECK(s->bound, RTX_ESTATE, "svae: parameters are not bound (rtx_svae_bind)");
    RTX_CHECK(!train || s->can_train, RTX_ESTATE, "svae: gradient / Adam buffers are not bound");
    RTX_CHECK(items && T >= 1 && T <= s->Tmax, RTX_EINVAL, "svae: sequence length %d outside [1, max_len = %d]", T, s->Tmax);
    return RTX_OK;
}

extern "C" {

int rtx_svae_create(const rtx_svae_cfg* cfg, rtx_svae** out)
{
    RTX_CHECK(cfg && out, RTX_EINVAL, "svae_create: NULL argument");
    RTX_CHECK(cfg->n_items > 0 && cfg->embed_size > 0 && cfg->rnn_size > 0 && cfg->max_len > 0, RTX_EINVAL, "svae_create: bad sizes");
    RTX_CHECK(cfg->n_enc >= 1 && cfg->n_dec >= 1 && cfg->n_enc <= SV_MAX_LAYERS && cfg->n_dec <= SV_MAX_LAYERS, RTX_EINVAL,
              "svae_create: bad layer counts %d/%d", cfg->n_enc, cfg->n_dec);
    RTX_CHECK(cfg->enc_dims[0] == cfg->rnn_size, RTX_EINVAL, "svae_create: enc_dims[0] = %d must equal rnn_size = %d", cfg->enc_dims[0],
              cfg->rnn_size);
    RTX_CHECK(cfg->enc_dims[cfg->n_enc] == cfg->dec_dims[0], RTX_EINVAL, "svae_create: latent size mismatch");
    RTX_CHECK(cfg->dec_dims[cfg->n_dec] == cfg->n_items, RTX_EINVAL, "svae_create: dec_dims[-1] = %d must equal n_items = %d",
              cfg->dec_dims[cfg->n_dec], cfg->n_items);
    RTX_CHECK(cfg->rnn_size <= 1024, RTX_EINVAL, "svae_create: rnn_size %d exceeds the LDS budget of the recurrent kernels (1024)", cfg->rnn_size);
    RTX_CHECK(2 * (cfg->n_enc + cfg->n_dec) + 5 <= RTX_MAX_TENSORS, RTX_EINVAL, "svae_create: too many parameter tensors for one Adam launch");
    int ndev = 0;
    hipError_t h = hipGetDeviceCount(&ndev);
    RTX_CHECK(h == hipSuccess && ndev > 0, RTX_EHIP, "no HIP device available (%s): librectorch_hip has no CPU path", hipGetErrorString(h));
    rtx_svae* s = new rtx_svae();
    s->cfg = *cfg;
    s->I = cfg->n_items; s->E = cfg->embed_size; s->R = cfg->rnn_size; s->Z = cfg->enc_dims[cfg->n_enc];
    s->n_enc = cfg->n_enc; s->NL = cfg->n_enc + cfg->n_dec; s->Tmax = cfg->max_len;
    for (int i = 0; i < cfg->n_enc; ++i) {
        SvLayer l;
        l.in = cfg->enc_dims[i];
        l.out = (i == cfg->n_enc - 1) ? 2 * cfg->enc_dims[i + 1] : cfg->enc_dims[i + 1];   // mu | logvar (nets.py:262-265)
        l.tanh_act = (i != cfg->n_enc - 1);                                                   // VAE_net.encode (nets.py:287-294)
        s->L.push_back(l);
    }
    for (int i = 0; i < cfg->n_dec; ++i) {
        SvLayer l;
        l.in = cfg->dec_dims[i]; l.out = cfg->dec_dims[i + 1];
        l.tanh_act = (i != cfg->n_dec - 1);                                                   // SVAE_net.decode (nets.py:683-687)
        s->L.push_back(l);
    }
    s->n_tensors = 2 * s->NL + 5;
    s->params.assign(s->n_tensors, nullptr);
    s->grads = s->m = s->v = s->params;
    const size_t T = s->Tmax, R = s->R, E = s->E, Z = s->Z;
    int rc = RTX_OK;
#define SV_ALLOC(p, n) do { rc = sv_alloc(s, &(p), (n)); if (rc) { rtx_svae_destroy(s); return rc; } } while (0)
    SV_ALLOC(s->X, T * E); SV_ALLOC(s->GI, T * 3 * R); SV_ALLOC(s->Hout, T * R); SV_ALLOC(s->Hprev, T * R);
    SV_ALLOC(s->Gr, T * R); SV_ALLOC(s->Gz, T * R); SV_ALLOC(s->Gn, T * R); SV_ALLOC(s->Ghn, T * R);
    SV_ALLOC(s->mu, T * Z); SV_ALLOC(s->lv, T * Z); SV_ALLOC(s->eps, T * Z); SV_ALLOC(s->zl, T * Z); SV_ALLOC(s->dz, T * Z);
    SV_ALLOC(s->dH, T * R); SV_ALLOC(s->dGI, T * 3 * R); SV_ALLOC(s->dGH, T * 3 * R); SV_ALLOC(s->dX, T * E);
    SV_ALLOC(s->row_loss, T); SV_ALLOC(s->kl_rows, T);
    SV_ALLOC(s->WhhT, 3 * R * R);
    {
        size_t widest = std::max((size_t)std::max(R, E), Z);
        for (auto& l : s->L) widest = std::max(widest, (size_t)std::min(l.in, l.out));
        s->part_elems = 16 * T * widest;   // up to 16 K-splits of the widest [T, hidden] product
        SV_ALLOC(s->part, s->part_elems);
        SV_ALLOC(s->part2, s->part_elems);
    }
    for (auto& l : s->L) { SV_ALLOC(l.A, T * l.out); SV_ALLOC(l.D, T * l.out); }
#undef SV_ALLOC
    if (3 * R <= 1024) {
        // LDS of the weight-resident forward recurrence: h [Rp + 4] | gp [6R] | wlA [CA][1024][4] | wlB [CB][NE][4]
        const size_t Kh = (((R + 1) / 2) + 3) & ~(size_t)3, HR = 6 * R, NE = HR > 1024 ? HR - 1024 : 0;
        const size_t CA = Kh > SV_GRU_KR ? (Kh - SV_GRU_KR + 3) / 4 : 0, CB = (Kh + 3) / 4;
        const size_t lds = sizeof(float) * ((size_t)sv_gru_fwd_hs((int)R, (int)Kh, SV_GRU_KR) + ((HR + 3) & ~(size_t)3) + CA * 1024 * 4 + CB * NE * 4);
        if (Kh <= R && lds <= 160 * 1024 &&
            hipFuncSetAttribute((const void*)k_sv_gru_fwd_all<SV_GRU_KR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess) {
            s->gru_fwd_lds = lds;
            s->gru_kh = (int)Kh;
        }
    }
    if (3 * R <= 1024 && R <= SV_GRU_KR2 + 4 * 64) {
        // the 512-thread whole-row form: rows beyond the 512th must find their two halves a thread each (2 NE <= 512)
        const size_t lds = sv_gru_rows_lds((int)R, SV_GRU_KR2);
        const long ne = 3 * (long)R > 512 ? 3 * (long)R - 512 : 0;
        const char* off = getenv("RTX_SVAE_GRU_ROWS");
        s->opt_gru_rows = !(off && off[0] == '0');
        if (2 * ne <= 512 && lds <= 160 * 1024 &&
            hipFuncSetAttribute((const void*)k_sv_gru_fwd_rows<SV_GRU_KR2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess)
            s->gru_rows_lds = lds;
    }
    if (R > 128 && R <= 8 * SV_KS_SL && 3 * R <= 64 * SV_KS_NR) {   // (narrower GRUs: the whole-row form wastes less)
        const char* off = getenv("RTX_SVAE_GRU_KS");
        const size_t lds = sv_gru_ks_lds();
        if (!(off && off[0] == '0') && lds <= 160 * 1024 &&
            hipFuncSetAttribute((const void*)k_sv_gru_fwd_ks<SV_KS_SL, SV_KS_NR, SV_KS_KG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess)
            s->gru_ks_lds = lds;
        const char* offb = getenv("RTX_SVAE_GRU_BWD_KS");
        const size_t ldsb = sizeof(float) * (64 * ((SV_KS_NR + 3) & ~3) + 8 * 8 * SV_KS_SL + (size_t)((SV_KS_NR * SV_KS_SL - SV_KS_KG + 3) / 4) * 512 * 4);
        if (!(offb && offb[0] == '0') && ldsb <= 160 * 1024 &&
            hipFuncSetAttribute((const void*)k_sv_gru_bwd_ks<SV_KS_SL, SV_KS_NR, SV_KS_KG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb) == hipSuccess)
            s->gru_bwd_ks_lds = ldsb;
    }
    if (R <= 1024) {
        // LDS of the weight-resident backward recurrence: dh [Rp] | dgh [NC * RP] | part [NC * R] | wl [CL][1024][4]
        const size_t NC = std::max<size_t>(1, std::min<size_t>(16, 1024 / R)), RP = (((3 * R + NC - 1) / NC) + 3) & ~(size_t)3;
        const size_t CL = RP > SV_GRU_KRB ? (RP - SV_GRU_KRB + 3) / 4 : 0;
        const size_t lds = sizeof(float) * (((R + 3) & ~(size_t)3) + NC * RP + SV_GRU_KRB + 4 + ((NC * R + 3) & ~(size_t)3) + CL * 1024 * 4);
        if (lds <= 160 * 1024 &&
            hipFuncSetAttribute((const void*)k_sv_gru_bwd_all<SV_GRU_KRB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess) {
            s->gru_bwd_lds = lds;
            s->gru_nc = (int)NC;
            s->gru_rp = (int)RP;
        }
    }
    const size_t lds_bwd = sizeof(float) * (4 * R + 16 * R);
    if (hipFuncSetAttribute((const void*)k_sv_gru_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bwd) != hipSuccess) {
        rtx_set_error("svae_create: cannot reserve %zu bytes of LDS for the GRU backward kernel", lds_bwd);
        rtx_svae_destroy(s);
        return RTX_EHIP;
    }
    if (hipStreamCreateWithFlags(&s->side, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&s->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&s->ev_join, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&s->ev_fork2, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&s->ev_join2, hipEventDisableTiming) != hipSuccess) {
        rtx_set_error("svae_create: cannot create the side stream");
        rtx_svae_destroy(s);
        return RTX_EHIP;
    }
    *out = s;
    return RTX_OK;
}

int rtx_svae_destroy(rtx_svae* s)
{
    if (!s) return RTX_OK;
    if (s->side) { (void)hipStreamSynchronize(s->side); (void)hipStreamDestroy(s->side); }
    if (s->ev_fork) (void)hipEventDestroy(s->ev_fork);
    if (s->ev_join) (void)hipEventDestroy(s->ev_join);
    if (s->ev_fork2) (void)hipEventDestroy(s->ev_fork2);
    if (s->ev_join2) (void)hipEventDestroy(s->ev_join2);
    for (void* p : s->allocs) (void)hipFree(p);
    if (s->loss_mailbox) { (void)hipDeviceSynchronize(); (void)hipHostFree(s->loss_mailbox); }
    delete s;
    return RTX_OK;
}

// The step's loss without draining the stream (ABI 8; the engine's rtx_engine_loss_mailbox / rtx_engine_wait_loss for the sequence model):
// `SVAE.train_batch` ends in `return loss.item()` (reference models.py:835), which made the host wait for the WHOLE step -- the loss is
// final before the backward recurrence starts -- and the GPU then idle while the host prepared the next user (60 us of 843 per user).
int rtx_svae_loss_mailbox(rtx_svae* s, int32_t enable)
{
    RTX_CHECK(s, RTX_EINVAL, "svae is NULL");
    if (enable && !s->loss_mailbox) {
        void* p = nullptr;
        RTX_HIP(hipHostMalloc(&p, 64, hipHostMallocCoherent | hipHostMallocMapped));
        memset(p, 0, 64);
        s->loss_mailbox = (uint32_t*)p;
        s->loss_ticket = 0;
    } else if (!enable && s->loss_mailbox) {
        RTX_HIP(hipDeviceSynchronize());
        (void)hipHostFree(s->loss_mailbox);
        s->loss_mailbox = nullptr;
    }
    return RTX_OK;
}

int rtx_svae_wait_loss(rtx_svae* s, float* loss_host, double timeout_s)
{
    RTX_CHECK(s && loss_host, RTX_EINVAL, "svae_wait_loss: NULL argument");
    RTX_CHECK(s->loss_mailbox && s->loss_ticket != 0, RTX_ESTATE, "svae_wait_loss: no training step has reported to the mailbox (rtx_svae_loss_mailbox(s, 1) first)");
    volatile uint32_t* mb = s->loss_mailbox;
    const uint32_t want = s->loss_ticket;       // the LAST step enqueued: steps are waited for in order
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spin = 0;; ++spin) {
        if (__atomic_load_n(&mb[1], __ATOMIC_ACQUIRE) == want) break;
        if ((spin & 0x3ff) == 0x3ff) {
            const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            RTX_CHECK(el < (timeout_s > 0 ? timeout_s : 60.0), RTX_EHIP, "svae_wait_loss: the step did not report its loss within %.1f s (ticket %u of %u)", el,
                      (unsigned)mb[1], (unsigned)want);
            if (el > 0.002) sched_yield();
        }
    }
    const uint32_t bits = __atomic_load_n(&mb[0], __ATOMIC_RELAXED);
    memcpy(loss_host, &bits, 4);
    return RTX_OK;
}

int32_t rtx_svae_n_tensors(const rtx_svae* s) { return s ? s->n_tensors : 0; }

int rtx_svae_set_option(rtx_svae* s, const char* key, int32_t value)
{
    RTX_CHECK(s && key, RTX_EINVAL, "svae_set_option: NULL argument");
    if (!strcmp(key, "gemm_bf16")) s->opt_gemm_bf16 = value != 0;
    else {
        rtx_set_error("svae_set_option: unknown key '%s' (gemm_bf16)", key);
        return RTX_EINVAL;
    }
    return RTX_OK;
}

int rtx_svae_tensor_shape(const rtx_svae* s, int32_t t, int32_t* rows, int32_t* cols)
{
    RTX_CHECK(s && t >= 0 && t < s->n_tensors && rows && cols, RTX_EINVAL, "svae_tensor_shape: bad arguments");
    int r, c;
    sv_shape(s, t, &r, &c);
    *rows = r; *cols = c;
    return RTX_OK;
}

int rtx_svae_bind(rtx_svae* s, float* const* params, float* const* grads, float* const* exp_avg, float* const* exp_avg_sq)
{
    RTX_CHECK(s && params, RTX_EINVAL, "svae_bind: NULL argument");
    for (int t = 0; t < s->n_tensors; ++t) {
        RTX_CHECK(params[t], RTX_EINVAL, "svae_bind: parameter %d is NULL", t);
        s->params[t] = params[t];
    }
    s->bound = true;
    s->can_train = grads && exp_avg && exp_avg_sq;
    if (s->can_train)
        for (int t = 0; t < s->n_tensors; ++t) {
            RTX_CHECK(grads[t] && exp_avg[t] && exp_avg_sq[t], RTX_EINVAL, "svae_bind: training buffer %d is NULL", t);
            s->grads[t] = grads[t]; s->m[t] = exp_avg[t]; s->v[t] = exp_avg_sq[t];
        }
    return RTX_OK;
}

int rtx_svae_forward(rtx_svae* s, const int32_t* items, int32_t T, const float* eps_noise, uint64_t seed, uint64_t offset,
                     int32_t remove_train, float* logits_all, float* logits_last, float* mu, float* logvar, void* stream)
{
    RTX_TRY(sv_check(s, items, T, false));
    hipStream_t st = (hipStream_t)stream;
    RTX_TRY(sv_forward(s, items, T, nullptr, 1, eps_noise, seed, offset, st));
    const float* Y = s->L.back().A;
    const size_t I = s->I;
    if (logits_all) RTX_HIP(hipMemcpyAsync(logits_all, Y, sizeof(float) * T * I, hipMemcpyDeviceToDevice, st));
    if (logits_last) {
        RTX_HIP(hipMemcpyAsync(logits_last, Y + (size_t)(T - 1) * I, sizeof(float) * I, hipMemcpyDeviceToDevice, st));
        if (remove_train) hipLaunchKernelGGL(k_sv_mask_items, dim3(1), dim3(256), 0, st, items, T, logits_last);   // models.py:1633-1634
    }
    if (mu) RTX_HIP(hipMemcpyAsync(mu, s->mu, sizeof(float) * T * s->Z, hipMemcpyDeviceToDevice, st));
    if (logvar) RTX_HIP(hipMemcpyAsync(logvar, s->lv, sizeof(float) * T * s->Z, hipMemcpyDeviceToDevice, st));
    RTX_HIP(hipGetLastError());
    return RTX_OK;
}

// One optimizer step on T rows: one sequence (seq_ptr NULL; the reference's step) or n_seq packed sequences whose rows carry
// their own loss factors (nll_scale / kl_scale, device arrays of T floats).
static int sv_train(rtx_svae* s, const int32_t* items, int T, const int32_t* seq_ptr, int n_seq, const float* nll_scale, const float* kl_scale,
                    const int64_t* target_indptr, const int32_t* target_indices, const float* target_dense, const rtx_step* step, float* loss_out,
                    float* loss_accum, hipStream_t st)
{
    const int E = s->E, R = s->R, Z = s->Z, I = s->I, NL = s->NL;
    const float inv_d = step->inv_batch;                 // 1 / (number of ones in the target), models.py:1623
    const float beta_over_T = step->beta / (float)T;     // beta * mean over the time steps, models.py:1624-1625
    RTX_TRY(sv_forward(s, items, T, seq_ptr, n_seq, step->eps_noise, step->seed, step->offset, st));
    // ---- loss and dlogits
    SvLayer& last = s->L[NL - 1];
    hipLaunchKernelGGL(k_sv_loss, dim3(T), dim3(256), 0, st, last.A, T, I, target_indptr, target_indices, target_dense, inv_d, nll_scale, last.D,
                       s->row_loss);
    // ---- backward through decoder and encoder.  D of layer l = gradient w.r.t. its pre-activation.  Only the chain of
    //      input gradients is on the critical path (it feeds the GRU's backward pass) ...
    for (int li = NL - 1; li >= 0; --li) {
        SvLayer& l = s->L[li];
        // gradient w.r.t. the layer input: [T][in] = D [T][out] x W [out][in]
        if (li == 0) {
            RTX_TRY(sv_gemm(s, st, l.D, l.out, 1, s->params[0], 1, l.in, s->dH, R, T, R, l.out));
        } else if (li == s->n_enc) {
            RTX_TRY(sv_gemm(s, st, l.D, l.out, 1, s->params[2 * li], 1, l.in, s->dz, Z, T, Z, l.out));
            hipLaunchKernelGGL(k_sv_reparam_bwd, dim3(T), dim3(256), 0, st, s->dz, s->mu, s->lv, s->eps, T, Z, beta_over_T, kl_scale,
                               s->L[li - 1].D, s->kl_rows);
        } else {
            SvLayer& p = s->L[li - 1];   // tanh layer: D_prev = (D W) * (1 - A_prev^2)
            RTX_TRY(sv_gemm(s, st, l.D, l.out, 1, s->params[2 * li], 1, l.in, p.D, p.out, T, p.out, l.out, SV_EPI_TANH_GRAD, nullptr, p.A, p.out));
        }
    }
    // ... the weight gradients of the MLPs only feed Adam: on a side stream they run under the GRU's backward pass, which
    //     is ONE workgroup for ~0.85 ms while 255 compute units would otherwise idle
    RTX_HIP(hipEventRecord(s->ev_fork, st));
    RTX_HIP(hipStreamWaitEvent(s->side, s->ev_fork, 0));
    for (int li = NL - 1; li >= 0; --li) {
        SvLayer& l = s->L[li];
        const float* in;
        long ld_in;
        if (li == 0) { in = s->Hout; ld_in = R; }
        else if (li == s->n_enc) { in = s->zl; ld_in = Z; }
        else { in = s->L[li - 1].A; ld_in = s->L[li - 1].out; }
        // dW[out][in] = sum_t D[t][out] * in[t][in];  db = column sums
        RTX_TRY(sv_gemm(s, s->side, l.D, 1, l.out, in, 1, ld_in, s->grads[2 * li], l.in, l.out, l.in, T, SV_EPI_NONE, nullptr, nullptr, 0, 1));
        RTX_TRY(sv_colsum(s->side, l.D, (long)l.out, T, l.out, s->grads[2 * li + 1]));
    }
    // (the zero fill of the embedding gradient -- I x E floats -- rides along: nothing reads it before the scatter-add at the end)
    RTX_HIP(hipMemsetAsync(s->grads[sv_tail(s, SV_T_EMB)], 0, sizeof(float) * (size_t)I * E, s->side));
    RTX_HIP(hipEventRecord(s->ev_join, s->side));
    hipLaunchKernelGGL(k_sv_final_loss, dim3(1), dim3(256), 0, st, s->row_loss, s->kl_rows, T, inv_d, beta_over_T, nll_scale, kl_scale, loss_out,
                       loss_accum, s->loss_mailbox, s->loss_mailbox ? ++s->loss_ticket : 0u);
    // ---- GRU backward through time, then its weight gradients over all steps at once
    if (s->gru_bwd_ks_lds > 0)
        hipLaunchKernelGGL((k_sv_gru_bwd_ks<SV_KS_SL, SV_KS_NR, SV_KS_KG>), dim3(seq_ptr ? n_seq : 1), dim3(512), s->gru_bwd_ks_lds, st, s->dH,
                           s->params[sv_tail(s, SV_T_WHH)], seq_ptr, T, R, s->Hprev, s->Gr, s->Gz, s->Gn, s->Ghn, s->dGI, s->dGH);
    else if (s->gru_bwd_lds > 0)
        hipLaunchKernelGGL(k_sv_gru_bwd_all<SV_GRU_KRB>, dim3(seq_ptr ? n_seq : 1), dim3(1024), s->gru_bwd_lds, st, s->dH, s->params[sv_tail(s, SV_T_WHH)],
                           seq_ptr, T, R, s->gru_nc, s->gru_rp, s->Hprev, s->Gr, s->Gz, s->Gn, s->Ghn, s->dGI, s->dGH);
    else
        hipLaunchKernelGGL(k_sv_gru_bwd, dim3(seq_ptr ? n_seq : 1), dim3(1024), sizeof(float) * 20 * R, st, s->dH, s->params[sv_tail(s, SV_T_WHH)], seq_ptr,
                           T, R, s->Hprev, s->Gr, s->Gz, s->Gn, s->Ghn, s->dGI, s->dGH);
    // the hidden-side gradients (dW_hh, db_hh) go to the side stream, the input-side ones and the embedding's stay here: five
    // independent ~12-us launches become two chains of 2 and 4
    RTX_HIP(hipEventRecord(s->ev_fork2, st));
    RTX_HIP(hipStreamWaitEvent(s->side, s->ev_fork2, 0));
    RTX_TRY(sv_gemm(s, s->side, s->dGH, 1, 3 * R, s->Hprev, 1, R, s->grads[sv_tail(s, SV_T_WHH)], R, 3 * R, R, T, SV_EPI_NONE, nullptr, nullptr, 0, 1));   // dW_hh = dGH^T H_prev
    RTX_TRY(sv_colsum(s->side, s->dGH, (long)3 * R, T, 3 * R, s->grads[sv_tail(s, SV_T_BHH)]));
    RTX_HIP(hipEventRecord(s->ev_join2, s->side));
    RTX_TRY(sv_gemm(s, st, s->dGI, 1, 3 * R, s->X, 1, E, s->grads[sv_tail(s, SV_T_WIH)], E, 3 * R, E, T));          // dW_ih = dGI^T X
    RTX_TRY(sv_colsum(st, s->dGI, (long)3 * R, T, 3 * R, s->grads[sv_tail(s, SV_T_BIH)]));
    RTX_TRY(sv_gemm(s, st, s->dGI, 3 * R, 1, s->params[sv_tail(s, SV_T_WIH)], 1, E, s->dX, E, T, E, 3 * R));        // dX = dGI W_ih
    RTX_HIP(hipStreamWaitEvent(st, s->ev_join, 0));     // the MLPs' gradients and the zeroed embedding gradient (done long ago)
    hipLaunchKernelGGL(k_sv_embed_grad, dim3(T), dim3(256), 0, st, items, T, E, s->dX, s->grads[sv_tail(s, SV_T_EMB)]);
    RTX_HIP(hipGetLastError());
    RTX_HIP(hipStreamWaitEvent(st, s->ev_join2, 0));
    // ---- torch.optim.Adam (coupled weight decay 5e-3, models.py:1618-1620) over every tensor
    RtxAdamArgs a = {};
    a.n = 0;
    for (int t = 0; t < s->n_tensors; ++t) {
        RtxAdamTensor& w = a.t[a.n++];
        int r, c;
        sv_shape(s, t, &r, &c);
        w.p = s->params[t]; w.g = s->grads[t]; w.g16 = nullptr; w.m = s->m[t]; w.v = s->v[t];
        w.sh = nullptr; w.shT = nullptr; w.ld_sh = 0; w.ld_shT = 0;
        if (c == 1) { w.rows = 1; w.cols = r; } else { w.rows = r; w.cols = c; }
    }
    a.update = 1;
    const double bc1 = 1.0 - pow((double)step->beta1, (double)step->step);
    const double bc2 = 1.0 - pow((double)step->beta2, (double)step->step);
    a.step_size = (float)((double)step->lr / bc1);
    a.bc2_sqrt = (float)sqrt(bc2);
    a.beta1 = step->beta1; a.beta2 = step->beta2; a.eps = step->eps; a.weight_decay = step->weight_decay;
    a.grad_scale = 1.f; a.lam = 0.f;
    return rtx_launch_adam(a, 0, st);
}

int rtx_svae_train_step(rtx_svae* s, const int32_t* items, int32_t T, const int64_t* target_indptr, const int32_t* target_indices,
                        const float* target_dense, const rtx_step* step, float* loss_out, float* loss_accum, void* stream)
{
    RTX_TRY(sv_check(s, items, T, true));
    RTX_CHECK(step && step->step >= 1, RTX_EINVAL, "svae_train_step: step count must be >= 1");
    RTX_CHECK((target_indptr && target_indices) || target_dense, RTX_EINVAL, "svae_train_step: no target");
    return sv_train(s, items, T, nullptr, 1, nullptr, nullptr, target_indptr, target_indices, target_dense, step, loss_out, loss_accum,
                    (hipStream_t)stream);
}

// Several users per optimizer step (NOT in the reference, which takes one Adam step per user): the sequences are concatenated,
// row t belongs to the sequence whose [seq_ptr[u], seq_ptr[u + 1]) contains it, and the step minimises
//     sum_t nll_scale[t] * NLL_t + sum_t kl_scale[t] * KL_t
// -- with nll_scale = 1 / (d_user * n_seq) and kl_scale = beta / (T_user * n_seq) the mean over the pack of the reference's
// per-user loss, i.e. gradient accumulation over the pack followed by ONE Adam step.  Every product becomes a [sum T, .] GEMM;
// the recurrences run one workgroup per sequence, side by side.
int rtx_svae_train_pack(rtx_svae* s, const int32_t* items, int32_t total_steps, const int32_t* seq_ptr, int32_t n_seq, const float* nll_scale,
                        const float* kl_scale, const int64_t* target_indptr, const int32_t* target_indices, const rtx_step* step, float* loss_out,
                        float* loss_accum, void* stream)
{
    RTX_TRY(sv_check(s, items, total_steps, true));
    RTX_CHECK(step && step->step >= 1, RTX_EINVAL, "svae_train_pack: step count must be >= 1");
    RTX_CHECK(seq_ptr && n_seq >= 1 && n_seq <= total_steps, RTX_EINVAL, "svae_train_pack: %d sequences over %d steps", n_seq, total_steps);
    RTX_CHECK(nll_scale && kl_scale && target_indptr && target_indices, RTX_EINVAL, "svae_train_pack: NULL argument");
    return sv_train(s, items, total_steps, seq_ptr, n_seq, nll_scale, kl_scale, target_indptr, target_indices, nullptr, step, loss_out, loss_accum,
                    (hipStream_t)stream);
}

}  // extern "C"
