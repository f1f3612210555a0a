// kernels.hip -- the non-GEMM kernels of the Mult-VAE / Mult-DAE step for gfx950 (HBM-bound work):
// sparse-row gather (K1), activation/transposition post kernels, VAE head, multinomial loss, dlogits,
// fused multi-tensor Adam with shadow refresh, dense<->CSR conversions.  Wave = 64 lanes throughout.
#include "rtx_kernels.h"
#include <stdlib.h>

// ------------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// block-wide sum for 256-thread blocks; red must hold >= 4 floats; result broadcast to all threads
__device__ __forceinline__ float block_sum(float v, float* red)
{
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ float block_max(float v, float* red)
{
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

__device__ __forceinline__ int64_t csr_row(const RtxCsrView& v, int b) { return v.row_ids ? (int64_t)v.row_ids[b] : (int64_t)b; }

// ------------------------------------------------------------------------------------------------
// K1: gather.  One workgroup per (padded) batch row.  The row's stored entries are read coalesced
// from the CSR arrays, normalised (F.normalize), dropped out, scattered into an LDS image of a chunk of
// the dense row, and the chunk is streamed to HBM with 16-byte stores.  Column Iin of a real row is set
// to one: read K-major by the weight-gradient kernel it turns into the bias-gradient column.
// ------------------------------------------------------------------------------------------------
#define RTX_GATHER_CHUNK 4096  // elements per LDS chunk (16 KB fp32 / 8 KB bf16)

template <typename T>
__global__ __launch_bounds__(256) void k_gather(const RtxGatherArgs a)
{
    constexpr int CH = 16384 / sizeof(T);   // elements per 16-KB LDS chunk
    __shared__ __attribute__((aligned(16))) T row[CH];
    __shared__ float red[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    T* X = (T*)a.X + (size_t)b * a.ldx;
    if (b >= a.B) {  // padding row of the batch: zeros (it multiplies nothing that is kept)
        for (int i = tid * 4; i < a.ldx; i += 256 * 4) store4<T>(X + i, 0.f, 0.f, 0.f, 0.f);
        if (tid == 0) a.tsum[b] = 0.f;
        return;
    }
    const int64_t u = csr_row(a.in, b);
    const int64_t beg = a.in.indptr[u], end = a.in.indptr[u + 1];
    // ||x||_2 over the stored entries (F.normalize: x / max(||x||, 1e-12)); a conditioned row (Iin > I) is normalised
    // over its item columns only (CMultiVAE_net.encode, reference nets.py:467-471).  Implicit feedback (no value array,
    // no condition columns): the squared norm and the target sum are just the numbers of stored entries.
    const bool cond = a.Iin > a.I;
    float ss;
    if (!a.in.values && !cond) {
        ss = (float)(end - beg);
    } else {
        ss = 0.f;
        for (int64_t k = beg + tid; k < end; k += 256) {
            const float v = a.in.values ? a.in.values[k] : 1.f;
            if (!cond || a.in.indices[k] < a.I) ss += v * v;
        }
        ss = block_sum(ss, red);
    }
    const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);
    // s_b = sum of the TARGET row
    {
        const int64_t ut = csr_row(a.target, b);
        const int64_t tb = a.target.indptr[ut], te = a.target.indptr[ut + 1];
        float ts;
        if (!a.target.values && !cond) {
            ts = (float)(te - tb);
        } else {
            ts = 0.f;
            for (int64_t k = tb + tid; k < te; k += 256)
                if (!cond || a.target.indices[k] < a.I) ts += a.target.values ? a.target.values[k] : 1.f;
            ts = block_sum(ts, red);
        }
        if (tid == 0) a.tsum[b] = ts;
    }
    const bool drop = a.training && a.dropout_p > 0.f;
    const float scale = drop ? (a.dropout_p < 1.f ? 1.f / (1.f - a.dropout_p) : 0.f) : 1.f;
    for (int c0 = 0; c0 < a.ldx; c0 += CH) {
        const int cn = min(CH, a.ldx - c0);
        for (int i = tid * 4; i < cn; i += 256 * 4) store4<T>(row + i, 0.f, 0.f, 0.f, 0.f);
        __syncthreads();
        for (int64_t k = beg + tid; k < end; k += 256) {
            const int i = a.in.indices[k];
            if (i >= c0 && i < c0 + cn) {
                float v = a.in.values ? a.in.values[k] : 1.f;
                if (i < a.I) v *= inv;
                if (drop && i < a.I) {   // condition columns are concatenated after the dropout (nets.py:469-471)
                    const uint64_t e = (uint64_t)b * (uint64_t)a.I + (uint64_t)i;
                    const bool keep = a.mask ? (a.mask[e] != 0) : rtx_dropout_keep(a.seed, a.offset, e, a.dropout_p);
                    v = keep ? v * scale : 0.f;
                }
                row[i - c0] = Elem<T>::from(v);
            }
        }
        if (tid == 0 && a.Iin >= c0 && a.Iin < c0 + cn) row[a.Iin - c0] = Elem<T>::from(1.f);   // ones column -> bias gradient
        __syncthreads();
        if (sizeof(T) == 2) {
            for (int i = tid * 8; i < cn; i += 256 * 8) *(uint4*)(X + c0 + i) = *(const uint4*)(row + i);
        } else {
            for (int i = tid * 4; i < cn; i += 256 * 4) *(uint4*)(X + c0 + i) = *(const uint4*)(row + i);
        }
        __syncthreads();
    }
}

// The same batch image by SCATTER (RtxGatherArgs::written): one workgroup per row slot clears what the previous launch wrote
// there, then writes this user's stored entries.  Everything k_gather computes (norm, target sum, dropout, ones column) is
// computed the same way; only the zeros are not written again.  Reference: samplers.py:99-100 (.toarray()) + nets.py:395-399.
template <typename T>
__global__ __launch_bounds__(256) void k_gather_scatter(const RtxGatherArgs a)
{
    __shared__ float red[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    T* X = (T*)a.X + (size_t)b * a.ldx;
    int32_t* wr = a.written + (size_t)b * a.written_cap;
    const int n_old = a.n_written[b];
    for (int k = tid; k < n_old; k += 256) X[wr[k]] = Elem<T>::from(0.f);
    __syncthreads();   // (a workgroup-scope fence: the clears are ordered before this launch's writes of the same columns)
    if (b >= a.B) {    // padding row of the batch: all zero, ones column included
        if (tid == 0) { a.n_written[b] = 0; a.tsum[b] = 0.f; }
        return;
    }
    const int64_t u = csr_row(a.in, b);
    const int64_t beg = a.in.indptr[u], end = a.in.indptr[u + 1];
    const bool cond = a.Iin > a.I;
    float ss;
    if (!a.in.values && !cond) {
        ss = (float)(end - beg);
    } else {
        ss = 0.f;
        for (int64_t k = beg + tid; k < end; k += 256) {
            const float v = a.in.values ? a.in.values[k] : 1.f;
            if (!cond || a.in.indices[k] < a.I) ss += v * v;
        }
        ss = block_sum(ss, red);
    }
    const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);
    {
        const int64_t ut = csr_row(a.target, b);
        const int64_t tb = a.target.indptr[ut], te = a.target.indptr[ut + 1];
        float ts;
        if (!a.target.values && !cond) {
            ts = (float)(te - tb);
        } else {
            ts = 0.f;
            for (int64_t k = tb + tid; k < te; k += 256)
                if (!cond || a.target.indices[k] < a.I) ts += a.target.values ? a.target.values[k] : 1.f;
            ts = block_sum(ts, red);
        }
        if (tid == 0) a.tsum[b] = ts;
    }
    const bool drop = a.training && a.dropout_p > 0.f;
    const float scale = drop ? (a.dropout_p < 1.f ? 1.f / (1.f - a.dropout_p) : 0.f) : 1.f;
    const int n = (int)(end - beg);   // < written_cap (the launcher checked the matrix's longest row)
    for (int k = tid; k < n; k += 256) {
        const int i = a.in.indices[beg + k];
        float v = a.in.values ? a.in.values[beg + k] : 1.f;
        if (i < a.I) v *= inv;
        if (drop && i < a.I) {
            const uint64_t e = (uint64_t)b * (uint64_t)a.I + (uint64_t)i;
            const bool keep = a.mask ? (a.mask[e] != 0) : rtx_dropout_keep(a.seed, a.offset, e, a.dropout_p);
            v = keep ? v * scale : 0.f;
        }
        if (i < a.ldx) X[i] = Elem<T>::from(v);
        wr[k] = i < a.ldx ? i : 0;
    }
    if (tid == 0) {
        X[a.Iin] = Elem<T>::from(1.f);   // ones column -> bias gradient
        wr[n] = a.Iin;
        a.n_written[b] = n + 1;
    }
}

int rtx_launch_gather(const RtxGatherArgs& a, int is_bf16, hipStream_t stream)
{
    RTX_CHECK(a.ldx % 8 == 0, RTX_EINVAL, "gather: ldx must be a multiple of 8");
    if (a.written) {
        RTX_CHECK(a.n_written && a.in.max_row_len > 0 && a.in.max_row_len < a.written_cap && a.Iin < a.ldx, RTX_EINVAL,
                  "gather: the scatter form needs the matrix's longest row (%d) below the list capacity (%d)", a.in.max_row_len, a.written_cap);
        if (is_bf16)
            hipLaunchKernelGGL(k_gather_scatter<bf16_t>, dim3(a.Bp), dim3(256), 0, stream, a);
        else
            hipLaunchKernelGGL(k_gather_scatter<float>, dim3(a.Bp), dim3(256), 0, stream, a);
        RTX_HIP(hipGetLastError());
        return RTX_OK;
    }
    if (is_bf16)
        hipLaunchKernelGGL(k_gather<bf16_t>, dim3(a.Bp), dim3(256), 0, stream, a);
    else
        hipLaunchKernelGGL(k_gather<float>, dim3(a.Bp), dim3(256), 0, stream, a);
    RTX_HIP(hipGetLastError());
    return RTX_OK;
}

// DataSampler densify (samplers.py:99-105): rows -> float32 [B][I], ld = I (arbitrary alignment)
__global__ __launch_bounds__(256) void k_csr_to_dense(const RtxCsrView v, int I, float* out)
{
    __shared__ float row[RTX_GATHER_CHUNK];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int64_t u = csr_row(v, b);
    const int64_t beg = v.indptr[u], end = v.indptr[u + 1];
    float* o = out + (size_t)b * I;
    for (int c0 = 0; c0 < I; c0 += RTX_GATHER_CHUNK) {
        const int cn = min(RTX_GATHER_CHUNK, I - c0);
        for (int i = tid; i < cn; i += 256) row[i] = 0.f;
        __syncthreads();
        for (int64_t k = beg + tid; k < end; k += 256) {
            const int i = v.indices[k];
            if (i >= c0 && i < c0 + cn) row[i - c0] = v.values ? v.values[k] : 1.f;
        }
        __syncthreads();
        for (int i = tid; i < cn; i += 256) o[c0 + i] = row[i];
        __syncthreads();
    }
}

int rtx_launch_csr_to_dense(const RtxCsrView& v, int B, int I, float* out, hipStream_t stream)
{
    if (B <= 0) return RTX_OK;
    hipLaunchKernelGGL(k_csr_to_dense, dim3(B), dim3(256), 0, stream, v, I, out);
    RTX_HIP(hipGetLastError());
    return RTX_OK;
}

// ------------------------------------------------------------------------------------------------
// post kernels: fp32 GEMM output (split-K slabs) -> bias + tanh (forward) / x (1 - o^2) (backward), written
// row-major in the compute type.  16 x 64 tiles, one float4 per thread and slab; latency-bound, so small
// tiles = many workgroups.  (Round 1 also wrote every result transposed through LDS; the K-major operand
// reads of the weight-gradient kernel made those copies unnecessary.)
// ------------------------------------------------------------------------------------------------
// BURST: slab loads in flight per thread (16 or 32): 88 or 153 registers.  <= 16 slabs (the data-gradient product's) take the 16-deep form.
// (Built to test whether register occupancy is what makes this kernel queue beside the weight kernel: it is not -- DESIGN 4.1 -- but the
// smaller form costs nothing.)
template <typename T, int MODE, int BURST = 32>
__global__ __launch_bounds__(256) void k_post(const RtxPostArgs a)
{
    const int tid = threadIdx.x;
    const int n = blockIdx.x * 64 + (tid & 15) * 4, b = blockIdx.y * 16 + (tid >> 4);
    const float* __restrict__ C = a.C + (size_t)b * a.ldc + n;
    // slab sums, up to 32 slabs per round trip: every load of a batch is issued before the first add (the data-gradient chain
    // runs beside a streaming weight kernel, where a dependent load costs 3-5 us: 25 slabs four at a time made this kernel 32 us).
    // Same order of additions as a plain loop (masked slabs add +0).
    float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s0 = 0; s0 < a.splits; s0 += BURST) {
        float4 t[BURST];
#pragma unroll
        for (int k = 0; k < BURST; ++k) t[k] = *(const float4*)(C + (size_t)min(s0 + k, a.splits - 1) * a.slab_stride);
#pragma unroll
        for (int k = 0; k < BURST; ++k) {
            const bool on = s0 + k < a.splits;
            c.x += on ? t[k].x : 0.f; c.y += on ? t[k].y : 0.f; c.z += on ? t[k].z : 0.f; c.w += on ? t[k].w : 0.f;
        }
    }
    float v[4] = {c.x, c.y, c.z, c.w};
    if (MODE == RTX_POST_BWD && a.tanh_act) {
        const float4 o = *(const float4*)(a.O32 + (size_t)b * a.Np + n);
        v[0] *= (1.f - o.x * o.x); v[1] *= (1.f - o.y * o.y);
        v[2] *= (1.f - o.z * o.z); v[3] *= (1.f - o.w * o.w);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const bool valid = (b < a.B) && (n + e < a.N_real);
        float x = v[e];
        if (MODE == RTX_POST_FWD && valid) {
            x += a.bias[n + e];
            if (a.tanh_act) x = tanhf(x);
        }
        v[e] = valid ? x : 0.f;
    }
    if (MODE == RTX_POST_FWD && a.O32) *(float4*)(a.O32 + (size_t)b * a.Np + n) = make_float4(v[0], v[1], v[2], v[3]);
    if (a.R) {
        if (MODE == RTX_POST_FWD && a.ones_col && b < a.B && a.N_real >= n && a.N_real < n + 4) v[a.N_real - n] = 1.f;
        store4<T>((T*)a.R + (size_t)b * a.Np + n, v[0], v[1], v[2], v[3]);
    }
}

int rtx_launch_post(const RtxPostArgs& a, int mode, int is_bf16, hipStream_t stream)
{
    RTX_CHECK(a.Np % 64 == 0 && a.Bp % 16 == 0 && a.ldc % 4 == 0, RTX_EINVAL, "post: bad padding");
    const dim3 block(256), grid(a.Np / 64, a.Bp / 16);
#define RTX_P(T, M) do { if (a.splits <= 16) hipLaunchKernelGGL((k_post<T, M, 16>), grid, block, 0, stream, a); \
                         else hipLaunchKernelGGL((k_post<T, M, 32>), grid, block, 0, stream, a); } while (0)
    if (is_bf16) {
        if (mode == RTX_POST_FWD) RTX_P(bf16_t, RTX_POST_FWD);
        else RTX_P(bf16_t, RTX_POST_BWD);
    } else {
        if (mode == RTX_POST_FWD) RTX_P(float, RTX_POST_FWD);
        else RTX_P(float, RTX_POST_BWD);
    }
#undef RTX_P
    RTX_HIP(hipGetLastError());
    return RTX_OK;
}

// ------------------------------------------------------------------------------------------------
// VAE head.  forward: [mu | logvar] = C + b ; z = mu + eps * exp(logvar / 2)  (eval: z = mu)
//            backward: dmu = dz + beta*mu/B ; dlogvar = dz*eps*std/2 + beta*(exp(logvar)-1)/(2B)
// 16 x 64 tiles, 4 elements per thread; all loads of a thread are issued before the first use.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_vae_fwd(const RtxVaeFwdArgs a)
{
    const int tid = threadIdx.x;
    const int j = blockIdx.x * 64 + (tid & 63), b0 = blockIdx.y * 16;
    const float* __restrict__ C = a.C;
    // slab sums: every load of a split is issued before the first add (out-of-range threads read a valid, clamped
    // address and are masked later: a branch around the loads would serialise them)
    float m[4] = {0.f, 0.f, 0.f, 0.f}, lv[4] = {0.f, 0.f, 0.f, 0.f};
    {
        const int jc = min(j, a.Z - 1);
        const float* base[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) base[k] = C + (size_t)min(b0 + k * 4 + (tid >> 6), a.Bp - 1) * a.ldc + jc;
#pragma unroll 2
        for (int s = 0; s < a.splits; ++s) {
            float t0[4], t1[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { t0[k] = base[k][(size_t)s * a.slab_stride]; t1[k] = base[k][(size_t)s * a.slab_stride + a.Z]; }
#pragma unroll
            for (int k = 0; k < 4; ++k) { m[k] += t0[k]; lv[k] += t1[k]; }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int b = b0 + k * 4 + (tid >> 6);
        float z = 0.f;
        if (b < a.B && j < a.Z) {
            const float mm = m[k] + a.bias[j], l = lv[k] + a.bias[a.Z + j];
            float eps = 0.f;
            if (a.training)
                eps = a.eps_in ? a.eps_in[(size_t)b * a.Z + j] : rtx_normal(a.seed, a.offset, (uint64_t)b * a.Z + j);
            z = a.training ? mm + eps * expf(0.5f * l) : mm;
            const size_t o = (size_t)b * a.Z + j;
            a.mu32[o] = mm;
            a.lv32[o] = l;
            a.eps32[o] = eps;
            if (a.mu_out) a.mu_out[o] = mm;
            if (a.lv_out) a.lv_out[o] = l;
        }
        if (b < a.B && j == a.Z) z = 1.f;   // ones column -> bias gradient of the first decoder layer
        ((T*)a.Zr)[(size_t)b * a.Zp + j] = Elem<T>::from(z);
    }
}

int rtx_launch_vae_fwd(const RtxVaeFwdArgs& a, int is_bf16, hipStream_t stream)
{
    const dim3 grid(a.Zp / 64, a.Bp / 16), block(256);
    if (is_bf16)
        hipLaunchKernelGGL(k_vae_fwd<bf16_t>, grid, block, 0, stream, a);
    else
        hipLaunchKernelGGL(k_vae_fwd<float>, grid, block, 0, stream, a);
    RTX_HIP(hipGetLastError());
    return RTX_OK;
}

template <typename T>
__global__ __launch_bounds__(256) void k_vae_bwd(const RtxVaeBwdArgs a)
{
    const int tid = threadIdx.x;
    const int n = blockIdx.x * 64 + (tid & 63), b0 = blockIdx.y * 16;
    const float* __restrict__ C = a.C;
    const int j = (n < a.Z) ? n : n - a.Z;
    float dz[4] = {0.f, 0.f, 0.f, 0.f}, mu[4], lv[4], ep[4];
    {
        const int jc = min(j, a.Z - 1);
        const float* base[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int bc = min(b0 + k * 4 + (tid >> 6), a.Bp - 1);
            base[k] = C + (size_t)bc * a.ldc + jc;
            const size_t o = (size_t)min(bc, a.B - 1) * a.Z + jc;
            mu[k] = a.mu32[o]; lv[k] = a.lv32[o]; ep[k] = a.eps32[o];
        }
#pragma unroll 2
        for (int s = 0; s < a.splits; ++s) {
            float t[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) t[k] = base[k][(size_t)s * a.slab_stride];
#pragma unroll
            for (int k = 0; k < 4; ++k) dz[k] += t[k];
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int b = b0 + k * 4 + (tid >> 6);
        float d = 0.f;
        if (b < a.B && n < 2 * a.Z) {
            if (n < a.Z) {
                d = dz[k] + a.beta * mu[k] * a.inv_batch;
            } else {
                d = a.beta * 0.5f * (expf(lv[k]) - 1.f) * a.inv_batch;
                if (a.training) d += dz[k] * ep[k] * 0.5f * expf(0.5f * lv[k]);
            }
        }
        ((T*)a.D)[(size_t)b * a.Np + n] = Elem<T>::from(d);
    }
}

int rtx_launch_vae_bwd(const RtxVaeBwdArgs& a, int is_bf16, hipStream_t stream)
{
    const dim3 grid(a.Np / 64, a.Bp / 16), block(256);
    if (is_bf16)
        hipLaunchKernelGGL(k_vae_bwd<bf16_t>, grid, block, 0, stream, a);
    else
        hipLaunchKernelGGL(k_vae_bwd<float>, grid, block, 0, stream, a);
    RTX_HIP(hipGetLastError());
    return RTX_OK;
}

// ------------------------------------------------------------------------------------------------
// multinomial log-likelihood: per row  lse_b = logsumexp(Y_b) ;  row_loss_b = (s_b*lse_b - <t_b,Y_b>)/B
//  (+ beta * KL_b / B for the VAE).   One workgroup per user, online max/sum, float4 reads.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void online_add(float& m, float& s, float x)
{
    if (x > m) {
        s = s * __expf(m - x) + 1.f;
        m = x;
    } else {
        s += __expf(x - m);
    }
}
__device__ __forceinline__ void online_merge(float& m, float& s, float m2, float s2)
{
    const float mm = fmaxf(m, m2);
    if (mm == -INFINITY) { m = mm; s = 0.f; return; }
    s = s * __expf(m - mm) + s2 * __expf(m2 - mm);
    m = mm;
}

// block-wide logsumexp of row y[0..I) (y 16-byte aligned); scratch: >= 8 floats
__device__ float block_lse(const float* y, int I, float* red)
{
    const int tid = threadIdx.x;
    float m = -INFINITY, s = 0.f;
    const int I4 = I & ~3;
    for (int i = tid * 4; i < I4; i += 256 * 4) {
        const float4 t = *(const float4*)(y + i);
        online_add(m, s, t.x); online_add(m, s, t.y); online_add(m, s, t.z); online_add(m, s, t.w);
    }
    for (int i = I4 + tid; i < I; i += 256) online_add(m, s, y[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64);
        online_merge(m, s, m2, s2);
    }
    __syncthreads();
    if ((tid & 63) == 0) { red[tid >> 6] = m; red[4 + (tid >> 6)] = s; }
    __syncthreads();
    float M = red[0], S = red[4];
    online_merge(M, S, red[1], red[5]);
    online_merge(M, S, red[2], red[6]);
    online_merge(M, S, red[3], red[7]);
    return M + logf(S);
}

// `mailbox` (optional): three 32-bit words of COHERENT HOST memory, {loss, ticket, tag}: the loss and the caller's tag (the step
// count), then -- released at system scope -- the engine's ticket of this reduction (monotonic over the engine's life, so a step
// counter that restarts or repeats can never match a stale entry).  A host that wants THIS step's loss (the reference's
// `return loss.item()`) spins on the ticket word instead of draining the stream (rtx_engine_wait_loss).
__global__ __launch_bounds__(256) void k_reduce_loss(const float* row_loss, int B, float lam, const float* sumsq, int nt,
                                                     float* loss_out, float* loss_accum, uint32_t* mailbox, uint32_t seq, uint32_t tag)
{
    __shared__ float red[4];
    // fixed summation order -> bit-reproducible loss for a given batch
    float s = 0.f;
    for (int b = threadIdx.x; b < B; b += 256) s += row_loss[b];
    s = block_sum(s, red);
    if (threadIdx.x == 0) {
        if (sumsq)
            for (int t = 0; t < nt; ++t) s += lam * sqrtf(sumsq[t]);
        if (loss_out) loss_out[0] = s;
        if (loss_accum) loss_accum[0] += s;
        if (mailbox) {
            __hip_atomic_store(mailbox, __float_as_uint(s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(mailbox + 2, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(mailbox + 1, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

int rtx_launch_reduce_loss(const float* row_loss, int B, float lam, const float* sumsq, int n_tensors, float* loss_out,
                           float* loss_accum, hipStream_t stream, uint32_t* mailbox, uint32_t seq, uint32_t tag)
{
    hipLaunchKernelGGL(k_reduce_loss, dim3(1), dim3(256), 0, stream, row_loss, B, lam, sumsq, n_tensors, loss_out, loss_accum, mailbox, seq, tag);
    RTX_HIP(hipGetLastError());
    return RTX_OK;
}

// ------------------------------------------------------------------------------------------------
// Loss and its gradient w.r.t. the logits, one pass over Y.  One workgroup per (user, 4096-column chunk):
//   lse_b       = logsumexp(Y_b)            from the strip partials of the logits GEMM (or from k_row_lse)
//   D[b][i]     = (s_b * exp(Y_bi - lse_b) - t_bi) * inv_batch         (reference models.py:813-815 through autograd)
//   row_part[b][c] = -<t_b, Y_b>_chunk * inv_batch   (+ s_b * lse_b * inv_batch + beta * KL_b * inv_batch in chunk 0)
// The target row is sparse: its stored entries are scattered into an LDS image of the chunk (as k_gather does for the
// input), so the pass over Y is purely streaming: 16-byte loads of Y, 8 / 16-byte stores of D.  The loss is the
// fixed-order sum of row_part (k_reduce_loss).  Replaces round 1's k_lse_loss + dlogits post kernel (which also wrote D
// transposed) + k_target_fixup.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_row_lse(const float* Y, int ldy, int I, float* lse)
{
    __shared__ float red[8];
    const float v = block_lse(Y + (size_t)blockIdx.x * ldy, I, red);
    if (threadIdx.x == 0) lse[blockIdx.x] = v;
}

template <typename T>
__global__ __launch_bounds__(256) void k_dlogits(const RtxDlogitsArgs a)
{
    __shared__ __attribute__((aligned(16))) float timg[RTX_GATHER_CHUNK];
    __shared__ float red[8];
    const RtxLossArgs& L = a.loss;
    const int b = blockIdx.x, chunk = blockIdx.y, tid = threadIdx.x;
    const int c0 = chunk * RTX_GATHER_CHUNK;
    const int cn = min(RTX_GATHER_CHUNK, a.ldd - c0);
    T* Drow = (T*)a.D + (size_t)b * a.ldd + c0;
    if (b >= L.B) {
        for (int i = tid * 4; i < cn; i += 256 * 4) store4<T>(Drow + i, 0.f, 0.f, 0.f, 0.f);
        return;
    }
    const float* y = L.Y + (size_t)b * L.ldy + c0;
    float lse;
    if (L.part) {
        float m = -INFINITY, s = 0.f;
        for (int k = tid; k < L.n_strips; k += 256) {
            const float2 pr = L.part[(size_t)b * L.part_ld + k];
            online_merge(m, s, pr.x, pr.y);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64);
            online_merge(m, s, m2, s2);
        }
        if ((tid & 63) == 0) { red[tid >> 6] = m; red[4 + (tid >> 6)] = s; }
        __syncthreads();
        float M = red[0], S = red[4];
        online_merge(M, S, red[1], red[5]);
        online_merge(M, S, red[2], red[6]);
        online_merge(M, S, red[3], red[7]);
        lse = M + logf(S);
    } else {
        lse = L.lse[b];     // k_row_lse ran first
    }
    const float sc = L.tsum[b] * L.inv_batch;
    const int64_t u = csr_row(L.target, b);
    const int64_t tb = L.target.indptr[u], te = L.target.indptr[u + 1];
    for (int i = tid * 4; i < cn; i += 256 * 4) *(float4*)(timg + i) = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    for (int64_t k = tb + tid; k < te; k += 256) {
        const int i = L.target.indices[k];
        if (i >= c0 && i < c0 + cn && i < L.I) timg[i - c0] = L.target.values ? L.target.values[k] : 1.f;
    }
    __syncthreads();
    float dot = 0.f;
    if (sizeof(T) == 2 && a.Y16) {
        // half-precision logits, possibly in place (Y16 == D): 8 elements = 16 bytes in, 16 bytes out per thread and pass
        typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
        const _Float16* y16 = (const _Float16*)a.Y16 + (size_t)b * a.ldd + c0;
#pragma unroll 2
        for (int i = tid * 8; i < cn; i += 256 * 8) {
            const int col = c0 + i;
            const f16x8_t yy = *(const f16x8_t*)(y16 + i);
            const float4 t0 = *(const float4*)(timg + i), t1 = *(const float4*)(timg + i + 4);
            const float tv[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
            float d[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const bool valid = col + e < L.I;
                const float yv = (float)yy[e];
                d[e] = valid ? sc * __expf(yv - lse) - tv[e] * L.inv_batch : 0.f;
                if (valid) dot += tv[e] * yv;
            }
            uint4 o;
            o.x = pack_bf16x2(d[0], d[1]);
            o.y = pack_bf16x2(d[2], d[3]);
            o.z = pack_bf16x2(d[4], d[5]);
            o.w = pack_bf16x2(d[6], d[7]);
            *(uint4*)((bf16_t*)Drow + i) = o;
        }
    } else
#pragma unroll 4
    for (int i = tid * 4; i < cn; i += 256 * 4) {
        const int col = c0 + i;
        float4 yy = make_float4(0.f, 0.f, 0.f, 0.f);
        if (col < L.ldy) yy = *(const float4*)(y + i);   // ldy is a multiple of 4: a group is inside the row or past it
        const float4 tt = *(const float4*)(timg + i);
        const float yv[4] = {yy.x, yy.y, yy.z, yy.w}, tv[4] = {tt.x, tt.y, tt.z, tt.w};
        float d[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool valid = col + e < L.I;
            d[e] = valid ? sc * __expf(yv[e] - lse) - tv[e] * L.inv_batch : 0.f;
            if (valid) dot += tv[e] * yv[e];
        }
        store4<T>(Drow + i, d[0], d[1], d[2], d[3]);
    }
    dot = block_sum(dot, red);
    float kl = 0.f;
    if (chunk == 0 && L.mu32) {
        for (int j = tid; j < L.Z; j += 256) {
            const float m = L.mu32[(size_t)b * L.Z + j], lv = L.lv32[(size_t)b * L.Z + j];
            kl += 1.f + lv - m * m - expf(lv);
        }
        kl = block_sum(kl, red);
    }
    if (tid == 0) {
        float part = -dot * L.inv_batch;
        if (chunk == 0) {
            if (L.part) L.lse[b] = lse;
            part += L.tsum[b] * lse * L.inv_batch + L.beta * (-0.5f * kl) * L.inv_batch;
        }
        L.row_loss[(size_t)b * gridDim.y + chunk] = part;
    }
}

// The same, ONE WORKGROUP PER USER ROW (round 5; the training step's half-precision logits in place, D == Y16).  The chunked kernel
// above runs 5 workgroups per user; each re-merges the row's 316 strip partials, zeroes and fills a 16-KB target image in LDS, and
// the 2560 of them need 1.25 rounds of the chip: 16.4 us for 41 MB.  Here a row's logits (NV 16-byte loads per thread) leave for
// the registers in ONE burst at kernel entry and stay there; the partials are merged once per row; the target needs no dense image:
// its <= RTX_DLR_CAP stored entries are read (logit still in place), corrected and parked in LDS, the dense pass writes every element
// from the registers, and behind a barrier the corrected entries overwrite theirs.  512 workgroups, all resident, one round.
#define RTX_DLR_CAP 4096
template <int NV>
__global__ __launch_bounds__(256) void k_dlogits_row(const RtxDlogitsArgs a)
{
    typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
    __shared__ float red[8];
    __shared__ int32_t t_idx[RTX_DLR_CAP];
    __shared__ float t_val[RTX_DLR_CAP];
    const RtxLossArgs& L = a.loss;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n8 = a.ldd >> 3;
    // gridDim.y workgroups share a row: this one owns the 16-byte groups [j0, j1) (<= NV * 256 of them) = columns [8 j0, 8 j1)
    const int S = gridDim.y, part_y = blockIdx.y;
    const int j0 = (int)((long)n8 * part_y / S), j1 = (int)((long)n8 * (part_y + 1) / S);
    bf16_t* Drow = (bf16_t*)a.D + (size_t)b * a.ldd;
    if (b >= L.B) {
        for (int j = j0 + tid; j < j1; j += 256) *(uint4*)(Drow + (size_t)j * 8) = make_uint4(0u, 0u, 0u, 0u);
        return;
    }
    // (1) the row's logits: every load in flight before anything else -- behind the one load the longest dependent chain starts
    //     with (row number -> row bounds -> stored entries -> their logits: four round trips, all of them under the merge below)
    const int64_t uu = csr_row(L.target, b);
    const _Float16* y16 = (const _Float16*)a.Y16 + (size_t)b * a.ldd;
    f16x8_t yy[NV];
#pragma unroll
    for (int u = 0; u < NV; ++u) yy[u] = *(const f16x8_t*)(y16 + (size_t)min(j0 + tid + u * 256, n8 - 1) * 8);
    // (round 6) everything the tail of this kernel needs -- the row's target sum, the first 256 latent means / log-variances of the KL
    // term -- is requested here too: loaded where it is used, each was one more dependent round trip at the very end of the kernel
    const float tsum_b = L.tsum[b];
    float kl_m0 = 0.f, kl_lv0 = 0.f;
    const bool kl_here = L.mu32 && part_y == 0;
    if (kl_here && tid < L.Z) { kl_m0 = L.mu32[(size_t)b * L.Z + tid]; kl_lv0 = L.lv32[(size_t)b * L.Z + tid]; }
    float2 pr0 = make_float2(-INFINITY, 0.f), pr1 = make_float2(-INFINITY, 0.f);     // this thread's strip partials (n_strips <= 512 here)
    if (tid < L.n_strips) pr0 = L.part[(size_t)b * L.part_ld + tid];
    if (tid + 256 < L.n_strips) pr1 = L.part[(size_t)b * L.part_ld + tid + 256];
    const int64_t tb = L.target.indptr[uu], te = L.target.indptr[uu + 1];
    const int nt = (int)min((int64_t)RTX_DLR_CAP, te - tb);       // (the launcher checked the matrix's longest row)
    // the first 256 stored entries (nearly every row has fewer): index, value and logit, requested before the log-sum-exp exists
    int i0 = -1;
    float tv0 = 0.f, yv0 = 0.f;
    if (tid < nt) {
        const int i = L.target.indices[tb + tid];
        if (i < L.I && i >= 8 * j0 && i < 8 * j1) {      // (the entries of this workgroup's columns)
            i0 = i;
            tv0 = L.target.values ? L.target.values[tb + tid] : 1.f;
            yv0 = (float)y16[i];
        }
    }
    // (2) log-sum-exp of the row from the strip partials of the logits product
    float lse;
    {
        float m = pr0.x, s = pr0.y;
        online_merge(m, s, pr1.x, pr1.y);
        for (int k = tid + 512; k < L.n_strips; k += 256) {       // (rows of more than 32 768 items)
            const float2 pr = L.part[(size_t)b * L.part_ld + k];
            online_merge(m, s, pr.x, pr.y);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64);
            online_merge(m, s, m2, s2);
        }
        if ((tid & 63) == 0) { red[tid >> 6] = m; red[4 + (tid >> 6)] = s; }
        __syncthreads();
        float M = red[0], S = red[4];
        online_merge(M, S, red[1], red[5]);
        online_merge(M, S, red[2], red[6]);
        online_merge(M, S, red[3], red[7]);
        lse = M + logf(S);
        __syncthreads();          // (red is reused by the sums below)
    }
    const float sc = tsum_b * L.inv_batch;
    // (3) the target's stored entries, while their logits are still in place: <t, y>, and the corrected gradient parked in LDS
    float dot = 0.f;
    if (tid < nt) {               // the entries requested at kernel entry
        t_idx[tid] = i0;
        t_val[tid] = i0 >= 0 ? sc * __expf(yv0 - lse) - tv0 * L.inv_batch : 0.f;
        dot += tv0 * yv0;
    }
    for (int k = tid + 256; k < nt; k += 256) {
        const int i = L.target.indices[tb + k];
        int idx = -1;
        float d = 0.f;
        if (i < L.I && i >= 8 * j0 && i < 8 * j1) {
            const float tv = L.target.values ? L.target.values[tb + k] : 1.f;
            const float yv = (float)y16[i];
            dot += tv * yv;
            d = sc * __expf(yv - lse) - tv * L.inv_batch;
            idx = i;
        }
        t_idx[k] = idx;
        t_val[k] = d;
    }
    __syncthreads();              // every target logit of these columns has been read: they may be overwritten now
    // (4) the dense pass, from the registers: 16 bytes out per thread and load
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int j = j0 + tid + u * 256;
        if (j < j1) {
            const int col = j * 8;
            float d[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) d[e] = (col + e < L.I) ? sc * __expf((float)yy[u][e] - lse) : 0.f;
            uint4 o;
            o.x = pack_bf16x2(d[0], d[1]);
            o.y = pack_bf16x2(d[2], d[3]);
            o.z = pack_bf16x2(d[4], d[5]);
            o.w = pack_bf16x2(d[6], d[7]);
            *(uint4*)(Drow + (size_t)col) = o;
        }
    }
    // (5) the stored entries' values replace what the dense pass wrote there (same workgroup: release, barrier, then the scatter)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    for (int k = tid; k < nt; k += 256)
        if (t_idx[k] >= 0) Drow[t_idx[k]] = f32_to_bf16(t_val[k]);
    // (6) row loss: -<t, y> / B + s lse / B + beta KL / B, all in partial 0 of the row (the others are zero)
    dot = block_sum(dot, red);
    float kl = 0.f;
    if (kl_here) {
        if (tid < L.Z) kl += 1.f + kl_lv0 - kl_m0 * kl_m0 - expf(kl_lv0);
        for (int j = tid + 256; j < L.Z; j += 256) {
            const float m = L.mu32[(size_t)b * L.Z + j], lv = L.lv32[(size_t)b * L.Z + j];
            kl += 1.f + lv - m * m - expf(lv);
        }
        kl = block_sum(kl, red);
    }
    // partial `part_y` of the row: this workgroup's share of -<t, y> / B; partial 0 also carries s lse / B + beta KL / B; the row's
    // remaining partials (the chunked kernel writes rtx_dlogits_chunks of them, and the loss sum reads them all) are zero
    const int chunks = (a.ldd + RTX_GATHER_CHUNK - 1) / RTX_GATHER_CHUNK;
    if (tid == 0) {
        float part = -dot * L.inv_batch;
        if (part_y == 0) {
            L.lse[b] = lse;
            part += tsum_b * lse * L.inv_batch + L.beta * (-0.5f * kl) * L.inv_batch;
        }
        L.row_loss[(size_t)b * chunks + part_y] = part;
    } else if (part_y == 0 && tid >= S && tid < chunks) {
        L.row_loss[(size_t)b * chunks + tid] = 0.f;
    }
}

int rtx_dlogits_chunks(int ldd) { return (ldd + RTX_GATHER_CHUNK - 1) / RTX_GATHER_CHUNK; }

// a.loss.row_loss receives B * rtx_dlogits_chunks(a.ldd) partial sums (row-major [B][chunks]): sum them with
// rtx_launch_reduce_loss(row_loss, B * chunks, ...)
int rtx_launch_dlogits(const RtxDlogitsArgs& a, int is_bf16, hipStream_t stream)
{
    if (a.Bp <= 0) return RTX_OK;
    RTX_CHECK(a.loss.ldy % 4 == 0 && a.ldd % 8 == 0 && a.ldd >= a.loss.I, RTX_EINVAL, "dlogits: bad leading dimensions");
    RTX_CHECK(!a.Y16 || (is_bf16 && a.loss.part && (((uintptr_t)a.Y16 | (uintptr_t)a.D) & 15) == 0), RTX_EINVAL,
              "dlogits: half-precision logits need bf16 deltas, the log-sum-exp partials of the logits product and 16-byte aligned images");
    if (!a.loss.part && a.loss.B > 0) {
        hipLaunchKernelGGL(k_row_lse, dim3(a.loss.B), dim3(256), 0, stream, a.loss.Y, a.loss.ldy, a.loss.I, a.loss.lse);
        RTX_HIP(hipGetLastError());
    }
    const dim3 grid(a.Bp, rtx_dlogits_chunks(a.ldd));
    // the training step's in-place half logits: one workgroup per row (k_dlogits_row) when the row fits its registers (<= 20 480
    // columns) and the target matrix's longest row its LDS list; RTX_DLOGITS_ROW=0 keeps the chunked kernel (A/B)
    static const bool row_kernel = [] { const char* v = getenv("RTX_DLOGITS_ROW"); return !(v && v[0] == '0'); }();
    if (is_bf16 && a.Y16 && a.loss.part && row_kernel && a.ldd <= 20 * 1024 && a.loss.target.max_row_len > 0 &&
        a.loss.target.max_row_len <= RTX_DLR_CAP && rtx_dlogits_chunks(a.ldd) <= 256) {
        // NV = 5 loads x 256 threads x 8 columns = 10 240 columns per workgroup: two workgroups share a longer row
        static const int s_env = [] { const char* v = getenv("RTX_DLOGITS_SPLIT"); return v ? atoi(v) : 0; }();   // (measurement: more workgroups per row)
        const int S = std::min(rtx_dlogits_chunks(a.ldd), std::max((a.ldd + 10239) / 10240, s_env));
        hipLaunchKernelGGL(k_dlogits_row<5>, dim3(a.Bp, S), dim3(256), 0, stream, a);
        RTX_HIP(hipGetLastError());
        return RTX_OK;
    }
    if (is_bf16)
        hipLaunchKernelGGL(k_dlogits<bf16_t>, grid, dim3(256), 0, stream, a);
    else
        hipLaunchKernelGGL(k_dlogits<float>, grid, dim3(256), 0, stream, a);
    RTX_HIP(hipGetLastError());
    return RTX_OK;
}

// n_items bounds the masked columns: a conditioned input row carries its condition columns after the items
// (CMultiVAE.predict masks x[:, :-cond_dim].nonzero() only, reference models.py:952-953)
__global__ __launch_bounds__(256) void k_neg_inf(const RtxCsrView v, float* logits, long ld, int n_items)
{
    const int b = blockIdx.x;
    const int64_t u = csr_row(v, b);
    for (int64_t k = v.indptr[u] + threadIdx.x; k < v.indptr[u + 1]; k += 256) {
        const float val = v.values ? v.values[k] : 1.f;
        const int i = v.indices[k];
        if (val != 0.f && i < n_items) logits[(size_t)b * ld + i] = -INFINITY;
    }
}

int rtx_launch_neg_inf(const RtxCsrView& in, int B, float* logits, long ld, int n_items, hipStream_t stream)
{
    if (B <= 0) return RTX_OK;
    hipLaunchKernelGGL(k_neg_inf, dim3(B), dim3(256), 0, stream, in, logits, ld, n_items);
    RTX_HIP(hipGetLastError());
    return RTX_OK;
}

// loss_function(recon_x, x, mu, logvar, beta) on dense tensors (reference models.py:813-815)
__global__ __launch_bounds__(256) void k_dense_loss(const float* Y, const float* X, int I, const float* mu, const float* lv, int Z,
                                                    float beta, float inv_batch, float* row_loss)
{
    __shared__ float red[8];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* y = Y + (size_t)b * I;
    const float* x = X + (size_t)b * I;
    // rows of an [B][I] tensor are not 16-byte aligned in general: scalar online pass
    float m = -INFINITY, s = 0.f, dot = 0.f, sx = 0.f;
    for (int i = tid; i < I; i += 256) {
        const float yi = y[i], xi = x[i];
        online_add(m, s, yi);
        dot += xi * yi;
        sx += xi;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64);
        online_merge(m, s, m2, s2);
    }
    __syncthreads();
    if ((tid & 63) == 0) { red[tid >> 6] = m; red[4 + (tid >> 6)] = s; }
    __syncthreads();
    float M = red[0], S = red[4];
    online_merge(M, S, red[1], red[5]);
    online_merge(M, S, red[2], red[6]);
    online_merge(M, S, red[3], red[7]);
    const float lse = M + logf(S);
    dot = block_sum(dot, red);
    sx = block_sum(sx, red);
    float kl = 0.f;
    if (mu) {
        for (int j = tid; j < Z; j += 256) {
            const float mm = mu[(size_t)b * Z + j], l = lv[(size_t)b * Z + j];
            kl += 1.f + l - mm * mm - expf(l);
        }
        kl = block_sum(kl, red);
    }
    if (tid == 0) row_loss[b] = (sx * lse - dot) * inv_batch + beta * (-0.5f * kl) * inv_batch;
}

int rtx_launch_dense_loss(const float* Y, const float* X, int B, int I, const float* mu, const float* lv, int Z, float beta,
                          float inv_batch, float* row_loss, hipStream_t stream)
{
    if (B <= 0) return RTX_OK;
    hipLaunchKernelGGL(k_dense_loss, dim3(B), dim3(256), 0, stream, Y, X, I, mu, lv, Z, beta, inv_batch, row_loss);
    RTX_HIP(hipGetLastError());
    return RTX_OK;
}

// ------------------------------------------------------------------------------------------------
// dense [B][I] float32 -> CSR (stored entries = non-zeros, column order preserved)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_dense_count(const float* X, int I, int32_t* counts)
{
    __shared__ float red[4];
    const int b = blockIdx.x;
    float c = 0.f;
    for (int i = threadIdx.x; i < I; i += 256) c += (X[(size_t)b * I + i] != 0.f) ? 1.f : 0.f;
    c = block_sum(c, red);
    if (threadIdx.x == 0) counts[b] = (int32_t)c;
}

__global__ void k_scan_counts(const int32_t* counts, int B, int64_t* indptr)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        int64_t acc = 0;
        indptr[0] = 0;
        for (int b = 0; b < B; ++b) {
            acc += counts[b];
            indptr[b + 1] = acc;
        }
    }
}

__global__ __launch_bounds__(256) void k_dense_fill(const float* X, int I, const int64_t* indptr, int32_t* indices, float* values)
{
    __shared__ int wave_cnt[4];
    __shared__ int base_sh;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) base_sh = 0;
    __syncthreads();
    const int64_t out0 = indptr[b];
    for (int c0 = 0; c0 < I; c0 += 256) {
        const int i = c0 + tid;
        const float v = (i < I) ? X[(size_t)b * I + i] : 0.f;
        const bool nz = v != 0.f;
        const unsigned long long bal = __ballot(nz);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wave_cnt[wave] = __popcll(bal);
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wave; ++w) woff += wave_cnt[w];
        const int base = base_sh;
        if (nz) {
            const int64_t o = out0 + base + woff + before;
            indices[o] = i;
            values[o] = v;
        }
        __syncthreads();
        if (tid == 0) base_sh = base + wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        __syncthreads();
    }
}

int rtx_launch_dense_count(const float* X, int B, int I, int32_t* counts, hipStream_t stream)
{
    if (B <= 0) return RTX_OK;
    hipLaunchKernelGGL(k_dense_count, dim3(B), dim3(256), 0, stream, X, I, counts);
    RTX_HIP(hipGetLastError());
    return RTX_OK;
}
int rtx_launch_scan_counts(const int32_t* counts, int B, int64_t* indptr, hipStream_t stream)
{
    hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(64), 0, stream, counts, B, indptr);
    RTX_HIP(hipGetLastError());
    return RTX_OK;
}
int rtx_launch_dense_fill(const float* X, int B, int I, const int64_t* indptr, int32_t* indices, float* values, hipStream_t stream)
{
    if (B <= 0) return RTX_OK;
    hipLaunchKernelGGL(k_dense_fill, dim3(B), dim3(256), 0, stream, X, I, indptr, indices, values);
    RTX_HIP(hipGetLastError());
    return RTX_OK;
}

// ------------------------------------------------------------------------------------------------
// fused multi-tensor Adam (torch.optim.Adam, amsgrad off, coupled weight decay) + refresh of the
// compute-precision shadows in both orientations.  One launch for all tensors: 64x64 tiles.
//   HBM per parameter: read p,g,m,v (16 B) + write p,m,v (12 B) + shadows.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_adam(const RtxAdamArgs a)
{
    __shared__ float tile[64][65];
    const int tid = threadIdx.x;
    // XCD-aware order: workgroup b runs on XCD b % 8; give every XCD one CONTIGUOUS run of tiles so that the
    // 128-byte lines straddling two neighbouring column tiles (rows are not line-aligned: 2400-B and 80432-B
    // strides) are re-read from that XCD's L2 instead of being fetched from HBM by two different L2s
    // (PMC: FETCH_SIZE was 1.37x the algorithmic read bytes with the plain order).
    const int per_xcd = (a.total_tiles + 7) / 8;
    const int tile_id = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (tile_id >= a.total_tiles) return;
    int ti = 0;
#pragma unroll 1
    for (int k = 1; k < a.n; ++k)
        if (tile_id >= a.t[k].tile_start) ti = k;
    const RtxAdamTensor& t = a.t[ti];
    const int local = tile_id - t.tile_start;
    float reg = 0.f;
    if (a.lam != 0.f && t.sumsq) {
        const float nrm = sqrtf(*t.sumsq);
        reg = nrm > 0.f ? a.lam / nrm : 0.f;
    }
    if (t.flat) {
        // Rows whose length is not a multiple of 4 floats (n_items = 17 769 of the Netflix shape) start at every 16-byte
        // phase, so the 2-D tiles fall back to 4-byte accesses (measured: 2x the time of the whole launch).  Without a
        // transposed copy to produce, the tensor is walked as ONE contiguous array instead: float4 everywhere, and the
        // compute copy gets its (row, column) back from the flat index.
        const long n = (long)t.rows * t.cols;
        if ((long)local * 4096 + 4096 <= n) {
            // a whole 4096-element tile (all but the tensor's last): EVERY load of the four passes is issued before the first update is
            // computed -- one round trip per tile, as the 2-D walk below does.  Round 6: this walk now serves every tensor without a
            // transposed copy, also rows of whole float4s.  The state-stream micro-benchmark (tests/native/test_gemm.cpp "streams")
            // measures the same six streams at 5.9-6.2 TB/s walked flat and at 3.8-4.7 TB/s as 64 x 128 tiles (64 x 64 here: worse).
            float4 P4[4], M4[4], V4[4], G4[4];
            const long f0 = (long)local * 4096 + tid * 4;
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                const long f = f0 + pass * 1024;
                P4[pass] = *(const float4*)(t.p + f);
                if (a.update) {
                    M4[pass] = *(const float4*)(t.m + f);
                    V4[pass] = *(const float4*)(t.v + f);
                    if (t.g16) {
                        const uint2 u = *(const uint2*)(t.g16 + f);
                        G4[pass] = make_float4(bf16_to_f32((bf16_t)(u.x & 0xffff)), bf16_to_f32((bf16_t)(u.x >> 16)),
                                               bf16_to_f32((bf16_t)(u.y & 0xffff)), bf16_to_f32((bf16_t)(u.y >> 16)));
                    } else {
                        G4[pass] = *(const float4*)(t.g + f);
                    }
                }
            }
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                const long f = f0 + pass * 1024;
                float pv[4] = {P4[pass].x, P4[pass].y, P4[pass].z, P4[pass].w};
                if (a.update) {
                    const float gv[4] = {G4[pass].x, G4[pass].y, G4[pass].z, G4[pass].w};
                    float mv[4] = {M4[pass].x, M4[pass].y, M4[pass].z, M4[pass].w};
                    float vv[4] = {V4[pass].x, V4[pass].y, V4[pass].z, V4[pass].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float g = gv[e] * a.grad_scale + reg * pv[e];
                        if (a.weight_decay != 0.f) g += a.weight_decay * pv[e];
                        const float m = mv[e] + (g - mv[e]) * (1.f - a.beta1);
                        const float v = vv[e] * a.beta2 + (1.f - a.beta2) * g * g;
                        const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
                        pv[e] = pv[e] - a.step_size * (m / denom);
                        mv[e] = m;
                        vv[e] = v;
                    }
                    *(float4*)(t.p + f) = make_float4(pv[0], pv[1], pv[2], pv[3]);
                    *(float4*)(t.m + f) = make_float4(mv[0], mv[1], mv[2], mv[3]);
                    *(float4*)(t.v + f) = make_float4(vv[0], vv[1], vv[2], vv[3]);
                }
                if (t.sh) {
                    int r = (int)(f / t.cols), c = (int)(f - (long)r * t.cols);
                    if ((t.cols & 3) == 0) {      // the four elements share a row, ld_sh is a multiple of 128: one 8- / 16-byte store
                        store4<T>((T*)t.sh + (size_t)r * t.ld_sh + c, pv[0], pv[1], pv[2], pv[3]);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            ((T*)t.sh)[(size_t)r * t.ld_sh + c] = Elem<T>::from(pv[e]);
                            if (++c == t.cols) { c = 0; ++r; }
                        }
                    }
                }
            }
            return;
        }
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const long f = (long)local * 4096 + pass * 1024 + tid * 4;
            if (f >= n) continue;
            const int nv = (int)min((long)4, n - f);
            float pv[4] = {0.f, 0.f, 0.f, 0.f}, gv[4] = {0.f, 0.f, 0.f, 0.f}, mv[4] = {0.f, 0.f, 0.f, 0.f}, vv[4] = {0.f, 0.f, 0.f, 0.f};
            if (nv == 4) {
                const float4 p4 = *(const float4*)(t.p + f);
                pv[0] = p4.x; pv[1] = p4.y; pv[2] = p4.z; pv[3] = p4.w;
                if (a.update) {
                    const float4 m4 = *(const float4*)(t.m + f), v4 = *(const float4*)(t.v + f);
                    if (t.g16) {
                        const uint2 u = *(const uint2*)(t.g16 + f);
                        gv[0] = bf16_to_f32((bf16_t)(u.x & 0xffff)); gv[1] = bf16_to_f32((bf16_t)(u.x >> 16));
                        gv[2] = bf16_to_f32((bf16_t)(u.y & 0xffff)); gv[3] = bf16_to_f32((bf16_t)(u.y >> 16));
                    } else {
                        const float4 g4 = *(const float4*)(t.g + f);
                        gv[0] = g4.x; gv[1] = g4.y; gv[2] = g4.z; gv[3] = g4.w;
                    }
                    mv[0] = m4.x; mv[1] = m4.y; mv[2] = m4.z; mv[3] = m4.w;
                    vv[0] = v4.x; vv[1] = v4.y; vv[2] = v4.z; vv[3] = v4.w;
                }
            } else {
                for (int e = 0; e < nv; ++e) {
                    pv[e] = t.p[f + e];
                    if (a.update) { gv[e] = t.g16 ? bf16_to_f32(t.g16[f + e]) : t.g[f + e]; mv[e] = t.m[f + e]; vv[e] = t.v[f + e]; }
                }
            }
            if (a.update) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (e < nv) {
                        float g = gv[e] * a.grad_scale + reg * pv[e];
                        if (a.weight_decay != 0.f) g += a.weight_decay * pv[e];
                        const float m = mv[e] + (g - mv[e]) * (1.f - a.beta1);
                        const float v = vv[e] * a.beta2 + (1.f - a.beta2) * g * g;
                        const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
                        pv[e] = pv[e] - a.step_size * (m / denom);
                        mv[e] = m;
                        vv[e] = v;
                    }
                }
                if (nv == 4) {
                    *(float4*)(t.p + f) = make_float4(pv[0], pv[1], pv[2], pv[3]);
                    *(float4*)(t.m + f) = make_float4(mv[0], mv[1], mv[2], mv[3]);
                    *(float4*)(t.v + f) = make_float4(vv[0], vv[1], vv[2], vv[3]);
                } else {
                    for (int e = 0; e < nv; ++e) { t.p[f + e] = pv[e]; t.m[f + e] = mv[e]; t.v[f + e] = vv[e]; }
                }
            }
            if (t.sh) {
                int r = (int)(f / t.cols), c = (int)(f - (long)r * t.cols);
                for (int e = 0; e < nv; ++e) {
                    ((T*)t.sh)[(size_t)r * t.ld_sh + c] = Elem<T>::from(pv[e]);
                    if (++c == t.cols) { c = 0; ++r; }
                }
            }
        }
        return;
    }
    const int tiles_c = (t.cols + 63) / 64;
    const int r0 = (local / tiles_c) * 64, c0 = (local % tiles_c) * 64;
    const int cl = (tid & 15) * 4;
    const bool vec = (t.cols & 3) == 0;
    if (vec && t.cols >= 4) {
        // Rows of whole float4s: EVERY load of the thread's four passes (p, m, v, g: 16 x 16 B) is issued before the first
        // update is computed -- one round trip per tile instead of four (the passes' stores may alias the next pass's loads as
        // far as the compiler knows, so written pass by pass each pass waits for the one before).  Out-of-range threads load a
        // clamped (valid) address and skip the stores: no branch around a load.  A rank of the sharded optimizer runs this on
        // 1/8 of the rows, where the launch is all latency: 35 -> ~15 us in the data-parallel step (profiles/r3_adam_probe.txt).
        float4 P4[4], M4[4], V4[4], G4[4];
        const int cc = min(c0 + cl, t.cols - 4);
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int rc = min(r0 + pass * 16 + (tid >> 4), t.rows - 1);
            const size_t o = (size_t)rc * t.cols + cc;
            P4[pass] = *(const float4*)(t.p + o);
            if (a.update) {
                M4[pass] = *(const float4*)(t.m + o);
                V4[pass] = *(const float4*)(t.v + o);
                if (t.g16) {   // data parallel, bf16 exchange: the reduced gradient arrives as bf16
                    const uint2 u = *(const uint2*)(t.g16 + o);
                    G4[pass] = make_float4(bf16_to_f32((bf16_t)(u.x & 0xffff)), bf16_to_f32((bf16_t)(u.x >> 16)),
                                           bf16_to_f32((bf16_t)(u.y & 0xffff)), bf16_to_f32((bf16_t)(u.y >> 16)));
                } else {
                    G4[pass] = *(const float4*)(t.g + o);
                }
            }
        }
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int rl = pass * 16 + (tid >> 4);
            const int r = r0 + rl, c = c0 + cl;
            const bool ok = r < t.rows && c < t.cols;
            float pv[4] = {P4[pass].x, P4[pass].y, P4[pass].z, P4[pass].w};
            if (ok) {
                const size_t o = (size_t)r * t.cols + c;
                if (a.update) {
                    const float gv[4] = {G4[pass].x, G4[pass].y, G4[pass].z, G4[pass].w};
                    float mv[4] = {M4[pass].x, M4[pass].y, M4[pass].z, M4[pass].w};
                    float vv[4] = {V4[pass].x, V4[pass].y, V4[pass].z, V4[pass].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float g = gv[e] * a.grad_scale + reg * pv[e];
                        if (a.weight_decay != 0.f) g += a.weight_decay * pv[e];
                        const float m = mv[e] + (g - mv[e]) * (1.f - a.beta1);
                        const float v = vv[e] * a.beta2 + (1.f - a.beta2) * g * g;
                        const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
                        pv[e] = pv[e] - a.step_size * (m / denom);
                        mv[e] = m;
                        vv[e] = v;
                    }
                    *(float4*)(t.p + o) = make_float4(pv[0], pv[1], pv[2], pv[3]);
                    *(float4*)(t.m + o) = make_float4(mv[0], mv[1], mv[2], mv[3]);
                    *(float4*)(t.v + o) = make_float4(vv[0], vv[1], vv[2], vv[3]);
                }
                if (t.sh) store4<T>((T*)t.sh + (size_t)r * t.ld_sh + c, pv[0], pv[1], pv[2], pv[3]);   // ld_sh is a multiple of 128 -> aligned
            }
            if (t.shT) {
#pragma unroll
                for (int e = 0; e < 4; ++e) tile[rl][cl + e] = ok ? pv[e] : 0.f;
            }
        }
    } else {
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int rl = pass * 16 + (tid >> 4);
        const int r = r0 + rl, c = c0 + cl;
        float pv[4] = {0.f, 0.f, 0.f, 0.f};
        if (r < t.rows && c < t.cols) {
            const size_t o = (size_t)r * t.cols + c;
            const int nv = min(4, t.cols - c);
            float gv[4], mv[4], vv[4];
            for (int e = 0; e < nv; ++e) {
                pv[e] = t.p[o + e];
                if (a.update) { gv[e] = t.g16 ? bf16_to_f32(t.g16[o + e]) : t.g[o + e]; mv[e] = t.m[o + e]; vv[e] = t.v[o + e]; }
            }
            if (a.update) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (e < nv) {
                        float g = gv[e] * a.grad_scale + reg * pv[e];
                        if (a.weight_decay != 0.f) g += a.weight_decay * pv[e];
                        const float m = mv[e] + (g - mv[e]) * (1.f - a.beta1);
                        const float v = vv[e] * a.beta2 + (1.f - a.beta2) * g * g;
                        const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
                        pv[e] = pv[e] - a.step_size * (m / denom);
                        mv[e] = m;
                        vv[e] = v;
                    }
                }
                for (int e = 0; e < nv; ++e) { t.p[o + e] = pv[e]; t.m[o + e] = mv[e]; t.v[o + e] = vv[e]; }
            }
            if (t.sh) {
                T* s = (T*)t.sh + (size_t)r * t.ld_sh + c;
                if (nv == 4) store4<T>(s, pv[0], pv[1], pv[2], pv[3]);
                else for (int e = 0; e < nv; ++e) s[e] = Elem<T>::from(pv[e]);
            }
        }
        if (t.shT) {
#pragma unroll
            for (int e = 0; e < 4; ++e) tile[rl][cl + e] = (r < t.rows && c + e < t.cols) ? pv[e] : 0.f;
        }
    }
    }
    if (t.shT) {
        __syncthreads();
        // transposed shadow [cols_p][ld_shT]: thread -> column c0+nl, 16 consecutive rows
        const int nl = tid >> 2, rq = (tid & 3) * 16;
        if (c0 + nl < t.cols) {
            T* dst = (T*)t.shT + (size_t)(c0 + nl) * t.ld_shT + r0 + rq;
            if (r0 + rq + 16 <= t.rows) {
#pragma unroll
                for (int e = 0; e < 16; e += 4)
                    store4<T>(dst + e, tile[rq + e][nl], tile[rq + e + 1][nl], tile[rq + e + 2][nl], tile[rq + e + 3][nl]);
            } else {
                for (int e = 0; e < 16; ++e)
                    if (r0 + rq + e < t.rows) dst[e] = Elem<T>::from(tile[rq + e][nl]);
            }
        }
    }
}

int rtx_launch_adam(RtxAdamArgs& a, int is_bf16, hipStream_t stream)
{
    RTX_CHECK(a.n > 0 && a.n <= RTX_MAX_TENSORS, RTX_EINVAL, "adam: bad tensor count %d", a.n);
    int tiles = 0;
    for (int k = 0; k < a.n; ++k) {
        a.t[k].tile_start = tiles;
        // flat walk: no transposed copy to produce, rows not 16-byte periodic, buffers 16-byte aligned
        const RtxAdamTensor& tk = a.t[k];
        const bool aligned = (((uintptr_t)tk.p | (uintptr_t)tk.m | (uintptr_t)tk.v | (uintptr_t)tk.g | (uintptr_t)tk.g16) & 15) == 0;
        // ... and a bias (one row) always: a handful of 4096-element tiles instead of one 64-column tile per workgroup
        // (round 6: every tensor without a transposed copy -- rows of whole float4s too: the flat walk streams at the rate of a copy, tiles do not)
        a.t[k].flat = (!tk.shT && aligned && ((long)tk.rows * tk.cols & 3) == 0) || (!tk.shT && aligned && (((tk.cols & 3) != 0 && tk.rows > 1) || tk.rows == 1)) ? 1 : 0;
        if (a.t[k].flat) tiles += (int)(((long)tk.rows * tk.cols + 4095) / 4096);
        else tiles += ((a.t[k].rows + 63) / 64) * ((a.t[k].cols + 63) / 64);
    }
    if (tiles == 0) return RTX_OK;
    a.total_tiles = tiles;
    const int grid = 8 * ((tiles + 7) / 8);
    if (is_bf16)
        hipLaunchKernelGGL(k_adam<bf16_t>, dim3(grid), dim3(256), 0, stream, a);
    else
        hipLaunchKernelGGL(k_adam<float>, dim3(grid), dim3(256), 0, stream, a);
    RTX_HIP(hipGetLastError());
    return RTX_OK;
}

// sum of squares of each parameter tensor (Mult-DAE's lam * sum_W ||W||_2, models.py:702-706)
__global__ __launch_bounds__(256) void k_sumsq(const float* p, long n, float* out)
{
    __shared__ float red[4];
    float s = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) s += p[i] * p[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) atomicAdd(out, s);
}

int rtx_launch_sumsq(const float* const* params_host, const long* sizes, int n, float* sumsq, hipStream_t stream)
{
    RTX_HIP(hipMemsetAsync(sumsq, 0, sizeof(float) * n, stream));
    for (int t = 0; t < n; ++t) {
        const int blocks = (int)((sizes[t] + 256 * 16 - 1) / (256 * 16));
        hipLaunchKernelGGL(k_sumsq, dim3(blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks)), dim3(256), 0, stream, params_host[t],
                           sizes[t], sumsq + t);
    }
    RTX_HIP(hipGetLastError());
    return RTX_OK;
}

// ------------------------------------------------------------------------------------------------
// Device-side ranking metrics for evaluate() (reference rectorch/metrics.py:136-147, 187-196):
// per user, exact top-K of the score row on order-preserving keys (a lower bound from per-thread maxima, the few hundred elements above
// it ranked by counting; a radix select only for rows of > 1024 ties), then nDCG@k / Recall@k for every requested k <= K against the held-out CSR row.  Only
// [n_k][B] doubles leave the GPU instead of the [B, n_items] score matrix (40 MB per 500 users at ml-20m).
// ------------------------------------------------------------------------------------------------
#define RTX_TOPK_MAX 1024

__device__ __forceinline__ uint32_t score_key(float f)
{
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);   // ascending in the float order, -inf lowest
}

struct RtxTopkArgs {
    const float* scores;
    long ld;
    int n_items, K, Kp2;
    RtxCsrView held;
    int n_k;
    int ks[16];
    double* ndcg;     // [n_k][B]  (nullable)
    double* recall;   // [n_k][B]  (nullable)
    int32_t* topk;    // [B][K]    (nullable)
    int B;
    RtxCsrView excl;  // has_excl: the users' rows of the TRAIN matrix -- their stored items (< n_items) rank as -inf without the
    int has_excl;     //   scores being touched (predict(remove_train=True) folded into the selection: rtx_engine_evaluate_topk)
    long out_ld;      // doubles between two cut-offs' rows of ndcg / recall (>= B; B = one [n_k][B] block per call)
    int dbg_stop;     // measurement (env RTX_TOPK_STOP at launch): the kernel returns after stage dbg_stop (0 = runs to the end)
};

// Exact top-K of a score row + the ranking metrics.  Round 4: the selection no longer histograms the row.  The 4-pass radix
// select of rounds 1-3 put 20 108 LDS atomics per pass on one or two bins (the scores of a row share sign and exponent bits, so
// the first digits are the same for nearly all of them): 285 us per 500 users, half of evaluate_device (profiles/r4_eval_kernel_stats.txt).
//   1. every thread keeps the c = ceil(K / 256) largest keys of its 79 elements (registers, no atomics);
//   2. the K-th largest of these 256 c keys -- all distinct elements -- is a LOWER BOUND L of the K-th largest score of the row
//      (rounds 4-5: a bitonic sort of <= 1024 keys in LDS; round 6: counting, see k_topk_metrics);
//   3. the elements >= L (a few hundred) are collected and ranked; the first K are the answer.
// More than RTX_TOPK_MAX elements >= L (a row of ties): the radix select below, unchanged, takes over.
// f(key, index) for every element of a score row.  16-byte loads, eight of them in flight per thread, wherever the row is 16-byte
// aligned (round 5: the 4-byte loads of rounds 1-4 walked the 80-KB row in 79 dependent round trips per thread -- with two
// workgroups per CU there is nothing to hide them behind; 71 -> see profiles/r5_eval_kernel_stats.txt)
template <typename F> __device__ __forceinline__ void topk_scan_row(const float* __restrict__ row, int n_items, int tid, F&& f)
{
    int done = 0;
    if ((((uintptr_t)row) & 15) == 0) {
        const int n4 = n_items >> 2;
        const float4* __restrict__ r4 = (const float4*)row;
        int j = tid;
        for (; j + 7 * 256 < n4; j += 8 * 256) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = r4[j + u * 256];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i0 = (j + u * 256) * 4;
                f(score_key(v[u].x), i0); f(score_key(v[u].y), i0 + 1); f(score_key(v[u].z), i0 + 2); f(score_key(v[u].w), i0 + 3);
            }
        }
        for (; j < n4; j += 256) {
            const float4 v = r4[j];
            const int i0 = j * 4;
            f(score_key(v.x), i0); f(score_key(v.y), i0 + 1); f(score_key(v.z), i0 + 2); f(score_key(v.w), i0 + 3);
        }
        done = n4 * 4;
    }
    for (int i = done + tid; i < n_items; i += 256) f(score_key(row[i]), i);
}

// NV > 0: the whole row (<= NV * 1024 items, 16-byte aligned) is loaded ONCE, NV 16-byte loads per thread in one burst, and stays in
// registers for both passes over it (maxima, collection); NV = 0: the row is streamed twice (any length / alignment).
//
// Round 6: the kernel is ONE workgroup's latency (500 rows = 500 workgroups, all resident at once), and that latency was barriers:
// two bitonic sorts in LDS (36 + 36..45 compare-exchange steps, a __syncthreads each), six two-barrier block sums per cut-off,
// a serial atomic counter for the candidates and four double-precision log2 per thread.  Now
//   * order statistics by COUNTING: every thread ranks its own keys against all the others through broadcast LDS reads (16 bytes
//     = 4 keys per instruction, every lane the same address: no bank conflict) -- one barrier per "sort";
//   * candidates placed by a block prefix sum of per-thread counts (no atomics; the order is irrelevant, they are ranked next);
//   * the held-out row (indices, values) requested at kernel entry together with the score row, parked in LDS: the relevance look-up
//     searches LDS, and its sum / positive count are block sums instead of a serial loop per thread;
//   * every cut-off's three sums reduced together: one barrier for all of them;  log2 only for ranks < K.
// (reference: rectorch/metrics.py:136-147, 187-196; evaluation.py:100-106)
// log2(r + 2) for every rank r < RTX_TOPK_MAX, written once per device by the host (glibc's log2, the function numpy calls in the
// reference's metrics): four software double-precision logarithms per thread were ~8 us of this kernel
__device__ double g_topk_log2[RTX_TOPK_MAX];
#define RTX_TOPK_HELD_CAP 512      // held-out entries of a row parked in LDS (longer rows: the global-memory look-up of rounds 1-5)

// rank (0 = first) of element (k, id) among the n (key, id) pairs in LDS, ordered by key descending, id ascending among equal keys;
// n4 = ceil(n / 4): the arrays are padded to a multiple of 4 with (key 0, id INT_MAX): below every real element
__device__ __forceinline__ uint32_t topk_rank_of(const uint32_t* __restrict__ keys, const int32_t* __restrict__ ids, int n4, uint32_t k, int32_t id)
{
    uint32_t r0 = 0, r1 = 0;
    const uint4* k4 = (const uint4*)keys;
    const int4* i4 = (const int4*)ids;
    for (int i = 0; i < n4; ++i) {
        const uint4 q = k4[i];
        const int4 d = i4[i];
        r0 += (q.x > k) + ((q.x == k) & (d.x < id)) + (q.y > k) + ((q.y == k) & (d.y < id));
        r1 += (q.z > k) + ((q.z == k) & (d.z < id)) + (q.w > k) + ((q.w == k) & (d.w < id));
    }
    return r0 + r1;
}

// number of keys greater than k among the n4 * 4 keys in LDS (two instructions per key: a compare and an add-with-carry)
__device__ __forceinline__ uint32_t topk_count_gt(const uint32_t* __restrict__ keys, int n4, uint32_t k)
{
    uint32_t g0 = 0, g1 = 0;
    const uint4* k4 = (const uint4*)keys;
    for (int i = 0; i < n4; ++i) {
        const uint4 q = k4[i];
        g0 += (q.x > k) + (q.y > k);
        g1 += (q.z > k) + (q.w > k);
    }
    return g0 + g1;
}
__device__ __forceinline__ uint32_t topk_max4(const uint4& q) { return max(max(q.x, q.y), max(q.z, q.w)); }

template <int NV>
__global__ __launch_bounds__(256) void k_topk_metrics(const RtxTopkArgs a)
{
    __shared__ uint32_t hist[256];
    __shared__ __attribute__((aligned(16))) uint32_t ckey[RTX_TOPK_MAX];
    __shared__ __attribute__((aligned(16))) int32_t cidx[RTX_TOPK_MAX];
    __shared__ int32_t sidx[RTX_TOPK_MAX];       // the ranked items
    __shared__ uint32_t relb[RTX_TOPK_MAX];      // step 4: a counter per rank; from step 5 on: the relevance of the ranked item as float bits
    __shared__ int32_t hidx[RTX_TOPK_HELD_CAP];
    __shared__ float hval[RTX_TOPK_HELD_CAP];
    __shared__ double dred[16 * 12];
    __shared__ double hred[8];
    __shared__ uint32_t wsum[4];
    __shared__ uint32_t sh_prefix, sh_mask, sh_need, sh_cnt_gt, sh_cnt_eq, sh_L, sh_tie;
    __shared__ uint32_t excl_bm[NV > 0 ? NV * 32 : 1];      // one bit per item of the row: stored in the user's train row
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* row = a.scores + (size_t)b * a.ld;
    const int K = a.K;
    // ---- 0. everything this workgroup will read from memory is requested here: the held-out row's bounds (uniform: scalar loads),
    //         the score row (NV x 16 B per thread), the held-out entries (<= 2 per thread)
    const int64_t u = csr_row(a.held, b);
    const int64_t hb = a.held.indptr[u], he = a.held.indptr[u + 1];
    const int hn = (int)(he - hb);
    const bool held_in_lds = hn <= RTX_TOPK_HELD_CAP;
    float4 rv[NV > 0 ? NV : 1];
    const int n4 = a.n_items >> 2;
    if constexpr (NV > 0) {
        const float4* __restrict__ r4 = (const float4*)row;
#pragma unroll
        for (int q = 0; q < NV; ++q) rv[q] = r4[min(tid + q * 256, n4 > 0 ? n4 - 1 : 0)];   // (clamped: the guard is at the use)
    }
    double l2r[RTX_TOPK_MAX / 256];             // log2(r + 2) of this thread's ranks (only ranks < K are ever used)
#pragma unroll
    for (int m = 0; m < RTX_TOPK_MAX / 256; ++m) l2r[m] = g_topk_log2[tid + 256 * m];
    int32_t hi0 = 0x7fffffff, hi1 = 0x7fffffff;
    float hv0 = 0.f, hv1 = 0.f;
    if (held_in_lds) {
        if (tid < hn) { hi0 = a.held.indices[hb + tid]; hv0 = a.held.values ? a.held.values[hb + tid] : 1.f; }
        if (tid + 256 < hn) { hi1 = a.held.indices[hb + tid + 256]; hv1 = a.held.values ? a.held.values[hb + tid + 256] : 1.f; }
    }
    // the row as order-preserving keys (registers); f4(keys of four neighbours, index of the first) / f1(key, index) visit every element
    uint4 kv[NV > 0 ? NV : 1];
    if constexpr (NV > 0) {
#pragma unroll
        for (int q = 0; q < NV; ++q) kv[q] = make_uint4(score_key(rv[q].x), score_key(rv[q].y), score_key(rv[q].z), score_key(rv[q].w));
    }
    // the user's train items rank as -inf (reference models.py:470-471, 952-953: recon_x[x.nonzero()] = -inf): a bitmap of the row in
    // LDS, set from the train row's stored entries, read back four bits per group of neighbours -- instead of a scatter kernel of its
    // own over the score matrix (5 us per batch of 500)
    const bool use_excl = NV > 0 && a.has_excl;
    if constexpr (NV > 0) {
        if (use_excl) {
            for (int i = tid; i < NV * 32; i += 256) excl_bm[i] = 0u;
            __syncthreads();
            const int64_t ue = csr_row(a.excl, b);
            for (int64_t k = a.excl.indptr[ue] + tid; k < a.excl.indptr[ue + 1]; k += 256) {
                const float val = a.excl.values ? a.excl.values[k] : 1.f;
                const int i = a.excl.indices[k];
                if (val != 0.f && i < a.n_items) atomicOr(&excl_bm[i >> 5], 1u << (i & 31));
            }
            __syncthreads();
            const uint32_t NEG = score_key(-INFINITY);
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                const int j = tid + q * 256;          // items 4 j .. 4 j + 3: bits (4 j) & 31 .. of word j >> 3
                const uint32_t nib = (excl_bm[j >> 3] >> ((j & 7) * 4)) & 15u;
                if (nib) {
                    if (nib & 1u) kv[q].x = NEG;
                    if (nib & 2u) kv[q].y = NEG;
                    if (nib & 4u) kv[q].z = NEG;
                    if (nib & 8u) kv[q].w = NEG;
                }
            }
        }
    }
    auto tail_key = [&](int i) __attribute__((always_inline)) -> uint32_t {     // (NV > 0: the < 4 elements behind the last whole group)
        if (use_excl && ((excl_bm[i >> 5] >> (i & 31)) & 1u)) return score_key(-INFINITY);
        return score_key(row[i]);
    };
    auto scan = [&](auto&& f4, auto&& f1) __attribute__((always_inline)) {
        if constexpr (NV > 0) {
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                const int j = tid + q * 256;
                if (j < n4) f4(kv[q], j * 4);
            }
            for (int i = n4 * 4 + tid; i < a.n_items; i += 256) f1(tail_key(i), i);
        } else {
            topk_scan_row(row, a.n_items, tid, f1);
        }
    };
    // ---- 1. per-thread maxima
    const int c = (K + 255) / 256;              // 1 .. 4
    uint32_t t0 = 0, t1 = 0, t2 = 0, t3 = 0;    // this thread's largest keys, descending (0 = below every real key)
    auto ins = [&](uint32_t k) __attribute__((always_inline)) {
        if (k > t0) { const uint32_t x = t0; t0 = k; k = x; }
        if (k > t1) { const uint32_t x = t1; t1 = k; k = x; }
        if (k > t2) { const uint32_t x = t2; t2 = k; k = x; }
        if (k > t3) t3 = k;
    };
    if (c == 1) scan([&](const uint4& q, int) { t0 = max(t0, topk_max4(q)); }, [&](uint32_t k, int) { t0 = max(t0, k); });
    else scan([&](const uint4& q, int) { ins(q.x); ins(q.y); ins(q.z); ins(q.w); }, [&](uint32_t k, int) { ins(k); });
    if (a.dbg_stop == 1) { if (t0 == 1u) a.ndcg[b] = 0.0; return; }
    // ---- 2. L = K-th largest of the 256 c thread maxima: a LOWER BOUND of the row's K-th largest score (they are distinct elements).
    //         L = the smallest key that fewer than K keys exceed: only "greater than" counts are needed, ties included
    const int n1 = c == 1 ? 256 : (c == 2 ? 512 : 1024);
    ckey[tid] = t0;
    if (c >= 2) ckey[256 + tid] = t1;
    if (c >= 3) { ckey[512 + tid] = t2; ckey[768 + tid] = c >= 4 ? t3 : 0u; }
    if (held_in_lds) { hidx[tid] = hi0; hval[tid] = hv0; hidx[tid + 256] = hi1; hval[tid + 256] = hv1; }
    if (tid == 0) { sh_L = 0xffffffffu; sh_tie = 0; }
    __syncthreads();
    {
        uint32_t cand = 0xffffffffu;
        if (topk_count_gt(ckey, n1 / 4, t0) < (uint32_t)K) cand = t0;
        if (c >= 2) {
            if (topk_count_gt(ckey, n1 / 4, t1) < (uint32_t)K) cand = min(cand, t1);
            if (c >= 3) {
                if (topk_count_gt(ckey, n1 / 4, t2) < (uint32_t)K) cand = min(cand, t2);
                if (c >= 4 && topk_count_gt(ckey, n1 / 4, t3) < (uint32_t)K) cand = min(cand, t3);
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cand = min(cand, (uint32_t)__shfl_xor((int)cand, o, 64));
        if (lane == 0) atomicMin(&sh_L, cand);
    }
    __syncthreads();
    const uint32_t L = sh_L;                    // (a row of fewer than K elements: a padding key, 0 -- everything is collected)
    if (a.dbg_stop == 2) { if (L == 1u) a.ndcg[b] = 0.0; return; }
    // ---- 3. collect the elements >= L: per-thread counts, block prefix sum, placement.  One element in a hundred qualifies: a group of
    //         four neighbours is looked at only when its maximum does
    uint32_t mine = 0;
    scan([&](const uint4& q, int) { if (topk_max4(q) >= L) mine += (q.x >= L) + (q.y >= L) + (q.z >= L) + (q.w >= L); },
         [&](uint32_t k, int) { mine += k >= L; });
    uint32_t incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = __shfl_up(incl, o, 64);
        if (lane >= o) incl += v;
    }
    if (lane == 63) wsum[wave] = incl;
    for (int i = tid; i < RTX_TOPK_MAX; i += 256) { ckey[i] = 0; cidx[i] = 0x7fffffff; relb[i] = 0u; }   // (everybody has read the maxima: barrier above)
    __syncthreads();
    uint32_t base = incl - mine;
    for (int w = 0; w < wave; ++w) base += wsum[w];
    const uint32_t n_cand = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    if (n_cand <= (uint32_t)RTX_TOPK_MAX) {
        uint32_t pos = base;
        auto put = [&](uint32_t k, int i) __attribute__((always_inline)) { if (k >= L) { ckey[pos] = k; cidx[pos] = i; ++pos; } };
        scan([&](const uint4& q, int i0) { if (topk_max4(q) >= L) { put(q.x, i0); put(q.y, i0 + 1); put(q.z, i0 + 2); put(q.w, i0 + 3); } }, put);
    }
    __syncthreads();
    int n_rank = (int)n_cand;                   // candidates to rank
    if (a.dbg_stop == 3) { if (ckey[tid] == 1u) a.ndcg[b] = 0.0; return; }
    if (n_cand > (uint32_t)RTX_TOPK_MAX) {
        // ---- a row with more than RTX_TOPK_MAX elements tied at / above the bound: radix select of the K-th largest key T, then
        //      everything above T and need_eq of the ties (ties at the K-th place are arbitrary in the reference's argpartition too)
        if (tid == 0) { sh_prefix = 0; sh_mask = 0; sh_need = (uint32_t)K; }
        __syncthreads();
        for (int shift = 24; shift >= 0; shift -= 8) {
            hist[tid] = 0;
            __syncthreads();
            const uint32_t prefix = sh_prefix, mask = sh_mask;
            for (int i = tid; i < a.n_items; i += 256) {
                const uint32_t k = score_key(row[i]);
                if ((k & mask) == prefix) atomicAdd(&hist[(k >> shift) & 255u], 1u);
            }
            __syncthreads();
            if (tid == 0) {
                uint32_t need = sh_need, d = 255;
                for (;; --d) {            // from the largest digit down
                    if (hist[d] >= need || d == 0) break;
                    need -= hist[d];
                }
                sh_need = need;           // rank inside digit d
                sh_prefix = prefix | (d << shift);
                sh_mask = mask | (255u << shift);
            }
            __syncthreads();
        }
        const uint32_t T = sh_prefix;
        const uint32_t need_eq = sh_need;   // how many elements equal to T belong to the top K
        if (tid == 0) { sh_cnt_gt = 0; sh_cnt_eq = 0; }
        for (int i = tid; i < RTX_TOPK_MAX; i += 256) { ckey[i] = 0; cidx[i] = 0x7fffffff; }
        __syncthreads();
        for (int i = tid; i < a.n_items; i += 256) {
            const uint32_t k = score_key(row[i]);
            if (k > T) {
                const uint32_t p = atomicAdd(&sh_cnt_gt, 1u);
                if (p < (uint32_t)K) { ckey[p] = k; cidx[p] = i; }
            }
        }
        __syncthreads();
        const uint32_t n_gt = sh_cnt_gt;
        for (int i = tid; i < a.n_items; i += 256) {
            const uint32_t k = score_key(row[i]);
            if (k == T) {
                const uint32_t p = atomicAdd(&sh_cnt_eq, 1u);
                if (p < need_eq && n_gt + p < (uint32_t)K) { ckey[n_gt + p] = k; cidx[n_gt + p] = i; }
            }
        }
        __syncthreads();
        n_rank = K;
    }
    // ---- 4. rank the candidates (key descending, index ascending among equal keys): the ranks < K are the answer, in order.
    //         First by "greater than" counts alone (two instructions per pair); two candidates of one rank below K -- equal scores
    //         among the ranked items: rare -- are noticed through a counter per rank, and only then the exact ranks (ties by index)
    //         are computed.
    for (int i = tid; i < K; i += 256) sidx[i] = 0x7fffffff;     // (a row of fewer than K elements: the tail stays "no item")
    __syncthreads();
    const int r4n = (n_rank + 3) >> 2;
    for (int p = tid; p < n_rank; p += 256) {
        const uint32_t r = topk_count_gt(ckey, r4n, ckey[p]);
        if (r < (uint32_t)K) {
            if (atomicAdd(&relb[r], 1u) != 0u) sh_tie = 1;
            sidx[r] = cidx[p];
        }
    }
    __syncthreads();
    if (sh_tie) {
        for (int p = tid; p < n_rank; p += 256) {
            const int32_t id = cidx[p];
            const uint32_t r = topk_rank_of(ckey, cidx, r4n, ckey[p], id);
            if (r < (uint32_t)K) sidx[r] = id;
        }
        __syncthreads();
    }
    if (a.dbg_stop == 4) { if (sidx[tid] == -5) a.ndcg[b] = 0.0; return; }
    // ---- 5. relevance of every ranked item: value of the held-out row at that item (0 if absent)
    for (int r = tid; r < K; r += 256) {
        const int item = sidx[r];
        float v = 0.f;
        if (held_in_lds) {
            int lo = 0, hi = hn;          // binary search (column ids are sorted within a row)
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (hidx[mid] < item) lo = mid + 1; else hi = mid;
            }
            if (lo < hn && hidx[lo] == item) v = hval[lo];
        } else {
            int64_t lo = hb, hi = he;
            while (lo < hi) {
                const int64_t mid = (lo + hi) >> 1;
                const int cc = a.held.indices[mid];
                if (cc < item) lo = mid + 1; else hi = mid;
            }
            if (lo < he && a.held.indices[lo] == item) v = a.held.values ? a.held.values[lo] : 1.f;
        }
        relb[r] = __float_as_uint(v);
        if (a.topk) a.topk[(size_t)b * K + r] = item;
    }
    if (a.dbg_stop == 5) return;
    // ---- 6. metrics.  The terms are the reference's (metrics.py:136-147, 187-196: rel / log2(r + 2), 1 / log2(r + 2)), one rank per
    //      thread and pass, summed in double by a fixed-order block reduction (64-lane butterfly, then the four waves pairwise).
    //      Sum and positive count of the held-out row: block sums over its parked entries (float values are small integers or
    //      ratings: exact in double in any order).
    double gs = 0.0, np = 0.0;
    if (held_in_lds) {
        gs = (double)hv0 + (double)hv1;
        np = (hv0 > 0.f ? 1.0 : 0.0) + (hv1 > 0.f ? 1.0 : 0.0);
    } else {
        for (int64_t k = hb + tid; k < he; k += 256) {
            const float v = a.held.values ? a.held.values[k] : 1.f;
            gs += (double)v;
            np += v > 0.f ? 1.0 : 0.0;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { gs += __shfl_xor(gs, o, 64); np += __shfl_xor(np, o, 64); }
    if (lane == 0) { hred[wave] = gs; hred[4 + wave] = np; }
    __syncthreads();    // (also: the relevances are complete)
    const double gsum = (hred[0] + hred[1]) + (hred[2] + hred[3]);
    const long npos = (long)((hred[4] + hred[5]) + (hred[6] + hred[7]));
    for (int q = 0; q < a.n_k; ++q) {
        const int kk = min(min(a.ks[q], a.n_items), K);
        const long nid = min((long)gsum, (long)min(a.ks[q], a.n_items));      // tp[:min(int(n), k)].sum()   (metrics.py:146)
        double dcg = 0.0, idcg = 0.0, hits = 0.0;
#pragma unroll
        for (int m = 0; m < RTX_TOPK_MAX / 256; ++m) {
            const int r = tid + 256 * m;
            const double l2 = l2r[m];
            if (r < kk) { const float rl = __uint_as_float(relb[r]); dcg += (double)rl / l2; hits += rl > 0.f ? 1.0 : 0.0; }
            if (r < nid && r < K) idcg += 1.0 / l2;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { dcg += __shfl_xor(dcg, o, 64); idcg += __shfl_xor(idcg, o, 64); hits += __shfl_xor(hits, o, 64); }
        if (lane == 0) { dred[q * 12 + wave] = dcg; dred[q * 12 + 4 + wave] = idcg; dred[q * 12 + 8 + wave] = hits; }
    }
    __syncthreads();
    if (tid < a.n_k) {
        const int q = tid;
        const double* d = dred + q * 12;
        const double dcg = (d[0] + d[1]) + (d[2] + d[3]), idcg = (d[4] + d[5]) + (d[6] + d[7]), hits = (d[8] + d[9]) + (d[10] + d[11]);
        if (a.ndcg) a.ndcg[(size_t)q * a.out_ld + b] = dcg / idcg;
        if (a.recall) a.recall[(size_t)q * a.out_ld + b] = (double)(float)hits / (double)min((long)min(a.ks[q], a.n_items), npos);   // metrics.py:194-195
    }
}

int rtx_launch_topk_metrics(const float* scores, long ld, int B, int n_items, const RtxCsrView& held, const int* ks, int n_k,
                            int kmax, double* ndcg, double* recall, int32_t* topk, hipStream_t stream, long out_ld, const RtxCsrView* excl)
{
    if (B <= 0) return RTX_OK;
    RTX_CHECK(n_k >= 1 && n_k <= 16, RTX_EINVAL, "topk_metrics: 1..16 cut-offs supported, got %d", n_k);
    const int K = kmax < n_items ? kmax : n_items;
    RTX_CHECK(K >= 1 && K <= RTX_TOPK_MAX, RTX_EINVAL, "topk_metrics: k must be in [1, %d], got %d", RTX_TOPK_MAX, K);
    RtxTopkArgs a = {};
    a.scores = scores; a.ld = ld; a.n_items = n_items; a.K = K;
    a.Kp2 = 2;
    while (a.Kp2 < K) a.Kp2 <<= 1;
    a.held = held; a.n_k = n_k;
    for (int q = 0; q < n_k; ++q) {
        RTX_CHECK(ks[q] >= 1, RTX_EINVAL, "topk_metrics: cut-off must be >= 1");
        a.ks[q] = ks[q];
    }
    a.ndcg = ndcg; a.recall = recall; a.topk = topk; a.B = B;
    a.out_ld = out_ld > 0 ? out_ld : B;
    const bool burst = (((uintptr_t)scores) & 15) == 0 && (ld & 3) == 0 && n_items <= 20 * 1024;
    if (excl) {
        if (burst) { a.excl = *excl; a.has_excl = 1; }
        else RTX_TRY(rtx_launch_neg_inf(*excl, B, (float*)scores, ld, n_items, stream));   // (the streamed form of the kernel: the scatter kernel first)
    }
    if (const char* dbg = getenv("RTX_TOPK_STOP")) a.dbg_stop = atoi(dbg);
    {
        static bool table_ready[64] = {};
        int devid = 0;
        RTX_HIP(hipGetDevice(&devid));
        if (devid >= 0 && devid < 64 && !table_ready[devid]) {
            std::vector<double> t(RTX_TOPK_MAX);
            for (int r = 0; r < RTX_TOPK_MAX; ++r) t[r] = std::log2((double)(r + 2));
            RTX_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_topk_log2), t.data(), sizeof(double) * RTX_TOPK_MAX));
            RTX_HIP(hipStreamSynchronize(nullptr));   // (the copy runs on the NULL stream, which a non-blocking caller's stream does not wait for: engine.hip dev_alloc has the story)
            table_ready[devid] = true;
        }
        RTX_CHECK(devid >= 0 && devid < 64, RTX_EINVAL, "topk_metrics: device index %d", devid);
    }
    if (burst)
        hipLaunchKernelGGL(k_topk_metrics<20>, dim3(B), dim3(256), 0, stream, a);
    else
        hipLaunchKernelGGL(k_topk_metrics<0>, dim3(B), dim3(256), 0, stream, a);
    RTX_HIP(hipGetLastError());
    return RTX_OK;
}


// f32 -> bf16 (round to nearest even) of a gradient range before its RCCL all-reduce (data parallel, bf16 exchange)
__global__ __launch_bounds__(256) void k_cast_f32_bf16(const float* __restrict__ src, bf16_t* __restrict__ dst, long n)
{
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i + 8 <= n) {
        const float4 a = *(const float4*)(src + i), b = *(const float4*)(src + i + 4);
        uint4 o;
        o.x = (uint32_t)f32_to_bf16(a.x) | ((uint32_t)f32_to_bf16(a.y) << 16);
        o.y = (uint32_t)f32_to_bf16(a.z) | ((uint32_t)f32_to_bf16(a.w) << 16);
        o.z = (uint32_t)f32_to_bf16(b.x) | ((uint32_t)f32_to_bf16(b.y) << 16);
        o.w = (uint32_t)f32_to_bf16(b.z) | ((uint32_t)f32_to_bf16(b.w) << 16);
        *(uint4*)(dst + i) = o;
    } else {
        for (long k = i; k < n; ++k) dst[k] = f32_to_bf16(src[k]);
    }
}

// Split-K slabs of a small weight-gradient product -> the gradient tensors (float32 parity mode, round 4):
//   gW[m * N_real + n] = sum_s C[s * slab_stride + m * ldc + n]  (m < M_real, n < N_real);  gbias[m] = the same at n == N_real.
// Fixed summation order (s ascending): the step stays bit-reproducible.
__global__ __launch_bounds__(256) void k_dw_slab_reduce(const float* __restrict__ C, int splits, long slab_stride, long ldc, int M_real, int N_real,
                                                        float* __restrict__ gW, float* __restrict__ gbias)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)M_real * (N_real + 1);
    if (i >= total) return;
    const int m = (int)(i / (N_real + 1)), n = (int)(i - (long)m * (N_real + 1));
    const float* c = C + (size_t)m * ldc + n;
    float v = 0.f;
    for (int s2 = 0; s2 < splits; ++s2) v += c[(size_t)s2 * slab_stride];
    if (n < N_real) gW[(size_t)m * N_real + n] = v;
    else if (gbias) gbias[m] = v;
}

// the tail region of a float32 gradient product (gemm_f32_km, RtxGemm::tail_*): [splits][rows][ldc] partial sums in tile-local
// coordinates -> rows row0.., columns col0.. of gW [M_real][N_real] (column N_real of the product = the bias gradient); fixed order
__global__ __launch_bounds__(256) void k_tail_reduce(const float* __restrict__ C, int splits, long slab_stride, long ldc, int rows, int cols, int row0,
                                                     int col0, int M_real, int N_real, float* __restrict__ gW, float* __restrict__ gbias)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)rows * cols) return;
    const int m = (int)(i / cols), n = (int)(i - (long)m * cols);
    const int gm = row0 + m, gn = col0 + n;
    if (gm >= M_real || gn > N_real) return;
    const float* c = C + (size_t)m * ldc + n;
    float v = 0.f;
    for (int s2 = 0; s2 < splits; ++s2) v += c[(size_t)s2 * slab_stride];
    if (gn < N_real) gW[(size_t)gm * N_real + gn] = v;
    else if (gbias) gbias[gm] = v;
}

int rtx_launch_tail_reduce(const float* C, int splits, long slab_stride, long ldc, int rows, int cols, int row0, int col0, int M_real, int N_real,
                           float* gW, float* gbias, hipStream_t stream)
{
    const long total = (long)rows * cols;
    if (total <= 0) return RTX_OK;
    hipLaunchKernelGGL(k_tail_reduce, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, C, splits, slab_stride, ldc, rows, cols, row0, col0,
                       M_real, N_real, gW, gbias);
    RTX_HIP(hipGetLastError());
    return RTX_OK;
}

int rtx_launch_dw_slab_reduce(const float* C, int splits, long slab_stride, long ldc, int M_real, int N_real, float* gW, float* gbias, hipStream_t stream)
{
    const long total = (long)M_real * (N_real + 1);
    if (total <= 0) return RTX_OK;
    hipLaunchKernelGGL(k_dw_slab_reduce, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, C, splits, slab_stride, ldc, M_real, N_real, gW, gbias);
    RTX_HIP(hipGetLastError());
    return RTX_OK;
}

int rtx_launch_cast_f32_bf16(const float* src, bf16_t* dst, long n, hipStream_t stream)
{
    if (n <= 0) return RTX_OK;
    RTX_CHECK(((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0, RTX_EINVAL, "cast: buffers must be 16-byte aligned");
    hipLaunchKernelGGL(k_cast_f32_bf16, dim3((unsigned)((n + 2047) / 2048)), dim3(256), 0, stream, src, dst, n);
    RTX_HIP(hipGetLastError());
    return RTX_OK;
}
