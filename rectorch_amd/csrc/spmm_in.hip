// spmm_in.hip -- the FIRST encoder layer of the Mult-VAE / Mult-DAE step as a sparse product (gfx950, bf16 numerics).
//
// Reference: `h = tanh(W1 F.normalize(dropout(x)) + b1)` (nets.py:394-399, 219-223) on a dense [B, n_items] tensor.  A batch
// row holds ~70-150 stored entries out of 20 108 columns, so the dense product moves 44 MB (the dense input image and the
// whole weight matrix) through split-K slabs plus a post kernel to do 2 * 36 000 * 600 useful multiply-adds.  Here:
//
//   k_in_chunks   one workgroup per user: the user's stored entries -- normalised, dropped out (the same Philox decision
//                 per (user, item) the dense gather takes), rounded to bf16 -- packed as (item | value << 16) words into
//                 one stream of 64-entry chunks, users back to back, each user's last chunk zero-padded; desc[chunk] =
//                 user; wsplit[] = a 16-way split of the stream at user boundaries (balanced by chunks).  In a training
//                 step the same kernel also writes what k_gather would: the dense bf16 image of the row (the weight-
//                 gradient kernel still reads it K-major) and the target row sums -- each entry is computed once.
//   k_spmm_in     one workgroup per FOUR output features: their four weight rows (4 x n_items bf16 = 157 KB of the
//                 160 KB LDS at n_items = 20 108) are interleaved into LDS once -- item i -> 8 bytes (w0 w1 | w2 w3) --
//                 then every wave streams its share of the chunk stream (coalesced 256-B loads, two blocks of 16
//                 chunks in flight while a third is consumed); each lane gathers its entry's four weights with ONE ds_read_b64 and accumulates them with
//                 four v_dot2c_f32_bf16 (the value sits in one half of the second operand, zero in the other: no
//                 bf16 -> f32 unpacking).  At a user boundary the four per-lane sums are reduced across the wave
//                 together (v_permlane32_swap, v_permlane16_swap, four DPP steps: 13 instructions for the four sums) and
//                 parked in one lane each; every 16 users the wave adds the bias, applies tanh and writes the float32
//                 activation and the bf16 operand row of the next layer with all 64 lanes busy.
//
// Measured at the ml-20m shape (B = 500; profiles/r2_bench_kernel_stats.txt, r2_spmm_native_test.log): k_in_chunks 12 us (the
// gather kernel it replaces: 12-14), k_spmm_in 20 us (staging 5, sums 9 -- VALU-bound: 8 instructions per chunk, 13 per user
// end -- the rest launch ramp and tail) against 22 + 9 us for the dense split-K product and its post kernel: 9-13 us per step.  Tried and dropped: the dense image as zeros streamed out at kernel start +
// 2-byte scattered stores of the values (17 us instead of 15; with an agent-scope fence 79 us: it writes the L2 back); the
// dense image rebuilt from the chunk stream on the step's side stream (no gain: profiles/r2_defer_image_experiment.log).
//
// The weight matrix is read from HBM exactly once (24 MB), the chunk stream (~360 KB) from L2 by every workgroup.
// Products are bf16 x bf16 exact in float32 like the MFMA's; only the order of the float32 additions differs from the dense
// kernel (per lane over the user's chunks, then across the wave) -- fixed, so results are reproducible run to run.
#include "rtx_kernels.h"

#define SPMM_ROWS 4
#define SPMM_WAVES 16     // waves per workgroup = parts of the chunk stream (k_in_chunks writes wsplit for this number)
#define SPMM_DB 16        // chunks per register buffer (three buffers: two blocks in flight while one is consumed)
#define SPMM_FILL_PASSES 3

static_assert(SPMM_WAVES == RTX_SPMM_WAVES, "k_in_chunks and k_spmm_in agree on the split");

// ------------------------------------------------------------------------------------------------
// k_in_chunks
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int in_chunks_of(const RtxCsrView& v, int b)
{
    const int64_t u = v.row_ids ? (int64_t)v.row_ids[b] : (int64_t)b;
    const int len = (int)(v.indptr[u + 1] - v.indptr[u]);
    return max(1, (len + 63) >> 6);   // an empty row still owns one (all-zero) chunk: its output is tanh(bias)
}

__global__ __launch_bounds__(512) void k_in_chunks(const RtxInChunksArgs a)
{
    constexpr int CH = 8192;   // bf16 elements per LDS image of a stretch of the dense row (16 KB)
    __shared__ __attribute__((aligned(16))) bf16_t img[CH];
    __shared__ int red_i[2][8];
    __shared__ float red_f[8];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (b >= a.B) {   // padding row of the dense image (only launched when X is given)
        bf16_t* X = a.X + (size_t)b * a.ldx;
        for (int i = tid * 8; i < a.ldx; i += 512 * 8) *(uint4*)(X + i) = make_uint4(0, 0, 0, 0);
        if (tid == 0 && a.tsum) a.tsum[b] = 0.f;
        return;
    }
    // this user's row first: its loads (row id -> row bounds -> first entries) are on their way while the chunk offsets of
    // all users are summed below -- five dependent HBM round trips folded into three
    const int64_t u = a.in.row_ids ? (int64_t)a.in.row_ids[b] : (int64_t)b;
    const int64_t beg = a.in.indptr[u];
    const int len = (int)(a.in.indptr[u + 1] - beg);
    const int nch = max(1, (len + 63) >> 6);
    const int first_item = tid < len ? a.in.indices[beg + tid] : 0;
    // chunks of the users before this one, and of all users
    int before = 0, total = 0;
    for (int j = tid; j < a.B; j += 512) {
        const int n = in_chunks_of(a.in, j);
        total += n;
        if (j < b) before += n;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        before += __shfl_xor(before, o, 64);
        total += __shfl_xor(total, o, 64);
    }
    if (lane == 0) { red_i[0][wv] = before; red_i[1][wv] = total; }
    __syncthreads();
    before = 0; total = 0;
#pragma unroll
    for (int w = 0; w < 8; ++w) { before += red_i[0][w]; total += red_i[1][w]; }
    // the engine sizes the stream from the matrix's longest row; a caller that hands a shorter bound must not corrupt memory
    if (a.cap_chunks > 0 && (int64_t)total + 1 > a.cap_chunks) return;
    // 1 / max(||x||, 1e-12) over the item columns (condition columns stay raw), as k_gather
    const bool cond = a.Iin > a.I;
    float ss;
    if (!a.in.values && !cond) {
        ss = (float)len;
    } else {
        ss = 0.f;
        for (int k = tid; k < len; k += 512) {
            const float v = a.in.values ? a.in.values[beg + k] : 1.f;
            if (!cond || a.in.indices[beg + k] < a.I) ss += v * v;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
        __syncthreads();
        if (lane == 0) red_f[wv] = ss;
        __syncthreads();
        ss = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) ss += red_f[w];
    }
    if (a.tsum) {   // s_b = sum of the TARGET row over the item columns (the multinomial likelihood's weight)
        const int64_t ut = a.target.row_ids ? (int64_t)a.target.row_ids[b] : (int64_t)b;
        const int64_t tb = a.target.indptr[ut], te = a.target.indptr[ut + 1];
        float ts;
        if (!a.target.values && !cond) {
            ts = (float)(te - tb);
        } else {
            ts = 0.f;
            for (int64_t k = tb + tid; k < te; k += 512)
                if (!cond || a.target.indices[k] < a.I) ts += a.target.values ? a.target.values[k] : 1.f;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) ts += __shfl_xor(ts, o, 64);
            __syncthreads();
            if (lane == 0) red_f[wv] = ts;
            __syncthreads();
            ts = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) ts += red_f[w];
        }
        if (tid == 0) a.tsum[b] = ts;
    }
    const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);
    const bool drop = a.training && a.dropout_p > 0.f;
    const float scale = drop ? (a.dropout_p < 1.f ? 1.f / (1.f - a.dropout_p) : 0.f) : 1.f;
    uint32_t* __restrict__ ent = a.ent + (size_t)before * 64;
    // entry t of the row -> its bf16 value (normalised, dropped out): computed once, wherever it is needed
    auto entry = [&](int t, int i) -> bf16_t {
        float v = a.in.values ? a.in.values[beg + t] : 1.f;
        if (i < a.I) v *= inv;
        if (drop && i < a.I) {   // condition columns are concatenated after the dropout (nets.py:469-471)
            const uint64_t e = (uint64_t)b * (uint64_t)a.I + (uint64_t)i;
            const bool keep = a.mask ? (a.mask[e] != 0) : rtx_dropout_keep(a.seed, a.offset, e, a.dropout_p);
            v = keep ? v * scale : 0.f;
        }
        return f32_to_bf16(v);
    };
    if (!a.X) {
        for (int t = tid; t < len; t += 512) {
            const int i = t == tid ? first_item : a.in.indices[beg + t];
            ent[t] = (uint32_t)i | ((uint32_t)entry(t, i) << 16);
        }
    } else {
        // ... and the dense image of the row (the weight-gradient kernel's operand), stretch by stretch through LDS as
        // k_gather builds it; an entry joins the chunk stream in the pass that owns its column
        bf16_t* X = a.X + (size_t)b * a.ldx;
        for (int c0 = 0; c0 < a.ldx; c0 += CH) {
            const int cn = min(CH, a.ldx - c0);
            for (int i = tid * 8; i < cn; i += 512 * 8) *(uint4*)(img + i) = make_uint4(0, 0, 0, 0);
            __syncthreads();
            for (int t = tid; t < len; t += 512) {
                const int i = t == tid ? first_item : a.in.indices[beg + t];
                if (i >= c0 && i < c0 + cn) {
                    const bf16_t v = entry(t, i);
                    img[i - c0] = v;
                    ent[t] = (uint32_t)i | ((uint32_t)v << 16);
                }
            }
            if (tid == 0 && a.Iin >= c0 && a.Iin < c0 + cn) img[a.Iin - c0] = f32_to_bf16(1.f);   // ones column -> bias gradient
            __syncthreads();
            for (int i = tid * 8; i < cn; i += 512 * 8) *(uint4*)(X + c0 + i) = *(const uint4*)(img + i);
            __syncthreads();
        }
    }
    for (int t = len + tid; t < nch * 64; t += 512) ent[t] = 0;   // zero padding of the user's last chunk
    for (int c = tid; c < nch; c += 512) a.desc[before + c] = b;
    if (b == a.B - 1 && tid == 0) a.desc[total] = -1;
    // the 16-way split: part w starts at the first user whose chunk offset reaches total * w / 16
    if (tid <= SPMM_WAVES) {
        const int w = tid;
        const long tw = (long)total * w, mine = (long)before * SPMM_WAVES;
        if (w == 0) {
            if (b == 0) a.wsplit[0] = 0;
        } else if (w == SPMM_WAVES) {
            if (b == a.B - 1) a.wsplit[w] = total;
        } else {
            const long prev = b > 0 ? (long)(before - in_chunks_of(a.in, b - 1)) * SPMM_WAVES : -1;
            if (b > 0 && mine >= tw && prev < tw) a.wsplit[w] = before;
            else if (b == a.B - 1 && mine < tw) a.wsplit[w] = total;   // the last user is longer than a whole part
        }
    }
}

int rtx_launch_in_chunks(const RtxInChunksArgs& a, hipStream_t stream)
{
    RTX_CHECK(a.B >= 1 && a.Iin >= a.I && a.Iin <= 65536, RTX_EINVAL, "in_chunks: bad shape");
    RTX_CHECK(!a.X || (a.ldx % 8 == 0 && a.Bp >= a.B), RTX_EINVAL, "in_chunks: bad dense image");
    hipLaunchKernelGGL(k_in_chunks, dim3(a.X ? a.Bp : a.B), dim3(512), 0, stream, a);
    RTX_HIP(hipGetLastError());
    return RTX_OK;
}

// ------------------------------------------------------------------------------------------------
// k_spmm_in
// ------------------------------------------------------------------------------------------------
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
// acc += w.lo * v.lo + w.hi * v.hi   (bf16 pairs; products exact in float32)
__device__ __forceinline__ float dot2_bf16(uint32_t w, uint32_t v, float acc)
{
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, w), __builtin_bit_cast(bf16x2_t, v), acc, false);
}
// Sums of four per-lane values over the wave at once: afterwards every lane of 16-lane row k holds the total of
// a[k == 0 ? 0 : k == 1 ? 2 : k == 2 ? 1 : 3]  (rows 0..3 <-> values 0, 2, 1, 3).
// (inline asm: this hipcc's __builtin_amdgcn_permlane{16,32}_swap hands back element 0 of the result pair twice.  The s_nop
// cover the VALU-write -> permlane-read wait states the compiler would insert for the builtin.)
__device__ __forceinline__ void permlane32_swap(float& x, float& y)   // rows 2,3 of x <-> rows 0,1 of y
{
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
}
__device__ __forceinline__ void permlane16_swap(float& x, float& y)   // odd rows of x <-> even rows of y
{
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
}
__device__ __forceinline__ float wave_sum4(float a0, float a1, float a2, float a3)
{
    permlane32_swap(a0, a1);
    const float s01 = a0 + a1;      // rows 0,1: 32 partial sums of a0; rows 2,3: of a1
    permlane32_swap(a2, a3);
    const float s23 = a2 + a3;
    float x = s01, y = s23;
    permlane16_swap(x, y);
    float v = x + y;                // row 0 = a0, row 1 = a2, row 2 = a1, row 3 = a3 (16 partial sums each)
    v += dpp_f<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_f<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_f<0x141>(v);   // row_half_mirror
    v += dpp_f<0x140>(v);   // row_mirror
    return v;
}

// feature of this lane's 16-lane row: rows 0..3 hold features 0, 2, 1, 3 (see wave_sum4)
__device__ __forceinline__ int spmm_feature(int lane) { return (((lane >> 4) & 1) << 1) | (lane >> 5); }

// The finished users of a wave wait here, 16 at a time: lane (slot + 16 * row) holds the row's pre-activation sum of the
// slot-th of them.  A wave's users are consecutive batch rows (the stream is in batch order and every user owns a chunk).
struct SpmmOut {
    float sum;
    int first;     // batch row of slot 0 (uniform)
    int n;         // users parked (uniform)
};

__device__ __forceinline__ void spmm_flush(const RtxSpmmInArgs& a, SpmmOut& o, int lane, int n, float bias)
{
    if ((lane & 15) < o.n && n < a.N_real) {
        float x = o.sum + bias;
        if (a.tanh_act) x = tanhf(x);
        const size_t at = (size_t)(o.first + (lane & 15)) * a.Np + n;
        if (a.O32) a.O32[at] = x;
        if (a.R) a.R[at] = f32_to_bf16(x);
    }
    o.first += o.n;
    o.n = 0;
}

__global__ __launch_bounds__(1024) void k_spmm_in(const RtxSpmmInArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];   // [K8] x (w0 w1 | w2 w3)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int o0 = blockIdx.x * SPMM_ROWS;
    const int nreal = min(max(a.N_real - o0, 0), SPMM_ROWS);
    const int K8 = (a.Kin + 7) & ~7;
    // measurement hook (tests/native/test_spmm.cpp): shader-clock stamps of the first and the last wave of every workgroup
    uint64_t* const stamp = (a.stamps && lane == 0 && (wave == 0 || wave == SPMM_WAVES - 1))
                                ? a.stamps + ((size_t)blockIdx.x * 2 + (wave != 0)) * 4 : nullptr;
    if (stamp) stamp[0] = __builtin_amdgcn_s_memtime();

    if (nreal > 0) {
        const int g0 = __builtin_amdgcn_readfirstlane(a.wsplit[wave]);
        const int g1 = __builtin_amdgcn_readfirstlane(a.wsplit[wave + 1]);
        uint32_t bufA[SPMM_DB], bufB[SPMM_DB], bufC[SPMM_DB];
        int2 endA, endB, endC;       // user of chunk (block + lane) and of the next one: they differ at a user's last chunk
        // block G of the stream -> registers (scalar base + lane: one address computation per block), and its user ends
#define SPMM_LOAD(BUF, END, G)                                                                                     \
    {                                                                                                              \
        const uint32_t* blk = a.ent + (size_t)(G) * 64;                                                            \
        _Pragma("unroll") for (int j = 0; j < SPMM_DB; ++j) BUF[j] = blk[j * 64 + lane];                           \
        const int* dsc = a.desc + (G);                                                                             \
        END.x = dsc[lane];                                                                                         \
        END.y = dsc[lane + 1];                                                                                     \
    }
        // the first two blocks of the stream are on their way while the weight rows are staged
        SPMM_LOAD(bufA, endA, g0)
        SPMM_LOAD(bufB, endB, g0 + SPMM_DB)
        SpmmOut out = {0.f, __builtin_amdgcn_readfirstlane(a.desc[g0]), 0};

        // stage the four weight rows: 8 consecutive inputs per thread, row and pass, interleaved per input.  All loads of
        // the (at most three) passes are issued before the first LDS store: one HBM round trip instead of three.
        static_assert(SPMM_FILL_PASSES * 1024 * 8 >= 160 * 1024 / (2 * SPMM_ROWS), "the passes cover the largest row that fits");
        {
            uint4 r[SPMM_FILL_PASSES][SPMM_ROWS];
#pragma unroll
            for (int ps = 0; ps < SPMM_FILL_PASSES; ++ps) {
                const int i = tid * 8 + ps * 8192;
                const int ic = i < K8 ? i : 0;
#pragma unroll
                for (int q = 0; q < SPMM_ROWS; ++q)   // a feature beyond N_real re-reads the last real row (zeroed below)
                    r[ps][q] = *(const uint4*)(a.W + (size_t)(o0 + min(q, nreal - 1)) * a.ldw + ic);
            }
#pragma unroll
            for (int ps = 0; ps < SPMM_FILL_PASSES; ++ps) {
                const int i = tid * 8 + ps * 8192;
                if (i >= K8) continue;
                uint32_t x[SPMM_ROWS][4];
#pragma unroll
                for (int q = 0; q < SPMM_ROWS; ++q) {
                    const bool real = q < nreal;
                    x[q][0] = real ? r[ps][q].x : 0u; x[q][1] = real ? r[ps][q].y : 0u;
                    x[q][2] = real ? r[ps][q].z : 0u; x[q][3] = real ? r[ps][q].w : 0u;
                }
                uint4* dst = (uint4*)(lds + 2 * i);
#pragma unroll
                for (int k = 0; k < 4; ++k) {   // inputs i + 2k (low halves) and i + 2k + 1 (high halves)
                    uint4 o;
                    o.x = (x[0][k] & 0xffffu) | (x[1][k] << 16);
                    o.y = (x[2][k] & 0xffffu) | (x[3][k] << 16);
                    o.z = (x[0][k] >> 16) | (x[1][k] & 0xffff0000u);
                    o.w = (x[2][k] >> 16) | (x[3][k] & 0xffff0000u);
                    dst[k] = o;
                }
            }
        }
        __syncthreads();
        if (stamp) stamp[1] = __builtin_amdgcn_s_memtime();

        float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
        const int n_mine = o0 + spmm_feature(lane);
        const float bias_mine = n_mine < a.N_real ? a.bias[n_mine] : 0.f;   // (a load inside the loop would drain the prefetch)
        const uint2* __restrict__ wl = (const uint2*)lds;
        // one block of chunks: all gathers first (16 LDS reads in flight), then the multiply-adds; a user is closed at its
        // last chunk: the four sums are reduced together and parked until 16 users are ready for the activation
#define SPMM_CONSUME(BUF, END, G)                                                                                  \
    {                                                                                                              \
        uint2 wq[SPMM_DB];                                                                                         \
        _Pragma("unroll") for (int j = 0; j < SPMM_DB; ++j) wq[j] = wl[BUF[j] & 0xffffu];                          \
        /* chunks at and beyond g1 belong to the next wave: they follow this wave's last user end, so what they add */ \
        /* to the accumulators is never summed; only their user ends must not be seen                               */ \
        const int nlive = min(SPMM_DB, g1 - (G));                                                                  \
        const uint64_t ends = __builtin_amdgcn_ballot_w64(END.x != END.y) & ((1ull << max(nlive, 0)) - 1ull);      \
        _Pragma("unroll") for (int j = 0; j < SPMM_DB; ++j)                                                        \
        {                                                                                                          \
            const uint32_t vlo = BUF[j] >> 16, vhi = BUF[j] & 0xffff0000u;                                         \
            acc0 = dot2_bf16(wq[j].x, vlo, acc0);                                                                  \
            acc1 = dot2_bf16(wq[j].x, vhi, acc1);                                                                  \
            acc2 = dot2_bf16(wq[j].y, vlo, acc2);                                                                  \
            acc3 = dot2_bf16(wq[j].y, vhi, acc3);                                                                  \
            if ((ends >> j) & 1) {                                                                                 \
                const float z = wave_sum4(acc0, acc1, acc2, acc3);                                                 \
                out.sum = (lane & 15) == out.n ? z : out.sum;                                                      \
                acc0 = acc1 = acc2 = acc3 = 0.f;                                                                   \
                if (++out.n == 16) spmm_flush(a, out, lane, n_mine, bias_mine);                                    \
            }                                                                                                      \
        }                                                                                                          \
    }
        for (int g = g0; g < g1; g += 3 * SPMM_DB) {
            SPMM_LOAD(bufC, endC, g + 2 * SPMM_DB)
            SPMM_CONSUME(bufA, endA, g)
            SPMM_LOAD(bufA, endA, g + 3 * SPMM_DB)
            SPMM_CONSUME(bufB, endB, g + SPMM_DB)
            SPMM_LOAD(bufB, endB, g + 4 * SPMM_DB)
            SPMM_CONSUME(bufC, endC, g + 2 * SPMM_DB)
        }
#undef SPMM_CONSUME
#undef SPMM_LOAD
        spmm_flush(a, out, lane, n_mine, bias_mine);
        if (stamp) stamp[2] = __builtin_amdgcn_s_memtime();
    }
    // what no user's sum lands on: the padding rows, the padding columns and the ones column of the next layer's operand
    for (int i = tid; i < a.Bp * SPMM_ROWS; i += 1024) {
        const int b = i / SPMM_ROWS, n = o0 + (i - b * SPMM_ROWS);
        if (n >= a.Np || (b < a.B && n < a.N_real)) continue;
        const size_t at = (size_t)b * a.Np + n;
        if (a.O32) a.O32[at] = 0.f;
        if (a.R) a.R[at] = f32_to_bf16((a.ones_col && b < a.B && n == a.N_real) ? 1.f : 0.f);
    }
    if (stamp) stamp[3] = __builtin_amdgcn_s_memtime();
}

size_t rtx_spmm_in_lds_bytes(int Kin)
{
    return (size_t)((Kin + 7) & ~7) * 2 * SPMM_ROWS;
}

int rtx_launch_spmm_in(const RtxSpmmInArgs& a, hipStream_t stream)
{
    const size_t LDS = rtx_spmm_in_lds_bytes(a.Kin);
    RTX_CHECK(LDS <= 160 * 1024, RTX_EINVAL, "spmm_in: %zu bytes of LDS needed for %d input columns", LDS, a.Kin);
    RTX_CHECK(a.ldw % 8 == 0 && a.Kin <= 65536 && ((a.Kin + 7) & ~7) <= a.ldw, RTX_EINVAL, "spmm_in: bad weight layout");
    static bool lds_set = false;
    if (!lds_set) {
        RTX_HIP(hipFuncSetAttribute((const void*)k_spmm_in, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        lds_set = true;
    }
    const int grid = (a.Np + SPMM_ROWS - 1) / SPMM_ROWS;
    hipLaunchKernelGGL(k_spmm_in, dim3(grid), dim3(1024), LDS, stream, a);
    RTX_HIP(hipGetLastError());
    return RTX_OK;
}
