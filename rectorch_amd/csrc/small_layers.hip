// small_layers.hip -- the hidden layers of the forward AND the backward pass in ONE launch each (gfx950, bf16 numerics).
//
// Reference: the `nn.Linear` + tanh of every hidden layer and the mu / logvar head with the reparameterisation
// (nets.py:262-265, 394-417).  At B = 500 these are [500 x 600] x [600 x 400] and [500 x 200] x [200 x 600] products: a
// few MFLOP each, so the general GEMM + its post kernel (8-9 us per launch, almost all of it launch ramp and the latency of
// walking K slice by slice through an LDS pipeline) cost 31 us per step for four launches.  Here a wave owns 16 batch rows
// x 32 output features and K is a compile-time constant (the padded input width / 32): every fragment of the wave's A and B
// operands is loaded straight from global memory into registers in one burst -- ONE round trip, no LDS, no K loop to wait
// in -- then 2 x K/32 v_mfma_f32_16x16x32_bf16 run back to back and the epilogue (bias, tanh or the VAE head, the float32
// activation, the next layer's bf16 operand row with its ones column) is applied from the accumulators.
//
// Backward (the data-gradient chain between the decoder matrix's product and the encoder matrix's weight kernel): the same
// scheme on dA = D W with the TRANSPOSED compute copy W^T [in][outp] (kept for the hidden layers only, 0.7 MB; written by the
// optimizer wherever it writes the compute copy), so the contraction index is contiguous for both operands again; epilogues
// of k_post (backward: x (1 - o^2)) and k_vae_bwd.  Four launches of the chain become two.
//
// Operands: A [Bp][K] bf16 row-major (K-contiguous; column `in` = 1, beyond it 0), W [outp][K] bf16 compute copy (row n =
// output feature n; its column `in` is zero, so the ones column adds nothing in the forward product).
#include "rtx_kernels.h"
#include <algorithm>

typedef __attribute__((ext_vector_type(8))) __bf16 sf_bf16x8;
typedef __attribute__((ext_vector_type(4))) float sf_f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned sf_u32x4;

// fragment s of a 16-row operand: lane (r = lane & 15, g = lane >> 4) holds elements [r][32 s + 8 g .. + 8)
template <int KS>
__device__ __forceinline__ void sf_load(sf_u32x4 (&f)[KS], const bf16_t* __restrict__ base, int ld, int row0, int lane)
{
    const bf16_t* p = base + (size_t)(row0 + (lane & 15)) * ld + (lane >> 4) * 8;
#pragma unroll
    for (int s = 0; s < KS; ++s) f[s] = *(const sf_u32x4*)(p + s * 32);
}
template <int KS>
__device__ __forceinline__ sf_f32x4 sf_dot(const sf_u32x4 (&a)[KS], const sf_u32x4 (&b)[KS])
{
    sf_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KS; ++s)
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(sf_bf16x8, a[s]), __builtin_bit_cast(sf_bf16x8, b[s]), acc, 0, 0, 0);
    return acc;   // element i: row (lane >> 4) * 4 + i, column lane & 15
}

// K split over KW waves of a workgroup (round 6, an experiment kept as a knob).  The one-burst scheme costs registers in proportion to K:
// 188-284 per lane at K = 640.  Hypothesis: beside the decoder matrix's weight kernel a CU has only the registers that kernel's retiring
// workgroups free (240 per SIMD lane at a time), so smaller waves would start sooner.  With KW = 2 / 4 every wave loads and
// multiplies a half / quarter of the fragments (108-176 / 70-100 registers); wave kw > 0 parks its partial accumulators in LDS, wave 0 adds
// them in the fixed order kw = 1 .. KW - 1 and runs the epilogue (sums of several MFMA chains: last bits differ from KW = 1).  Result: no
// change in the step (numbers at g_small_kw below) -- the hypothesis is refuted at this granularity.
template <int NACC>
__device__ __forceinline__ void sf_combine(sf_f32x4 (&acc)[NACC], int kw, int KW, int row_wave, int lane, float* red)
{
    // red: [row_wave][kw - 1][NACC][64 lanes][4]
    if (KW == 1) return;
    if (kw > 0) {
#pragma unroll
        for (int t = 0; t < NACC; ++t) *(sf_f32x4*)(red + ((((size_t)row_wave * (KW - 1) + (kw - 1)) * NACC + t) * 64 + lane) * 4) = acc[t];
    }
    __syncthreads();
    if (kw == 0) {
        for (int k = 1; k < KW; ++k)
#pragma unroll
            for (int t = 0; t < NACC; ++t) acc[t] += *(const sf_f32x4*)(red + ((((size_t)row_wave * (KW - 1) + (k - 1)) * NACC + t) * 64 + lane) * 4);
    }
}

// ---- hidden layer: R / O32 [Bp][Np] = act(A W^T + bias), conventions of k_post (forward) -------------------------------
template <int KS, int KW>
__global__ __launch_bounds__(KW == 4 ? 512 : 256) void k_fwd_hidden(const RtxSmallFwdArgs a)
{
    constexpr int KH = KS / KW;
    extern __shared__ float sf_red[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, row_wave = wave / KW, kw = wave % KW;
    const int r0 = (blockIdx.y * ((blockDim.x >> 6) / KW) + row_wave) * 16, c0 = blockIdx.x * 32;
    sf_u32x4 fa[KH], fb0[KH], fb1[KH];
    sf_load<KH>(fa, a.A + kw * KH * 32, a.lda, r0, lane);
    sf_load<KH>(fb0, a.W + kw * KH * 32, a.ldw, c0, lane);
    sf_load<KH>(fb1, a.W + kw * KH * 32, a.ldw, c0 + 16, lane);
    __builtin_amdgcn_sched_barrier(0);   // every load is in flight before the first MFMA waits (the scheduler would interleave
                                         // them twelve at a time to save registers: a pipeline paced by the memory latency)
    sf_f32x4 acc[2] = {sf_dot<KH>(fa, fb0), sf_dot<KH>(fa, fb1)};
    sf_combine<2>(acc, kw, KW, row_wave, lane, sf_red);
    if (kw != 0) return;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int n = c0 + t * 16 + (lane & 15);
        const float bias = n < a.N_real ? a.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int b = r0 + (lane >> 4) * 4 + i;
            float x = 0.f;
            if (b < a.B && n < a.N_real) {
                x = acc[t][i] + bias;
                if (a.tanh_act) x = tanhf(x);
            }
            const size_t at = (size_t)b * a.Np + n;
            if (a.O32) a.O32[at] = x;
            a.R[at] = f32_to_bf16((b < a.B && n == a.N_real) ? 1.f : x);
        }
    }
}

// ---- VAE head: [mu | logvar] = A W^T + bias; z = mu + eps exp(logvar / 2) (eval: z = mu); conventions of k_vae_fwd -----
template <int KS, int KW>
__global__ __launch_bounds__(KW == 4 ? 512 : 256) void k_fwd_head(const RtxSmallFwdArgs a)
{
    constexpr int KH = KS / KW;
    extern __shared__ float sf_red[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, row_wave = wave / KW, kw = wave % KW;
    const int r0 = (blockIdx.y * ((blockDim.x >> 6) / KW) + row_wave) * 16, c0 = blockIdx.x * 16;
    const int j = c0 + (lane & 15);
    sf_f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};   // mu, logvar
    float eps[4] = {0.f, 0.f, 0.f, 0.f};
    const float bm = j < a.Z ? a.bias[j] : 0.f, bl = j < a.Z ? a.bias[a.Z + j] : 0.f;
    if (c0 < a.Z) {   // (a tile beyond the latent width only carries the ones column / zero padding of the next operand)
        sf_u32x4 fa[KH], fm[KH], fl[KH];
        sf_load<KH>(fa, a.A + kw * KH * 32, a.lda, r0, lane);
        sf_load<KH>(fm, a.W + kw * KH * 32, a.ldw, c0, lane);
        sf_load<KH>(fl, a.W + kw * KH * 32, a.ldw, a.Z + c0, lane);
        // the noise (Philox + Box-Muller, ~200 instructions per element) is drawn while the operands are in flight
        // (pure ALU here: a load in this stretch makes the compiler wait for ALL loads at the branch; injected noise is read below)
        if (kw == 0 && a.training && !a.eps_in) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int b = r0 + (lane >> 4) * 4 + i;
                eps[i] = rtx_normal(a.seed, a.offset, (uint64_t)b * a.Z + j);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        acc[0] = sf_dot<KH>(fa, fm);
        acc[1] = sf_dot<KH>(fa, fl);
    }
    sf_combine<2>(acc, kw, KW, row_wave, lane, sf_red);   // (every wave of the workgroup reaches the barrier: c0 is per workgroup)
    if (kw != 0) return;
    const sf_f32x4 mu = acc[0], lv = acc[1];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int b = r0 + (lane >> 4) * 4 + i;
        float z = 0.f;
        if (b < a.B && j < a.Z) {
            const float mm = mu[i] + bm, l = lv[i] + bl;
            if (a.training && a.eps_in) eps[i] = a.eps_in[(size_t)b * a.Z + j];
            z = a.training ? mm + eps[i] * expf(0.5f * l) : mm;
            const size_t o = (size_t)b * a.Z + j;
            a.mu32[o] = mm;
            a.lv32[o] = l;
            a.eps32[o] = eps[i];
            if (a.mu_out) a.mu_out[o] = mm;
            if (a.lv_out) a.lv_out[o] = l;
        }
        if (b < a.B && j == a.Z) z = 1.f;   // ones column -> bias gradient of the first decoder layer
        a.R[(size_t)b * a.Np + j] = f32_to_bf16(z);
    }
}

// ---- backward through a hidden layer: Dout [Bp][Np] = (D W)[b][n] x (1 - o[b][n]^2), conventions of k_post (backward) ------
template <int KS, int KW>
__global__ __launch_bounds__(KW == 4 ? 512 : 256) void k_bwd_hidden(const RtxSmallBwdArgs a)
{
    constexpr int KH = KS / KW;
    extern __shared__ float sf_red[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, row_wave = wave / KW, kw = wave % KW;
    const int r0 = (blockIdx.y * ((blockDim.x >> 6) / KW) + row_wave) * 16, c0 = blockIdx.x * 32;
    sf_u32x4 fa[KH], fb0[KH], fb1[KH];
    sf_load<KH>(fa, a.D + kw * KH * 32, a.ld, r0, lane);
    sf_load<KH>(fb0, a.WT + kw * KH * 32, a.ld, c0, lane);
    sf_load<KH>(fb1, a.WT + kw * KH * 32, a.ld, c0 + 16, lane);
    float o[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i)
            o[t][i] = (a.tanh_act && kw == 0) ? a.O32[(size_t)(r0 + (lane >> 4) * 4 + i) * a.Np + c0 + t * 16 + (lane & 15)] : 0.f;
    __builtin_amdgcn_sched_barrier(0);
    sf_f32x4 acc[2] = {sf_dot<KH>(fa, fb0), sf_dot<KH>(fa, fb1)};
    sf_combine<2>(acc, kw, KW, row_wave, lane, sf_red);
    if (kw != 0) return;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int n = c0 + t * 16 + (lane & 15);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int b = r0 + (lane >> 4) * 4 + i;
            const float x = (b < a.B && n < a.N_real) ? acc[t][i] * (1.f - o[t][i] * o[t][i]) : 0.f;
            a.Dout[(size_t)b * a.Np + n] = f32_to_bf16(x);
        }
    }
}

// ---- backward through the VAE head: dz = D W; Dout [Bp][Np] = [dmu | dlogvar | 0], conventions of k_vae_bwd ------------
template <int KS, int KW>
__global__ __launch_bounds__(KW == 4 ? 512 : 256) void k_bwd_head(const RtxSmallBwdArgs a)
{
    constexpr int KH = KS / KW;
    extern __shared__ float sf_red[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, row_wave = wave / KW, kw = wave % KW;
    const int r0 = (blockIdx.y * ((blockDim.x >> 6) / KW) + row_wave) * 16, c0 = blockIdx.x * 16;
    const int j = c0 + (lane & 15);
    sf_u32x4 fa[KH], fb[KH];
    sf_load<KH>(fa, a.D + kw * KH * 32, a.ld, r0, lane);
    sf_load<KH>(fb, a.WT + kw * KH * 32, a.ld, c0, lane);
    float mu[4] = {0.f, 0.f, 0.f, 0.f}, lv[4] = {0.f, 0.f, 0.f, 0.f}, ep[4] = {0.f, 0.f, 0.f, 0.f};
    if (kw == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {   // (clamped addresses, masked below: no branch around the loads)
            const size_t at = (size_t)min(r0 + (lane >> 4) * 4 + i, a.B - 1) * a.Z + min(j, a.Z - 1);
            mu[i] = a.mu32[at]; lv[i] = a.lv32[at]; ep[i] = a.eps32[at];
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    sf_f32x4 acc[1] = {sf_dot<KH>(fa, fb)};
    sf_combine<1>(acc, kw, KW, row_wave, lane, sf_red);
    if (kw != 0) return;
    const sf_f32x4 dz = acc[0];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int b = r0 + (lane >> 4) * 4 + i;
        bf16_t* out = a.Dout + (size_t)b * a.Np;
        if (j < a.Z) {
            float dm = 0.f, dl = 0.f;
            if (b < a.B) {
                dm = dz[i] + a.beta * mu[i] * a.inv_batch;
                dl = a.beta * 0.5f * (expf(lv[i]) - 1.f) * a.inv_batch;
                if (a.training) dl += dz[i] * ep[i] * 0.5f * expf(0.5f * lv[i]);
            }
            out[j] = f32_to_bf16(dm);
            out[a.Z + j] = f32_to_bf16(dl);
        }
        for (int n = 2 * a.Z + j; n < a.Np; n += gridDim.x * 16) out[n] = 0;   // the padding columns, shared out over the tiles
    }
}

bool rtx_small_fwd_ok(int K) { return K >= 128 && K <= 1024 && K % 128 == 0; }

// waves (16 batch rows each) per workgroup.  Rounds 3-5 launched 64-row workgroups (4 waves): a B = 500 layer was 104-160 workgroups, i.e.
// ~100 CUs each pulling 120-480 KB of fragments through a 64 B/clk L1 while the rest of the chip idled.  One wave per workgroup is 4x the
// workgroups: every CU takes a share (the B fragments the four waves used to share come from L2 either way).  Round 6, one box, alternating:
// 251.6 / 255.8 / 251.6 us per step with 4 waves, 248.3 / 251.7 / 247.0 with 1, 248.6 / 249.9 / 247.9 with 2 (knob "small_waves").
static int g_small_waves = 1;
void rtx_small_set_waves(int w) { g_small_waves = (w == 1 || w == 2 || w == 4) ? w : 1; }
// K split over this many waves per 16-row block (sf_combine; knob "small_kw").  Measured in round 6 (three alternating rounds, one box):
// 1: 251.4 / 248.1 / 246.6 us per step, 2: 249.1 / 248.3 / 246.5, 4: 247.3 / 247.9 / 255.4 -- the registers a wave needs are NOT what makes
// these kernels queue beside the weight kernel (188 -> 108 -> 70 changes nothing).  Default 1: one MFMA chain per output, the bits of rounds 3-5.
static int g_small_kw = 1;
void rtx_small_set_kw(int kw) { g_small_kw = (kw == 2 || kw == 4) ? kw : 1; }

template <int KS>
static void small_fwd_launch(const RtxSmallFwdArgs& a, hipStream_t stream)
{
    const int kw = g_small_kw, w = kw > 1 ? std::min(g_small_waves, 2) : g_small_waves;   // (256 threads at KW <= 2, 512 at KW = 4)
    const size_t lds = (size_t)w * (kw - 1) * 2 * 64 * 4 * sizeof(float);
    if (a.Z > 0) {
        if (kw == 4) hipLaunchKernelGGL((k_fwd_head<KS, 4>), dim3(a.Np / 16, a.Bp / (16 * w)), dim3(64 * w * 4), lds, stream, a);
        else if (kw == 2) hipLaunchKernelGGL((k_fwd_head<KS, 2>), dim3(a.Np / 16, a.Bp / (16 * w)), dim3(64 * w * 2), lds, stream, a);
        else hipLaunchKernelGGL((k_fwd_head<KS, 1>), dim3(a.Np / 16, a.Bp / (16 * w)), dim3(64 * w), 0, stream, a);
    } else {
        if (kw == 4) hipLaunchKernelGGL((k_fwd_hidden<KS, 4>), dim3(a.Np / 32, a.Bp / (16 * w)), dim3(64 * w * 4), lds, stream, a);
        else if (kw == 2) hipLaunchKernelGGL((k_fwd_hidden<KS, 2>), dim3(a.Np / 32, a.Bp / (16 * w)), dim3(64 * w * 2), lds, stream, a);
        else hipLaunchKernelGGL((k_fwd_hidden<KS, 1>), dim3(a.Np / 32, a.Bp / (16 * w)), dim3(64 * w), 0, stream, a);
    }
}

template <int KS>
static void small_bwd_launch(const RtxSmallBwdArgs& a, hipStream_t stream)
{
    const int kw = g_small_kw, w = kw > 1 ? std::min(g_small_waves, 2) : g_small_waves;
    const size_t lds = (size_t)w * (kw - 1) * 2 * 64 * 4 * sizeof(float);
    if (a.Z > 0) {
        if (kw == 4) hipLaunchKernelGGL((k_bwd_head<KS, 4>), dim3((a.Z + 15) / 16, a.Bp / (16 * w)), dim3(64 * w * 4), lds, stream, a);
        else if (kw == 2) hipLaunchKernelGGL((k_bwd_head<KS, 2>), dim3((a.Z + 15) / 16, a.Bp / (16 * w)), dim3(64 * w * 2), lds, stream, a);
        else hipLaunchKernelGGL((k_bwd_head<KS, 1>), dim3((a.Z + 15) / 16, a.Bp / (16 * w)), dim3(64 * w), 0, stream, a);
    } else {
        if (kw == 4) hipLaunchKernelGGL((k_bwd_hidden<KS, 4>), dim3(a.Np / 32, a.Bp / (16 * w)), dim3(64 * w * 4), lds, stream, a);
        else if (kw == 2) hipLaunchKernelGGL((k_bwd_hidden<KS, 2>), dim3(a.Np / 32, a.Bp / (16 * w)), dim3(64 * w * 2), lds, stream, a);
        else hipLaunchKernelGGL((k_bwd_hidden<KS, 1>), dim3(a.Np / 32, a.Bp / (16 * w)), dim3(64 * w), 0, stream, a);
    }
}

int rtx_launch_small_bwd(const RtxSmallBwdArgs& a, hipStream_t stream)
{
    RTX_CHECK(rtx_small_fwd_ok(a.ld), RTX_EINVAL, "small_bwd: output width %d not in {128, 256, .. 1024}", a.ld);
    RTX_CHECK(a.Bp % 64 == 0 && a.Np % 32 == 0 && a.Dout && a.B >= 1, RTX_EINVAL, "small_bwd: bad padding");
    RTX_CHECK(a.Z > 0 ? ((a.Z + 15) & ~15) <= a.wt_rows : a.Np <= a.wt_rows, RTX_EINVAL, "small_bwd: the transposed weight copy has %d rows", a.wt_rows);
    switch (a.ld / 128) {
        case 1: small_bwd_launch<4>(a, stream); break;
        case 2: small_bwd_launch<8>(a, stream); break;
        case 3: small_bwd_launch<12>(a, stream); break;
        case 4: small_bwd_launch<16>(a, stream); break;
        case 5: small_bwd_launch<20>(a, stream); break;
        case 6: small_bwd_launch<24>(a, stream); break;
        case 7: small_bwd_launch<28>(a, stream); break;
        default: small_bwd_launch<32>(a, stream); break;
    }
    RTX_HIP(hipGetLastError());
    return RTX_OK;
}

int rtx_launch_small_fwd(const RtxSmallFwdArgs& a, hipStream_t stream)
{
    RTX_CHECK(rtx_small_fwd_ok(a.lda) && a.ldw == a.lda, RTX_EINVAL, "small_fwd: input width %d not in {128, 256, .. 1024}", a.lda);
    RTX_CHECK(a.Bp % 64 == 0 && a.Np % 32 == 0 && a.R, RTX_EINVAL, "small_fwd: bad padding");
    RTX_CHECK(a.Z > 0 ? a.Z + ((a.Z + 15) & ~15) <= a.w_rows : a.Np <= a.w_rows, RTX_EINVAL, "small_fwd: the weight copy has %d rows", a.w_rows);
    switch (a.lda / 128) {
        case 1: small_fwd_launch<4>(a, stream); break;
        case 2: small_fwd_launch<8>(a, stream); break;
        case 3: small_fwd_launch<12>(a, stream); break;
        case 4: small_fwd_launch<16>(a, stream); break;
        case 5: small_fwd_launch<20>(a, stream); break;
        case 6: small_fwd_launch<24>(a, stream); break;
        case 7: small_fwd_launch<28>(a, stream); break;
        default: small_fwd_launch<32>(a, stream); break;
    }
    RTX_HIP(hipGetLastError());
    return RTX_OK;
}
