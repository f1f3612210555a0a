// Native (no Python) check of the sparse first layer (spmm_in.hip: k_in_chunks + k_spmm_in) against a double-precision host
// loop over the same stored entries, the same Philox dropout decisions and the same bf16-rounded values, plus a timing of
// the pair at the ml-20m shape with the weight matrix rotating through more copies than the last-level cache holds.
#include "../../rectorch_amd/csrc/rtx_kernels.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

void rtx_set_error(const char* fmt, ...);
const char* rtx_last_error_str();

static uint32_t rng_state = 777;
static uint32_t urand()
{
    rng_state = rng_state * 1664525u + 1013904223u;
    return rng_state >> 8;
}
static float frand() { return (urand() * (1.0f / 16777216.0f)) * 2.f - 1.f; }

#define CK(x)                                                                            \
    do {                                                                                 \
        hipError_t e = (x);                                                              \
        if (e != hipSuccess) {                                                           \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
            exit(2);                                                                     \
        }                                                                                \
    } while (0)
#define RT(x)                                                                  \
    do {                                                                       \
        if ((x) != RTX_OK) {                                                   \
            printf("launch failed at %s:%d: %s\n", __FILE__, __LINE__, rtx_last_error_str()); \
            exit(2);                                                           \
        }                                                                      \
    } while (0)

template <typename T>
static T* to_dev(const std::vector<T>& h)
{
    T* d = nullptr;
    CK(hipMalloc(&d, std::max<size_t>(h.size(), 1) * sizeof(T)));
    if (!h.empty()) CK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return d;
}

struct Case {
    const char* name;
    int B, I, cond, N_real, n_rows;
    bool values, ids, training, tanh_act;
    int long_rows;   // rows of several thousand entries
};

static int run_case(const Case& c, int perf_iters)
{
    const int Iin = c.I + c.cond, Bp = (c.B + 127) / 128 * 128, Np = (c.N_real + 1 + 127) / 128 * 128;
    const int ldw = (Iin + 1 + 127) / 128 * 128;
    // the matrix: log-normal row lengths, a few empty rows, a few very long ones
    std::vector<int64_t> indptr(c.n_rows + 1, 0);
    std::vector<int32_t> indices;
    std::vector<float> values;
    int longest = 0;
    for (int r = 0; r < c.n_rows; ++r) {
        int len = (int)expf(4.3f + 1.0f * frand() * 1.7f);
        if (r % 97 == 5) len = 0;
        if (r % 61 == 7 && c.long_rows) len = 2000 + (int)(urand() % 3000);
        len = std::min(len, c.I);
        std::vector<int32_t> row;
        // distinct sorted items: walk the columns with random strides
        int col = (int)(urand() % std::max(1, c.I / std::max(len, 1)));
        for (int k = 0; k < len && col < c.I; ++k) {
            row.push_back(col);
            col += 1 + (int)(urand() % std::max(1, 2 * (c.I - col) / std::max(1, len - k) - 1));
        }
        if (c.cond && r % 3 != 0) row.push_back(c.I + (int)(urand() % c.cond));   // one condition column, raw
        for (int32_t i : row) {
            indices.push_back(i);
            if (c.values) values.push_back(0.5f * (1 + (int)(urand() % 10)));
        }
        indptr[r + 1] = (int64_t)indices.size();
        longest = std::max(longest, (int)row.size());
    }
    std::vector<int32_t> ids(c.B);
    for (int b = 0; b < c.B; ++b) ids[b] = c.ids ? (int32_t)(urand() % c.n_rows) : b;
    if (c.long_rows) ids[c.B - 1] = 7;   // a long row closes the batch: the last part of the split is one user
    std::vector<bf16_t> W((size_t)Np * ldw, 0);
    for (int o = 0; o < c.N_real; ++o)
        for (int i = 0; i < Iin; ++i) W[(size_t)o * ldw + i] = f32_to_bf16(0.05f * frand());
    std::vector<float> bias(c.N_real);
    for (auto& x : bias) x = 0.1f * frand();

    RtxCsrView v = {};
    v.indptr = to_dev(indptr); v.indices = to_dev(indices); v.values = c.values ? to_dev(values) : nullptr;
    v.row_ids = c.ids || c.long_rows ? to_dev(ids) : nullptr;
    v.max_row_len = longest;
    const int64_t cap = (int64_t)Bp * std::max(1, (longest + 63) / 64) + 64;
    uint32_t* ent; int32_t *desc, *wsplit;
    CK(hipMalloc(&ent, cap * 256)); CK(hipMemset(ent, 0, cap * 256));
    CK(hipMalloc(&desc, (cap + 128) * 4)); CK(hipMemset(desc, 0, (cap + 128) * 4));
    CK(hipMalloc(&wsplit, 17 * 4));
    const int NW = perf_iters ? 12 : 1;   // 12 x 26 MB of weights: more than the 256 MB last-level cache
    std::vector<bf16_t*> dW(NW);
    for (auto& p : dW) p = to_dev(W);
    float* dbias = to_dev(bias);
    float* O32; bf16_t* R;
    CK(hipMalloc(&O32, (size_t)Bp * Np * 4)); CK(hipMemset(O32, 0xff, (size_t)Bp * Np * 4));
    CK(hipMalloc(&R, (size_t)Bp * Np * 2)); CK(hipMemset(R, 0xff, (size_t)Bp * Np * 2));

    RtxInChunksArgs ca = {};
    ca.in = v; ca.B = c.B; ca.I = c.I; ca.Iin = Iin; ca.training = c.training; ca.dropout_p = 0.5f;
    ca.seed = 1234; ca.offset = 99; ca.ent = ent; ca.desc = desc; ca.wsplit = wsplit;
    RtxSpmmInArgs sa = {};
    sa.ent = ent; sa.desc = desc; sa.wsplit = wsplit; sa.B = c.B; sa.Bp = Bp; sa.W = dW[0]; sa.ldw = ldw; sa.Kin = Iin;
    sa.bias = dbias; sa.N_real = c.N_real; sa.Np = Np; sa.tanh_act = c.tanh_act; sa.O32 = O32; sa.R = R; sa.ones_col = 1;
    // training: the same launch also writes the dense image of the rows and the target row sums (what k_gather would)
    bf16_t* X = nullptr; float* tsum = nullptr;
    if (c.training) {
        CK(hipMalloc(&X, (size_t)Bp * ldw * 2)); CK(hipMemset(X, 0xff, (size_t)Bp * ldw * 2));
        CK(hipMalloc(&tsum, Bp * 4)); CK(hipMemset(tsum, 0xff, Bp * 4));
        ca.target = v; ca.tsum = tsum; ca.X = X; ca.ldx = ldw; ca.Bp = Bp;
    }
    RT(rtx_launch_in_chunks(ca, 0));
    RT(rtx_launch_spmm_in(sa, 0));
    CK(hipDeviceSynchronize());

    std::vector<float> o32((size_t)Bp * Np);
    std::vector<bf16_t> r((size_t)Bp * Np);
    std::vector<int32_t> ws(17);
    CK(hipMemcpy(o32.data(), O32, o32.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(r.data(), R, r.size() * 2, hipMemcpyDeviceToHost));
    CK(hipMemcpy(ws.data(), wsplit, 17 * 4, hipMemcpyDeviceToHost));
    // the split: monotone, from 0 to the number of chunks
    int bad = 0;
    int total = 0;
    for (int b = 0; b < c.B; ++b) {
        const int64_t u = v.row_ids ? ids[b] : b;
        total += std::max(1, (int)((indptr[u + 1] - indptr[u] + 63) / 64));
    }
    if (ws[0] != 0 || ws[16] != total) { printf("  split ends %d..%d, expected 0..%d\n", ws[0], ws[16], total); ++bad; }
    for (int w = 0; w < 16; ++w) if (ws[w] > ws[w + 1]) { printf("  split not monotone at %d\n", w); ++bad; }
    std::vector<bf16_t> hx;
    std::vector<float> hts;
    if (X) {
        hx.resize((size_t)Bp * ldw); hts.resize(Bp);
        CK(hipMemcpy(hx.data(), X, hx.size() * 2, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hts.data(), tsum, Bp * 4, hipMemcpyDeviceToHost));
    }
    // host reference
    double worst = 0;
    const float scale = c.training ? 2.f : 1.f;
    for (int b = 0; b < Bp; ++b) {
        std::vector<double> h(c.N_real, 0.0);
        std::vector<bf16_t> xrow(X ? ldw : 0, 0);
        double ts = 0;
        if (b < c.B) {
            const int64_t u = v.row_ids ? ids[b] : b;
            double ss = 0;
            for (int64_t k = indptr[u]; k < indptr[u + 1]; ++k)
                if (indices[k] < c.I) { const float x = c.values ? values[k] : 1.f; ss += (double)x * x; }
            const float inv = 1.f / fmaxf(sqrtf((float)ss), 1e-12f);
            for (int64_t k = indptr[u]; k < indptr[u + 1]; ++k) {
                const int i = indices[k];
                float x = c.values ? values[k] : 1.f;
                if (i < c.I) x *= inv;
                if (c.training && i < c.I)
                    x = rtx_dropout_keep(1234, 99, (uint64_t)b * c.I + i, 0.5f) ? x * scale : 0.f;
                if (X) xrow[i] = f32_to_bf16(x);
                if (i < c.I) ts += c.values ? values[k] : 1.f;
                const double xv = bf16_to_f32(f32_to_bf16(x));
                if (xv != 0)
                    for (int o = 0; o < c.N_real; ++o) h[o] += xv * bf16_to_f32(W[(size_t)o * ldw + i]);
            }
        }
        if (X) {
            if (b < c.B) xrow[Iin] = f32_to_bf16(1.f);
            for (int i = 0; i < ldw; ++i)
                if (hx[(size_t)b * ldw + i] != xrow[i]) { if (bad < 8) printf("  X[%d][%d] = %04x, expected %04x\n", b, i, hx[(size_t)b * ldw + i], xrow[i]); ++bad; }
            if (fabs(hts[b] - ts) > 1e-3 * (1 + fabs(ts))) { if (bad < 8) printf("  tsum[%d] = %g, expected %g\n", b, hts[b], ts); ++bad; }
        }
        for (int n = 0; n < Np; ++n) {
            double want = 0;
            if (b < c.B && n < c.N_real) { want = h[n] + bias[n]; if (c.tanh_act) want = tanh(want); }
            const float got = o32[(size_t)b * Np + n];
            const double d = fabs(got - want);
            if (!(d <= 2e-5 + 1e-5 * fabs(want))) { if (bad < 8) printf("  O32[%d][%d] = %g, expected %g\n", b, n, got, want); ++bad; }
            worst = std::max(worst, d);
            const bf16_t rw = (b < c.B && n == c.N_real) ? f32_to_bf16(1.f) : f32_to_bf16(got);
            if (r[(size_t)b * Np + n] != rw) { if (bad < 8) printf("  R[%d][%d] = %04x, expected %04x\n", b, n, r[(size_t)b * Np + n], rw); ++bad; }
        }
    }
    printf("%-34s B=%d I=%d+%d N=%d chunks=%d longest=%d  max|err|=%.2e  %s\n", c.name, c.B, c.I, c.cond, c.N_real, total, longest, worst,
           bad ? "FAIL" : "ok");
    if (perf_iters && !bad) {
        hipEvent_t e0, e1, e2;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
        float t_prep = 0, t_spmm = 0;
        for (int it = -3; it < perf_iters; ++it) {
            sa.W = dW[(it + 3) % NW];
            ca.offset = 100 + it;
            CK(hipEventRecord(e0, 0));
            RT(rtx_launch_in_chunks(ca, 0));
            CK(hipEventRecord(e1, 0));
            RT(rtx_launch_spmm_in(sa, 0));
            CK(hipEventRecord(e2, 0));
            CK(hipEventSynchronize(e2));
            float a, b2;
            CK(hipEventElapsedTime(&a, e0, e1)); CK(hipEventElapsedTime(&b2, e1, e2));
            if (it >= 0) { t_prep += a; t_spmm += b2; }
        }
        // back to back, no events in between
        CK(hipEventRecord(e0, 0));
        for (int it = 0; it < perf_iters; ++it) {
            sa.W = dW[it % NW];
            RT(rtx_launch_in_chunks(ca, 0));
            RT(rtx_launch_spmm_in(sa, 0));
        }
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float both; CK(hipEventElapsedTime(&both, e0, e1));
        {   // where the time of one launch goes: shader-clock stamps of the first and last wave of every workgroup
            const int grid = (Np + 3) / 4;
            uint64_t* st;
            CK(hipMalloc(&st, (size_t)grid * 8 * 8)); CK(hipMemset(st, 0, (size_t)grid * 8 * 8));
            sa.stamps = st; sa.W = dW[1 % NW];
            RT(rtx_launch_in_chunks(ca, 0));
            RT(rtx_launch_spmm_in(sa, 0));
            CK(hipDeviceSynchronize());
            sa.stamps = nullptr;
            std::vector<uint64_t> h((size_t)grid * 8);
            CK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
            uint64_t first = ~0ull, last = 0;
            double ph[3] = {0, 0, 0}, mx[3] = {0, 0, 0};
            int n = 0;
            for (int w = 0; w < grid * 2; ++w) {
                const uint64_t* t = &h[(size_t)w * 4];
                if (!t[0] || !t[2]) continue;   // a padding-column workgroup
                first = std::min(first, t[0]); last = std::max(last, t[3]);
                for (int k = 0; k < 3; ++k) { const double d = (double)(t[k + 1] - t[k]); ph[k] += d; mx[k] = std::max(mx[k], d); }
                ++n;
            }
            printf("    stamps (ticks; %d waves): stage mean %.0f max %.0f | sum mean %.0f max %.0f | tail mean %.0f max %.0f | first start -> last end %.0f\n",
                   n, ph[0] / n, mx[0], ph[1] / n, mx[1], ph[2] / n, mx[2], (double)(last - first));
        }
        printf("    perf: in_chunks %.1f us, spmm_in %.1f us (event-bracketed); pair back to back %.1f us\n", 1e3 * t_prep / perf_iters,
               1e3 * t_spmm / perf_iters, 1e3 * both / perf_iters);
    }
    return bad;
}

int main()
{
    int bad = 0;
    const Case cases[] = {
        {"ml-20m shape, train, binary", 500, 20108, 0, 600, 4000, false, true, true, true, 1},
        {"eval, ratings, natural rows", 130, 5000, 0, 70, 130, true, false, false, true, 0},
        {"conditioned, ratings, train", 257, 3001, 16, 33, 900, true, true, true, true, 1},
        {"one user", 1, 777, 0, 8, 5, false, false, true, false, 0},
        {"linear, short rows", 64, 400, 0, 600, 64, false, false, true, false, 0},
    };
    for (const Case& c : cases) bad += run_case(c, 0) != 0;
    bad += run_case(cases[0], 40) != 0;
    printf(bad ? "SPMM TESTS FAILED (%d cases)\n" : "SPMM TESTS PASSED\n", bad);
    return bad ? 1 : 0;
}
