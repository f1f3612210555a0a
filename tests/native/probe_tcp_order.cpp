// probe_tcp_order.cpp -- does a wave's L2-HIT load wait behind ANOTHER wave's HBM-miss loads on the same CU?  (round 6)
//
// Why: a weight-gradient + Adam workgroup spends 8.4 of its 20 us in a K walk of eight slices (~1 us per slice) although every operand
// slice is an L2 hit two slices ahead -- while the same waves (and the CU's other workgroup) have 48-96 KB of optimizer state in flight
// from HBM.  If the CU's vector memory pipe returns data in request order ACROSS waves, no assignment of roles to waves can decouple
// the two; if only within a wave, "matrix" waves without state loads would walk K at L2 speed.
//
// One workgroup of two waves per CU.  Wave 0 chases a pointer chain through a 256-KB ring that lives in L2 (each load depends on the
// one before: the time per load IS the latency), wave 1 -- by mode --
//   A  idles
//   B  streams a cold 64-MB region with 8 x 16-byte loads per lane in flight (HBM misses), on EVERY CU
//   C  the same, but only on odd-numbered workgroups while the chain is timed on even ones (same fabric load per streaming CU, no
//      streaming wave on the timed CU)
//   D  as B, but the streaming loads are issued by wave 0 ITSELF between two links of its chain (the in-order case by construction)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef __attribute__((ext_vector_type(4))) float f4;

__global__ __launch_bounds__(1024) void k_probe(const unsigned* __restrict__ ring, int ring_words, const f4* __restrict__ cold, size_t cold_per_wg, int mode, int links,
                                               int stream_iters, unsigned long long* out, float* sink)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned* myring = ring + (size_t)blockIdx.x * ring_words;
    const f4* mycold = cold + (size_t)blockIdx.x * cold_per_wg;
    const int n_str = (int)(blockDim.x >> 6) - 1;          // streaming waves of this workgroup (each walks its own part of the region)
    const bool streamer_cu = mode == 1 || mode == 3 || (mode == 2 && (blockIdx.x & 1));
    const bool timed_cu = mode != 2 || !(blockIdx.x & 1);
    __shared__ volatile int done;
    if (threadIdx.x == 0) done = 0;
    __syncthreads();
    if (wave == 0) {
        if (!timed_cu) return;
        unsigned idx = lane;       // 64 independent chains, one per lane (a wave instruction = 64 scattered 4-byte loads of the ring)
        f4 acc = {0.f, 0.f, 0.f, 0.f};
        size_t cpos = lane;
        // warm the ring into L2
        for (int i = 0; i < 2 * ring_words / 64; ++i) idx = myring[idx];
        const unsigned long long t0 = __builtin_readcyclecounter();
        for (int i = 0; i < links; ++i) {
            if (mode == 3) {       // the wave's own streaming loads in front of the link
#pragma unroll
                for (int u = 0; u < 4; ++u) { acc += __builtin_nontemporal_load(mycold + cpos); cpos += 64; if (cpos >= cold_per_wg) cpos = lane; }
            }
            idx = myring[idx];     // depends on the previous link
        }
        const unsigned long long t1 = __builtin_readcyclecounter();
        if (lane == 0) { out[blockIdx.x] = (t1 - t0); done = 1; }
        if (idx == 0xffffffffu || acc.x == 12345.f) sink[0] = acc.x + acc.y + acc.z + acc.w;
    } else {
        if (!streamer_cu || mode == 3) return;
        f4 acc = {0.f, 0.f, 0.f, 0.f};
        const size_t part = cold_per_wg / (size_t)n_str / 64 * 64;
        mycold += (size_t)(wave - 1) * part;
        size_t pos = lane;
        // a BOUNDED stream (8 KB per iteration; ~0.3 us each at a CU's share of HBM): long enough to outlast the chain, and a workgroup
        // whose own chain has finished stops early (the flag is in its LDS; other workgroups are never waited for)
        for (int it = 0; it < stream_iters; ++it) {
            f4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { v[u] = __builtin_nontemporal_load(mycold + pos); pos += 64; if (pos >= part) pos = lane; }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
            if (timed_cu && done) break;
        }
        if (acc.x == 12345.f) sink[1] = acc.x + acc.y + acc.z + acc.w;
    }
}

int main()
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int n_wg = prop.multiProcessorCount, ring_words = 65536 /* 256 KB */, links = 2000;
    const size_t cold_per_wg = (size_t)(64 << 20) / n_wg / 16 * 16;   // f4 elements per workgroup of a 1-GB region... (64 MB x 16 B)
    std::vector<unsigned> h((size_t)ring_words);
    // a random cyclic permutation per lane class: idx -> next, every link far from the previous one (different 128-byte lines)
    std::vector<unsigned> perm(ring_words);
    for (int i = 0; i < ring_words; ++i) perm[i] = i;
    srand(1);
    for (int i = ring_words - 1; i > 0; --i) { int j = rand() % (i + 1); std::swap(perm[i], perm[j]); }
    for (int i = 0; i < ring_words; ++i) h[perm[i]] = perm[(i + 1) % ring_words];
    unsigned* ring; f4* cold; unsigned long long* out; float* sink;
    CK(hipMalloc(&ring, (size_t)n_wg * ring_words * 4));
    for (int w = 0; w < n_wg; ++w) CK(hipMemcpy(ring + (size_t)w * ring_words, h.data(), (size_t)ring_words * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&cold, cold_per_wg * n_wg * sizeof(f4)));
    CK(hipMemset(cold, 0, cold_per_wg * n_wg * sizeof(f4)));
    CK(hipMalloc(&out, n_wg * 8)); CK(hipMalloc(&sink, 16));
    const char* names[4] = {"A  chain alone", "B  a streaming wave beside it on every CU", "C  streaming waves on the OTHER CUs only", "D  the chain's own wave streams"};
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 4; ++mode)
            for (int n_str : {1, 3, 7, 15}) {        // streaming waves beside the chain's wave (8 KB in flight each)
                if ((mode == 0 || mode == 3) && n_str != 1) continue;
                CK(hipMemset(out, 0, n_wg * 8));
                hipLaunchKernelGGL(k_probe, dim3(n_wg), dim3(64 * (1 + n_str)), 0, 0, ring, ring_words, cold, cold_per_wg, mode, links, 12000, out, sink);
                CK(hipDeviceSynchronize());
                std::vector<unsigned long long> r(n_wg);
                CK(hipMemcpy(r.data(), out, n_wg * 8, hipMemcpyDeviceToHost));
                double s = 0; int n = 0;
                for (int w = 0; w < n_wg; ++w) if (r[w]) { s += (double)r[w] / links; ++n; }
                printf("[tcp-order] %-46s x %2d streaming wave(s): %7.0f shader cycles per dependent L2 load (mean over %d CUs)\n", names[mode], (mode == 0 || mode == 3) ? 0 : n_str, s / n, n);
            }
    return 0;
}
