// ltr_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels for the pytorchltr
// ranking-loss / ranking-metric hot path, behind the C ABI of include/ltr_hip.h.
//
// Design:
//   * one workgroup per query; the query's (score, label) row is staged ONCE into LDS,
//     int64 labels are narrowed to fp32 in registers on the way in;
//   * the O(n^2) pair expansion of the reference (utils/tensor_operations.py:94-119) is
//     never materialised: a lane OWNS document(s) k (score, label, accumulators in VGPRs)
//     and streams every other document m of the query from LDS (wave-uniform address ->
//     LDS broadcast, conflict-free).  Each unordered pair is visited from both ends, so a
//     lane accumulates d loss / d s_k without atomics or cross-lane traffic;
//   * the reference's n-mask (loss/pairwise_additive.py:75-81) is the loop bound m < n[b],
//     k < n[b]: padded documents are neither loaded nor visited;
//   * rank positions (Lambda losses, metrics) come from an O(n^2) counting rank in the
//     same owner/stream structure -- exact, deterministic, no sort network;
//   * per-query reductions: DPP/shuffle wave reduction + one LDS hop across waves.
//
// Written for gfx950 only: wave size 64 is hard-coded.
#include <mutex>

#include "ltr_common.inc"

namespace {

// The device status page (see ltr_common.inc): ONE pinned, device-mapped host page per process, created
// under a mutex on the first multi-workgroup launch outside stream capture (two host threads may make
// their first launch concurrently); its device-side address is looked up once per device.
struct StatusPage { int *host; int *dev[kMaxDevices]; bool tried; };
inline StatusPage &status_page_ref() { static StatusPage sp = {}; return sp; }
inline std::mutex &status_mutex() { static std::mutex m; return m; }
inline int *status_device_ptr_impl(hipStream_t stream)
{
    StatusPage &sp = status_page_ref();
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= kMaxDevices) return nullptr;
    if (int *d = __atomic_load_n(&sp.dev[dev], __ATOMIC_ACQUIRE)) return d;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
        (void)hipGetLastError();
        return nullptr;
    }
    std::lock_guard<std::mutex> lock(status_mutex());
    if (sp.dev[dev]) return sp.dev[dev];
    if (!sp.host) {
        if (sp.tried) return nullptr;
        sp.tried = true;
        void *h = nullptr;
        if (hipHostMalloc(&h, 64, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        for (int i = 0; i < 16; ++i) reinterpret_cast<volatile int *>(h)[i] = 0;
        __atomic_store_n(&sp.host, reinterpret_cast<int *>(h), __ATOMIC_RELEASE);
    }
    void *d = nullptr;
    if (hipHostGetDevicePointer(&d, sp.host, 0) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    __atomic_store_n(&sp.dev[dev], reinterpret_cast<int *>(d), __ATOMIC_RELEASE);
    return sp.dev[dev];
}
inline int status_peek_impl()
{
    int *h = __atomic_load_n(&status_page_ref().host, __ATOMIC_ACQUIRE);
    return h ? *reinterpret_cast<volatile int *>(h) : 0;
}

}  // namespace

extern "C" __attribute__((visibility("hidden"))) int *ltr_internal_status_device_ptr(void *stream)
{
    return status_device_ptr_impl((hipStream_t)stream);
}
extern "C" __attribute__((visibility("hidden"))) int ltr_internal_status_peek(void) { return status_peek_impl(); }

namespace {

// NW > 0 (symmetric pass only): the workgroup has NW waves, known at compile time (1 / 2 / 4 / 8 / 16 by batch size and
// list length) -- the pair pass then needs no dispatch-packet read, no divisions and one barrier less.
template <int KIND, int DPT, int NW>
__device__ __forceinline__ void pairwise_loss_body(const LossParams &p);

template <int KIND, int DPT, int NW = 0>
__global__ void __launch_bounds__(NW > 0 ? NW * 64 : 1024)
pairwise_loss_kernel(LossParams p)
{
    pairwise_loss_body<KIND, DPT, NW>(p);
}

// The LambdaNDCG kinds' symmetric pass, left to itself, takes 93-96 SGPRs at 34-42 VGPRs: SEVEN waves per SIMD (the SGPR cliff,
// DESIGN 4.6 -- more than 80 SGPRs + the trap handler's 16 do not fit eight times into a SIMD's 800).  Capped (the attribute wants
// a literal: an entry point of its own around the same body).
template <int KIND, int NW>
__global__ void __launch_bounds__(NW * 64) __attribute__((amdgpu_num_sgpr(80)))
pairwise_loss_kernel80(LossParams p)
{
    pairwise_loss_body<KIND, 0, NW>(p);
}

template <int KIND, int DPT, int NW>
__device__ __forceinline__ void pairwise_loss_body(const LossParams &p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int L = p.L;
    const int tid = threadIdx.x;
    const int T = NW > 0 ? NW * 64 : (int)blockDim.x;
    int b, nb;
    if (p.sched) {
        b = sched_query_sampled(p.n, p.B, L, p.sched, tid, nb, (int)blockIdx.x);
    } else {
        b = (int)blockIdx.x;
        nb = clamp_n(p.n[b], L);
    }
    constexpr bool kSym = (DPT == 0);              // DPT == 0 selects the symmetric pair pass
    // LDS row stride: list_len rounded to 4 (both-ends pass) or to whole 64-wide tiles (symmetric)
    const int L4 = kSym ? ((L + 63) & ~63) : ((L + 3) & ~3);
    const int msplit = kSym ? (T >> 6) : p.msplit; // gradient slices: per wave / per m-slice
    const QueryLds q = carve_query_lds<KIND>(smem, L4, msplit);
#if defined(LTR_DEBUG_STOP) && LTR_DEBUG_STOP == 0
    if (p.B >= 0) return;                        // tuning: launch + dispatch floor
#endif

    // ---- stage this query's (score, label) row: coalesced 4 B / 8 B per lane.  The loads do
    // not wait for n[b] (they are issued for the whole row and only the first n[b] entries
    // are kept), so the n, score and label fetches overlap instead of chaining latencies ----
    const size_t row = (size_t)b * L;
    stage_rows(q.sy, p.scores + row, p.rel, p.rel_dtype, row, L, nb, tid, T);
    if (KIND == LTR_NDCG1 || KIND == LTR_NDCG2)
        for (int m = tid; m < 2 * L4; m += T) q.rank_s[m] = 0;
    __syncthreads();
#if defined(LTR_DEBUG_STOP) && LTR_DEBUG_STOP == 1
    if (p.B >= 0) { if (tid == 0) p.loss[b] = q.sy[0].x; return; }   // tuning: + staging
#endif

    float gscale, gsum;
    float total;
    constexpr bool kDefer = NW > 0 && NW <= 8;     // 1 / maxDCG applied once, after the pair pass
    if constexpr (kSym) {
        if (KIND == LTR_NDCG1 || KIND == LTR_NDCG2) {
            int owners = 64;
            while (owners < L4 && owners < T) owners *= 2;
            const int ms = T / owners;
            const int mlen = (nb + ms - 1) / ms;
            const int m0 = __builtin_amdgcn_readfirstlane(min(nb, (tid / owners) * mlen));
            const int m1 = __builtin_amdgcn_readfirstlane(min(nb, m0 + mlen));
            prepare_ndcg<KIND, 1, NW, kDefer>(q, nb, owners, tid % owners, m0, m1, ms > 1);
        }
        const bool intlab = (KIND == LTR_HINGE || KIND == LTR_DCG_HINGE) && p.rel_dtype != LTR_LABEL_F32;
        total = intlab ? pairwise_core_sym<KIND, NW, kDefer, true>(q, nb, L4, p.sigma, gscale)
                       : pairwise_core_sym<KIND, NW, kDefer, false>(q, nb, L4, p.sigma, gscale);
        gsum = 0.f;
    } else {
        total = pairwise_core<KIND, (DPT > 0 ? DPT : 1)>(q, nb, L4, msplit, p.sigma, gscale, gsum);
    }
    (void)gsum;
#if defined(LTR_DEBUG_STOP) && LTR_DEBUG_STOP == 2
    if (p.B >= 0) { if (tid == 0) p.loss[b] = total; return; }       // tuning: + pair pass
#endif

    if (tid == 0) p.loss[b] = total;
    if (p.dscores != nullptr) {
        for (int k = tid; k < L; k += T) {
            float g = 0.f;
            if (k < nb) {
                for (int s = 0; s < msplit; ++s) g += q.gpart[(size_t)s * L4 + k];
                g *= gscale;
            }
            p.dscores[row + k] = g;
        }
    }
}

// ---------------------------------------------------------------------------------
// backward: out[b, j] = grad_out[b] * dscores[b, j]
// ---------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------
// Split-query launch for long lists on small batches.  One workgroup per query leaves the time of
// a launch to its longest query (C4: 256 queries on 256 CUs, n = 1000 takes 55 us, the mean 18);
// here up to `nsplit` workgroups share one query's pair units (the same unit space the waves of
// one workgroup share in pairwise_core_sym), each writes its raw pair sum and its gradient
// slice to the workspace, and a finish kernel adds the parts in order, applies the loss modifier
// and scales the gradient.  Parts beyond what a short query can use exit at once.
// workspace: float loss_part[B][nsplit], then float grad_part[B][nsplit][L].
// ---------------------------------------------------------------------------------
// Rankings of the NDCG kinds for the split-query launch, once per query (the parts of a query
// would each repeat two 1024-element sorts): one 1024-thread workgroup stages the row, runs
// prepare_ndcg and leaves per document (a_k, 0) for LambdaNDCG1 or (G_k / maxDCG, rank_k) for
// LambdaNDCG2 in the workspace.
template <int KIND>
__global__ void __launch_bounds__(1024)
ndcg_prepare_kernel(LossParams p, float2 *prep)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int b = blockIdx.x;
    const int L = p.L;
    const int tid = threadIdx.x;
    const int T = blockDim.x;
    const int L4 = (L + 63) & ~63;
    const int nb = clamp_n(p.n[b], L);
    // (no gradient slices here: beyond 1024 documents the region behind the rank arrays is sized for the sort alone)
    const QueryLds q = carve_query_lds<KIND>(smem, L4, L4 > kSymMaxLen ? 4 : (T >> 6));
    const size_t row = (size_t)b * L;
    stage_rows(q.sy, p.scores + row, p.rel, p.rel_dtype, row, L, nb, tid, T);
    for (int m = tid; m < 2 * L4; m += T) q.rank_s[m] = 0;
    __syncthreads();
    int owners = 64;
    while (owners < L4 && owners < T) owners *= 2;
    const int ms = T / owners;
    const int mlen = (nb + ms - 1) / ms;
    const int m0 = __builtin_amdgcn_readfirstlane(min(nb, (tid / owners) * mlen));
    const int m1 = __builtin_amdgcn_readfirstlane(min(nb, m0 + mlen));
    prepare_ndcg<KIND, 1>(q, nb, owners, tid % owners, m0, m1, ms > 1);
    float2 *out = prep + row;
    for (int k = tid; k < nb; k += T) {
        if (KIND == LTR_NDCG1) {
            out[k] = make_float2(q.sy[k].y, 0.f);
        } else {
            const float4 v = q.q4[k];
            out[k] = make_float2(v.z, v.w);
        }
    }
}

template <int KIND>
__global__ void __launch_bounds__(1024)
pairwise_loss_split_kernel(LossParams p, int nsplit, int part_major, float *ws, const float2 *prep)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // Query-major by default.  When all the parts fit the chip in one go (part_major), the parts of
    // one query are launched B blocks apart instead: with B % 8 == 0 they land on one XCD and share
    // its L2 for the staged rows (measured: 27.9 -> 24.0 us at 32 x 1000, hinge).
    // Otherwise query-major over the list-length order (sched_query_sampled): block ids 8r .. 8r+7
    // are the parts of the query of rank r, so a CU (ids 256 apart) hosts parts of queries whose
    // ranks are 32 apart -- one of every length class -- instead of 8 random queries.
    const int L = p.L;
    const int tid = threadIdx.x;
    const int T = blockDim.x;
    int b, part, nb;
    if (part_major) {
        part = blockIdx.x / p.B;
        b = blockIdx.x - part * p.B;
        nb = clamp_n(p.n[b], L);
    } else {
        const int qpos = blockIdx.x / nsplit;
        part = blockIdx.x - qpos * nsplit;
        if (p.sched) {
            b = sched_query_sampled(p.n, p.B, L, p.sched, tid, nb, qpos);
        } else {
            b = qpos;
            nb = clamp_n(p.n[b], L);
        }
    }
    const int L4 = (L + 63) & ~63;
    const int msplit = T >> 6;
    const int parts = split_parts_for(nb, nsplit, msplit);
    if (part >= parts) return;                           // uniform: nothing for this workgroup
    const QueryLds q = carve_query_lds<KIND>(smem, L4, msplit);
    const size_t row = (size_t)b * L;
    stage_rows(q.sy, p.scores + row, p.rel, p.rel_dtype, row, L, nb, tid, T);
    __syncthreads();
    if (KIND == LTR_NDCG1 || KIND == LTR_NDCG2) {
        // the rankings were taken once per query by ndcg_prepare_kernel: prep[k] = (a_k, -) for
        // LambdaNDCG1, (gain G_k / maxDCG, rank_k) for LambdaNDCG2
        const float2 *pr = prep + row;
        for (int k = tid; k < nb; k += T) {
            const float2 v = pr[k];
            if (KIND == LTR_NDCG1) {
                q.sy[k].y = v.x;
            } else {
                const float2 sv = q.sy[k];
                q.q4[k] = make_float4(sv.x, sv.y, v.x, v.y);
            }
        }
        if (KIND == LTR_NDCG2) fill_ndcg2_delta(q.delta, nb, L4, tid, T);   // pairwise_lambda.py:206-211
        __syncthreads();
    }
    float unused = 1.0f;
    const float raw = pairwise_core_sym<KIND>(q, nb, L4, p.sigma, unused, part, parts, true);
    float *wl = ws + (size_t)b * nsplit + part;
    float *wg = ws + (size_t)p.B * nsplit + ((size_t)b * nsplit + part) * L;
    if (tid == 0) *wl = raw;
    if (p.dscores != nullptr) {
        for (int k = tid; k < nb; k += T) {
            float g = 0.f;
            for (int s = 0; s < msplit; ++s) g += q.gpart[(size_t)s * L4 + k];
            wg[k] = g;
        }
    }
}

template <int KIND>
__global__ void __launch_bounds__(256)
pairwise_loss_finish_kernel(LossParams p, int nsplit, int waves, const float *ws)
{
    // grid (B, ceil(L / 256)): one gradient entry per thread, the parts' slices read as independent
    // loads (one workgroup per query walked them one after the other: 9.8 us at C4, now ~3)
    const int b = blockIdx.x;
    const int L = p.L;
    const int nb = clamp_n(p.n[b], L);
    const int parts = split_parts_for(nb, nsplit, waves);
    const float *wl = ws + (size_t)b * nsplit;
    float total = 0.f;
    for (int s = 0; s < parts; ++s) total += wl[s];      // every thread, same order
    float gscale = 1.0f;
    if (KIND == LTR_DCG_HINGE) {
        const float lg = logf(2.0f + total);
        gscale = 1.0f / ((2.0f + total) * lg * lg);
        total = -1.0f / lg;
    } else if (KIND != LTR_HINGE) {
        gscale = p.sigma / kLn2;
    }
    if (threadIdx.x == 0 && blockIdx.y == 0) p.loss[b] = total;
    if (p.dscores != nullptr) {
        const float *wg = ws + (size_t)p.B * nsplit + (size_t)b * nsplit * L;
        const int k = blockIdx.y * blockDim.x + threadIdx.x;
        if (k < L) {
            float g = 0.f;
            if (k < nb) {
                for (int s0 = 0; s0 < parts; s0 += 8) {             // eight independent loads at a time
                    float v[8];
#pragma unroll
                    for (int s = 0; s < 8; ++s) v[s] = (s0 + s < parts) ? wg[(size_t)(s0 + s) * L + k] : 0.f;
#pragma unroll
                    for (int s = 0; s < 8; ++s) g += v[s];          // fixed order; the extra terms are +0
                }
                g *= gscale;
            }
            p.dscores[(size_t)b * L + k] = g;
        }
    }
}

__global__ void scale_rows_kernel(const float *__restrict__ ds, const float *__restrict__ go,
                                  size_t total, int L, float *__restrict__ out)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride)
        out[i] = ds[i] * go[i / (size_t)L];
}

__global__ void scale_rows_vec4_kernel(const float4 *__restrict__ ds, const float *__restrict__ go,
                                       size_t total4, int L4, float4 *__restrict__ out)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += stride) {
        const float g = go[i / (size_t)L4];
        float4 v = ds[i];
        v.x *= g; v.y *= g; v.z *= g; v.w *= g;
        out[i] = v;
    }
}

// the same with one upstream gradient for all rows (`.mean().backward()`: autograd hands over an
// expanded scalar), read from device memory
__global__ void scale_uniform_kernel(const float *__restrict__ ds, const float *__restrict__ go,
                                     size_t total, float *__restrict__ out)
{
    const float g = go[0];
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride)
        out[i] = ds[i] * g;
}

__global__ void scale_uniform_vec4_kernel(const float4 *__restrict__ ds, const float *__restrict__ go,
                                          size_t total4, float4 *__restrict__ out)
{
    const float g = go[0];
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += stride) {
        float4 v = ds[i];
        v.x *= g; v.y *= g; v.z *= g; v.w *= g;
        out[i] = v;
    }
}

// ---------------------------------------------------------------------------------
// Listwise softmax cross-entropy (ListNet top-one) -- named by the project brief, absent from the
// reference (pytorchltr/loss/__init__.py:1-7): the specification is include/ltr_hip.h.
//   loss = ln Z_s - sum_j P_y(j) (s_j - max s),  Z_s = sum_j exp(s_j - max s),  P_y = softmax(y)
//   d loss / d s_j = exp(s_j - max s) / Z_s - P_y(j)
// O(n) per query: one WAVE per query, four queries per workgroup, the row re-read from L1/L2 for
// the three passes (max, sums, gradient) -- no LDS, no barriers.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, kWave));
    return v;
}

template <typename LabelT>
__global__ void __launch_bounds__(256)
listwise_softmax_kernel(const float *__restrict__ scores, const LabelT *__restrict__ rel,
                        const int64_t *__restrict__ n, int B, int L, float *__restrict__ loss,
                        float *__restrict__ dscores)
{
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;                                   // wave-uniform
    const int nb = clamp_n(n[b], L);
    const size_t row = (size_t)b * L;
    float ms = -INFINITY, my = -INFINITY;
    for (int j = lane; j < nb; j += 64) {
        ms = fmaxf(ms, scores[row + j]);
        my = fmaxf(my, (float)rel[row + j]);
    }
    ms = wave_max(ms);
    my = wave_max(my);
    float zs = 0.f, zy = 0.f, dot = 0.f;
    for (int j = lane; j < nb; j += 64) {
        const float ds = scores[row + j] - ms;
        const float ey = expf((float)rel[row + j] - my);
        zs += expf(ds);
        zy += ey;
        dot += ey * ds;
    }
    zs = wave_sum(zs);
    zy = wave_sum(zy);
    dot = wave_sum(dot);
    const float inv_zs = nb > 0 ? 1.0f / zs : 0.f;
    const float inv_zy = nb > 0 ? 1.0f / zy : 0.f;
    if (lane == 0) loss[b] = nb > 0 ? (logf(zs) - dot * inv_zy) : 0.f;
    if (dscores != nullptr) {
        for (int j = lane; j < L; j += 64) {
            float g = 0.f;
            if (j < nb)
                g = expf(scores[row + j] - ms) * inv_zs - expf((float)rel[row + j] - my) * inv_zy;
            dscores[row + j] = g;
        }
    }
}

// ---------------------------------------------------------------------------------
// rank_by_score / dcg / ndcg / arp
// ---------------------------------------------------------------------------------
struct MetricParams {
    const float *scores;
    const void *rel;
    const int64_t *n;
    const int32_t *tie;   // (L) tie priorities (a permutation of 0..L-1) or null = index order
    unsigned long long tie_seed;   // use_seed: tie words hashed from (seed, position), see tie_hash_word
    const int64_t *tie_seed_dev;   //   the seed read from device memory instead (a device generator's draw)
    int use_seed;
    void *out;
    int B, L;
    int rel_dtype;
    int k;           // dcg: cutoff (0 = full curve)
    int use_exp;
    int normalize;
    int msplit;
};

__host__ __device__ inline size_t metric_lds_bytes(int L)
{
    const size_t L4 = (size_t)((L + 3) & ~3);
    return 8 * L4 + 8 * L4 + 16 * L4 + 32 * 4 + 64 * 4;  // sy, ranks, two curves (also: packed keys), red, scan
}

__host__ __device__ inline size_t metric_lds_bytes_sort(int L)
{
    const size_t L4 = (size_t)((L + 3) & ~3);
    // ... + the inverse tie map int[L4] behind everything else
    return 8 * L4 + 8 * L4 + 8 * (size_t)sort_pow2(L) + 32 * 4 + 64 * 4 + 4 * L4;
}

// Inclusive prefix sum of buf[0..L) in place (LDS).  Thread t owns a contiguous chunk.
__device__ void block_inclusive_scan(float *buf, int L, float *scan_scratch)
{
    const int T = blockDim.x, tid = threadIdx.x;
    const int ch = (L + T - 1) / T;
    const int lo = min(L, tid * ch), hi = min(L, lo + ch);
    float s = 0.f;
    for (int i = lo; i < hi; ++i) s += buf[i];
    // exclusive scan of per-thread sums: wave scan + cross-wave offsets
    float incl = s;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        const float up = __shfl_up(incl, off, kWave);
        if ((tid & 63) >= off) incl += up;
    }
    const int w = tid >> 6, nw = T >> 6;
    __syncthreads();
    if ((tid & 63) == 63) scan_scratch[w] = incl;
    __syncthreads();
    float woff = 0.f;
    for (int i = 0; i < w; ++i) woff += scan_scratch[i];
    (void)nw;
    float run = woff + incl - s;
    for (int i = lo; i < hi; ++i) { run += buf[i]; buf[i] = run; }
    __syncthreads();
}

enum { METRIC_RANK = 0, METRIC_DCG = 1, METRIC_ARP = 2 };

// (the sort path, DPT <= 0, under 64 VGPRs -- eight waves per SIMD: its workgroups have up to 1024 threads, and at the 68-69 VGPRs
// the compiler takes when left alone ONE of those fits a CU instead of two.  ndcg@10, 16 384 x 1000: 1167 -> 746 us, arp 631 -> 399,
// 65 536 x 512: 1260 -> 1103, 8192 x 2000: 1096 -> 746; lists of 128: unchanged.  Round 6.)
template <int OP, int DPT>
__global__ void __launch_bounds__(1024, (DPT <= 0 ? 8 : 4))
metric_kernel(MetricParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int b = blockIdx.x;
    const int L = p.L;
    const int L4 = (L + 3) & ~3;
    const int tid = threadIdx.x;
    const int T = blockDim.x;
    const int msplit = p.msplit;
    const int owners = T / msplit;
    const int o = tid % owners;
    const int slice = tid / owners;
    const int nb = clamp_n(p.n[b], L);

    float2 *sy = reinterpret_cast<float2 *>(smem);
    int *rank_s = reinterpret_cast<int *>(smem + 8 * (size_t)L4);
    int *rank_y = rank_s + L4;
    float *curve = reinterpret_cast<float *>(smem + 16 * (size_t)L4);
    float *icurve = curve + L4;
    float *red = icurve + L4;
    float *scan_scratch = red + 32;

    const size_t row = (size_t)b * L;
    const bool need_labels = (OP != METRIC_RANK);
    // dcg keeps the reference's quirk: padded labels are read and counted (dcg.py:85-94)
    const int nload = (OP == METRIC_DCG) ? L : nb;
    for (int m = tid; m < nload; m += T)
        sy[m] = make_float2(p.scores[row + m],
                            need_labels ? load_label(p.rel, p.rel_dtype, row + m) : 0.f);
    for (int m = tid; m < 2 * L4; m += T) rank_s[m] = 0;
    __syncthreads();

    const int mlen = (nb + msplit - 1) / msplit;
    const int m0 = __builtin_amdgcn_readfirstlane(slice * mlen);
    const int m1 = __builtin_amdgcn_readfirstlane(min(nb, m0 + mlen));
    const bool with_y = (OP == METRIC_DCG) && p.normalize;
    if (DPT <= 0) {
        // long lists: bitonic sort of (score, index) keys -- and of (label, index) for the ideal
        // ranking -- through the curve buffer; T = min(1024, P), E = P / T registers per thread
        // (DPT = 0, -2, -4 stands for E = 1, 2, 4)
        constexpr int E = DPT == 0 ? 1 : (DPT == -2 ? 2 : 4);
        unsigned long long *xbuf = reinterpret_cast<unsigned long long *>(curve);
        int Pq = 64;                                    // smallest power of two >= n of this query
        while (Pq < nb) Pq <<= 1;
        // random tie-break: the low key word is the document's tie priority; the inverse map
        // (priority -> document) sits behind the other arrays
        int *invt = nullptr;
        const unsigned long long seed = p.use_seed ? (p.tie_seed_dev ? (unsigned long long)p.tie_seed_dev[0] : p.tie_seed) : 0ull;
        const int low_mask = p.use_seed ? 0xFFF : 0;
        auto tie_word = [&](int i) { return p.use_seed ? (int)tie_hash_word(seed, (unsigned)i) : (p.tie ? p.tie[i] : i); };
        if (p.tie && !p.use_seed) {
            invt = reinterpret_cast<int *>(smem + metric_lds_bytes_sort(L) - 4 * (size_t)L4);
            for (int j = tid; j < L; j += T) invt[p.tie[j]] = j;
            __syncthreads();
        }
        unsigned long long v[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int i = e * T + tid;
            v[e] = (i < nb) ? rank_key(sy[i].x, tie_word(i)) : ~0ull;
        }
        sort_ranks<E>(v, Pq, nb, rank_s, xbuf, invt, low_mask);
        if (with_y) {
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int i = e * T + tid;
                v[e] = (i < nb) ? rank_key(sy[i].y, tie_word(i)) : ~0ull;
            }
            sort_ranks<E>(v, Pq, nb, rank_y, xbuf, invt, low_mask);
        }
    } else {
        // counting rank on packed keys (see count_ranks_keyed); the keys live in the curve region
        ulonglong2 *keys = reinterpret_cast<ulonglong2 *>(curve);
        for (int k = tid; k < nb; k += T) {
            const float2 v = sy[k];
            // tie word: hashed from the seed, a caller-drawn priority, or the index
            const int t = p.use_seed ? (int)tie_hash_word(p.tie_seed_dev ? (unsigned long long)p.tie_seed_dev[0] : p.tie_seed, (unsigned)k)
                                     : (p.tie ? p.tie[k] : k);
            keys[k] = make_ulonglong2(rank_key(v.x, t), rank_key(v.y, t));
        }
        __syncthreads();
        if (with_y)
            count_ranks_keyed<(DPT > 0 ? DPT : 1), true>(keys, nb, owners, o, m0, m1, msplit > 1, rank_s, rank_y);
        else
            count_ranks_keyed<(DPT > 0 ? DPT : 1), false>(keys, nb, owners, o, m0, m1, msplit > 1, rank_s, rank_y);
    }
    __syncthreads();

    if (OP == METRIC_RANK) {
        // invert: ranking[rank_k] = k; padded documents keep their index (tail in index order)
        int *inv = rank_y;
        for (int k = tid; k < L; k += T) inv[k < nb ? rank_s[k] : k] = k;
        __syncthreads();
        int64_t *out = reinterpret_cast<int64_t *>(p.out) + row;
        for (int r = tid; r < L; r += T) out[r] = (int64_t)inv[r];
        return;
    }

    if (OP == METRIC_ARP) {
        // arp.py:31-42: sum((r+1) * rel_r) / sum(rel_r) over ranks r < n; 0 -> 1 guard
        float srp = 0.f, nrp = 0.f;
        for (int k = tid; k < nb; k += T) {
            const float y = sy[k].y;
            srp += (float)(rank_s[k] + 1) * y;
            nrp += y;
        }
        srp = block_sum(srp, red);
        nrp = block_sum(nrp, red);
        if (nrp == 0.0f) nrp = 1.0f;
        if (tid == 0) reinterpret_cast<float *>(p.out)[b] = srp / nrp;
        return;
    }

    // ---- dcg / ndcg ----
    const int kk = p.k > 0 ? min(p.k, L) : 0;
    float part = 0.f, ipart = 0.f;
    for (int k = tid; k < L; k += T) {
        const float y = sy[k].y;
        const float gain = p.use_exp ? (exp2f(y) - 1.0f) : y;        // dcg.py:91-92
        const int r = k < nb ? rank_s[k] : k;
        const float term = gain / log2f((float)r + 2.0f);             // dcg.py:93
        int ry = 0;
        float iterm = 0.f;
        if (p.normalize) {
            ry = k < nb ? rank_y[k] : k;                              // ideal ranking, dcg.py:36
            iterm = gain / log2f((float)ry + 2.0f);
        }
        if (kk > 0) {
            part += (r < kk) ? term : 0.f;
            ipart += (p.normalize && ry < kk) ? iterm : 0.f;
        } else {
            curve[r] = term;
            if (p.normalize) icurve[ry] = iterm;
        }
    }
    if (kk > 0) {
        part = block_sum(part, red);
        if (p.normalize) {
            ipart = block_sum(ipart, red);
            if (ipart == 0.0f) ipart = 1.0f;                           // dcg.py:37
            part = part / ipart;
        }
        if (tid == 0) reinterpret_cast<float *>(p.out)[b] = part;
        return;
    }
    __syncthreads();
    block_inclusive_scan(curve, L, scan_scratch);                      // cumsum, dcg.py:94
    if (p.normalize) block_inclusive_scan(icurve, L, scan_scratch);
    float *out = reinterpret_cast<float *>(p.out) + row;
    for (int r = tid; r < L; r += T) {
        float v = curve[r];
        if (p.normalize) {
            float id = icurve[r];
            if (id == 0.0f) id = 1.0f;
            v /= id;
        }
        out[r] = v;
    }
}

// ---------------------------------------------------------------------------------
// mask_padded_values / batch_pairs
// ---------------------------------------------------------------------------------
__global__ void mask_padded_kernel(const float *__restrict__ xs, const int64_t *__restrict__ n,
                                   size_t total, int L, float mask_value, float *__restrict__ out)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const size_t b = i / (size_t)L;
        const int j = (int)(i - b * (size_t)L);
        out[i] = ((int64_t)j >= n[b]) ? mask_value : xs[i];
    }
}

// out[b,i,j,0] = x[b,i]; out[b,i,j,1] = x[b,j].  One (i,j) pair per thread-iteration,
// written as a single 2-element vector store (8 B or 16 B per lane, coalesced along j).
template <typename T, typename T2>
__global__ void batch_pairs_kernel(const T *__restrict__ x, int L, T2 *__restrict__ out)
{
    const int b = blockIdx.y;
    const T *xr = x + (size_t)b * L;
    T2 *o = out + (size_t)b * L * L;
    const size_t LL = (size_t)L * L;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < LL; e += stride) {
        const int i = (int)(e / (size_t)L);
        const int j = (int)(e - (size_t)i * L);
        T2 v;
        v.x = xr[i];
        v.y = xr[j];
        o[e] = v;
    }
}

// ---------------------------------------------------------------------------------
// Plackett-Luce sampling keys and the PBM click simulator (both sit on the
// rank kernel).  Randomness comes in as a uniform(0,1) tensor from the caller (torch's device
// Philox generator), so both kernels are deterministic functions of their inputs.
// ---------------------------------------------------------------------------------
// rank_by_plackettluce, utils/tensor_operations.py:67-91: r = log(-log u) - log_softmax(masked
// scores); the sampled ranking is the ASCENDING argsort of r.  Writes key = -r (so the
// descending rank kernel applies), row by row; padded documents are masked by the rank kernel.
__global__ void __launch_bounds__(256)
plackettluce_keys_kernel(const float *__restrict__ scores, const int64_t *__restrict__ n,
                         const float *__restrict__ u, int L, float *__restrict__ keys)
{
    __shared__ float red[8];
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int nb = clamp_n(n[b], L);
    const size_t row = (size_t)b * L;
    float mx = -INFINITY;
    for (int j = tid; j < nb; j += 256) mx = fmaxf(mx, scores[row + j]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, kWave));
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float se = 0.f;
    for (int j = tid; j < nb; j += 256) se += expf(scores[row + j] - mx);
    se = wave_sum(se);
    if ((tid & 63) == 0) red[4 + (tid >> 6)] = se;
    __syncthreads();
    const float lse = mx + logf((red[4] + red[5]) + (red[6] + red[7]));
    for (int j = tid; j < L; j += 256) {
        float key = -INFINITY;
        if (j < nb) key = (scores[row + j] - lse) - logf(-logf(u[row + j]));
        keys[row + j] = key;
    }
}

// simulate_pbm, click_simulation/pbm.py:12-63.  For rank position r of query b:
//   obs = r < min(n, cutoff) ? 1 / (r + 2)^eta : 0;   doc = rankings[b, r];
//   click = u[b, r] < relevance_probs[ys[b, doc]] * obs;
// clicks and propensities are written at the DOCUMENT's slot (the reference's final gather
// through the inverted ranking, :57-63).
__global__ void pbm_clicks_kernel(const int64_t *__restrict__ rankings, const int64_t *__restrict__ ys,
                                  const int64_t *__restrict__ n, const float *__restrict__ rel_probs,
                                  int n_probs, const float *__restrict__ u, size_t total, int L,
                                  int cutoff, float eta, int64_t *__restrict__ clicks,
                                  float *__restrict__ props)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const size_t b = i / (size_t)L;
        const int r = (int)(i - b * (size_t)L);
        int64_t lim = n[b];
        if (cutoff >= 0 && (int64_t)cutoff < lim) lim = cutoff;
        const float obs = ((int64_t)r < lim) ? powf(1.0f / (2.0f + (float)r), eta) : 0.0f;
        int64_t doc = rankings[i];
        if (doc < 0) doc = 0;
        if (doc >= L) doc = L - 1;
        int64_t y = ys[b * (size_t)L + doc];
        if (y < 0) y = 0;
        if (y >= n_probs) y = n_probs - 1;
        const float pclick = rel_probs[y] * obs;
        clicks[b * (size_t)L + doc] = (u[i] < pclick) ? 1 : 0;
        props[b * (size_t)L + doc] = obs;
    }
}

// ---------------------------------------------------------------------------------
// collate / pad (reference: SVMRankDataset.collate_fn dense path,
// datasets/svmrank/svmrank.py:126-207): ragged rows + offsets -> zero-padded (B, L, F),
// (B, L), n.  Pure gather/pad copy: one workgroup column per query, 16-byte vectors when F % 4
// == 0, rows >= min(n_q, L) written as zeros, truncated queries gathered through `sel`.
// ---------------------------------------------------------------------------------
template <typename V>
__global__ void collate_pad_kernel(const V *__restrict__ xs, const int64_t *__restrict__ ys,
                                   const int64_t *__restrict__ offsets,
                                   const int64_t *__restrict__ qidx,
                                   const int64_t *__restrict__ sel, int Q, int L, int C,
                                   V *__restrict__ out_x, int64_t *__restrict__ out_y,
                                   int64_t *__restrict__ out_n)
{
    const int b = blockIdx.x;
    int64_t q = qidx[b];
    if (q < 0) q = 0;
    if (q >= Q) q = Q - 1;
    const int64_t off = offsets[q];
    const int64_t cnt = offsets[q + 1] - off;
    const int nout = (int)(cnt < (int64_t)L ? cnt : (int64_t)L);
    const int64_t *srow = sel ? sel + (size_t)b * L : nullptr;
    V zero;
    memset(&zero, 0, sizeof(V));
    const size_t total = (size_t)L * C;
    V *ox = out_x + (size_t)b * total;
    for (size_t e = (size_t)blockIdx.y * blockDim.x + threadIdx.x; e < total;
         e += (size_t)gridDim.y * blockDim.x) {
        const int l = (int)(e / (size_t)C);
        const int c = (int)(e - (size_t)l * C);
        V v = zero;
        if (l < nout) {
            const int64_t src = off + (srow ? srow[l] : (int64_t)l);
            v = xs[(size_t)src * C + c];
        }
        ox[e] = v;
    }
    if (blockIdx.y == 0) {
        for (int l = threadIdx.x; l < L; l += blockDim.x)
            out_y[(size_t)b * L + l] = (l < nout) ? ys[off + (srow ? srow[l] : (int64_t)l)] : 0;
        if (threadIdx.x == 0) out_n[b] = nout;
    }
}

// The sparse branch of the same collate (svmrank.py:162-176,197-202: sparse SVMRankItems; the reference
// returns a torch sparse COO batch whose dense form is the padded (B, L, F) batch).  Here the split lives
// in HBM as CSR -- row pointers over all documents, column indices, values -- and is gathered straight
// into the DENSE padded batch the loss kernels consume: one wave per output row builds the row in LDS
// (zero fill, scatter its non-zeros) and stores it coalesced; duplicate column entries of a row add up,
// as torch's coalesce() does.
__global__ void __launch_bounds__(256)
collate_pad_csr_kernel(const int64_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                       const float *__restrict__ values, const int64_t *__restrict__ ys,
                       const int64_t *__restrict__ offsets, const int64_t *__restrict__ qidx,
                       const int64_t *__restrict__ sel, int Q, int L, int F, float *__restrict__ out_x,
                       int64_t *__restrict__ out_y, int64_t *__restrict__ out_n)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    int64_t q = qidx[b];
    if (q < 0) q = 0;
    if (q >= Q) q = Q - 1;
    const int64_t off = offsets[q];
    const int64_t cnt = offsets[q + 1] - off;
    const int nout = (int)(cnt < (int64_t)L ? cnt : (int64_t)L);
    const int64_t *srow = sel ? sel + (size_t)b * L : nullptr;
    float *rowbuf = reinterpret_cast<float *>(smem) + (size_t)wave * F;
    for (int l = blockIdx.y * nw + wave; l < L; l += gridDim.y * nw) {
        for (int f = lane; f < F; f += 64) rowbuf[f] = 0.f;
        if (l < nout) {
            const int64_t src = off + (srow ? srow[l] : (int64_t)l);
            const int64_t p0 = indptr[src], p1 = indptr[src + 1];
            for (int64_t e = p0 + lane; e < p1; e += 64) {
                const int col = indices[e];
                if (col >= 0 && col < F) atomicAdd(&rowbuf[col], values[e]);      // (LDS; duplicates add up)
            }
        }
        float *ox = out_x + ((size_t)b * L + l) * F;
        for (int f = lane; f < F; f += 64) ox[f] = rowbuf[f];
    }
    if (blockIdx.y == 0) {
        for (int l = threadIdx.x; l < L; l += blockDim.x)
            out_y[(size_t)b * L + l] = (l < nout) ? ys[off + (srow ? srow[l] : (int64_t)l)] : 0;
        if (threadIdx.x == 0) out_n[b] = nout;
    }
}

struct LaunchShape { int owners, dpt, msplit; };

// dpt == 0: symmetric pair pass -- LDS rows padded to 64-wide tiles, one gradient slice per wave
inline size_t loss_lds_bytes_cfg(int kind, int L, const LaunchShape &s)
{
    if (s.dpt == 0) return loss_lds_bytes(kind, (L + 63) & ~63, (s.owners * s.msplit) / 64);
    return loss_lds_bytes(kind, L, s.msplit);
}


LaunchShape choose_shape(int B, int L)
{
    LaunchShape s;
    if (L <= 64) { s.owners = 64; s.dpt = 1; }
    else if (L <= 128) { s.owners = 128; s.dpt = 1; }
    else if (L <= 256) { s.owners = 256; s.dpt = 1; }
    else if (L <= 512) { s.owners = 256; s.dpt = 2; }
    else { s.owners = 256; s.dpt = 4; }
    // Split the streamed index over more waves while the grid would leave SIMDs empty:
    // 256 CUs x 4 SIMDs, aim for >= 4 waves per SIMD.
    s.msplit = 1;
    const long target_waves = 4096;
    while ((long)B * (s.owners / 64) * s.msplit < target_waves && s.owners * s.msplit * 2 <= 1024 &&
           (L / (s.msplit * 2)) >= 16)
        s.msplit *= 2;
    return s;
}

// Loss kernels: lists up to kLossSymMaxLen take the symmetric pair pass (dpt == 0; measured on MI355X:
// C2 hinge 7.7 -> 6.5 us, NDCG2 20.3 -> 16.4 us, C5 35 -> 26 us, C4 62 -> 56 us).
// Waves per query: while the batch leaves CUs short of work a query is spread wide -- 4 waves for L <= 128, 8 up to 256, 16 above
// (the latency of ONE query is the launch's).  Once every CU has several rounds of queries the launch is bound by how many
// queries a CU has IN FLIGHT (a query is a chain of one HBM round trip, a staging barrier, the pair pass and a reduction; eight
// waves per SIMD whatever their grouping), and narrow workgroups win: round 6, ragged lists, us, default -> best,
//   hinge      128: 65 536 queries 130 -> 100 (1 wave), 262 144: 482 -> 336;   64: 262 144: 343 -> 153;   256: 65 536: 354 -> 280 (4)
//              512: 65 536: 1226 -> 904 (4 waves; 8: 959), 1000: 65 536: 3620 -> 3198 (4), 16 384: 959 -> 884
//   LambdaNDCG2 128: 65 536: 318 -> 273 (1), 256: 65 536: 976 -> 774 (4), 512: 1024 queries 80 -> 70 (8), 65 536: 3947 -> 2751 (8),
//              1000: 4096: 940 -> 716 (8), 65 536: 15 655 -> 11 621 (8)
//   logistic / LambdaARP 128: 262 144: 895 -> 827 / 868 -> 811 (1); 256: 4096: 56.9 -> 54.7 (4); 512: 16 384: 708 -> 652 (8)
// (scripts/dev/loss_waves.py; a pair evaluation costs the LambdaNDCG kinds several times what it costs the hinge kinds, which
// is why they want their ~8 k pairs per wave where the hinge kinds take 32 k.)
LaunchShape choose_loss_shape(int kind, int B, int L)
{
#ifndef LTR_NO_SYM
    if (L <= kLossSymMaxLen) {
        LaunchShape s;
        s.dpt = 0;
        s.owners = 64;
        // (beyond 1024 documents: eight -- sixteen gradient slices of 2048 floats do not fit the LDS)
        int waves = (L <= 128) ? 4 : (L <= 256 ? 8 : (L <= kSymMaxLen ? 16 : 8));
        const long cus = device_cu_count();
        const bool cheap = kind == LTR_HINGE || kind == LTR_DCG_HINGE;
        const bool ndcg = kind == LTR_NDCG1 || kind == LTR_NDCG2;
        if (L <= 64) {
            if (B >= 16 * cus) waves = 1;
        } else if (L <= 128) {
            if (cheap) waves = B >= 128 * cus ? 1 : (B >= 16 * cus ? 2 : 4);
            else if (ndcg) { if (B >= 64 * cus) waves = 2; }              // (65 536: 2 waves 272, 1 wave 291, 4 waves 292)
            else if (B >= 256 * cus) waves = 1;
        } else if (L <= 256) {
            if (B >= 16 * cus) waves = 4;
        } else if (L <= kSymMaxLen) {
            if (ndcg) { if (B >= 4 * cus) waves = 8; }
            else if (B >= 16 * cus) waves = (cheap && B >= 64 * cus) ? 4 : 8;
        }
        s.msplit = waves;
        return s;
    }
#endif
    (void)kind;
    return choose_shape(B, L);
}

template <int KIND>
int launch_loss_kind(const LossParams &p, const LaunchShape &s, hipStream_t stream)
{
    const dim3 grid((unsigned)p.B), block((unsigned)(s.owners * s.msplit));
    const size_t lds = loss_lds_bytes_cfg(KIND, p.L, s);
#define LTR_LAUNCH(D)                                                                           \
    do {                                                                                        \
        LTR_ENSURE_LDS((pairwise_loss_kernel<KIND, D>), lds);          \
        hipLaunchKernelGGL((pairwise_loss_kernel<KIND, D>), grid, block, lds, stream, p);       \
    } while (0)
#define LTR_LAUNCH_SYM(NWAVES)                                                                   \
    do {                                                                                        \
        if constexpr (KIND == LTR_NDCG1 || KIND == LTR_NDCG2) {                                 \
            LTR_ENSURE_LDS((pairwise_loss_kernel80<KIND, NWAVES>), lds);                        \
            hipLaunchKernelGGL((pairwise_loss_kernel80<KIND, NWAVES>), grid, block, lds, stream, p); \
        } else {                                                                                \
            LTR_ENSURE_LDS((pairwise_loss_kernel<KIND, 0, NWAVES>), lds);                       \
            hipLaunchKernelGGL((pairwise_loss_kernel<KIND, 0, NWAVES>), grid, block, lds, stream, p); \
        }                                                                                       \
    } while (0)
    switch (s.dpt) {
    case 0:
        // the shapes choose_loss_shape picks get a compile-time wave count; explicit (_cfg) shapes
        // with another block size run the run-time variant
        if (s.owners * s.msplit == 64) LTR_LAUNCH_SYM(1);
        else if (s.owners * s.msplit == 128) LTR_LAUNCH_SYM(2);
        else if (s.owners * s.msplit == 256) LTR_LAUNCH_SYM(4);
        else if (s.owners * s.msplit == 512) LTR_LAUNCH_SYM(8);
        else if (s.owners * s.msplit == 1024) LTR_LAUNCH_SYM(16);
        else LTR_LAUNCH(0);
        break;
    case 1: LTR_LAUNCH(1); break;
    case 2: LTR_LAUNCH(2); break;
    case 4: LTR_LAUNCH(4); break;
    default: return LTR_ERR_CONFIG;
    }
#undef LTR_LAUNCH
#undef LTR_LAUNCH_SYM
    return (int)hipGetLastError();
}

int launch_loss(int kind, const LossParams &p, const LaunchShape &s, hipStream_t stream)
{
    switch (kind) {
    case LTR_HINGE: return launch_loss_kind<LTR_HINGE>(p, s, stream);
    case LTR_DCG_HINGE: return launch_loss_kind<LTR_DCG_HINGE>(p, s, stream);
    case LTR_LOGISTIC: return launch_loss_kind<LTR_LOGISTIC>(p, s, stream);
    case LTR_ARP1: return launch_loss_kind<LTR_ARP1>(p, s, stream);
    case LTR_ARP2: return launch_loss_kind<LTR_ARP2>(p, s, stream);
    case LTR_NDCG1: return launch_loss_kind<LTR_NDCG1>(p, s, stream);
    case LTR_NDCG2: return launch_loss_kind<LTR_NDCG2>(p, s, stream);
    default: return LTR_ERR_KIND;
    }
}

// How many workgroups share a query in the split launch (1 = use the one-kernel path): long lists
// only, and only while the batch alone cannot give every CU several queries to balance with.
constexpr int kSplitWaves = LTR_SPLIT_WAVES;   // waves per part: small workgroups, many per CU
static int choose_loss_splits(int kind, int B, int L)
{
    if (L <= 256 || L > kLossSymMaxLen) return 1;
    const int cus = device_cu_count();
    // measured (hinge / logistic, us, plain kernel with the list-length order -> split launch):
    // 384 x 1000: 35/66 either way; 512 x 1000: 58/135 -> 43/90; 768 x 1000: 71/163 -> 69/149;
    // 1024 x 1000: 72/175 -> 78/180; 512 x 512: 18/37 -> 20/35; 768 x 300: 14/21 -> 17/27
    // The NDCG kinds: rankings once per query by ndcg_prepare_kernel, then the same split (single
    // kernel -> split, LambdaNDCG1/2: 32 x 1000: 130/144 -> 51/53 us; 128 x 600: 65/71 -> 46/48;
    // C4 256 x 1000: 132/145 -> 82/90; 384 x 1000: 132/146 -> 114/123; 300 x 400: 41/43 -> 44/46)
    const bool ndcg = (kind == LTR_NDCG1 || kind == LTR_NDCG2);
    // Round 4 audit of the shorter lists (plain -> split, us; hinge / logistic / LambdaNDCG2): 256 x 300 7.9 / 13.5 / 28.1 ->
    // 11.7 / 14.0 / 31.7 (!), 128 x 300 7.9 / 13.3 / 28.1 -> 9.5 / 10.9 / 28.6, 256 x 400 11.7 / 22.7 / 36.9 -> 12.9 / 16.7 / 35.8,
    // 128 x 400 11.6 / 22.6 / 36.8 -> 9.9 / 12.5 / 29.9, 384 x 512 15.5 / 32.2 -> 17.5 / 27.2: the hinge kinds' pass is too
    // short to be worth a second launch below 400 documents or beyond half a chip of 400-document queries
    {
        const bool hinge = (kind == LTR_HINGE || kind == LTR_DCG_HINGE);
        if (L < 400 && (hinge || ndcg || 2 * B > cus)) return 1;
        if (hinge && L < 512 && 2 * B > cus) return 1;
        if (hinge && L < 640 && B > cus) return 1;
    }
    if (ndcg && B > cus && L <= 512) return 1;
    if (2 * B > 3 * cus && !(!ndcg && B <= 2 * cus && L > 640)) return 1;
    // up to 8 parts per query; 16 on the smallest batches (64 x 1000: hinge 22.4 -> 16.5 us, logistic
    // 28.6 -> 20.7, LambdaNDCG2 55 -> 45; at 128 x 600 and above 8 is better)
    const int cap = (4 * B <= cus) ? 2 * LTR_SPLIT_MAX : LTR_SPLIT_MAX;
    int s = (LTR_SPLIT_MAX * cus) / (B > 0 ? B : 1);
    if (s > cap) s = cap;
    return s < 2 ? 1 : s;
}

template <int KIND>
static int launch_loss_split(const LossParams &p, int nsplit, float *ws, hipStream_t stream)
{
    // 4-wave parts, many per CU; 8-wave parts when all of them fit the chip one per CU
    // (32 x 1000: hinge 21.5 -> 15.7 us, logistic 27.7 -> 22.0, LambdaNDCG2 52.9 -> 46.4)
    const int waves = ((long long)p.B * nsplit <= device_cu_count()) ? 2 * kSplitWaves : kSplitWaves;
    const size_t lds = loss_lds_bytes(KIND, (p.L + 63) & ~63, waves);
    LTR_ENSURE_LDS((pairwise_loss_split_kernel<KIND>), lds);
    const int part_major = (long long)p.B * nsplit <= device_cu_count() ? 1 : 0;
    float2 *prep = nullptr;
    if (KIND == LTR_NDCG1 || KIND == LTR_NDCG2) {
        // (behind the raw sums and gradient slices; 8-byte aligned: an even number of floats precedes it)
        prep = reinterpret_cast<float2 *>(ws + (((size_t)p.B * nsplit * ((size_t)p.L + 1) + 1) & ~(size_t)1));
        constexpr int PK = (KIND == LTR_NDCG1 || KIND == LTR_NDCG2) ? KIND : LTR_NDCG1;
        const size_t plds = loss_lds_bytes(PK, (p.L + 63) & ~63, ((p.L + 63) & ~63) > kSymMaxLen ? 4 : 16);
        LTR_ENSURE_LDS((ndcg_prepare_kernel<PK>), plds);
        hipLaunchKernelGGL((ndcg_prepare_kernel<PK>), dim3((unsigned)p.B), dim3(1024), plds, stream, p, prep);
    }
    hipLaunchKernelGGL((pairwise_loss_split_kernel<KIND>), dim3((unsigned)(p.B * nsplit)), dim3(64 * waves),
                       lds, stream, p, nsplit, part_major, ws, (const float2 *)prep);
    hipLaunchKernelGGL((pairwise_loss_finish_kernel<KIND>), dim3((unsigned)p.B, (unsigned)((p.L + 255) / 256)),
                       dim3(256), 0, stream, p, nsplit, waves, (const float *)ws);
    return (int)hipGetLastError();
}

template <int OP>
int launch_metric(const MetricParams &p0, hipStream_t stream)
{
    MetricParams p = p0;
#ifndef LTR_NO_SORT_RANK
    if (p.L > kSortRankMinLen) {
        const int P = sort_pow2(p.L);
        int T = P < 1024 ? P : 1024;                    // E = P / T <= 4 keys per thread
        // (many rounds of queries per CU: half the threads with two keys each -- narrower workgroups, more queries in flight, see
        // choose_loss_shape.  Round 6, us: ndcg@10 65 536 x 512 1105 -> 854, 65 536 x 300 953 -> 737, 16 384 x 1000 749 -> 552,
        // arp 65 536 x 300 530 -> 384; 1024 x 512: 25.9 / 25.5, 256 x 1000: 21.7 -> 26.7 -- hence from 16 queries per CU on.  Four
        // keys per thread: 65 536 x 512 889, 8192 x 2000 747 -> 804 -- not taken.)
        if (P <= 1024 && P >= 128 && (long)p.B >= 16L * device_cu_count()) T = P / 2;
        p.msplit = 1;
        const size_t lds = metric_lds_bytes_sort(p.L);
        const dim3 sgrid((unsigned)p.B), sblock((unsigned)T);
        if (P == T) {
            LTR_ENSURE_LDS((metric_kernel<OP, 0>), lds);
            hipLaunchKernelGGL((metric_kernel<OP, 0>), sgrid, sblock, lds, stream, p);
        } else if (P == 2 * T) {
            LTR_ENSURE_LDS((metric_kernel<OP, -2>), lds);
            hipLaunchKernelGGL((metric_kernel<OP, -2>), sgrid, sblock, lds, stream, p);
        } else {
            LTR_ENSURE_LDS((metric_kernel<OP, -4>), lds);
            hipLaunchKernelGGL((metric_kernel<OP, -4>), sgrid, sblock, lds, stream, p);
        }
        return (int)hipGetLastError();
    }
#endif
    LaunchShape s = choose_shape(p.B, p.L);
    // (many rounds of queries per CU: ONE wave per query, two documents per thread -- what bounds the launch then is the number of
    // queries a CU has in flight, see choose_loss_shape.  Lists of 128, round 6: ndcg@10 65 536 queries 116 -> 99 us, 2^20: 1667 ->
    // 1359, arp 2^20: 1216 -> 804; at 1024 queries the two-wave shape stays, 6.6 against 8.0)
    if (p.L > 64 && p.L <= 128 && (long)p.B >= 64L * device_cu_count()) { s.owners = 64; s.dpt = 2; s.msplit = 1; }
    p.msplit = s.msplit;
    const dim3 grid((unsigned)p.B), block((unsigned)(s.owners * s.msplit));
    const size_t lds = metric_lds_bytes(p.L);
#define LTR_LAUNCH(D)                                                                           \
    do {                                                                                        \
        LTR_ENSURE_LDS((metric_kernel<OP, D>), lds);          \
        hipLaunchKernelGGL((metric_kernel<OP, D>), grid, block, lds, stream, p);                \
    } while (0)
    switch (s.dpt) {
    case 1: LTR_LAUNCH(1); break;
    case 2: LTR_LAUNCH(2); break;
    default: LTR_LAUNCH(4); break;
    }
#undef LTR_LAUNCH
    return (int)hipGetLastError();
}

}  // namespace

// ---------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------
extern "C" {

int ltr_version(void) { return LTR_VERSION; }

int ltr_max_list_len(void) { return kMaxListLen; }

int ltr_device_status(int clear)
{
    LTR_CLEAR_STALE_ERROR();
    StatusPage &sp = status_page_ref();
    if (!sp.host) return LTR_OK;
    volatile int *w = reinterpret_cast<volatile int *>(sp.host);
    const int v = *w;
    if (clear) {
        *w = 0;
        // the launch that gave up left counters of its exchange area behind: zero the areas before they are used again
        if (v == LTR_ERR_TIMEOUT) ltr_internal_exchange_mark_dirty();
    }
    return v;
}

const char *ltr_error_string(int code)
{
    switch (code) {
    case LTR_OK: return "ok";
    case LTR_ERR_NULL: return "ltr: required pointer is NULL";
    case LTR_ERR_SHAPE: return "ltr: invalid shape (need B >= 0, L > 0, F > 0, k >= 0)";
    case LTR_ERR_KIND: return "ltr: unknown loss kind or label dtype";
    case LTR_ERR_LIST_TOO_LONG: return "ltr: list_len exceeds ltr_max_list_len()";
    case LTR_ERR_WORKSPACE: return "ltr: workspace missing or too small";
    case LTR_ERR_CONFIG: return "ltr: invalid explicit launch configuration";
    case LTR_ERR_TIMEOUT: return "ltr: a multi-workgroup kernel gave up waiting for its partners; the outputs of that launch are invalid (ltr_device_status(1) clears the flag)";
    default: break;
    }
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "ltr: unknown error";
}

int ltr_pairwise_loss_f32_cfg(int kind, float sigma, const float *scores, const void *rel,
                              int rel_dtype, const int64_t *n, int B, int L, float *loss,
                              float *dscores, int owners, int dpt, int msplit, void *stream)
{
    LTR_CLEAR_STALE_ERROR();
    if (kind < LTR_HINGE || kind > LTR_NDCG2 || bad_label_dtype(rel_dtype)) return LTR_ERR_KIND;
    if (B < 0 || L <= 0) return LTR_ERR_SHAPE;
    if (L > kMaxListLen) return LTR_ERR_LIST_TOO_LONG;
    if (B == 0) return LTR_OK;
    if (!scores || !rel || !n || !loss) return LTR_ERR_NULL;
    if (owners <= 0 || owners % 64 != 0 || msplit <= 0 || owners * msplit > 1024 ||
        (dpt != 0 && dpt != 1 && dpt != 2 && dpt != 4) || (dpt == 0 && L > kLossSymMaxLen))
        return LTR_ERR_CONFIG;
    {
        const LaunchShape chk{owners, dpt, msplit};
        if (loss_lds_bytes_cfg(kind, L, chk) > kLdsBudget) return LTR_ERR_CONFIG;
    }
    LossParams p;
    p.scores = scores; p.rel = rel; p.n = n; p.loss = loss; p.dscores = dscores;
    p.B = B; p.L = L; p.sigma = sigma; p.rel_dtype = rel_dtype; p.msplit = msplit;
    p.sched = sched_groups_for_lists(B, L);
    LaunchShape s{owners, dpt, msplit};
    return launch_loss(kind, p, s, (hipStream_t)stream);
}

int ltr_pairwise_loss_f32(int kind, float sigma, const float *scores, const void *rel,
                          int rel_dtype, const int64_t *n, int B, int L, float *loss,
                          float *dscores, void *stream)
{
    LTR_CLEAR_STALE_ERROR();
    if (B < 0 || L <= 0) return LTR_ERR_SHAPE;
    LaunchShape s = choose_loss_shape(kind, B, L);
    if (s.dpt == 0 && loss_lds_bytes_cfg(kind, L, s) > kLdsBudget) s = choose_shape(B, L);
    while (s.dpt != 0 && s.msplit > 1 && loss_lds_bytes(kind, L, s.msplit) > kLdsBudget) s.msplit /= 2;
    return ltr_pairwise_loss_f32_cfg(kind, sigma, scores, rel, rel_dtype, n, B, L, loss, dscores,
                                     s.owners, s.dpt, s.msplit, stream);
}

size_t ltr_pairwise_loss_workspace_bytes(int kind, int B, int L)
{
    if (B <= 0 || L <= 0) return 0;
    const int nsplit = choose_loss_splits(kind, B, L);
    if (nsplit <= 1) return 0;
    size_t floats = (size_t)B * nsplit * ((size_t)L + 1);
    if (kind == LTR_NDCG1 || kind == LTR_NDCG2)
        floats = ((floats + 1) & ~(size_t)1) + 2 * (size_t)B * L;      // + the prepared (gain, rank) pairs
    return floats * sizeof(float);
}

int ltr_pairwise_loss_ws_f32(int kind, float sigma, const float *scores, const void *rel,
                             int rel_dtype, const int64_t *n, int B, int L, float *loss,
                             float *dscores, void *workspace, size_t workspace_bytes, void *stream)
{
    LTR_CLEAR_STALE_ERROR();
    if (kind < LTR_HINGE || kind > LTR_NDCG2 || bad_label_dtype(rel_dtype)) return LTR_ERR_KIND;
    if (B < 0 || L <= 0) return LTR_ERR_SHAPE;
    if (L > kMaxListLen) return LTR_ERR_LIST_TOO_LONG;
    if (B == 0) return LTR_OK;
    if (!scores || !rel || !n || !loss) return LTR_ERR_NULL;
    const int nsplit = choose_loss_splits(kind, B, L);
    if (nsplit <= 1)
        return ltr_pairwise_loss_f32(kind, sigma, scores, rel, rel_dtype, n, B, L, loss, dscores, stream);
    if (!workspace || workspace_bytes < ltr_pairwise_loss_workspace_bytes(kind, B, L)) return LTR_ERR_WORKSPACE;
    LossParams p;
    p.scores = scores; p.rel = rel; p.n = n; p.loss = loss; p.dscores = dscores;
    p.B = B; p.L = L; p.sigma = sigma; p.rel_dtype = rel_dtype; p.msplit = kSplitWaves;
#ifdef LTR_NO_SPLIT_SCHED
    p.sched = 0;
#else
    p.sched = (B >= 16) ? (B + 63) / 64 : 0;            // (ignored by the part-major order)
#endif
    float *ws = (float *)workspace;
    hipStream_t st = (hipStream_t)stream;
    switch (kind) {
    case LTR_HINGE: return launch_loss_split<LTR_HINGE>(p, nsplit, ws, st);
    case LTR_DCG_HINGE: return launch_loss_split<LTR_DCG_HINGE>(p, nsplit, ws, st);
    case LTR_LOGISTIC: return launch_loss_split<LTR_LOGISTIC>(p, nsplit, ws, st);
    case LTR_ARP1: return launch_loss_split<LTR_ARP1>(p, nsplit, ws, st);
    case LTR_ARP2: return launch_loss_split<LTR_ARP2>(p, nsplit, ws, st);
    case LTR_NDCG1: return launch_loss_split<LTR_NDCG1>(p, nsplit, ws, st);
    default: return launch_loss_split<LTR_NDCG2>(p, nsplit, ws, st);
    }
}


int ltr_scale_rows_f32(const float *dscores, const float *grad_out, int B, int L, float *out,
                       void *stream)
{
    LTR_CLEAR_STALE_ERROR();
    if (B < 0 || L <= 0) return LTR_ERR_SHAPE;
    if (B == 0) return LTR_OK;
    if (!dscores || !grad_out || !out) return LTR_ERR_NULL;
    const size_t total = (size_t)B * L;
    const bool vec = (L % 4 == 0) && (((uintptr_t)dscores | (uintptr_t)out) % 16 == 0);
    if (vec)
        hipLaunchKernelGGL(scale_rows_vec4_kernel, dim3(grid_for(total / 4, 256)), dim3(256), 0,
                           (hipStream_t)stream, (const float4 *)dscores, grad_out, total / 4, L / 4,
                           (float4 *)out);
    else
        hipLaunchKernelGGL(scale_rows_kernel, dim3(grid_for(total, 256)), dim3(256), 0,
                           (hipStream_t)stream, dscores, grad_out, total, L, out);
    return (int)hipGetLastError();
}

int ltr_scale_rows_uniform_f32(const float *dscores, const float *grad_scalar, int B, int L,
                               float *out, void *stream)
{
    LTR_CLEAR_STALE_ERROR();
    if (B < 0 || L <= 0) return LTR_ERR_SHAPE;
    if (B == 0) return LTR_OK;
    if (!dscores || !grad_scalar || !out) return LTR_ERR_NULL;
    const size_t total = (size_t)B * L;
    const bool vec = (total % 4 == 0) && (((uintptr_t)dscores | (uintptr_t)out) % 16 == 0);
    if (vec)
        hipLaunchKernelGGL(scale_uniform_vec4_kernel, dim3(grid_for(total / 4, 256)), dim3(256), 0,
                           (hipStream_t)stream, (const float4 *)dscores, grad_scalar, total / 4,
                           (float4 *)out);
    else
        hipLaunchKernelGGL(scale_uniform_kernel, dim3(grid_for(total, 256)), dim3(256), 0,
                           (hipStream_t)stream, dscores, grad_scalar, total, out);
    return (int)hipGetLastError();
}

int ltr_rank_by_score_tie_f32(const float *scores, const int64_t *n, const int32_t *tie, int B, int L,
                              int64_t *ranking, void *stream)
{
    LTR_CLEAR_STALE_ERROR();
    if (B < 0 || L <= 0) return LTR_ERR_SHAPE;
    if (L > kMaxListLen) return LTR_ERR_LIST_TOO_LONG;
    if (B == 0) return LTR_OK;
    if (!scores || !n || !ranking) return LTR_ERR_NULL;
    MetricParams p{};
    p.scores = scores; p.rel = nullptr; p.n = n; p.tie = tie; p.out = ranking; p.B = B; p.L = L;
    return launch_metric<METRIC_RANK>(p, (hipStream_t)stream);
}

int ltr_rank_by_score_f32(const float *scores, const int64_t *n, int B, int L, int64_t *ranking,
                          void *stream)
{
    LTR_CLEAR_STALE_ERROR();
    return ltr_rank_by_score_tie_f32(scores, n, nullptr, B, L, ranking, stream);
}

int ltr_dcg_tie_f32(const float *scores, const void *rel, int rel_dtype, const int64_t *n,
                    const int32_t *tie, int B, int L, int k, int use_exp, int normalize, float *out,
                    void *stream)
{
    LTR_CLEAR_STALE_ERROR();
    if (bad_label_dtype(rel_dtype)) return LTR_ERR_KIND;
    if (B < 0 || L <= 0 || k < 0) return LTR_ERR_SHAPE;
    if (L > kMaxListLen) return LTR_ERR_LIST_TOO_LONG;
    if (B == 0) return LTR_OK;
    if (!scores || !rel || !n || !out) return LTR_ERR_NULL;
    MetricParams p{};
    p.scores = scores; p.rel = rel; p.n = n; p.tie = tie; p.out = out; p.B = B; p.L = L;
    p.rel_dtype = rel_dtype; p.k = k; p.use_exp = use_exp; p.normalize = normalize;
    return launch_metric<METRIC_DCG>(p, (hipStream_t)stream);
}

int ltr_dcg_f32(const float *scores, const void *rel, int rel_dtype, const int64_t *n, int B,
                int L, int k, int use_exp, int normalize, float *out, void *stream)
{
    LTR_CLEAR_STALE_ERROR();
    return ltr_dcg_tie_f32(scores, rel, rel_dtype, n, nullptr, B, L, k, use_exp, normalize, out, stream);
}

int ltr_arp_tie_f32(const float *scores, const void *rel, int rel_dtype, const int64_t *n,
                    const int32_t *tie, int B, int L, float *out, void *stream)
{
    LTR_CLEAR_STALE_ERROR();
    if (bad_label_dtype(rel_dtype)) return LTR_ERR_KIND;
    if (B < 0 || L <= 0) return LTR_ERR_SHAPE;
    if (L > kMaxListLen) return LTR_ERR_LIST_TOO_LONG;
    if (B == 0) return LTR_OK;
    if (!scores || !rel || !n || !out) return LTR_ERR_NULL;
    MetricParams p{};
    p.scores = scores; p.rel = rel; p.n = n; p.tie = tie; p.out = out; p.B = B; p.L = L;
    p.rel_dtype = rel_dtype;
    return launch_metric<METRIC_ARP>(p, (hipStream_t)stream);
}

int ltr_arp_f32(const float *scores, const void *rel, int rel_dtype, const int64_t *n, int B,
                int L, float *out, void *stream)
{
    LTR_CLEAR_STALE_ERROR();
    return ltr_arp_tie_f32(scores, rel, rel_dtype, n, nullptr, B, L, out, stream);
}

// ---- the same three with the tie words hashed in the kernel from a seed (see tie_hash_word) ----
int ltr_rank_by_score_seed_f32(const float *scores, const int64_t *n, uint64_t seed, const int64_t *seed_dev,
                               int B, int L, int64_t *ranking, void *stream)
{
    LTR_CLEAR_STALE_ERROR();
    if (B < 0 || L <= 0) return LTR_ERR_SHAPE;
    if (L > kMaxListLen || L > kTieHashMaxLen) return LTR_ERR_LIST_TOO_LONG;
    if (B == 0) return LTR_OK;
    if (!scores || !n || !ranking) return LTR_ERR_NULL;
    MetricParams p{};
    p.scores = scores; p.rel = nullptr; p.n = n; p.tie = nullptr; p.out = ranking; p.B = B; p.L = L;
    p.use_seed = 1; p.tie_seed = seed; p.tie_seed_dev = seed_dev;
    return launch_metric<METRIC_RANK>(p, (hipStream_t)stream);
}

int ltr_dcg_seed_f32(const float *scores, const void *rel, int rel_dtype, const int64_t *n, uint64_t seed,
                     const int64_t *seed_dev, int B, int L, int k, int use_exp, int normalize, float *out,
                     void *stream)
{
    LTR_CLEAR_STALE_ERROR();
    if (bad_label_dtype(rel_dtype)) return LTR_ERR_KIND;
    if (B < 0 || L <= 0 || k < 0) return LTR_ERR_SHAPE;
    if (L > kMaxListLen || L > kTieHashMaxLen) return LTR_ERR_LIST_TOO_LONG;
    if (B == 0) return LTR_OK;
    if (!scores || !rel || !n || !out) return LTR_ERR_NULL;
    MetricParams p{};
    p.scores = scores; p.rel = rel; p.n = n; p.tie = nullptr; p.out = out; p.B = B; p.L = L;
    p.rel_dtype = rel_dtype; p.k = k; p.use_exp = use_exp; p.normalize = normalize;
    p.use_seed = 1; p.tie_seed = seed; p.tie_seed_dev = seed_dev;
    return launch_metric<METRIC_DCG>(p, (hipStream_t)stream);
}

int ltr_arp_seed_f32(const float *scores, const void *rel, int rel_dtype, const int64_t *n, uint64_t seed,
                     const int64_t *seed_dev, int B, int L, float *out, void *stream)
{
    LTR_CLEAR_STALE_ERROR();
    if (bad_label_dtype(rel_dtype)) return LTR_ERR_KIND;
    if (B < 0 || L <= 0) return LTR_ERR_SHAPE;
    if (L > kMaxListLen || L > kTieHashMaxLen) return LTR_ERR_LIST_TOO_LONG;
    if (B == 0) return LTR_OK;
    if (!scores || !rel || !n || !out) return LTR_ERR_NULL;
    MetricParams p{};
    p.scores = scores; p.rel = rel; p.n = n; p.tie = nullptr; p.out = out; p.B = B; p.L = L;
    p.rel_dtype = rel_dtype;
    p.use_seed = 1; p.tie_seed = seed; p.tie_seed_dev = seed_dev;
    return launch_metric<METRIC_ARP>(p, (hipStream_t)stream);
}

uint32_t ltr_tie_hash_word(uint64_t seed, uint32_t position) { return tie_hash_word(seed, position); }

int ltr_listwise_softmax_f32(const float *scores, const void *rel, int rel_dtype, const int64_t *n,
                             int B, int L, float *loss, float *dscores, void *stream)
{
    LTR_CLEAR_STALE_ERROR();
    if (bad_label_dtype(rel_dtype)) return LTR_ERR_KIND;
    if (B < 0 || L <= 0) return LTR_ERR_SHAPE;
    if (B == 0) return LTR_OK;
    if (!scores || !rel || !n || !loss) return LTR_ERR_NULL;
    const dim3 grid((unsigned)((B + 3) / 4)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (rel_dtype == LTR_LABEL_I64)
        hipLaunchKernelGGL((listwise_softmax_kernel<int64_t>), grid, block, 0, st, scores,
                           (const int64_t *)rel, n, B, L, loss, dscores);
    else if (rel_dtype == LTR_LABEL_F32)
        hipLaunchKernelGGL((listwise_softmax_kernel<float>), grid, block, 0, st, scores,
                           (const float *)rel, n, B, L, loss, dscores);
    else
        hipLaunchKernelGGL((listwise_softmax_kernel<int32_t>), grid, block, 0, st, scores,
                           (const int32_t *)rel, n, B, L, loss, dscores);
    return (int)hipGetLastError();
}

int ltr_mask_padded_values_f32(const float *xs, const int64_t *n, int B, int L, float mask_value,
                               float *out, void *stream)
{
    LTR_CLEAR_STALE_ERROR();
    if (B < 0 || L <= 0) return LTR_ERR_SHAPE;
    if (B == 0) return LTR_OK;
    if (!xs || !n || !out) return LTR_ERR_NULL;
    const size_t total = (size_t)B * L;
    hipLaunchKernelGGL(mask_padded_kernel, dim3(grid_for(total, 256)), dim3(256), 0,
                       (hipStream_t)stream, xs, n, total, L, mask_value, out);
    return (int)hipGetLastError();
}

int ltr_batch_pairs(const void *x, int elem_bytes, int B, int L, void *out, void *stream)
{
    LTR_CLEAR_STALE_ERROR();
    if (B < 0 || L <= 0) return LTR_ERR_SHAPE;
    if (elem_bytes != 4 && elem_bytes != 8) return LTR_ERR_KIND;
    if (B == 0) return LTR_OK;
    if (!x || !out) return LTR_ERR_NULL;
    if (B > 65535) return LTR_ERR_SHAPE;
    const size_t LL = (size_t)L * L;
    unsigned gx = grid_for(LL, 256);
    if (elem_bytes == 4)
        hipLaunchKernelGGL((batch_pairs_kernel<uint32_t, uint2>), dim3(gx, (unsigned)B), dim3(256), 0,
                           (hipStream_t)stream, (const uint32_t *)x, L, (uint2 *)out);
    else
        hipLaunchKernelGGL((batch_pairs_kernel<uint64_t, ulonglong2>), dim3(gx, (unsigned)B),
                           dim3(256), 0, (hipStream_t)stream, (const uint64_t *)x, L,
                           (ulonglong2 *)out);
    return (int)hipGetLastError();
}

int ltr_plackettluce_keys_f32(const float *scores, const int64_t *n, const float *u, int B, int L,
                              float *keys, void *stream)
{
    LTR_CLEAR_STALE_ERROR();
    if (B < 0 || L <= 0) return LTR_ERR_SHAPE;
    if (B == 0) return LTR_OK;
    if (!scores || !n || !u || !keys) return LTR_ERR_NULL;
    hipLaunchKernelGGL(plackettluce_keys_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream,
                       scores, n, u, L, keys);
    return (int)hipGetLastError();
}

int ltr_pbm_clicks(const int64_t *rankings, const int64_t *ys, const int64_t *n,
                   const float *relevance_probs, int n_probs, const float *u, int B, int L,
                   int cutoff, float eta, int64_t *clicks, float *propensities, void *stream)
{
    LTR_CLEAR_STALE_ERROR();
    if (B < 0 || L <= 0 || n_probs <= 0) return LTR_ERR_SHAPE;
    if (B == 0) return LTR_OK;
    if (!rankings || !ys || !n || !relevance_probs || !u || !clicks || !propensities) return LTR_ERR_NULL;
    const size_t total = (size_t)B * L;
    hipLaunchKernelGGL(pbm_clicks_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       rankings, ys, n, relevance_probs, n_probs, u, total, L, cutoff, eta, clicks,
                       propensities);
    return (int)hipGetLastError();
}

int ltr_collate_pad_f32(const float *xs, const int64_t *ys, const int64_t *offsets,
                        const int64_t *qidx, const int64_t *sel, int Q, int B, int L, int F,
                        float *out_x, int64_t *out_y, int64_t *out_n, void *stream)
{
    LTR_CLEAR_STALE_ERROR();
    if (B < 0 || L <= 0 || F <= 0 || Q <= 0) return LTR_ERR_SHAPE;
    if (B == 0) return LTR_OK;
    if (!xs || !ys || !offsets || !qidx || !out_x || !out_y || !out_n) return LTR_ERR_NULL;
    const bool vec = (F % 4 == 0) && (((uintptr_t)xs | (uintptr_t)out_x) % 16 == 0);
    const int C = vec ? F / 4 : F;
    const size_t per_query = (size_t)L * C;
    unsigned gy = (unsigned)((per_query + 256 * 8 - 1) / (256 * 8));
    if (gy < 1) gy = 1;
    if (gy > 64) gy = 64;
    const dim3 grid((unsigned)B, gy), block(256);
    if (vec)
        hipLaunchKernelGGL((collate_pad_kernel<float4>), grid, block, 0, (hipStream_t)stream,
                           (const float4 *)xs, ys, offsets, qidx, sel, Q, L, C, (float4 *)out_x, out_y,
                           out_n);
    else
        hipLaunchKernelGGL((collate_pad_kernel<float>), grid, block, 0, (hipStream_t)stream, xs, ys,
                           offsets, qidx, sel, Q, L, C, out_x, out_y, out_n);
    return (int)hipGetLastError();
}

int ltr_collate_pad_csr_f32(const int64_t *indptr, const int32_t *indices, const float *values,
                            const int64_t *ys, const int64_t *offsets, const int64_t *qidx, const int64_t *sel,
                            int Q, int B, int L, int F, float *out_x, int64_t *out_y, int64_t *out_n, void *stream)
{
    LTR_CLEAR_STALE_ERROR();
    if (B < 0 || L <= 0 || F <= 0 || Q <= 0) return LTR_ERR_SHAPE;
    if ((size_t)F * 4 * 4 > kLdsBudget) return LTR_ERR_SHAPE;          // four row buffers per workgroup
    if (B == 0) return LTR_OK;
    if (!indptr || !indices || !values || !ys || !offsets || !qidx || !out_x || !out_y || !out_n) return LTR_ERR_NULL;
    unsigned gy = (unsigned)((L + 3) / 4);
    if (gy > 64) gy = 64;
    const size_t lds = (size_t)F * 4 * 4;
    LTR_ENSURE_LDS(collate_pad_csr_kernel, lds);
    hipLaunchKernelGGL(collate_pad_csr_kernel, dim3((unsigned)B, gy), dim3(256), lds, (hipStream_t)stream,
                       indptr, indices, values, ys, offsets, qidx, sel, Q, L, F, out_x, out_y, out_n);
    return (int)hipGetLastError();
}

}  // extern "C"

#include "ltr_f64.inc"
