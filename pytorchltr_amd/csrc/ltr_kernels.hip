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
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ltr_hip.h"

namespace {

constexpr int kWave = 64;
constexpr int kMaxListLen = 4096;
constexpr int kSymMaxLen = 1024;   // longest list the symmetric pair pass takes (LDS: one slice per wave)
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

// ---------------------------------------------------------------------------------
// Device status word.  The only kernels that can fail at run time are the ones whose workgroups
// wait for each other (ltr_cluster.inc): a wait that gives up must not end as a silent NaN.  One
// pinned, device-mapped host page per process holds a sticky status word; a kernel that gives up
// stores LTR_ERR_TIMEOUT there (system-scope store, rare path), and the linear-scorer entry
// points return it -- without any synchronisation: a host read of pinned memory -- on the NEXT
// call; ltr_device_status() reads / clears it explicitly.  Allocated lazily outside stream capture;
// until then (or when the allocation fails) the kernels only poison their outputs with NaN.
// ---------------------------------------------------------------------------------
// Every launcher reports its launch with hipGetLastError(), which returns (and clears) the LAST error
// of the calling thread -- including one an unrelated earlier call left behind.  The entry points
// therefore drop stale state first, so that a non-zero return is about THIS call.
#define LTR_CLEAR_STALE_ERROR() ((void)hipGetLastError())

struct StatusPage { int *host; int *dev; };
inline StatusPage &status_page_ref() { static StatusPage sp = {nullptr, nullptr}; return sp; }
inline int *status_device_ptr(hipStream_t stream)
{
    StatusPage &sp = status_page_ref();
    if (sp.dev) return sp.dev;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
        (void)hipGetLastError();
        return nullptr;
    }
    static bool tried = false;
    if (tried) return nullptr;
    tried = true;
    void *h = nullptr, *d = nullptr;
    if (hipHostMalloc(&h, 64, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    for (int i = 0; i < 16; ++i) reinterpret_cast<volatile int *>(h)[i] = 0;
    if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) { (void)hipGetLastError(); (void)hipHostFree(h); return nullptr; }
    sp.host = reinterpret_cast<int *>(h);
    sp.dev = reinterpret_cast<int *>(d);
    return sp.dev;
}
inline int status_peek()
{
    const StatusPage &sp = status_page_ref();
    return sp.host ? *reinterpret_cast<volatile int *>(sp.host) : 0;
}

// ---------------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------------
__device__ __forceinline__ float load_label(const void *rel, int dtype, size_t idx)
{
    // wave-uniform switch; int64 -> fp32 narrowing happens in registers (no cast kernel)
    if (dtype == LTR_LABEL_I64) return (float)((const int64_t *)rel)[idx];
    if (dtype == LTR_LABEL_F32) return ((const float *)rel)[idx];
    return (float)((const int32_t *)rel)[idx];
}

// Stage one query's (score, label) pairs into LDS.  The label-dtype switch is hoisted out of
// the loop (wave-uniform), int64 -> fp32 narrowing happens in registers (no cast kernel).
template <typename LabelT>
__device__ __forceinline__ void stage_rows_t(float2 *sy, const float *__restrict__ srow,
                                             const LabelT *__restrict__ yrow, int L, int nb,
                                             int tid, int T)
{
    for (int m = tid; m < L; m += T) {
        const float sv = srow[m];
        const float yv = (float)yrow[m];
        if (m < nb) sy[m] = make_float2(sv, yv);
    }
}

__device__ __forceinline__ void stage_rows(float2 *sy, const float *srow, const void *rel,
                                           int dtype, size_t row, int L, int nb, int tid, int T)
{
    if (dtype == LTR_LABEL_I64)
        stage_rows_t(sy, srow, (const int64_t *)rel + row, L, nb, tid, T);
    else if (dtype == LTR_LABEL_F32)
        stage_rows_t(sy, srow, (const float *)rel + row, L, nb, tid, T);
    else
        stage_rows_t(sy, srow, (const int32_t *)rel + row, L, nb, tid, T);
}

// Labels only (the fused scorer kernels compute the scores themselves).
template <typename LabelT>
__device__ __forceinline__ void stage_labels_t(float2 *sy, const LabelT *__restrict__ yrow, int L,
                                               int nb, int tid, int T)
{
    for (int m = tid; m < L; m += T) {
        const float yv = (float)yrow[m];
        if (m < nb) sy[m].y = yv;
    }
}

__device__ __forceinline__ void stage_labels(float2 *sy, const void *rel, int dtype, size_t row,
                                             int L, int nb, int tid, int T)
{
    if (dtype == LTR_LABEL_I64) stage_labels_t(sy, (const int64_t *)rel + row, L, nb, tid, T);
    else if (dtype == LTR_LABEL_F32) stage_labels_t(sy, (const float *)rel + row, L, nb, tid, T);
    else stage_labels_t(sy, (const int32_t *)rel + row, L, nb, tid, T);
}

__device__ __forceinline__ int clamp_n(int64_t n, int L)
{
    return n < 0 ? 0 : (n > (int64_t)L ? L : (int)n);
}

// Which query a workgroup takes (one workgroup per query: the fused scorer+loss kernels and the
// loss kernel).  The hardware starts block ids in
// order, round-robin over the 8 XCDs and, inside an XCD, over its 32 CUs (a CU hosts ids i, i+256,
// i+512, ...), and a launch of a few workgroups per CU lasts as long as its most loaded CU / its
// last workgroup.  Measured at C2/C3 shapes (n ~ U[1,128], B = 1024; profiles/README.md): batch in
// random order 13.5-14.0 us (hinge) / 23.7 (LambdaNDCG2) / 18.2 (logistic); the same batch sorted
// by n descending 12.2 / 20.0 / 15.3 -- long lists first, and every CU gets one list of each
// quartile.  A full sort of n[] inside every workgroup costs more than that (tried: counting
// sort, 8 000-15 000 cycles), so the order is approximated with 64-query samples: the batch is cut
// into G = ceil(B/64) interleaved groups (group g = chunks of 8 consecutive ids, G chunks apart),
// each group is ranked by n descending, and block id `pos` -- member number m of its group -- takes
// the group's m-th longest list.  Early ids get every group's longest lists, late ids the shortest:
// 12.1 / 20.3 / 15.5 us with the permutation applied on the host.  ONE wave finds the query
// without LDS traffic: one n per lane (8 x 64-byte segments), a radix select over the bits of n
// with ballots (uniform control flow, ~100 instructions), ties by lane order.  A bijection inside
// each group and a pure function of n[]; every output is indexed by the query, so results do not
// depend on it.
#ifndef LTR_SCHED_MAX_PER_CU
#define LTR_SCHED_MAX_PER_CU 4        // register-tile kernel (B = 8 x #CUs: 22.1 -> 23.9 us with it)
#endif
#ifndef LTR_LOSS_SCHED_MAX_PER_CU
#define LTR_LOSS_SCHED_MAX_PER_CU 16   // loss kernel, general fused kernel
#endif
__device__ __forceinline__ int sched_query_sampled(const int64_t *__restrict__ n, int B, int L, int G, int tid,
                                                   int &nb_out, int pos)
{
    __shared__ int s_sel[2];
    if (tid < 64) {
        const int lane = tid;
        const int u = pos >> 3;
        const int jp = u / G;
        const int gam = u - jp * G;
        int rho = jp * 8 + (pos & 7);                            // this block's member number
        const int id = ((lane >> 3) * G + gam) * 8 + (lane & 7); // member `lane` of the group
        unsigned long long cand = __ballot(id < B);              // a prefix of the lanes (ids grow with lane)
        const int key = clamp_n(n[min(id, B - 1)], L);
        // (all lists equally long, e.g. full lists: cand stays the whole group and rho the lane)
        const bool flat = __ballot(key != __builtin_amdgcn_readfirstlane(key)) == 0ull;
        // (LTR_SCHED_LOW_BITS low bits of n are ignored: an approximate order balances as well and
        // every bit is one more dependent ballot round in front of the first load)
#ifndef LTR_SCHED_LOW_BITS
#define LTR_SCHED_LOW_BITS 3
#endif
        for (int bt = flat ? -1 : 31 - __builtin_clz(L); bt >= LTR_SCHED_LOW_BITS; --bt) {   // descending n: set bits first
            const unsigned long long m = __ballot(((key >> bt) & 1) != 0) & cand;
            const int c = __popcll(m);
            if (rho < c) cand = m;
            else { rho -= c; cand &= ~m; }
        }
        // cand = the lanes holding the selected n; the rho-th of them in lane order takes the block
        const int below = __popcll(cand & ((1ull << lane) - 1ull));
        if ((((cand >> lane) & 1ull) != 0ull) && below == rho) { s_sel[0] = id; s_sel[1] = key; }
    }
    __syncthreads();
    nb_out = __builtin_amdgcn_readfirstlane(s_sel[1]);           // saves the dependent n[b] load
    return __builtin_amdgcn_readfirstlane(s_sel[0]);
}

// Cross-lane adds on the DPP path (no LDS round trip; HIP's __shfl_* lower to ds_bpermute).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_move(float v)
{
    // lanes outside ROW_MASK (or without a valid source) receive 0
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}

// Sum over the 4 lanes of a quad; every lane of the quad gets the result.
__device__ __forceinline__ float quad_sum(float v)
{
    v += dpp_move<0xB1, 0xF>(v);      // quad_perm [1,0,3,2]
    v += dpp_move<0x4E, 0xF>(v);      // quad_perm [2,3,0,1]
    return v;
}

// Sum over the 64 lanes of the wave; every lane gets the result (DPP row ops + one readlane).
__device__ __forceinline__ float wave_sum(float v)
{
    v = quad_sum(v);
    v += dpp_move<0x141, 0xF>(v);     // row_half_mirror: 8-lane sums
    v += dpp_move<0x140, 0xF>(v);     // row_mirror: every lane holds its 16-lane row sum
    v += dpp_move<0x142, 0xA>(v);     // row_bcast15 -> rows 1 and 3 add the previous row
    v += dpp_move<0x143, 0xC>(v);     // row_bcast31 -> rows 2 and 3 add rows 0+1: lane 63 = total
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// Sum over the whole workgroup; every thread gets the result.  Deterministic order.
// `red` is an LDS scratch of >= 16 floats.  Contains barriers: call from uniform code.
__device__ __forceinline__ float block_sum(float v, float *red)
{
    v = wave_sum(v);
    const int nw = blockDim.x >> 6;
    if (nw == 1) return v;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}

// Two sums in one pass (same barriers): used for (loss, sum of gradients).
__device__ __forceinline__ void block_sum2(float &a, float &b, float *red)
{
    a = wave_sum(a);
    b = wave_sum(b);
    const int nw = blockDim.x >> 6;
    if (nw == 1) return;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { red[2 * (threadIdx.x >> 6)] = a; red[2 * (threadIdx.x >> 6) + 1] = b; }
    __syncthreads();
    float ta = 0.f, tb = 0.f;
    for (int i = 0; i < nw; ++i) { ta += red[2 * i]; tb += red[2 * i + 1]; }
    a = ta;
    b = tb;
}

// log2(1 + e) for e in [0, 1]; two-term series below 2^-10 where 1+e would round e away.
__device__ __forceinline__ float log2_1p(float e)
{
    const float direct = __builtin_amdgcn_logf(1.0f + e);      // v_log_f32 == log2
    const float series = e * (1.0f - 0.5f * e) * kLog2e;
    return e < 0.0009765625f ? series : direct;
}

// ---------------------------------------------------------------------------------
// counting rank: for every owned document k < nb,
//   rank_s[k] = #{m < nb : s_m > s_k or (s_m == s_k and m < k)}     (score, descending)
//   rank_y[k] = the same on labels (only if WITH_Y)
// This is rank_by_score (utils/tensor_operations.py:48-64) restricted to the real
// documents; padded documents rank at their own index (key -inf, index tie-break).
// rank arrays must be zeroed by the caller when msplit > 1 (partial counts are added).
// ---------------------------------------------------------------------------------
template <int DPT, bool WITH_Y>
__device__ __forceinline__ void count_ranks(const float2 *sy, int nb, int owners, int o,
                                            int m0, int m1, bool partial, int *rank_s,
                                            int *rank_y)
{
    for (int base = 0; base < nb; base += owners * DPT) {
        const int wave_first = base + (o & ~63);
        if (wave_first >= nb) continue;                       // wave-uniform
        float sk[DPT], yk[DPT];
        int cs[DPT], cy[DPT], kk[DPT];
#pragma unroll
        for (int c = 0; c < DPT; ++c) {
            kk[c] = base + o + c * owners;
            const bool valid = kk[c] < nb;
            const float2 v = valid ? sy[kk[c]] : make_float2(0.f, 0.f);
            sk[c] = v.x; yk[c] = v.y; cs[c] = 0; cy[c] = 0;
        }
#pragma unroll 4
        for (int m = m0; m < m1; ++m) {
            const float2 v = sy[m];
#pragma unroll
            for (int c = 0; c < DPT; ++c) {
                const bool before = m < kk[c];
                cs[c] += (v.x > sk[c]) | ((v.x == sk[c]) & before);
                if (WITH_Y) cy[c] += (v.y > yk[c]) | ((v.y == yk[c]) & before);
            }
        }
#pragma unroll
        for (int c = 0; c < DPT; ++c) {
            if (kk[c] < nb) {
                if (partial) {
                    atomicAdd(&rank_s[kk[c]], cs[c]);
                    if (WITH_Y) atomicAdd(&rank_y[kk[c]], cy[c]);
                } else {
                    rank_s[kk[c]] = cs[c];
                    if (WITH_Y) rank_y[kk[c]] = cy[c];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------
// pair terms.  Conventions: owner document k (registers), streamed document m (LDS).
//   c1 = sigma * log2(e)   so that exp(-sigma*d) = exp2(-c1*d)
// Gradient accumulators are in units of sigma/ln2 (applied once per document at the end).
// ---------------------------------------------------------------------------------

// Hinge (loss/pairwise_additive.py:107-113): term(i,j) = max(0, 1-(s_i-s_j)) if y_i > y_j.
// The reference zeroes `loss < 0` (strict), so at exactly the margin the gradient flows.
__device__ __forceinline__ void pair_hinge(float sk, float yk, float sm, float ym, float &g,
                                           float &l)
{
    const float d = sk - sm;
    const bool gt = yk > ym, lt = yk < ym;
    const float u = gt ? (1.0f - d) : (1.0f + d);
    const bool act = (gt | lt) & (u >= 0.0f);
    l += (gt & act) ? u : 0.0f;
    g += act ? (gt ? -1.0f : 1.0f) : 0.0f;
}

// Logistic / ARP2 / NDCG2: term(i,j) = W_ij * log2(1 + exp(-sigma (s_i - s_j))) if y_i > y_j
// (loss/pairwise_additive.py:158-163, loss/pairwise_lambda.py:135-140, :198-218), W symmetric.
// Stable for any finite input (the reference overflows beyond |sigma d| ~ 88).
__device__ __forceinline__ void pair_oriented(float sk, float yk, float sm, float ym, float W,
                                              float c1, float &g, float &l)
{
    const bool gt = yk > ym, lt = yk < ym;
    const float t = (sk - sm) * c1;
    const float z = gt ? t : -t;                       // log2e * sigma * (s_winner - s_loser)
    const float e = __builtin_amdgcn_exp2f(-fabsf(z));
    const float r = __builtin_amdgcn_rcpf(1.0f + e);
    const float psi = (z >= 0.0f) ? e * r : r;         // sigmoid(-sigma (s_win - s_lose))
    const float lg = log2_1p(e) + fmaxf(-z, 0.0f);     // log2(1 + exp(-sigma (s_win - s_lose)))
    const float Wm = (gt | lt) ? W : 0.0f;
    l += gt ? Wm * lg : 0.0f;
    g += (gt ? -Wm : Wm) * psi;
}

// ARP1 / NDCG1: term(i,j) = a_i * log2(1 + exp(-sigma (s_i - s_j))) for ALL i, j < n
// (-log2(sigmoid ** a_i), loss/pairwise_lambda.py:114-117, :165-173).
__device__ __forceinline__ void pair_rowweight(float sk, float ak, float sm, float am, float c1,
                                               float &g, float &l)
{
    const float t = (sk - sm) * c1;
    const float e = __builtin_amdgcn_exp2f(-fabsf(t));
    const float r = __builtin_amdgcn_rcpf(1.0f + e);
    const float sneg = (t >= 0.0f) ? e * r : r;        // sigmoid(-sigma (s_k - s_m))
    const float lg = log2_1p(e) + fmaxf(-t, 0.0f);
    l += ak * lg;
    g += am - (ak + am) * sneg;                        // -a_k sig(-x) + a_m sig(x)
}

// lists longer than this take the sort path (DPT == 0 instantiation of metric_kernel)
#ifndef LTR_SORT_RANK_MIN
#define LTR_SORT_RANK_MIN 256
#endif
constexpr int kSortRankMinLen = LTR_SORT_RANK_MIN;
__host__ __device__ inline int sort_pow2(int L)
{
    int P = 64;
    while (P < L) P <<= 1;
    return P;
}

// ---------------------------------------------------------------------------------
// Ranks of long lists by a bitonic sort instead of the O(n^2) counting rank.
// Every document becomes one 64-bit key: (order-reversed score bits << 32) | index, so that an
// ascending unsigned sort is exactly "score descending, index ascending" -- the tie rule of the
// counting rank (-0.0 is folded into +0.0 first so that it ties with it, as the float compare
// does).  Slots past n hold the all-ones sentinel and stay at the tail.  Element i = e*T + tid
// lives in register e of thread tid; a compare-exchange partner at distance j is in the same
// thread (j >= T), reached through LDS (64 <= j < T, one buffer, two barriers), or a lane
// shuffle (j < 64).  P = E*T is a power of two >= n.  O(P log^2 P) work: L = 1000 ranks in
// ~1/7 of the counting rank's time on MI355X.  NaN scores sort first (the counting rank gave
// them colliding ranks).
// ---------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long rank_key(float v, int idx)
{
    const unsigned bits = __float_as_uint(v + 0.0f);
    const unsigned asc = (bits & 0x80000000u) ? ~bits : (bits | 0x80000000u);
    return ((unsigned long long)(~asc) << 32) | (unsigned)idx;
}

// inv (LDS, may be null): the low key word is a tie priority, inv[priority] = document index.
template <int E>
__device__ __forceinline__ void sort_ranks(unsigned long long (&v)[E], int P, int nb, int *rank_out,
                                           unsigned long long *xbuf, const int *inv = nullptr)
{
    const int tid = threadIdx.x;
    const int T = blockDim.x;
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j >= T) {
                const int je = j / T;                          // partner register
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    if ((e & je) == 0 && (e | je) < E) {
                        const int i = e * T + tid;
                        const bool up = (i & k) == 0;
                        const unsigned long long a = v[e], b = v[e | je];
                        const bool swap = (a > b) == up;
                        v[e] = swap ? b : a;
                        v[e | je] = swap ? a : b;
                    }
                }
            } else {
                unsigned long long other[E];
                if (j >= 64) {
                    __syncthreads();                           // previous readers of xbuf are done
#pragma unroll
                    for (int e = 0; e < E; ++e) xbuf[e * T + tid] = v[e];
                    __syncthreads();
#pragma unroll
                    for (int e = 0; e < E; ++e) other[e] = xbuf[(e * T + tid) ^ j];
                } else {
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)v[e], j);
                        const unsigned hi = (unsigned)__shfl_xor((int)(unsigned)(v[e] >> 32), j);
                        other[e] = ((unsigned long long)hi << 32) | lo;
                    }
                }
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const int i = e * T + tid;
                    const bool keep_min = ((i & j) == 0) == ((i & k) == 0);
                    const bool smaller = v[e] < other[e];
                    v[e] = (smaller == keep_min) ? v[e] : other[e];
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
        if (v[e] != ~0ull) {
            const int low = (int)(unsigned)v[e];
            const int idx = inv ? inv[low] : low;
            if (idx < nb) rank_out[idx] = e * T + tid;
        }
    }
}

// Counting rank on packed keys: with (order-reversed value bits << 32 | index) keys (rank_key) the
// tie rule is part of the key, so "m comes before k" is ONE unsigned 64-bit compare and the count
// one add-with-carry -- about 2.5x fewer VALU instructions per pair than comparing floats and
// indices separately.  keys[k] = (score key, label key), built by the caller for k < nb.
template <int DPT, bool WITH_Y>
__device__ __forceinline__ void count_ranks_keyed(const ulonglong2 *keys, int nb, int owners, int o,
                                                  int m0, int m1, bool partial, int *rank_s,
                                                  int *rank_y)
{
    for (int base = 0; base < nb; base += owners * DPT) {
        const int wave_first = base + (o & ~63);
        if (wave_first >= nb) continue;                       // wave-uniform
        unsigned long long kx[DPT], ky[DPT];
        int cs[DPT], cy[DPT], kk[DPT];
#pragma unroll
        for (int c = 0; c < DPT; ++c) {
            kk[c] = base + o + c * owners;
            const ulonglong2 v = keys[kk[c] < nb ? kk[c] : 0];
            kx[c] = v.x; ky[c] = v.y; cs[c] = 0; cy[c] = 0;
        }
#pragma unroll 4
        for (int m = m0; m < m1; ++m) {
            const ulonglong2 v = keys[m];
#pragma unroll
            for (int c = 0; c < DPT; ++c) {
                cs[c] += (v.x < kx[c]);
                if (WITH_Y) cy[c] += (v.y < ky[c]);
            }
        }
#pragma unroll
        for (int c = 0; c < DPT; ++c) {
            if (kk[c] < nb) {
                if (partial) {
                    atomicAdd(&rank_s[kk[c]], cs[c]);
                    if (WITH_Y) atomicAdd(&rank_y[kk[c]], cy[c]);
                } else {
                    rank_s[kk[c]] = cs[c];
                    if (WITH_Y) rank_y[kk[c]] = cy[c];
                }
            }
        }
    }
}

struct LossParams {
    const float *scores;
    const void *rel;
    const int64_t *n;
    float *loss;
    float *dscores;
    int B, L;
    float sigma;
    int rel_dtype;
    int msplit;
    int sched;          // > 0: number of 64-query sample groups of the list-length scheduling
};

// LDS carve (bytes), L4 = L rounded up to 4:
//   sy    float2[L4]          (score, label) -- for NDCG1 the label slot is overwritten by a_k
//   q4    float4[L4]          NDCG2 only: (score, label, gain G, rank)
//   delta float [L4 + 4]      NDCG2 only: |1/D(d) - 1/D(d+1)|, D(d) = log2(2+d)
//   gpart float [msplit*L4]   per-slice gradient partials; aliased by the two int rank arrays
//   red   float [32]
__host__ __device__ inline size_t loss_lds_bytes(int kind, int L, int msplit)
{
    const size_t L4 = (size_t)((L + 3) & ~3);
    size_t bytes = 8 * L4;
    if (kind == LTR_NDCG2) bytes += 16 * L4 + 4 * (L4 + 4);
    size_t g = 4 * L4 * (size_t)msplit;
    if ((kind == LTR_NDCG1 || kind == LTR_NDCG2) && g < 8 * L4) g = 8 * L4;
    // room behind the rank arrays for the packed keys of the counting rank (16 B per document)
    if ((kind == LTR_NDCG1 || kind == LTR_NDCG2) && L4 <= 1024 && g < 24 * L4) g = 24 * L4;
    return bytes + g + 32 * 4;
}

// LDS views of one query, carved from the dynamic segment (see loss_lds_bytes).
struct QueryLds {
    float2 *sy;
    float4 *q4;
    float *delta;
    float *gpart;
    int *rank_s;
    int *rank_y;
    float *red;
    unsigned gbytes;      // bytes of the gpart region (the sort path borrows its tail)
};

template <int KIND>
__device__ __forceinline__ QueryLds carve_query_lds(unsigned char *base, int L4, int msplit)
{
    QueryLds q;
    q.sy = reinterpret_cast<float2 *>(base);
    unsigned char *cur = base + 8 * (size_t)L4;
    q.q4 = nullptr;
    q.delta = nullptr;
    if (KIND == LTR_NDCG2) {
        q.q4 = reinterpret_cast<float4 *>(cur);
        cur += 16 * (size_t)L4;
        q.delta = reinterpret_cast<float *>(cur);
        cur += 4 * (size_t)(L4 + 4);
    }
    q.gpart = reinterpret_cast<float *>(cur);
    q.rank_s = reinterpret_cast<int *>(cur);      // aliases gpart (dead before gpart is written)
    q.rank_y = q.rank_s + L4;
    size_t g = 4 * (size_t)L4 * msplit;
    if ((KIND == LTR_NDCG1 || KIND == LTR_NDCG2) && g < 8 * (size_t)L4) g = 8 * (size_t)L4;
    if ((KIND == LTR_NDCG1 || KIND == LTR_NDCG2) && L4 <= 1024 && g < 24 * (size_t)L4) g = 24 * (size_t)L4;
    cur += g;
    q.gbytes = (unsigned)g;
    q.red = reinterpret_cast<float *>(cur);
    return q;
}

// NDCG kinds: ranks by score and by label, maxDCG, gains.  On return (barrier passed)
//   NDCG1: sy[k].y = a_k = G_k / log2(2 + rank_k);   NDCG2: q4[k] = (s, y, G, rank), delta table.
// TW > 0: the workgroup has TW waves (compile-time): the maxDCG sum takes one barrier instead of two.
template <int KIND, int DPT, int TW = 0>
__device__ __forceinline__ void prepare_ndcg(const QueryLds &q, int nb, int owners, int o, int m0,
                                             int m1, bool partial)
{
    const int tid = threadIdx.x;
    const int T = TW > 0 ? TW * 64 : (int)blockDim.x;
    float2 *sy = q.sy;
    // Lists longer than kSortRankMinLen: both rankings by a bitonic sort of packed (key, index)
    // words (see sort_ranks) when the exchange buffer fits behind the rank arrays in the gpart
    // region -- it always does on the symmetric path (L <= 1024); otherwise the counting rank.
    const int L4r = (int)(q.rank_y - q.rank_s);
    int Pq = 64;
    while (Pq < nb) Pq <<= 1;
    // (every thread writes its register(s) to the exchange buffer: max(Pq, T) or 4T words)
    const unsigned xwords = (unsigned)(Pq <= T ? T : 4 * T);
    const bool sorted = nb > kSortRankMinLen && Pq <= 4 * T &&
                        q.gbytes >= 8u * (unsigned)L4r + 8u * xwords;
    if (sorted) {
        unsigned long long *xbuf = reinterpret_cast<unsigned long long *>(q.rank_s + 2 * L4r);
        if (Pq <= T) {
            unsigned long long v[1];
            v[0] = (tid < nb) ? rank_key(sy[tid].x, tid) : ~0ull;
            sort_ranks<1>(v, Pq, nb, q.rank_s, xbuf);
            v[0] = (tid < nb) ? rank_key(sy[tid].y, tid) : ~0ull;
            sort_ranks<1>(v, Pq, nb, q.rank_y, xbuf);
        } else {
            unsigned long long v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (e * T + tid < nb) ? rank_key(sy[e * T + tid].x, e * T + tid) : ~0ull;
            sort_ranks<4>(v, Pq, nb, q.rank_s, xbuf);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (e * T + tid < nb) ? rank_key(sy[e * T + tid].y, e * T + tid) : ~0ull;
            sort_ranks<4>(v, Pq, nb, q.rank_y, xbuf);
        }
    } else if (q.gbytes >= 24u * (unsigned)L4r) {
        ulonglong2 *keys = reinterpret_cast<ulonglong2 *>(q.rank_s + 2 * L4r);   // behind the ranks
        for (int k = tid; k < nb; k += T) {
            const float2 v = sy[k];
            keys[k] = make_ulonglong2(rank_key(v.x, k), rank_key(v.y, k));
        }
        __syncthreads();
        count_ranks_keyed<DPT, true>(keys, nb, owners, o, m0, m1, partial, q.rank_s, q.rank_y);
    } else {
        count_ranks<DPT, true>(sy, nb, owners, o, m0, m1, partial, q.rank_s, q.rank_y);
    }
    __syncthreads();
    // _max_dcg (loss/pairwise_lambda.py:231-241): labels sorted descending over the
    // first n documents, gains 2^y - 1, discounts log2(2 + r).
    float part = 0.f;
    for (int k = tid; k < nb; k += T)
        part += (exp2f(sy[k].y) - 1.0f) / log2f(2.0f + (float)q.rank_y[k]);
    float maxdcg;
    if (TW > 0) {
        const float ws = wave_sum(part);
        if ((tid & 63) == 0) q.red[tid >> 6] = ws;
        __syncthreads();
        maxdcg = 0.f;
#pragma unroll
        for (int i = 0; i < (TW > 0 ? TW : 1); ++i) maxdcg += q.red[i];
    } else {
        maxdcg = block_sum(part, q.red);
    }
    if (maxdcg == 0.0f) maxdcg = 1.0f;                     // pairwise_lambda.py:227
    const float inv_maxdcg = 1.0f / maxdcg;
    for (int k = tid; k < nb; k += T) {
        const float2 v = sy[k];
        const float G = (exp2f(v.y) - 1.0f) * inv_maxdcg;   // _ndcg_gains, :221-228
        const int r = q.rank_s[k];
        if (KIND == LTR_NDCG1)
            sy[k].y = G / log2f(2.0f + (float)r);           // a_k = G_k / D(rank_k)
        else
            q.q4[k] = make_float4(v.x, v.y, G, (float)r);
    }
    if (KIND == LTR_NDCG2)
        for (int d = tid; d < nb; d += T)                   // delta table, :206-211
            q.delta[d] = fabsf(1.0f / log2f(2.0f + (float)d) - 1.0f / log2f(3.0f + (float)d));
    __syncthreads();                                        // ranks dead from here: gpart may be written
}

// The per-query core shared by the loss kernel and the fused scorer kernel.
// Precondition: q.sy[0..nb) = (score, label) is staged and visible (barrier passed); for the
// NDCG kinds q.rank_s[0..2*L4) is zeroed.  Postcondition (after the trailing barrier):
// q.gpart[slice*L4 + k] holds the slice partials of d(pair sum)/d s_k in units of `gscale`;
// returns the final per-query loss (modifier applied) to every thread and, in `gsum`, the sum
// over documents of d loss / d s_k (= d loss / d bias of a linear scorer; ~0 by construction).
template <int KIND, int DPT, bool PIPE = true>
__device__ __forceinline__ float pairwise_core(const QueryLds &q, int nb, int L4, int msplit,
                                               float sigma, float &gscale, float &gsum)
{
    const int tid = threadIdx.x;
    const int T = blockDim.x;
    const int owners = T / msplit;                 // multiple of 64: slices are whole waves
    const int o = tid % owners;
    const int slice = tid / owners;
    float2 *sy = q.sy;

    // slice of the streamed index this thread's wave walks
    // (slices are whole waves, so these bounds are wave-uniform: keep them in SGPRs)
    const int mlen = ((nb + msplit - 1) / msplit + 1) & ~1;        // even: slices start 16-B aligned
    const int m0 = __builtin_amdgcn_readfirstlane(min(nb, slice * mlen));
    const int m1 = __builtin_amdgcn_readfirstlane(min(nb, m0 + mlen));
    const float c1 = sigma * kLog2e;

    if (KIND == LTR_NDCG1 || KIND == LTR_NDCG2)
        prepare_ndcg<KIND, DPT>(q, nb, owners, o, m0, m1, msplit > 1);

    // ---- pair pass ----
    float lacc = 0.f, gacc = 0.f;
    for (int base = 0; base < nb; base += owners * DPT) {
        const int wave_first = base + (o & ~63);
        if (wave_first >= nb) continue;                          // wave-uniform
        float sk[DPT], yk[DPT], Gk[DPT], rk[DPT], gk[DPT];
        int kk[DPT];
#pragma unroll
        for (int c = 0; c < DPT; ++c) {
            kk[c] = base + o + c * owners;
            const bool valid = kk[c] < nb;
            gk[c] = 0.f; Gk[c] = 0.f; rk[c] = 0.f;
            if (KIND == LTR_NDCG2) {
                const float4 v = valid ? q.q4[kk[c]] : make_float4(0.f, __builtin_nanf(""), 0.f, 0.f);
                sk[c] = v.x; yk[c] = v.y; Gk[c] = v.z; rk[c] = v.w;
            } else if (KIND == LTR_ARP1 || KIND == LTR_NDCG1) {
                const float2 v = valid ? sy[kk[c]] : make_float2(0.f, 0.f);   // a_k = 0: no loss
                sk[c] = v.x; yk[c] = v.y;
            } else {
                // NaN label: both orientation tests fail -> an idle owner contributes nothing
                const float2 v = valid ? sy[kk[c]] : make_float2(0.f, __builtin_nanf(""));
                sk[c] = v.x; yk[c] = v.y;
            }
        }
        // One streamed document against the DPT owned ones.
        auto visit = [&](float sm, float ym, float Gm, float rm) {
#pragma unroll
            for (int c = 0; c < DPT; ++c) {
                if (KIND == LTR_NDCG2) {
                    const int d = (int)fabsf(rk[c] - rm);
                    const float W = q.delta[d] * fabsf(Gk[c] - Gm);
                    pair_oriented(sk[c], yk[c], sm, ym, W, c1, gk[c], lacc);
                } else if (KIND == LTR_HINGE || KIND == LTR_DCG_HINGE) {
                    pair_hinge(sk[c], yk[c], sm, ym, gk[c], lacc);
                } else if (KIND == LTR_LOGISTIC) {
                    pair_oriented(sk[c], yk[c], sm, ym, 1.0f, c1, gk[c], lacc);
                } else if (KIND == LTR_ARP2) {
                    pair_oriented(sk[c], yk[c], sm, ym, fabsf(yk[c] - ym), c1, gk[c], lacc);
                } else {
                    pair_rowweight(sk[c], yk[c], sm, ym, c1, gk[c], lacc);
                }
            }
        };
        // Streamed documents come from LDS in 64-byte chunks (4 x ds_read_b128, wave-uniform
        // address = broadcast), software-pipelined one chunk ahead so the LDS latency hides under
        // the pair arithmetic instead of stalling every iteration.
        constexpr int MU = (KIND == LTR_NDCG2) ? 4 : 8;                  // documents per chunk
        const float4 *src = (KIND == LTR_NDCG2) ? q.q4 : reinterpret_cast<const float4 *>(sy);
        constexpr int VPD = (KIND == LTR_NDCG2) ? 1 : 2;                 // documents per float4
        // PIPE = false (register-tight callers): plain loop, no prefetch registers
        const int mfull = PIPE ? m0 + ((m1 - m0) / MU) * MU : m0;        // m0 is even (see mlen)
        if (mfull > m0) {
            float4 cur[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) cur[j] = src[m0 / VPD + j];
            for (int m = m0; m < mfull; m += MU) {
                const int mn = (m + MU < mfull) ? (m + MU) : m;          // last chunk: harmless re-read
                float4 nxt[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) nxt[j] = src[mn / VPD + j];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (KIND == LTR_NDCG2) {
                        visit(cur[j].x, cur[j].y, cur[j].z, cur[j].w);
                    } else {
                        visit(cur[j].x, cur[j].y, 0.f, 0.f);
                        visit(cur[j].z, cur[j].w, 0.f, 0.f);
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) cur[j] = nxt[j];
            }
        }
        for (int m = mfull; m < m1; ++m) {                               // remainder (< MU documents)
            if (KIND == LTR_NDCG2) {
                const float4 v = q.q4[m];
                visit(v.x, v.y, v.z, v.w);
            } else {
                const float2 v = sy[m];
                visit(v.x, v.y, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int c = 0; c < DPT; ++c)
            if (kk[c] < nb) { q.gpart[(size_t)slice * L4 + kk[c]] = gk[c]; gacc += gk[c]; }
    }

    // ---- per-query reduction (loss and gradient sum in one pass) and loss modifier ----
    float total = lacc;
    block_sum2(total, gacc, q.red);           // its barriers publish gpart when there are >= 2 waves
    if (T == kWave) __syncthreads();
    gscale = 1.0f;
    if (KIND == LTR_DCG_HINGE) {
        // _loss_modifier (loss/pairwise_additive.py:132-133): -1/ln(2+H);
        // d/dH = 1/((2+H) ln^2(2+H))
        const float lg = logf(2.0f + total);
        gscale = 1.0f / ((2.0f + total) * lg * lg);
        total = -1.0f / lg;
    } else if (KIND != LTR_HINGE) {
        gscale = sigma / kLn2;
    }
    gsum = gacc * gscale;
    return total;
}

// ---------------------------------------------------------------------------------
// Symmetric pair pass (list_len <= kSymMaxLen): every UNORDERED pair is evaluated once.
// Documents are cut into 64-wide tiles; a job is an unordered tile pair (a, b).  A wave keeps
// the "home" tile a in registers (lane i = document 64a+i) and a "visitor" tile b that ROTATES
// through the lanes: at step j lane i holds visitor 64b + ((i+j) & 63) -- its score, label and
// its gradient accumulator travel together, one v_mov_b32_dpp wave_rol:1 each per step (no LDS,
// no atomics).  One evaluation of the pair term updates the home gradient (+c) and the visitor
// gradient (-c): every loss on this path depends on score DIFFERENCES only, so the two
// contributions are exact negatives.  Diagonal jobs take j = 1..32 (the last step on half the
// lanes), off-diagonal jobs j = 0..63: 32*nt^2 steps in all, split evenly over the waves; each
// wave flushes into its PRIVATE gpart slice (deterministic).  This halves the VALU work of the
// both-ends formulation (pairwise_core), which remains the path for longer lists.
// Inert documents (index >= n[b]) carry NaN sentinels: NaN label for the oriented kinds (both
// label tests fail), NaN score for the row-weight kinds (detected with x == x).
// Precondition: sy[0..nb) staged (q4/delta/a_k prepared for the NDCG kinds), Lt = 64*ceil(L/64).
// Postcondition (after the trailing barrier): gpart[w*Lt + k], w < blockDim/64, hold partials of
// d(pair sum)/d s_k in units of gscale; returns the per-query loss.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ float wave_rol1(float v)
{
    // lane i <- lane (i+1) & 63; every lane has a valid source, so no "old" value is needed
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x134, 0xF, 0xF, true));
}

// part / parts: this workgroup is one of `parts` that share the query's pair units (split-query
// launch); raw: return the plain pair sum (no loss modifier, gscale untouched).
// TW > 0: the workgroup has TW waves (compile-time: no dispatch-packet read, shifts instead of
// divisions) and, when it owns the whole query, only as many of them take pair units as the query
// has work for (at least LTR_SYM_MIN_STEPS steps per wave, a power of two of waves).  Measured (C2,
// fused hinge step, us): 1 step per wave 11.3, 4: 11.6, 8: 11.9, 16: 12.3 -- the pass is latency-
// bound, more waves with fewer steps each win, so the default is 1 (every wave that can get a step).
// Idle waves still publish a zero gradient slice.
#ifndef LTR_SYM_MIN_STEPS
#define LTR_SYM_MIN_STEPS 1
#endif
#ifdef LTR_TRACE
#define LTR_SYM_STAMP(i) do { if (trace && threadIdx.x == 0) trace[i] = (long long)wall_clock64(); } while (0)
#else
#define LTR_SYM_STAMP(i) do { } while (0)
#endif
template <int KIND, int TW = 0>
__device__ __forceinline__ float pairwise_core_sym(const QueryLds &q, int nb, int Lt, float sigma,
                                                   float &gscale, int part = 0, int parts = 1,
                                                   bool raw = false, long long *trace = nullptr)
{
    const int tid = threadIdx.x;
    const int T = TW > 0 ? TW * 64 : (int)blockDim.x;
    const int lane = tid & 63;
    const int wl = __builtin_amdgcn_readfirstlane(tid >> 6);
    int w = part * (T >> 6) + wl;                            // wave index among all parts
    int W = parts * (T >> 6);
    const int nt = (nb + 63) >> 6;
    const float c1 = sigma * kLog2e;
    constexpr bool kRowWeight = (KIND == LTR_ARP1 || KIND == LTR_NDCG1);
    const float kNaN = __builtin_nanf("");
    float *gw = q.gpart + (size_t)wl * Lt;
    for (int i = lane; i < 64 * nt; i += 64) gw[i] = 0.f;

    float lacc = 0.f;
    // Steps per job: off-diagonal 64 (j = 0..63); diagonal 32 (j = 1..32), except that the last,
    // partially filled tile only has pairs at j < hc (hc = its document count) while hc <= 32.
    const int hc_last = nb - 64 * (nt - 1);
    const int dlast = (hc_last - 1 < 32) ? (hc_last - 1) : 32;
    const int S = (nt > 0) ? (32 * nt * nt - (32 - dlast)) : 0;
    int u, u1;
    if (TW > 0 && parts == 1 && (TW & (TW - 1)) == 0) {
        // waves that take units: the largest power of two <= S / LTR_SYM_MIN_STEPS, in [1, TW]
        int sh = 0;
        while ((2 << sh) <= TW && (2 << sh) * LTR_SYM_MIN_STEPS <= S) ++sh;
        W = 1 << sh;
        u = (wl * S) >> sh;
        u1 = (wl < W) ? ((wl + 1) * S) >> sh : u;
        if (wl >= W) u = u1 = 0;
        w = wl;
    } else if (TW > 0 && parts == 1) {
        u = (wl * S) / TW;                                   // (division by a constant)
        u1 = ((wl + 1) * S) / TW;
    } else {
        u = (w * S) / W;
        u1 = ((w + 1) * S) / W;
    }
    // decode the first unit of this wave into (a, b, j): row a of the job triangle holds its
    // diagonal job (32 steps, dlast in the last row) and 64 steps for every b > a -- walk the rows
    // (<= nt iterations), then the column is a division (a job-by-job walk cost up to nt^2/2
    // iterations per wave, a quarter of a split-launch part's whole work at n = 1000)
    int a = 0, rem = u;
    while (a < nt - 1) {
        const int rowlen = 32 + 64 * (nt - 1 - a);
        if (rem < rowlen) break;
        rem -= rowlen;
        ++a;
    }
    int b = a;
    {
        const int dl = (a == nt - 1) ? dlast : 32;
        if (rem >= dl && a < nt - 1) {
            rem -= dl;
            b = a + 1 + (rem >> 6);
            rem &= 63;
        }
    }
    int j = ((a == b) ? 1 : 0) + rem;
    LTR_SYM_STAMP(0);                                        // slices zeroed, units decoded

    while (u < u1) {
        const bool diag = (a == b);
        const int jend = diag ? (1 + ((a == nt - 1) ? dlast : 32)) : 64;
        // ---- load the home tile of this segment ----
        const int hidx = 64 * a + lane;
        const bool hvalid = hidx < nb;
        float sh, yh, Gh = 0.f, rh = 0.f;
        if (KIND == LTR_NDCG2) {
            const float4 hv = q.q4[hvalid ? hidx : 0];
            sh = hv.x; yh = hvalid ? hv.y : kNaN; Gh = hv.z; rh = hv.w;
        } else {
            const float2 hv = q.sy[hvalid ? hidx : 0];
            if (kRowWeight) { sh = hvalid ? hv.x : kNaN; yh = hv.y; }
            else { sh = hv.x; yh = hvalid ? hv.y : kNaN; }
        }
        float gh = 0.f;
        // A visitor chain: the visitor tile rotated by `rot` lanes, its gradient accumulator travelling
        // with it.  LambdaNDCG2: the discount difference delta(|rank_home - rank_visitor|) comes
        // from an LDS table; it is fetched ONE STEP AHEAD (the next visitor's rank is one rotation
        // away), so the LDS round trip is off the dependency chain of the step.
        struct Vis { float sv, yv, Gv, rv, gv, dcur; };
        auto load_vis = [&](Vis &V, int rot) {
            const int vidx = 64 * b + ((lane + rot) & 63);
            const bool vvalid = vidx < nb;
            V.Gv = 0.f; V.rv = 0.f; V.gv = 0.f; V.dcur = 0.f;
            if (KIND == LTR_NDCG2) {
                const float4 vv = q.q4[vvalid ? vidx : 0];
                V.sv = vv.x; V.yv = vvalid ? vv.y : kNaN; V.Gv = vv.z; V.rv = vv.w;
                V.dcur = q.delta[(int)fabsf(rh - V.rv)];
            } else {
                const float2 vv = q.sy[vvalid ? vidx : 0];
                if (kRowWeight) { V.sv = vvalid ? vv.x : kNaN; V.yv = vv.y; }
                else { V.sv = vv.x; V.yv = vvalid ? vv.y : kNaN; }
            }
        };
        // one evaluation of the pair (home, visitor); `half`: only lanes < 32 count (j == 32)
        auto visit = [&](Vis &V, bool half) {
            const float sv = V.sv, yv = V.yv;
            const bool on = half ? (lane < 32) : true;
            float c;                                             // d term / d s_home
            if (KIND == LTR_HINGE || KIND == LTR_DCG_HINGE) {
                // sgn = -1 when the home document is the higher-labelled one, +1 when the visitor is,
                // 0 when the labels are equal or one of them is the NaN sentinel (ordered compares);
                // margin term u = |sgn| + sgn*(s_home - s_vis): 0 for an inert pair; the pair counts
                // iff u >= 0 (non-strict, as the reference leaves the gradient at the margin) -- one
                // compare feeding selects, no scalar mask arithmetic in the dependency chain
                float sgn = (yh < yv) ? 1.0f : 0.0f;
                sgn = (yh > yv) ? -1.0f : sgn;
                if (half) sgn = on ? sgn : 0.0f;
                const float uu = __builtin_fmaf(sgn, sh - sv, __builtin_fabsf(sgn));
                lacc += __builtin_fmaxf(uu, 0.0f);
                c = (uu >= 0.0f) ? sgn : 0.0f;
            } else if (!kRowWeight) {
                const bool gt = yh > yv, lt = yh < yv;
                float Wp = 1.0f;
                if (KIND == LTR_ARP2) Wp = fabsf(yh - yv);
                if (KIND == LTR_NDCG2) Wp = V.dcur * fabsf(Gh - V.Gv);
                const float t = (sh - sv) * c1;
                const float z = gt ? t : -t;
                const float e = __builtin_amdgcn_exp2f(-fabsf(z));
                const float r = __builtin_amdgcn_rcpf(1.0f + e);
                const float psi = (z >= 0.0f) ? e * r : r;
                const float lg = log2_1p(e) + fmaxf(-z, 0.0f);
                const float Wm = (on & (gt | lt)) ? Wp : 0.0f;
                lacc += Wm * lg;
                c = (gt ? -Wm : Wm) * psi;
            } else {
                const float x = (sh - sv) * c1;
                const bool ok = on & (x == x);                   // NaN score = inert document
                const float e = __builtin_amdgcn_exp2f(-fabsf(x));
                const float r = __builtin_amdgcn_rcpf(1.0f + e);
                const float sneg = (x >= 0.0f) ? e * r : r;
                const float l1 = log2_1p(e);
                const float both = yh * (l1 + fmaxf(-x, 0.0f)) + yv * (l1 + fmaxf(x, 0.0f));
                lacc += ok ? both : 0.0f;
                c = ok ? (yv - (yh + yv) * sneg) : 0.0f;
            }
            gh += c;
            V.gv -= c;
        };
        auto rotate = [&](Vis &V) {
            V.sv = wave_rol1(V.sv); V.yv = wave_rol1(V.yv); V.gv = wave_rol1(V.gv);
            if (KIND == LTR_NDCG2) { V.Gv = wave_rol1(V.Gv); V.rv = wave_rol1(V.rv); }
        };
        // a full step of a chain (the discount of the NEXT step requested first)
        auto step = [&](Vis &V) {
            float dnext = 0.f;
            if (KIND == LTR_NDCG2) dnext = q.delta[(int)fabsf(rh - wave_rol1(V.rv))];
            visit(V, false);
            rotate(V);
            V.dcur = dnext;
        };
        auto flush_vis = [&](const Vis &V, int rot) {
            const int fidx = 64 * b + ((lane + rot) & 63);       // the visitor now in this lane
            if (fidx < nb) gw[fidx] += V.gv;
        };

        const int steps = min(jend - j, u1 - u);
        u += steps;
        const int jstop = j + steps;
        // full steps (every lane), then at most one half step (diagonal job, j == 32): the LAST step
        const bool has_half = diag && jstop == 33;
        // (Two visitor chains per lane -- the steps [j, j+n/2) and [j+n/2, jstop) interleaved, two
        // independent dependency chains -- were measured and dropped: correct, but 2-10 % slower in
        // every kernel (C2 fused hinge 10.5 -> 11.2 us, LambdaNDCG2 16.2 -> 16.6, 256 x 1000 hinge
        // loss 49.7 -> 56.2): the second chain's registers and moves cost more than its ILP buys.)
        {
            Vis A;
            load_vis(A, j);
            const int jfull = has_half ? 32 : jstop;
            for (; j < jfull; ++j) step(A);
            if (has_half) { visit(A, true); rotate(A); ++j; }
            if (hvalid) gw[hidx] += gh;
            flush_vis(A, j);
        }
        if (j == jend) {
            if (++b == nt) { ++a; b = a; }
            j = (a == b) ? 1 : 0;
        }
    }
    if (kRowWeight && part == 0)              // the i == j terms: a_i * log2(1 + e^0) = a_i
        for (int k = tid; k < nb; k += T) lacc += q.sy[k].y;
    LTR_SYM_STAMP(1);                                        // steps done, accumulators flushed (wave 0)

    float total;
    if (TW > 0) {
        // every wave parks its sum in its own slot (the upper half of `red`: no reuse hazard with the
        // block sums of prepare_ndcg), ONE barrier publishes the sums and the gradient slices
        const float ws = wave_sum(lacc);
        if (lane == 0) q.red[16 + wl] = ws;
        __syncthreads();
        LTR_SYM_STAMP(2);                                    // barrier passed
        total = 0.f;
#pragma unroll
        for (int i = 0; i < (TW > 0 ? TW : 1); ++i) total += q.red[16 + i];
    } else {
        total = block_sum(lacc, q.red);       // its barriers publish gpart when there are >= 2 waves
        if (T == kWave) __syncthreads();
    }
    if (raw) return total;
    gscale = 1.0f;
    if (KIND == LTR_DCG_HINGE) {
        const float lg = logf(2.0f + total);
        gscale = 1.0f / ((2.0f + total) * lg * lg);
        total = -1.0f / lg;
    } else if (KIND != LTR_HINGE) {
        gscale = sigma / kLn2;
    }
    return total;
}

// NW > 0 (symmetric pass only): the workgroup has NW waves, known at compile time (4 / 8 / 16 by
// list length) -- the pair pass then needs no dispatch-packet read, no divisions and one barrier less.
template <int KIND, int DPT, int NW = 0>
__global__ void __launch_bounds__(NW > 0 ? NW * 64 : 1024)
pairwise_loss_kernel(LossParams p)
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
    if constexpr (kSym) {
        if (KIND == LTR_NDCG1 || KIND == LTR_NDCG2) {
            int owners = 64;
            while (owners < L4 && owners < T) owners *= 2;
            const int ms = T / owners;
            const int mlen = (nb + ms - 1) / ms;
            const int m0 = __builtin_amdgcn_readfirstlane(min(nb, (tid / owners) * mlen));
            const int m1 = __builtin_amdgcn_readfirstlane(min(nb, m0 + mlen));
            prepare_ndcg<KIND, 1, NW>(q, nb, owners, tid % owners, m0, m1, ms > 1);
        }
        total = pairwise_core_sym<KIND, NW>(q, nb, L4, p.sigma, gscale);
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
#ifndef LTR_SPLIT_WAVES
#define LTR_SPLIT_WAVES 4
#endif
#ifndef LTR_SPLIT_MAX
#define LTR_SPLIT_MAX 8
#endif
#ifndef LTR_SPLIT_MIN_STEPS
#define LTR_SPLIT_MIN_STEPS 16
#endif
__host__ __device__ inline int split_parts_for(int nb, int nsplit, int waves)
{
    // a part should have at least ~16 pair steps per wave to be worth a workgroup (swept 8..96 with the
    // list-length order in place: C4 hinge 34.7 us at 96, 31.7 at 48, 28.4 at 16)
    const int nt = (nb + 63) >> 6;
    const int units = 32 * nt * nt;
    int e = units / (LTR_SPLIT_MIN_STEPS * waves);
    e = e < 1 ? 1 : e;
    return e < nsplit ? e : nsplit;
}

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
    const QueryLds q = carve_query_lds<KIND>(smem, L4, T >> 6);
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
        if (KIND == LTR_NDCG2)
            for (int d = tid; d < nb; d += T)               // delta table, pairwise_lambda.py:206-211
                q.delta[d] = fabsf(1.0f / log2f(2.0f + (float)d) - 1.0f / log2f(3.0f + (float)d));
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

template <int OP, int DPT>
__global__ void __launch_bounds__(1024)
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
        if (p.tie) {
            invt = reinterpret_cast<int *>(smem + metric_lds_bytes_sort(L) - 4 * (size_t)L4);
            for (int j = tid; j < L; j += T) invt[p.tie[j]] = j;
            __syncthreads();
        }
        unsigned long long v[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int i = e * T + tid;
            v[e] = (i < nb) ? rank_key(sy[i].x, p.tie ? p.tie[i] : i) : ~0ull;
        }
        sort_ranks<E>(v, Pq, nb, rank_s, xbuf, invt);
        if (with_y) {
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int i = e * T + tid;
                v[e] = (i < nb) ? rank_key(sy[i].y, p.tie ? p.tie[i] : i) : ~0ull;
            }
            sort_ranks<E>(v, Pq, nb, rank_y, xbuf, invt);
        }
    } else {
        // counting rank on packed keys (see count_ranks_keyed); the keys live in the curve region
        ulonglong2 *keys = reinterpret_cast<ulonglong2 *>(curve);
        for (int k = tid; k < nb; k += T) {
            const float2 v = sy[k];
            const int t = p.tie ? p.tie[k] : k;         // tie priority (random tie-break) or index
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

// ---------------------------------------------------------------------------------
// host side: launch-shape heuristic and dispatch
// ---------------------------------------------------------------------------------
// (64 bytes short of the CU's 160 KB: kernels that use sched_query_sampled carry 8 bytes of static
// LDS, and static + dynamic must fit together)
constexpr size_t kLdsBudget = 160 * 1024 - 64;
// Raise a kernel's dynamic-LDS limit above the 64 KiB default once per device (the attribute is
// sticky), so steady-state launches -- including hipGraph capture -- issue no extra API calls.
constexpr int kMaxDevices = 64;
#define LTR_ENSURE_LDS(KERNEL, BYTES)                                                           \
    do {                                                                                        \
        static size_t cfg_[kMaxDevices] = {};                                                   \
        int dev_ = 0;                                                                           \
        (void)hipGetDevice(&dev_);                                                              \
        if ((size_t)(BYTES) > 64 * 1024 && dev_ >= 0 && dev_ < kMaxDevices &&                   \
            (size_t)(BYTES) > cfg_[dev_]) {                                                     \
            hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void *>(&KERNEL),        \
                                                hipFuncAttributeMaxDynamicSharedMemorySize,     \
                                                (int)kLdsBudget);                               \
            if (e_ != hipSuccess) return (int)e_;                                               \
            cfg_[dev_] = kLdsBudget;                                                            \
        }                                                                                       \
    } while (0)

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

// Loss kernels: lists up to kSymMaxLen take the symmetric pair pass (dpt == 0; measured on MI355X:
// C2 hinge 7.7 -> 6.5 us, NDCG2 20.3 -> 16.4 us, C5 35 -> 26 us, C4 62 -> 56 us);
// 4 waves per query for L <= 128, 8 up to 256, 16 above.
LaunchShape choose_loss_shape(int B, int L)
{
#ifndef LTR_NO_SYM
    if (L <= kSymMaxLen) {
        LaunchShape s;
        s.dpt = 0;
        s.owners = 64;
        s.msplit = (L <= 128) ? 4 : (L <= 256 ? 8 : 16);   // waves per query
        (void)B;
        return s;
    }
#endif
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
        LTR_ENSURE_LDS((pairwise_loss_kernel<KIND, 0, NWAVES>), lds);                           \
        hipLaunchKernelGGL((pairwise_loss_kernel<KIND, 0, NWAVES>), grid, block, lds, stream, p); \
    } while (0)
    switch (s.dpt) {
    case 0:
        // the shapes choose_loss_shape picks get a compile-time wave count; explicit (_cfg) shapes
        // with another block size run the run-time variant
        if (s.owners * s.msplit == 256) LTR_LAUNCH_SYM(4);
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

inline int device_cu_count()
{
    static int cached[kMaxDevices] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= kMaxDevices) dev = 0;
    if (cached[dev] == 0) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0)
            v = 256;
        cached[dev] = v;
    }
    return cached[dev];
}

// Number of sample groups for sched_query_sampled, or 0 = plain order: worth it with more than one
// workgroup per CU and up to `max_per_cu` of them (beyond that the order stops mattering).
inline int sched_groups(int B, int max_per_cu)
{
#ifdef LTR_NO_SCHED
    (void)B; (void)max_per_cu;
    return 0;
#else
    const int cus = device_cu_count();
    return (B > cus + cus / 8 && (long long)B <= (long long)max_per_cu * cus) ? (B + 63) / 64 : 0;
#endif
}

// The loss kernel and the general fused kernel: the pass costs ~0.3 us per launch, which short
// lists do not win back (L = 64: 5.0 -> 5.2 us at B = 1024); lists of 128 need a full chip
// (B >= 4 x #CUs: hinge 7.0 -> 6.7, LambdaNDCG2 15.0 -> 13.2), longer ones always gain (L = 512,
// B = 1024: hinge 37 -> 28 us, logistic 75 -> 55, LambdaNDCG2 148 -> 120; profiles/README.md).
inline int sched_groups_for_lists(int B, int L)
{
    if (L <= 64) return 0;
    if (L < 256 && B < 4 * device_cu_count()) return 0;
    return sched_groups(B, LTR_LOSS_SCHED_MAX_PER_CU);
}

// How many workgroups share a query in the split launch (1 = use the one-kernel path): long lists
// only, and only while the batch alone cannot give every CU several queries to balance with.
constexpr int kSplitWaves = LTR_SPLIT_WAVES;   // waves per part: small workgroups, many per CU
static int choose_loss_splits(int kind, int B, int L)
{
    if (L <= 256 || L > kSymMaxLen) return 1;
    const int cus = device_cu_count();
    // measured (hinge / logistic, us, plain kernel with the list-length order -> split launch):
    // 384 x 1000: 35/66 either way; 512 x 1000: 58/135 -> 43/90; 768 x 1000: 71/163 -> 69/149;
    // 1024 x 1000: 72/175 -> 78/180; 512 x 512: 18/37 -> 20/35; 768 x 300: 14/21 -> 17/27
    // The NDCG kinds: rankings once per query by ndcg_prepare_kernel, then the same split (single
    // kernel -> split, LambdaNDCG1/2: 32 x 1000: 130/144 -> 51/53 us; 128 x 600: 65/71 -> 46/48;
    // C4 256 x 1000: 132/145 -> 82/90; 384 x 1000: 132/146 -> 114/123; 300 x 400: 41/43 -> 44/46)
    const bool ndcg = (kind == LTR_NDCG1 || kind == LTR_NDCG2);
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
        const size_t plds = loss_lds_bytes(PK, (p.L + 63) & ~63, 16);
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
        const int T = P < 1024 ? P : 1024;              // E = P / T <= 4 registers per thread
        p.msplit = 1;
        const size_t lds = metric_lds_bytes_sort(p.L);
        const dim3 sgrid((unsigned)p.B), sblock((unsigned)T);
        if (P <= 1024) {
            LTR_ENSURE_LDS((metric_kernel<OP, 0>), lds);
            hipLaunchKernelGGL((metric_kernel<OP, 0>), sgrid, sblock, lds, stream, p);
        } else if (P == 2048) {
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

inline bool bad_label_dtype(int d) { return d != LTR_LABEL_I64 && d != LTR_LABEL_F32 && d != LTR_LABEL_I32; }

inline unsigned grid_for(size_t items, int block)
{
    size_t g = (items + (size_t)block - 1) / (size_t)block;
    if (g < 1) g = 1;
    if (g > 2048) g = 2048;          // 256 CUs x 8 blocks, grid-stride the rest
    return (unsigned)g;
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
    if (clear) *w = 0;
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
        (dpt != 0 && dpt != 1 && dpt != 2 && dpt != 4) || (dpt == 0 && L > kSymMaxLen))
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
    LaunchShape s = choose_loss_shape(B, L);
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

}  // extern "C"

#include "ltr_linear.inc"
#include "ltr_f64.inc"
#include "ltr_mlp.inc"
#include "ltr_scorer.inc"
