// probe_r05.hip -- round-5 hardware probes (standalone; hipcc --offload-arch=gfx950 -O3 -o build/probe_r05 ...)
//   valu     : cycles per wave64 VALU instruction (v_fma_f32 / v_pk_fma_f32 / v_mov_dpp), 1..4 waves per SIMD
//   anyorder : does hipExtAnyOrderLaunch let a kernel start while its predecessor in the stream still runs?
//   xcc      : blockIdx -> HW_REG_XCC_ID map of a 1024-workgroup grid (alone and behind a running kernel)
//   pingpong : round-trip latency of a flag hand-off between two workgroups, same XCD / different XCD,
//              for every (store scope, load scope) pair and for RMW polling
// Every wait is bounded; nothing here can hang the GPU.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned xcc_id()
{
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}
__device__ __forceinline__ unsigned hw_id()
{
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v));
    return v;
}

// ------------------------------------------------------------------ valu
// MODE 0: v_fma_f32 (8 independent accumulators)   1: v_pk_fma_f32   2: v_fma_f32 dependent chain
//      3: v_mov_b32 dpp wave_rol:1 (8 independent)  4: v_add_f32 (8 independent)   5: v_exp_f32
template <int MODE>
__global__ void valu_kernel(long long *out, int iters)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float2 p0 = make_float2(a0, a1), p1 = make_float2(a2, a3), p2 = make_float2(a4, a5), p3 = make_float2(a6, a7);
    float2 p4 = p0, p5 = p1, p6 = p2, p7 = p3;
    const float m = 0.999f, c = 0.001f;
    const float2 m2 = make_float2(m, m), c2 = make_float2(c, c);
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    const long long w0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#define F8 "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n" \
           "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
            asm volatile(F8 F8 F8 F8 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
#undef F8
        } else if (MODE == 1) {
#define F8 "v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n" \
           "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
            asm volatile(F8 F8 F8 F8 : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(m2), "v"(c2));
#undef F8
        } else if (MODE == 2) {
#define F8 "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n" \
           "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
            asm volatile(F8 F8 F8 F8 : "+v"(a0) : "v"(m), "v"(c));
#undef F8
        } else if (MODE == 3) {
#define F8 "v_mov_b32_dpp %0, %0 wave_rol:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 wave_rol:1 row_mask:0xf bank_mask:0xf\n" \
           "v_mov_b32_dpp %2, %2 wave_rol:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 wave_rol:1 row_mask:0xf bank_mask:0xf\n" \
           "v_mov_b32_dpp %4, %4 wave_rol:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %5 wave_rol:1 row_mask:0xf bank_mask:0xf\n" \
           "v_mov_b32_dpp %6, %6 wave_rol:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %7 wave_rol:1 row_mask:0xf bank_mask:0xf\n"
            asm volatile(F8 F8 F8 F8 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
#undef F8
        } else if (MODE == 4) {
#define F8 "v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n" \
           "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
            asm volatile(F8 F8 F8 F8 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
#undef F8
        } else {
#define F8 "v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n" \
           "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
            asm volatile(F8 F8 F8 F8 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
#undef F8
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    const long long w1 = __builtin_amdgcn_s_memrealtime();
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.x + p2.x + p3.x + p4.y + p5.y + p6.y + p7.y;
    if ((threadIdx.x & 63) == 0) {
        const int w = threadIdx.x >> 6;
        out[w * 4 + 0] = t1 - t0;
        out[w * 4 + 1] = w1 - w0;
        out[w * 4 + 2] = (long long)s;
        out[w * 4 + 3] = hw_id();
    }
}

template <int MODE>
static void run_valu(const char *name, long long *d_out)
{
    const int iters = 2000;        // x 32 instructions
    for (int threads : {64, 256, 512, 1024}) {
        valu_kernel<MODE><<<1, threads>>>(d_out, iters);       // warm-up (clocks)
        valu_kernel<MODE><<<1, threads>>>(d_out, iters);
        CK(hipDeviceSynchronize());
        long long h[64];
        CK(hipMemcpy(h, d_out, sizeof(long long) * 4 * (threads / 64), hipMemcpyDeviceToHost));
        long long cyc = 0, wall = 0;
        for (int w = 0; w < threads / 64; ++w) { cyc = std::max(cyc, h[w * 4]); wall = std::max(wall, h[w * 4 + 1]); }
        const double n = (double)iters * 32;
        const int wps = std::max(1, threads / 256);            // waves per SIMD (a workgroup's waves go round-robin over 4 SIMDs)
        printf("valu %-12s threads %4d waves/SIMD %d: %.3f memtime-ticks/inst/wave, %.3f ns/inst/wave (100 MHz wall), "
               "=> SIMD issues one wave-inst per %.3f ns\n", name, threads, wps, cyc / n, wall * 10.0 / n, wall * 10.0 / n / wps);
    }
}

// ------------------------------------------------------------------ anyorder / xcc
__global__ void spin_kernel(long long *stamps, int us, int record_xcc)
{
    const long long t0 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        stamps[(size_t)blockIdx.x * 4 + 0] = t0;
        if (record_xcc) stamps[(size_t)blockIdx.x * 4 + 2] = ((long long)xcc_id() << 32) | (hw_id() & 0xffffffffu);
    }
    while (__builtin_amdgcn_s_memrealtime() - t0 < (long long)us * 100) __builtin_amdgcn_s_sleep(16);
    if (threadIdx.x == 0) stamps[(size_t)blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memrealtime();
}

static void minmax(const std::vector<long long> &v, int n, int field, long long &lo, long long &hi)
{
    lo = (1ll << 62); hi = 0;
    for (int i = 0; i < n; ++i) { lo = std::min(lo, v[(size_t)i * 4 + field]); hi = std::max(hi, v[(size_t)i * 4 + field]); }
}

static void run_anyorder(hipStream_t st)
{
    const int GA = 1024, GB = 1024;
    long long *dA, *dB;
    CK(hipMalloc(&dA, sizeof(long long) * 4 * GA));
    CK(hipMalloc(&dB, sizeof(long long) * 4 * GB));
    std::vector<long long> hA(4 * GA), hB(4 * GB);
    for (int flags = 0; flags <= 1; ++flags) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipMemsetAsync(dA, 0, sizeof(long long) * 4 * GA, st));
            CK(hipMemsetAsync(dB, 0, sizeof(long long) * 4 * GB, st));
            CK(hipStreamSynchronize(st));
            // A: 1024 workgroups x 512 threads spinning 30 us (fills the chip like the headline kernel); B right behind it
            hipLaunchKernelGGL(spin_kernel, dim3(GA), dim3(512), 0, st, dA, 30, 1);
            hipExtLaunchKernelGGL(spin_kernel, dim3(GB), dim3(512), 0, st, nullptr, nullptr, (unsigned)flags, dB, 5, 1);
            CK(hipGetLastError());
            CK(hipStreamSynchronize(st));
            CK(hipMemcpy(hA.data(), dA, sizeof(long long) * 4 * GA, hipMemcpyDeviceToHost));
            CK(hipMemcpy(hB.data(), dB, sizeof(long long) * 4 * GB, hipMemcpyDeviceToHost));
            long long a0, a1, ae0, ae1, b0, b1;
            minmax(hA, GA, 0, a0, a1); minmax(hA, GA, 1, ae0, ae1); minmax(hB, GB, 0, b0, b1);
            int xcc_ok = 0;
            for (int i = 0; i < GB; ++i) xcc_ok += ((hB[(size_t)i * 4 + 2] >> 32) == (i & 7));
            printf("anyorder flags=%d rep=%d: A starts %.2f..%.2f us, A ends %.2f..%.2f us; B starts %.2f..%.2f us (rel. to A's first start)"
                   "  -> B first start %s A's last end by %.2f us; B xcc==blockIdx%%8 for %d/%d\n",
                   flags, rep, 0.0, (a1 - a0) / 100.0, (ae0 - a0) / 100.0, (ae1 - a0) / 100.0, (b0 - a0) / 100.0, (b1 - a0) / 100.0,
                   b0 < ae1 ? "BEFORE" : "after", (b0 < ae1 ? (ae1 - b0) : (b0 - ae1)) / 100.0, xcc_ok, GB);
        }
    }
    // a small kernel A (does not fill the chip) and B any-order: does B overlap then?
    for (int flags = 0; flags <= 1; ++flags) {
        CK(hipMemsetAsync(dA, 0, sizeof(long long) * 4 * GA, st));
        CK(hipMemsetAsync(dB, 0, sizeof(long long) * 4 * GB, st));
        CK(hipStreamSynchronize(st));
        hipLaunchKernelGGL(spin_kernel, dim3(64), dim3(256), 0, st, dA, 30, 1);
        hipExtLaunchKernelGGL(spin_kernel, dim3(GB), dim3(512), 0, st, nullptr, nullptr, (unsigned)flags, dB, 5, 1);
        CK(hipGetLastError());
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(hA.data(), dA, sizeof(long long) * 4 * GA, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hB.data(), dB, sizeof(long long) * 4 * GB, hipMemcpyDeviceToHost));
        long long a0, a1, ae0, ae1, b0, b1;
        minmax(hA, 64, 0, a0, a1); minmax(hA, 64, 1, ae0, ae1); minmax(hB, GB, 0, b0, b1);
        printf("anyorder(small A) flags=%d: A ends %.2f us; B starts %.2f..%.2f us -> %s\n", flags, (ae1 - a0) / 100.0,
               (b0 - a0) / 100.0, (b1 - a0) / 100.0, b0 < ae1 ? "OVERLAP" : "serial");
    }
    // two streams: the same pair on different streams (what the hardware allows at all)
    {
        hipStream_t s2;
        CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
        CK(hipMemsetAsync(dA, 0, sizeof(long long) * 4 * GA, st));
        CK(hipMemsetAsync(dB, 0, sizeof(long long) * 4 * GB, st));
        CK(hipStreamSynchronize(st));
        hipLaunchKernelGGL(spin_kernel, dim3(GA), dim3(512), 0, st, dA, 30, 1);
        hipLaunchKernelGGL(spin_kernel, dim3(GB), dim3(512), 0, s2, dB, 5, 1);
        CK(hipStreamSynchronize(st));
        CK(hipStreamSynchronize(s2));
        CK(hipMemcpy(hA.data(), dA, sizeof(long long) * 4 * GA, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hB.data(), dB, sizeof(long long) * 4 * GB, hipMemcpyDeviceToHost));
        long long a0, a1, ae0, ae1, b0, b1;
        minmax(hA, GA, 0, a0, a1); minmax(hA, GA, 1, ae0, ae1); minmax(hB, GB, 0, b0, b1);
        int xcc_ok = 0;
        for (int i = 0; i < GB; ++i) xcc_ok += ((hB[(size_t)i * 4 + 2] >> 32) == (i & 7));
        printf("two streams: A ends %.2f..%.2f us; B starts %.2f..%.2f us; B xcc==blockIdx%%8 for %d/%d\n", (ae0 - a0) / 100.0,
               (ae1 - a0) / 100.0, (b0 - a0) / 100.0, (b1 - a0) / 100.0, xcc_ok, GB);
        CK(hipStreamDestroy(s2));
    }
    // back-to-back launch boundary of ordinary launches: 20 x (1024 x 512 threads, 2 us)
    {
        for (int flags = 0; flags <= 1; ++flags) {
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipStreamSynchronize(st));
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < 50; ++i)
                hipExtLaunchKernelGGL(spin_kernel, dim3(GA), dim3(512), 0, st, nullptr, nullptr, (unsigned)flags, dA, 2, 0);
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("50 back-to-back 2 us kernels (1024 x 512), flags=%d: %.2f us per launch\n", flags, ms * 1000.0 / 50);
        }
    }
    CK(hipFree(dA)); CK(hipFree(dB));
}

// ------------------------------------------------------------------ pingpong
// Memory operations with EXPLICIT scope bits (the compiler folds fetch_add(p, 0) into a load and prints agent-scope
// RMWs without sc1, so everything here is inline asm):
//   store kinds  0 plain  1 sc0  2 sc1  3 sc0 sc1  4 RMW swap (no bits)  5 RMW swap sc1
//   load kinds   0 plain  1 sc0  2 sc1  3 sc0 sc1  4 RMW add 0 returning (sc0)  5 RMW add 0 returning, sc0 sc1
template <int K> __device__ __forceinline__ void st_kind(unsigned *p, unsigned v)
{
    if (K == 0) asm volatile("global_store_dword %0, %1, off\n s_waitcnt vmcnt(0)" :: "v"(p), "v"(v) : "memory");
    else if (K == 1) asm volatile("global_store_dword %0, %1, off sc0\n s_waitcnt vmcnt(0)" :: "v"(p), "v"(v) : "memory");
    else if (K == 2) asm volatile("global_store_dword %0, %1, off sc1\n s_waitcnt vmcnt(0)" :: "v"(p), "v"(v) : "memory");
    else if (K == 3) asm volatile("global_store_dword %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" :: "v"(p), "v"(v) : "memory");
    else if (K == 4) asm volatile("global_atomic_swap %0, %1, off\n s_waitcnt vmcnt(0)" :: "v"(p), "v"(v) : "memory");
    else asm volatile("global_atomic_swap %0, %1, off sc1\n s_waitcnt vmcnt(0)" :: "v"(p), "v"(v) : "memory");
}
template <int K> __device__ __forceinline__ unsigned ld_kind(unsigned *p)
{
    unsigned v, z = 0;
    if (K == 0) asm volatile("global_load_dword %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (K == 1) asm volatile("global_load_dword %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (K == 2) asm volatile("global_load_dword %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (K == 3) asm volatile("global_load_dword %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (K == 4) asm volatile("global_atomic_add %0, %1, %2, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p), "v"(z) : "memory");
    else asm volatile("global_atomic_add %0, %1, %2, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p), "v"(z) : "memory");
    return v;
}
static const char *st_name[] = {"st plain", "st sc0", "st sc1", "st sc0sc1", "swap", "swap sc1"};
static const char *ld_name[] = {"ld plain", "ld sc0", "ld sc1", "ld sc0sc1", "rmw", "rmw sc1"};

// two workgroups (blockIdx 0 and `partner`) bounce a counter `rounds` times through flag[0]
template <int ST, int LD>
__global__ void pingpong_kernel(unsigned *flag, long long *out, int partner, int rounds, int spin_limit)
{
    if ((int)blockIdx.x != 0 && (int)blockIdx.x != partner) return;
    if (threadIdx.x != 0) return;
    const bool ping = blockIdx.x == 0;
    const long long t0 = __builtin_amdgcn_s_memrealtime();
    int fails = 0;
    for (int r = 0; r < rounds && !fails; ++r) {
        const unsigned want = ping ? 2u * r + 2u : 2u * r + 1u;
        if (ping) st_kind<ST>(flag, 2u * r + 1u);
        int spins = 0;
        while (ld_kind<LD>(flag) != want) { if (++spins > spin_limit) { fails = 1; break; } }
        if (!ping && !fails) st_kind<ST>(flag, 2u * r + 2u);
    }
    const long long t1 = __builtin_amdgcn_s_memrealtime();
    long long *o = out + (ping ? 0 : 4);
    o[0] = t1 - t0;
    o[1] = fails;
    o[2] = xcc_id();
    o[3] = hw_id();
}

template <int ST, int LD>
static void run_pp(unsigned *d_flag, long long *d_out)
{
    for (int partner : {8, 1}) {
        const int rounds = 200;
        double best = 1e30; int fails = 0; long long xa = -1, xb = -1;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemset(d_flag, 0, 256));
            CK(hipMemset(d_out, 0, 64));
            pingpong_kernel<ST, LD><<<16, 64>>>(d_flag, d_out, partner, rounds, 100000);
            CK(hipDeviceSynchronize());
            long long h[8];
            CK(hipMemcpy(h, d_out, 64, hipMemcpyDeviceToHost));
            fails |= (int)(h[1] | h[5]);
            xa = h[2]; xb = h[6];
            if (!(h[1] | h[5])) best = std::min(best, h[0] * 10.0 / rounds);
        }
        printf("pingpong %-10s / %-10s partner blk %d (xcc %lld <-> %lld): %s  %.0f ns per round trip\n", st_name[ST], ld_name[LD], partner, xa, xb,
               fails ? "NOT VISIBLE (gave up)" : "ok", fails ? 0.0 : best);
    }
}

// payload hand-off: producer writes `words` dwords then raises a flag; consumer polls the flag, reads the payload
// and checks it; the acknowledgement travels back the same way.  PST / PLD = kinds of the payload accesses,
// FST / FLD = kinds of the flag accesses.
template <int PST, int PLD, int FST, int FLD>
__global__ void handoff_kernel(unsigned *payload, unsigned *flag, long long *out, int partner, int words, int rounds, int spin_limit)
{
    if ((int)blockIdx.x != 0 && (int)blockIdx.x != partner) return;
    const bool prod = blockIdx.x == 0;
    const int tid = threadIdx.x;
    __shared__ int s_fail;
    if (tid == 0) s_fail = 0;
    __syncthreads();
    long long bad = 0;
    const long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int r = 0; r < rounds; ++r) {
        if (prod) {
            if (tid == 0) {
                int spins = 0;
                while (ld_kind<FLD>(flag + 64) != (unsigned)r) if (++spins > spin_limit) { s_fail = 1; break; }
            }
            __syncthreads();
            if (s_fail) break;
            for (int i = tid; i < words; i += blockDim.x) st_kind<PST>(payload + i, (unsigned)(r * 4096 + i));
            __syncthreads();
            if (tid == 0) st_kind<FST>(flag, (unsigned)(r + 1));
        } else {
            if (tid == 0) {
                int spins = 0;
                while (ld_kind<FLD>(flag) != (unsigned)(r + 1)) if (++spins > spin_limit) { s_fail = 1; break; }
            }
            __syncthreads();
            if (s_fail) break;
            for (int i = tid; i < words; i += blockDim.x) bad += (ld_kind<PLD>(payload + i) != (unsigned)(r * 4096 + i));
            __syncthreads();
            if (tid == 0) st_kind<FST>(flag + 64, (unsigned)(r + 1));
        }
    }
    const long long t1 = __builtin_amdgcn_s_memrealtime();
    if (bad) atomicAdd((unsigned long long *)(out + 6), (unsigned long long)bad);
    if (tid == 0) {
        long long *o = out + (prod ? 0 : 3);
        o[0] = t1 - t0;
        o[1] = s_fail;
        o[2] = xcc_id();
    }
}

template <int PST, int PLD, int FST, int FLD>
static void run_handoff(unsigned *d_payload, unsigned *d_flag, long long *d_out)
{
    for (int partner : {8, 1}) {
        const int rounds = 200, words = 1024;
        CK(hipMemset(d_flag, 0, 1024));
        CK(hipMemset(d_out, 0, 64));
        CK(hipMemset(d_payload, 0, words * 4));
        handoff_kernel<PST, PLD, FST, FLD><<<16, 256>>>(d_payload, d_flag, d_out, partner, words, rounds, 100000);
        CK(hipDeviceSynchronize());
        long long h[8];
        CK(hipMemcpy(h, d_out, 64, hipMemcpyDeviceToHost));
        printf("handoff payload %-10s / %-10s flag %-10s / %-10s partner blk %d (xcc %lld -> %lld): %s, %lld stale of %d, %.0f ns per round (flag + 4 KB + ack)\n",
               st_name[PST], ld_name[PLD], st_name[FST], ld_name[FLD], partner, h[2], h[5], (h[1] | h[4]) ? "GAVE UP" : "ok", h[6], rounds * words, h[0] * 10.0 / rounds);
    }
}

int main(int argc, char **argv)
{
    const char *what = argc > 1 ? argv[1] : "all";
    const bool all = !strcmp(what, "all");
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, %d CUs, clock %d kHz, wall-clock rate %d kHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate, 0);
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    long long *d_out;
    CK(hipMalloc(&d_out, 4096));
    if (all || !strcmp(what, "valu")) {
        run_valu<0>("v_fma_f32", d_out);
        run_valu<1>("v_pk_fma_f32", d_out);
        run_valu<4>("v_add_f32", d_out);
        run_valu<3>("v_mov_dpp", d_out);
        run_valu<5>("v_exp_f32", d_out);
        run_valu<2>("fma dependent", d_out);
    }
    if (all || !strcmp(what, "anyorder")) run_anyorder(st);
    if (all || !strcmp(what, "pingpong")) {
        unsigned *d_flag, *d_payload;
        CK(hipMalloc(&d_flag, 4096));
        CK(hipMalloc(&d_payload, 1 << 16));
        run_pp<0, 0>(d_flag, d_out);
        run_pp<0, 1>(d_flag, d_out);
        run_pp<0, 2>(d_flag, d_out);
        run_pp<0, 3>(d_flag, d_out);
        run_pp<0, 4>(d_flag, d_out);
        run_pp<1, 1>(d_flag, d_out);
        run_pp<2, 2>(d_flag, d_out);
        run_pp<3, 3>(d_flag, d_out);
        run_pp<4, 4>(d_flag, d_out);
        run_pp<5, 5>(d_flag, d_out);
        run_pp<4, 2>(d_flag, d_out);
        run_pp<2, 4>(d_flag, d_out);
        run_pp<2, 5>(d_flag, d_out);
        run_pp<3, 5>(d_flag, d_out);
        run_handoff<0, 0, 4, 4>(d_payload, d_flag, d_out);
        run_handoff<0, 1, 4, 4>(d_payload, d_flag, d_out);
        run_handoff<0, 2, 4, 4>(d_payload, d_flag, d_out);
        run_handoff<0, 3, 4, 4>(d_payload, d_flag, d_out);
        run_handoff<2, 2, 4, 4>(d_payload, d_flag, d_out);
        run_handoff<2, 2, 2, 2>(d_payload, d_flag, d_out);
        run_handoff<2, 2, 5, 5>(d_payload, d_flag, d_out);
        run_handoff<3, 3, 5, 5>(d_payload, d_flag, d_out);
        run_handoff<3, 3, 3, 3>(d_payload, d_flag, d_out);
    }
    printf("done\n");
    return 0;
}
