// LDS-DMA probe (gfx950): buffer_load_dwordx4 ... lds through __builtin_amdgcn_raw_ptr_buffer_load_lds -- where the data of lane l
// lands (M0 base + 16 l) and what an out-of-range lane writes (zeros?).   hipcc --offload-arch=gfx950 -O3 dma_probe.hip -o dma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
extern "C" __global__ void k(const float4 *src, float4 *dst, int nbytes)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    float4 *l4 = reinterpret_cast<float4 *>(smem);
    for (int i = 0; i < 4; ++i) l4[i * 256 + tid] = make_float4(-7.f, -7.f, -7.f, -7.f);
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, nbytes, 0x00020000);
    const int wl = __builtin_amdgcn_readfirstlane(tid >> 6);
    __attribute__((address_space(3))) unsigned char *lbase = (__attribute__((address_space(3))) unsigned char *)smem;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)(lbase + (i * 256 + wl * 64) * 16), 16, tid * 16, i * 256 * 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < 4; ++i) dst[i * 256 + tid] = l4[i * 256 + tid];
}
int main()
{
    const int N = 1024;
    std::vector<float> h(4 * N);
    for (int i = 0; i < 4 * N; ++i) h[i] = (float)i;
    float *s, *d;
    hipMalloc(&s, 16 * N); hipMalloc(&d, 16 * N);
    hipMemcpy(s, h.data(), 16 * N, hipMemcpyHostToDevice);
    for (int nbytes : {16 * N, 16 * 700, 16 * 130 + 8}) {
        hipMemset(d, 0xff, 16 * N);
        hipLaunchKernelGGL(k, dim3(1), dim3(256), 16 * N, 0, (const float4 *)s, (float4 *)d, nbytes);
        std::vector<float> o(4 * N);
        hipMemcpy(o.data(), d, 16 * N, hipMemcpyDeviceToHost);
        int same = 0, zero = 0, kept = 0, other = 0, first_other = -1;
        for (int i = 0; i < 4 * N; ++i) {
            if (o[i] == h[i]) ++same; else if (o[i] == 0.f) ++zero; else if (o[i] == -7.f) ++kept; else { if (first_other < 0) first_other = i; ++other; }
        }
        printf("nbytes %d: floats same %d zero %d kept(-7) %d other %d (first %d) expect same %d\n", nbytes, same, zero, kept, other, first_other, nbytes / 4);
    }
    return 0;
}
