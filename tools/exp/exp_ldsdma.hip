// Checks the LDS-DMA addressing the ring kernels rely on (gfx950):
//   buffer_load_dwordx4 voff, rsrc, soff offen offset:IMM lds
//     global address = base + soff + voff + IMM,   LDS address = M0 + IMM + 16 * lane
// with only the first 20 lanes active, and a descriptor base moved back by 4096 bytes.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

__global__ void k(const float *__restrict__ src, float *__restrict__ out, uint32_t soffA, uint32_t soffB)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    for (int i = lane; i < 1024; i += 64) lds[i] = -1.f;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)src - 4096), 0, 0x7fffffff, 0x00020000);
    const uint32_t ldsbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float *)lds;
    if (lane < 20) {
        const int voff = lane * 16;
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\t"
                     "buffer_load_dwordx4 %0, %1, %3 offen offset:640 lds\n\t"
                     "buffer_load_dwordx4 %0, %1, %4 offen offset:960 lds"
                     :: "v"(voff), "s"(r), "s"(ldsbase), "s"(soffA + 4096 - 640), "s"(soffB + 4096 - 960) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int i = lane; i < 1024; i += 64) out[i] = lds[i];
}

int main()
{
    const int n = 1 << 16;
    std::vector<float> h(n);
    for (int i = 0; i < n; ++i) h[i] = (float)i;
    float *d, *o;
    hipMalloc(&d, n * 4 + 8192);
    hipMalloc(&o, 1024 * 4);
    d += 1024;
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    const uint32_t a = 1237 * 4, b = 40001 * 4;     // unaligned float indices
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, d, o, a, b);
    std::vector<float> r(1024);
    hipMemcpy(r.data(), o, 1024 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 1024; ++i) {
        float want = -1.f;
        if (i >= 160 && i < 240) want = 1237 + (i - 160);
        if (i >= 240 && i < 320) want = 40001 + (i - 240);
        if (r[i] != want) { if (bad < 8) printf("lds[%d] = %g want %g\n", i, r[i], want); ++bad; }
    }
    printf("ldsdma check: %d bad\n", bad);
    return bad != 0;
}
