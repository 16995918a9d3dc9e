// Calibration of the FETCH_SIZE counter for the access pattern of the diagonal gather: every lane
// reads one dword per load (64 consecutive dwords per wave) through a buffer descriptor with a scalar
// offset.  Streams NBYTES (default 1 GiB, well past the 256 MiB Infinity Cache) exactly once;
// run under `rocprofv3 --pmc FETCH_SIZE` and compare the counter with the known byte count.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

__global__ void __launch_bounds__(256) stream_dwords(const float *p, size_t ndw, float *out)
{
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const size_t nwaves = (size_t)gridDim.x * 4;
    float s = 0.f;
    // each wave walks 256-byte pieces, 8 in flight
    for (size_t base = wave * 512; base + 512 <= ndw; base += nwaves * 512) {
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)(p + base), 0, 0x7fffffff, 0x00020000);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            s += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, lane * 4, j * 256, 0));
    }
    if (s == 12345.678f) out[0] = s;
}

int main(int argc, char **argv)
{
    const size_t nbytes = argc > 1 ? strtoull(argv[1], 0, 10) : (1ull << 30);
    float *p, *out;
    if (hipMalloc(&p, nbytes) != hipSuccess || hipMalloc(&out, 256) != hipSuccess) return 1;
    hipMemset(p, 0, nbytes);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(stream_dwords, dim3(4096), dim3(256), 0, 0, p, nbytes / 4, out);
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    printf("streamed %zu bytes per launch, 3 launches\n", nbytes);
    return 0;
}
