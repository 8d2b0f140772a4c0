// Micro-benchmark: cycles per dependent DPP move on gfx950 (one wave): wave_rol:1 vs row_shr:1 vs ds_bpermute vs plain v_mov.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void k(float* out, long long* cyc, int n)
{
    float v = threadIdx.x * 1.0f;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; i++)
    {
#pragma unroll
        for (int u = 0; u < 16; u++)
        {
            if (MODE == 0) v = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x134, 0xf, 0xf, false)) + 1.0f;      // wave_rol:1
            else if (MODE == 1) v = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xf, 0xf, false)) + 1.0f; // row_shr:1
            else if (MODE == 2) v = __int_as_float(__builtin_amdgcn_ds_bpermute(((threadIdx.x + 1) & 63) * 4, __float_as_int(v))) + 1.0f;
            else if (MODE == 3) v = v + 1.0f;
            else if (MODE == 4) v = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x13C, 0xf, 0xf, false)) + 1.0f; // wave_ror:1
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = v;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main()
{
    float* d; long long* c; hipMalloc(&d, 256); hipMalloc(&c, 8);
    const char* names[] = {"wave_rol:1 + add", "row_shr:1 + add", "ds_bpermute + add", "add only", "wave_ror:1 + add"};
    for (int m = 0; m < 5; m++)
    {
        long long h = 0; const int n = 1000;
        for (int rep = 0; rep < 2; rep++)
        {
            if (m == 0) hipLaunchKernelGGL(k<0>, 1, 64, 0, 0, d, c, n);
            if (m == 1) hipLaunchKernelGGL(k<1>, 1, 64, 0, 0, d, c, n);
            if (m == 2) hipLaunchKernelGGL(k<2>, 1, 64, 0, 0, d, c, n);
            if (m == 3) hipLaunchKernelGGL(k<3>, 1, 64, 0, 0, d, c, n);
            if (m == 4) hipLaunchKernelGGL(k<4>, 1, 64, 0, 0, d, c, n);
            hipDeviceSynchronize(); hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
        }
        printf("%-22s %.1f ticks per op pair\n", names[m], double(h) / (n * 16));
    }
    return 0;
}
