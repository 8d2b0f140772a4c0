// ds_add_f32 against v_add_f32: is an LDS float atomic add the IEEE round-to-nearest-even addition (subnormals kept) the VALU performs?
// Every lane adds a stream of random floats (all exponents, both signs, subnormals, zeros) to its own LDS slot with atomicAdd and to a
// register with +; the two running sums must agree bit for bit after every addition.  Prints the number of mismatching steps.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ inline uint32_t rng(uint32_t& s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }
__global__ void k(unsigned long long* bad, int steps, int mode)
{
    __shared__ float acc[256];
    uint32_t s = 0x9e3779b9u * (blockIdx.x * 256 + threadIdx.x + 1);
    unsigned long long nb = 0;
    for (int rep = 0; rep < 64; rep++)
    {
        float h = 0.f;
        acc[threadIdx.x] = 0.f;
        for (int i = 0; i < steps; i++)
        {
            uint32_t b = rng(s);
            if (mode == 1) b &= 0x7fffffffu;                       // non-negative addends (the histogram's case)
            if (mode == 2) b = (b & 0x807fffffu);                   // subnormals and zeros only
            if (mode == 3) b = (b & 0x007fffffu) | ((rng(s) % 40u + 100u) << 23); // a narrow band of exponents: frequent ties / cancellation-free sums
            if ((b & 0x7f800000u) == 0x7f800000u) b &= 0xbfffffffu; // no inf / nan
            const float v = __uint_as_float(b);
            h = h + v;
            atomicAdd(&acc[threadIdx.x], v);
            const float a = acc[threadIdx.x];
            if (__float_as_uint(a) != __float_as_uint(h)) { nb++; acc[threadIdx.x] = h; }
            if (!(fabsf(h) < 1e30f)) { h = 0.f; acc[threadIdx.x] = 0.f; }
        }
    }
    if (nb) atomicAdd(bad, nb);
}
int main()
{
    unsigned long long* d; hipMalloc(&d, 8);
    for (int mode = 0; mode < 4; mode++)
    {
        hipMemset(d, 0, 8);
        hipLaunchKernelGGL(k, dim3(1024), dim3(256), 0, 0, d, 64, mode);
        unsigned long long h = 0; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
        printf("mode %d: %llu mismatching steps of %llu\n", mode, h, 1024ull * 256 * 64 * 64);
    }
    return 0;
}
