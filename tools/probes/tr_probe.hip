// Probe: what does ds_read_b64_tr_b16 deliver to each lane?  LDS holds lds[i] = i (as f16, exact below 2048);
// lane l reads 8 bytes at byte address addr[l] (host supplied); out[l*4 + j] = element j the lane received.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef _Float16 f16;
typedef f16 f16x4 __attribute__((ext_vector_type(4)));

__global__ void tr_probe_kernel(const int* __restrict__ addr, float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) f16 lds[2048];
    for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = (f16)(float)i;
    __syncthreads();
    const unsigned a = (unsigned)(uintptr_t)lds + (unsigned)addr[threadIdx.x];
    f16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (float)v[j];
}

extern "C" int tr_probe(const int* addr_dev, float* out_dev) {
    hipLaunchKernelGGL(tr_probe_kernel, dim3(1), dim3(64), 0, 0, addr_dev, out_dev);
    return (int)hipDeviceSynchronize();
}
