// What does a kernel boundary cost between two big persistent-style launches (1000 x 256 threads, 32 KB of dynamic LDS)
// with and without private (scratch) memory?  rocprofv3 --kernel-trace gives the gaps.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/gap_probe tools/micro/gap_probe.hip && rocprofv3 --kernel-trace ... /tmp/gap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ long long wc() { return (long long)__builtin_readcyclecounter(); }
template <int SCR>
__global__ __launch_bounds__(256) void k_probe(double *out, int idx, int usec) {
    extern __shared__ double lds[];
    double buf[SCR > 0 ? SCR : 1];
    if (SCR > 0) {
        for (int i = 0; i < SCR; ++i) buf[i] = out[(threadIdx.x + i) & 1023];
    }
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (long long)usec * 100) __builtin_amdgcn_s_sleep(8);
    double acc = lds[(threadIdx.x + 1) & 255];
    if (SCR > 0) acc += buf[(idx + threadIdx.x) % SCR]; // dynamic index: the array lives in scratch
    if (acc == -1.0) out[0] = acc;
}
int main() {
    double *out;
    hipMalloc(&out, 8192);
    hipMemset(out, 0, 8192);
    hipFuncSetAttribute((const void *)k_probe<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 32768);
    hipFuncSetAttribute((const void *)k_probe<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 32768);
    for (int rep = 0; rep < 3; ++rep) {
        for (int i = 0; i < 4; ++i) k_probe<0><<<1000, 256, 32768>>>(out, i, 50);
        for (int i = 0; i < 4; ++i) k_probe<64><<<1000, 256, 32768>>>(out, i, 50);
        for (int i = 0; i < 4; ++i) k_probe<0><<<1000, 256, 1024>>>(out, i, 50);
        for (int i = 0; i < 4; ++i) k_probe<0><<<256, 256, 32768>>>(out, i, 50);
        hipDeviceSynchronize();
    }
    printf("done\n");
    return 0;
}
