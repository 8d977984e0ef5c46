// What rate does v_mfma_f64_16x16x4_f64 reach on this part, (a) from registers only, (b) with one ds_read_b64 of the B
// operand per instruction (the k_snode_* tile loops), for 1 / 2 / 4 waves per SIMD and 4 / 8 / 16 independent accumulators?
// build: hipcc -O3 --offload-arch=gfx950 -o mfma_f64_probe mfma_f64_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
template <int NACC, bool LDS, int TW, bool SYNC = false>
__global__ __launch_bounds__(TW) void k(double *out, int iters, const double *in) {
    extern __shared__ double W[];
    const int lane = threadIdx.x & 63;
    for (int t = threadIdx.x; t < 4096; t += blockDim.x) W[t] = in[t & 63];
    __syncthreads();
    v4d acc[NACC];
#pragma unroll
    for (int c = 0; c < NACC; ++c) acc[c] = v4d{0, 0, 0, 0};
    double a = in[lane], b = in[63 - lane];
    const double *p = W + (lane >> 4) * 256 + (lane & 15);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int c = 0; c < NACC; ++c) {
                double bw = b;
                if (LDS) bw = p[g * 1024 + 16 * c];
                acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bw, acc[c], 0, 0, 0);
            }
        }
        if (LDS) asm volatile("" ::: "memory");
        if (SYNC) { // (the tile loops' chunk boundary: 16 LDS writes per thread into the other half, then an LDS-only barrier)
#pragma unroll
            for (int q = 0; q < 16; ++q) W[4096 + ((it & 1) ? 4096 : 0) + q * TW + threadIdx.x] = a * q;
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < NACC; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC, bool LDS, int TW, bool SYNC = false, int WGPC = 1> void run(double *out, const double *in) {
    const int threads = TW, wpe = TW / 256 * WGPC;
    const int wg_per_cu = 1;
    const int grid = 256 * wg_per_cu * WGPC, iters = 2000;
    const size_t lds = WGPC == 1 ? 96 * 1024 : 72 * 1024; // (more than half a CU's LDS: one workgroup per CU; 72 KiB: two)
    hipFuncSetAttribute((const void *)k<NACC, LDS, TW, SYNC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC, LDS, TW, SYNC><<<grid, threads, lds>>>(out, 10, in);
    hipEventRecord(e0);
    k<NACC, LDS, TW, SYNC><<<grid, threads, lds>>>(out, iters, in);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * (threads / 64) * iters * 4.0 * NACC * 2048.0;
    std::printf("acc %2d  lds %d  waves/SIMD %d  chunk barrier %d  workgroups/CU %d: %7.2f TFLOP/s  (%.3f ms)\n", NACC, (int)LDS, wpe, (int)SYNC, WGPC, flops / ms / 1e9, ms);
}
int main() {
    double *out, *in;
    hipMalloc(&out, 1 << 24); hipMalloc(&in, 4096 * 8);
    hipMemset(in, 0, 4096 * 8);
    run<4, false, 256>(out, in); run<16, false, 256>(out, in); run<4, true, 256>(out, in); run<8, true, 256>(out, in); run<16, true, 256>(out, in);
    run<4, false, 512>(out, in); run<16, false, 512>(out, in); run<4, true, 512>(out, in); run<8, true, 512>(out, in); run<16, true, 512>(out, in);
    // the tile loops' shape: a barrier and 16 LDS writes per 64 matrix instructions of a wave
    run<16, true, 512, true>(out, in); run<16, true, 256, true, 2>(out, in); run<16, true, 256, false, 2>(out, in); run<4, true, 512, true, 2>(out, in);
    run<4, false, 1024>(out, in); run<4, true, 1024>(out, in); run<8, true, 1024>(out, in); // (4 waves per SIMD: 128 registers, no room for 16 accumulators)
    return 0;
}
