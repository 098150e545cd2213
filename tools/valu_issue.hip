// valu_issue.hip -- how fast does one SIMD of the MI355X issue plain fp32 vector instructions, and does a second wave on the SIMD
// add issue slots?  Independent v_fma_f32 chains, 1 / 2 / 4 waves per SIMD (blocks of 256 threads = one wave on each SIMD of a CU).
// build: hipcc --offload-arch=gfx950 -O2 -o tools/valu_issue tools/valu_issue.hip ; run: tools/valu_issue
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int PK>
__global__ __launch_bounds__(256) void k_fma(float *out, int iters, float a, float b) {
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = (float)threadIdx.x * 1e-3f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) x[i] = __builtin_fmaf(x[i], a, b);
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    float *out;
    hipMalloc(&out, 4096 * 256 * sizeof(float));
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wps = 1; wps <= 4; wps *= 2) {
        const int blocks = cus * wps;
        hipLaunchKernelGGL(k_fma<0>, dim3(blocks), dim3(256), 0, 0, out, 100, 1.0001f, 0.5f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_fma<0>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double instr_per_wave = (double)iters * 64;
        // cycles per instruction as seen by ONE SIMD: elapsed cycles / (instructions issued on that SIMD)
        const double clk_ghz = p.clockRate * 1e-6;
        printf("waves/SIMD %d: %.3f ms; per wave %.2f cycles/instr; per SIMD %.2f cycles/instr (clock %.2f GHz from props); %.1f TFLOP/s\n", wps, ms,
               ms * 1e-3 * clk_ghz * 1e9 / instr_per_wave, ms * 1e-3 * clk_ghz * 1e9 / (instr_per_wave * wps), clk_ghz,
               instr_per_wave * 128 * blocks * 4 / (ms * 1e-3) * 1e-12);
    }
    return 0;
}
