// Which device expf does torch-ROCm's exp() agree with bit for bit?  tools/probe_exp.py loads this and compares.
#include <hip/hip_runtime.h>
extern "C" {
__global__ void k_exp(const float *x, float *o0, float *o1, float *o2, float *o3, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = x[i];
    o0[i] = expf(v);                                   // OCML expf as hipcc links it
    o1[i] = __expf(v);                                 // fast path
    o2[i] = exp2f(v * 1.44269504088896340736f);        // exp2 of the scaled argument
    o3[i] = (float)exp((double)v);                     // correctly rounded (via double)
}
void probe_exp(const float *x, float *o0, float *o1, float *o2, float *o3, int n, void *stream) {
    hipLaunchKernelGGL(k_exp, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, o0, o1, o2, o3, n);
}
}
