// Checks the cross-lane primitives of isaacgymdyros_amd/csrc/dw_quad_wave.h on the device: every lane publishes its id, the
// exchanged value must be the id the comment of the primitive promises.  Build and run on the GPU box:
//   hipcc --offload-arch=gfx950 -I isaacgymdyros_amd/csrc tools/dpp_check.hip -o /tmp/dpp_check && /tmp/dpp_check
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "dw_quad_wave.h"

__global__ void k(int *out) {
    const int l = dwq::lane_id() + 64 * (int)(threadIdx.x >> 6);
    const float x = (float)(l & 63);
    out[l * 6 + 0] = (int)dwq::quad_bcast<2>(x);
    out[l * 6 + 1] = (int)dwq::quad_xor1(x);
    out[l * 6 + 2] = (int)dwq::quad_xor2(x);
    out[l * 6 + 3] = (int)dwq::oct_xor4(x);
    out[l * 6 + 4] = dwq::wave_any((l & 63) == 17) ? 1 : 0;
    out[l * 6 + 5] = (int)(dwq::wave_ballot(((l & 63) & 7) == 0) & 0xffffffffull);
}

int main() {
    int *d, h[128 * 6];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(128), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 128; ++t) {
        const int l = t & 63;
        const int want[6] = {(l & ~3) | 2, l ^ 1, l ^ 2, l ^ 4, 1, 0x01010101};
        for (int i = 0; i < 6; ++i) if (h[t * 6 + i] != want[i]) { if (bad < 10) printf("lane %d op %d: got %d want %d\n", t, i, h[t * 6 + i], want[i]); ++bad; }
    }
    printf(bad ? "dpp_check: %d mismatches\n" : "dpp_check: ok\n", bad);
    return bad ? 1 : 0;
}
