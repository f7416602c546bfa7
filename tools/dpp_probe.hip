#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ double dpp_up1(double v) {  // lane i <- lane i-1 (lane 0 keeps its own)
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double dpp_down1(double v) {  // lane i <- lane i+1 (lane 63 keeps its own)
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x130, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x130, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__global__ void k(double* out) {
    double v = 100.0 + threadIdx.x;
    out[threadIdx.x] = dpp_up1(v);
    out[64 + threadIdx.x] = dpp_down1(v);
    out[128 + threadIdx.x] = __shfl_up(v, 1, 64);
    out[192 + threadIdx.x] = __shfl_down(v, 1, 64);
}
int main() {
    double* d; hipMalloc(&d, 256 * 8); k<<<1, 64>>>(d); double h[256]; hipMemcpy(h, d, 256 * 8, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < 64; ++i) { if (h[i] != h[128 + i]) bad++; if (h[64 + i] != h[192 + i]) bad++; }
    printf("dpp mismatches: %d  (up: %g %g %g ... %g | down: %g ... %g %g)\n", bad, h[0], h[1], h[2], h[63], h[64], h[126], h[127]);
    return 0;
}
