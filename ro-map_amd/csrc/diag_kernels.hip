// diag_kernels.hip -- device side of libmon_core_diag.so: the MFMA fragment-layout self-test (the micro-benchmarks are in microbench.hip).
#include "device_common.h"
#include "model.h"

namespace mon {
void set_error(const char* fmt, ...);

// ------------------------------------------------------------------ MFMA fragment-layout self-test
// D[32x32] = A[32x16] * B[16x32]: A lane l -> row l&31, k = 8*(l>>5)+j ; B lane l -> col l&31, same k ;
// D lane l, reg r -> col l&31, row (r&3) + 8*(r>>2) + 4*(l>>5).   A, B, D row-major.
__global__ void __launch_bounds__(64) k_selftest_mfma(const uint16_t* __restrict__ A, const uint16_t* __restrict__ B, float* __restrict__ D) {
    const int l = threadIdx.x, i = l & 31, hk = l >> 5;
    half8_t a, b;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        a[j] = reinterpret_cast<const half_t*>(A)[i * 16 + 8 * hk + j];
        b[j] = reinterpret_cast<const half_t*>(B)[(8 * hk + j) * 32 + i];
    }
    float16_t c = { 0 };
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * hk) * 32 + i] = c[r];
}

int selftest_mfma(int device, const uint16_t* A, const uint16_t* B, float* D) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1 || use_device(device) != hipSuccess) { set_error("selftest: no HIP device");
        return MON_ERR_NO_DEVICE; }
    uint16_t *dA = nullptr, *dB = nullptr; float* dD = nullptr;
    if (hipMalloc((void**)&dA, 32 * 16 * 2) != hipSuccess || hipMalloc((void**)&dB, 16 * 32 * 2) != hipSuccess
            || hipMalloc((void**)&dD, 32 * 32 * 4) != hipSuccess) { set_error("selftest: hipMalloc failed"); return MON_ERR_HIP; }
    hipMemcpy(dA, A, 32 * 16 * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B, 16 * 32 * 2, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_selftest_mfma, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    const hipError_t e = hipMemcpy(D, dD, 32 * 32 * 4, hipMemcpyDeviceToHost);
    hipFree(dA); hipFree(dB); hipFree(dD);
    if (e != hipSuccess) { set_error("selftest: %s", hipGetErrorString(e)); return MON_ERR_HIP; }
    return MON_OK;
}


}  // namespace mon
