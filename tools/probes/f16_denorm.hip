// Does v_mfma_f32_32x32x16_f16 keep fp16 subnormal inputs?  A[i][k] = a (k == 0), B[k][j] = b (k == 0) -> D[i][j] = a * b.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float a, float b, float* o) {
    const int lane = threadIdx.x;
    f16x8 A = {}, B = {};
    if (lane < 32) { A[0] = (_Float16)a; B[0] = (_Float16)b; }   // hi = 0, element 0 -> k = 0
    f32x16 acc = {};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, acc, 0, 0, 0);
    if (lane == 0) o[0] = acc[0];
}
int main() {
    float* d; hipMalloc(&d, 4);
    const float as[] = {1.0f, 3.0517578125e-05f /*2^-15 subnormal*/, 5.9604644775390625e-08f /*2^-24 min subnormal*/, 1.5f * 5.9604644775390625e-08f * 1024};
    for (float a : as) {
        for (float b : {1.0f, 1024.0f, 3.0517578125e-05f}) {
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, b, d);
            float h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
            printf("a=%.10e b=%.10e  mfma=%.10e  exact=%.10e  %s\n", a, b, h, (double)(float)(_Float16)a * (double)(float)(_Float16)b,
                   h == (float)((double)(float)(_Float16)a * (double)(float)(_Float16)b) ? "KEPT" : "DIFF");
        }
    }
    return 0;
}
