// Do v_mfma_f32_32x32x2_f32 and v_mfma_f32_16x16x4_f32 round identically when fed the same k sequence?
// (If yes, a 16x16-tile kernel for small M could keep results bit-identical to the 32x32-tile one.)
//   hipcc --offload-arch=gfx950 -O2 tools/micro/mfma_order.hip -o tools/micro/mfma_order && tools/micro/mfma_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int K = 64;   // reduction length (multiple of 8)

// A [32][K], B [32][K] row-major; C [32][32] = A B^T.
// 32x32x2: lanes 0-31 supply k, lanes 32-63 supply k' per instruction (order given by ka[], kb[]).
__global__ void k32(const float *A, const float *B, float *C) {
    const int lane = threadIdx.x, r = lane & 31, half = lane >> 5;
    f32x16 acc = {0};
    for (int k8 = 0; k8 < K; k8 += 8)
        for (int s = 0; s < 4; ++s) {              // instruction s consumes k8+s (lanes<32) and k8+4+s (lanes>=32)
            const int k = k8 + s + 4 * half;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[r * K + k], B[r * K + k], acc, 0, 0, 0);
        }
    for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * half;
        C[row * 32 + r] = acc[i];
    }
}
// 16x16x4: lane group g = lane >> 4 supplies one k per instruction.  mode 0: groups 0..3 take
// (k, k+4, k+1, k+5) then (k+2, k+6, k+3, k+7) -- the same SEQUENCE as the 32x32x2 kernel if the hardware
// adds the groups in order.  Each (16x16) quadrant of the 32x32 result by one launch block.
__global__ void k16(const float *A, const float *B, float *C, int mode) {
    const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
    const int qm = blockIdx.x >> 1, qn = blockIdx.x & 1;
    f32x4 acc = {0};
    for (int k8 = 0; k8 < K; k8 += 8)
        for (int j = 0; j < 2; ++j) {
            int k;
            if (mode == 0) { const int seq[2][4] = {{0, 4, 1, 5}, {2, 6, 3, 7}}; k = k8 + seq[j][g]; }
            else { k = k8 + 4 * j + g; }           // plain order 0..3, 4..7
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(qm * 16 + r) * K + k], B[(qn * 16 + r) * K + k], acc, 0, 0, 0);
        }
    for (int i = 0; i < 4; ++i) C[(qm * 16 + 4 * g + i) * 32 + qn * 16 + r] = acc[i];
}

int main() {
    std::vector<float> A(32 * K), B(32 * K), C32(1024), C16(1024), C16b(1024);
    srand(7);
    for (auto &v : A) v = (rand() / (float)RAND_MAX - 0.5f) * 4.f;
    for (auto &v : B) v = (rand() / (float)RAND_MAX - 0.5f) * 4.f;
    float *dA, *dB, *dC;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 4096);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k32, dim3(1), dim3(64), 0, 0, dA, dB, dC);
    hipMemcpy(C32.data(), dC, 4096, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(k16, dim3(4), dim3(64), 0, 0, dA, dB, dC, 0);
    hipMemcpy(C16.data(), dC, 4096, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(k16, dim3(4), dim3(64), 0, 0, dA, dB, dC, 1);
    hipMemcpy(C16b.data(), dC, 4096, hipMemcpyDeviceToHost);
    // host references: sequential fmaf in the 32x32x2 sequence, and fp64
    int same0 = 0, same1 = 0, fmaf_ok = 0;
    double maxd = 0;
    for (int m = 0; m < 32; ++m)
        for (int n = 0; n < 32; ++n) {
            float s = 0.f;
            double d = 0;
            for (int k8 = 0; k8 < K; k8 += 8)
                for (int t = 0; t < 4; ++t) {
                    s = fmaf(A[m * K + k8 + t], B[n * K + k8 + t], s);
                    s = fmaf(A[m * K + k8 + 4 + t], B[n * K + k8 + 4 + t], s);
                }
            for (int k = 0; k < K; ++k) d += (double)A[m * K + k] * B[n * K + k];
            const int i = m * 32 + n;
            same0 += C32[i] == C16[i];
            same1 += C32[i] == C16b[i];
            fmaf_ok += C32[i] == s;
            if (fabs(C32[i] - d) > maxd) maxd = fabs(C32[i] - d);
        }
    printf("32x32x2 == sequential fmaf chain (k0,k4,k1,k5,..): %d / 1024\n", fmaf_ok);
    printf("16x16x4 (same sequence)  == 32x32x2: %d / 1024\n", same0);
    printf("16x16x4 (plain k order)  == 32x32x2: %d / 1024\n", same1);
    printf("max |32x32x2 - fp64| = %.3g\n", maxd);
    return 0;
}
