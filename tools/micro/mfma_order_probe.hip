// In which order does v_mfma_f32_16x16x4_f32 add its four products?  D = A(16x4) B(4x16) + C on random
// fp32 data, compared bit for bit with candidate orders on the host:
//   seq : fma(a3,b3, fma(a2,b2, fma(a1,b1, fma(a0,b0, c))))      (one fused step per k, ascending)
//   pair: ((a0 b0 + a1 b1) + (a2 b2 + a3 b3)) + c  (and variants)
// and the same for a chain of 32 such instructions (k = 128), which is what the owners' products are.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_order_probe mfma_order_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// A [16][K] row-major, B [K][16], D [16][16]; K a multiple of 4; one wavefront
__global__ void mfma_chain(const float* A, const float* B, float* D, int K) {
  const int l = threadIdx.x, r = l & 15, g = l >> 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < K; k0 += 4)
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[r * K + k0 + g], B[(k0 + g) * 16 + r], acc, 0, 0, 0);
  for (int v = 0; v < 4; ++v) D[(4 * g + v) * 16 + r] = acc[v];
}

// the same for v_mfma_f32_32x32x2_f32 (the first-layer tiles): A [32][K], B [K][32], two k per instruction
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void mfma32_chain(const float* A, const float* B, float* D, int K) {
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  for (int k0 = 0; k0 < K; k0 += 2)
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[r * K + k0 + h], B[(k0 + h) * 32 + r], acc, 0, 0, 0);
  for (int i = 0; i < 16; ++i) D[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = acc[i];
}

static void probe32() {
  const int K = 128;
  std::vector<float> A(32 * K), B(K * 32), D(1024);
  srand(99);
  for (auto& x : A) x = (float)rand() / RAND_MAX * 2.f - 1.f;
  for (auto& x : B) x = ((float)rand() / RAND_MAX * 2.f - 1.f) * 3.f;
  float *dA, *dB, *dD;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, 4096);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(mfma32_chain, dim3(1), dim3(64), 0, 0, dA, dB, dD, K);
  hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
  int eq_seq = 0, eq_pair = 0;
  for (int i = 0; i < 32; ++i)
    for (int j = 0; j < 32; ++j) {
      float seq = 0.f, pair = 0.f;
      for (int k = 0; k < K; ++k) seq = fmaf(A[i * K + k], B[k * 32 + j], seq);
      for (int k = 0; k < K; k += 2) pair = (A[i * K + k] * B[k * 32 + j] + A[i * K + k + 1] * B[(k + 1) * 32 + j]) + pair;
      float d = D[i * 32 + j];
      eq_seq += memcmp(&d, &seq, 4) == 0; eq_pair += memcmp(&d, &pair, 4) == 0;
    }
  printf("32x32x2, K = 128: of 1024 outputs bit-equal to  sequential fma chain: %d   pairwise products: %d\n", eq_seq, eq_pair);
}

int main() {
  probe32();
  for (int K : {4, 128}) {
    std::vector<float> A(16 * K), B(K * 16), D(256);
    srand(7 + K);
    for (auto& x : A) x = (float)rand() / RAND_MAX * 2.f - 1.f;
    for (auto& x : B) x = ((float)rand() / RAND_MAX * 2.f - 1.f) * 3.f;
    float *dA, *dB, *dD;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, 1024);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(mfma_chain, dim3(1), dim3(64), 0, 0, dA, dB, dD, K);
    hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
    int eq_seq = 0, eq_pair = 0, eq_sum = 0;
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < 16; ++j) {
        float seq = 0.f, pair = 0.f, sum = 0.f;
        for (int k0 = 0; k0 < K; k0 += 4) {
          float p[4];
          for (int e = 0; e < 4; ++e) p[e] = A[i * K + k0 + e] * B[(k0 + e) * 16 + j];
          for (int e = 0; e < 4; ++e) seq = fmaf(A[i * K + k0 + e], B[(k0 + e) * 16 + j], seq);
          pair = ((p[0] + p[1]) + (p[2] + p[3])) + pair;
          sum = (((p[0] + p[1]) + p[2]) + p[3]) + sum;
        }
        float d = D[i * 16 + j];
        eq_seq += memcmp(&d, &seq, 4) == 0; eq_pair += memcmp(&d, &pair, 4) == 0; eq_sum += memcmp(&d, &sum, 4) == 0;
      }
    printf("K = %3d: of 256 outputs bit-equal to  sequential fma chain: %d   pairwise products: %d   summed products: %d\n",
           K, eq_seq, eq_pair, eq_sum);
  }
  return 0;
}
