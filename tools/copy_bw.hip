// tools/copy_bw.hip -- which float4 copy shape reaches the HBM rate on this box?  (the measured roofline denominator)
//   hipcc --offload-arch=gfx950 -O3 tools/copy_bw.hip -o /tmp/copy_bw && /tmp/copy_bw
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4 __attribute__((ext_vector_type(4)));
template <int U, bool NT>
__global__ void __launch_bounds__(256) k_stride(const v4* __restrict__ s, v4* __restrict__ d, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + (U - 1) * stride < n; i += U * stride) {
    v4 r[U];
#pragma unroll
    for (int k = 0; k < U; ++k) r[k] = NT ? __builtin_nontemporal_load(s + i + k * stride) : s[i + k * stride];
#pragma unroll
    for (int k = 0; k < U; ++k) {
      if (NT) __builtin_nontemporal_store(r[k], d + i + k * stride);
      else d[i + k * stride] = r[k];
    }
  }
  for (; i < n; i += stride) d[i] = s[i];
}
template <int U, bool NT>
__global__ void __launch_bounds__(256) k_chunk(const v4* __restrict__ s, v4* __restrict__ d, size_t n) {
  const size_t per = (n + gridDim.x - 1) / gridDim.x;
  const size_t b0 = (size_t)blockIdx.x * per, b1 = b0 + per < n ? b0 + per : n;
  size_t i = b0 + threadIdx.x;
  for (; i + (U - 1) * 256 < b1; i += U * 256) {
    v4 r[U];
#pragma unroll
    for (int k = 0; k < U; ++k) r[k] = NT ? __builtin_nontemporal_load(s + i + k * 256) : s[i + k * 256];
#pragma unroll
    for (int k = 0; k < U; ++k) {
      if (NT) __builtin_nontemporal_store(r[k], d + i + k * 256);
      else d[i + k * 256] = r[k];
    }
  }
  for (; i < b1; i += 256) d[i] = s[i];
}
template <class F>
static void run(const char* name, F launch, size_t bytes) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  launch();
  hipEventRecord(e0, 0);
  for (int i = 0; i < 10; ++i) launch();
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-28s %7.1f GB/s (read + write)\n", name, 2.0 * bytes * 10 / (ms * 1e-3) / 1e9);
}
int main() {
  for (size_t bytes : {(size_t)1 << 30, (size_t)4 << 30}) {
    v4 *a, *b;
    hipMalloc(&a, bytes);
    hipMalloc(&b, bytes);
    hipMemset(a, 1, bytes);
    const size_t n = bytes / 16;
    printf("-- %zu MiB\n", bytes >> 20);
    for (int g : {2048, 4096, 8192, 16384}) {
      char nm[64];
      snprintf(nm, 64, "stride U1 grid %d", g); run(nm, [&] { k_stride<1, false><<<g, 256>>>(a, b, n); }, bytes);
      snprintf(nm, 64, "stride U4 grid %d", g); run(nm, [&] { k_stride<4, false><<<g, 256>>>(a, b, n); }, bytes);
      snprintf(nm, 64, "stride U4 nt grid %d", g); run(nm, [&] { k_stride<4, true><<<g, 256>>>(a, b, n); }, bytes);
      snprintf(nm, 64, "chunk U4 grid %d", g); run(nm, [&] { k_chunk<4, false><<<g, 256>>>(a, b, n); }, bytes);
      snprintf(nm, 64, "chunk U4 nt grid %d", g); run(nm, [&] { k_chunk<4, true><<<g, 256>>>(a, b, n); }, bytes);
      snprintf(nm, 64, "chunk U8 nt grid %d", g); run(nm, [&] { k_chunk<8, true><<<g, 256>>>(a, b, n); }, bytes);
    }
    hipFree(a);
    hipFree(b);
  }
  return 0;
}
