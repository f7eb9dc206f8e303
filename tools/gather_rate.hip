// tools/gather_rate.hip -- how many DIVERGENT 16-byte lane loads per cycle does a CU's vector memory path sustain?
// (round 5, review item 3: is the occupancy-independent part of k_trace's time the L1's divergent-address rate?)
//   hipcc --offload-arch=gfx950 -O3 tools/gather_rate.hip -o /tmp/gather_rate && /tmp/gather_rate
// Every lane walks its own pseudo-random sequence of 32-byte records (the BVH node size) of a table of a given size and
// reads one or both 16-byte halves of each (k_trace reads both: two global_load_dwordx4 of the same line per node visit).
// The loads of one lane are independent of each other (the index comes from a counter, not from the data), U of them are in
// flight per lane, and 7 workgroups of 256 threads are resident per CU as in k_trace: this measures THROUGHPUT of the
// address / tag / data-return path, not latency.  DEP=1 makes each index depend on the previous record (a pointer chase,
// the traversal's own shape) for comparison.  Reported: lane-loads per CU-cycle (at the 2.4 GHz nominal clock) and the
// record rate in GB/s counting 32 B per record.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
template <int HALVES, int U, bool DEP>
__global__ void __launch_bounds__(256, 7) k_gather(const uint4* __restrict__ tab, uint32_t mask, int iters, uint32_t* out) {
  uint32_t s = mix(blockIdx.x * 256u + threadIdx.x + 1u);
  uint32_t acc = 0;
  for (int i = 0; i < iters; ++i) {
    uint4 a[U], b[U];
#pragma unroll
    for (int k = 0; k < U; ++k) {
      s = mix(s + (DEP ? acc : 0u) + 0x9e3779b9u);
      const uint32_t rec = s & mask;
      a[k] = tab[2 * (size_t)rec];
      if (HALVES == 2) b[k] = tab[2 * (size_t)rec + 1];
      if (DEP) acc += a[k].x + (HALVES == 2 ? b[k].y : 0u);
    }
    if (!DEP) {
#pragma unroll
      for (int k = 0; k < U; ++k) acc += a[k].x + (HALVES == 2 ? b[k].y : 0u);
    }
  }
  if (acc == 0x12345678u) out[0] = acc;
}
template <int HALVES, int U, bool DEP>
static void run(const uint4* tab, size_t records, int numCU, uint32_t* out, const char* what) {
  const int grid = numCU * 7, iters = DEP ? 256 : 2048 / U;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k_gather<HALVES, U, DEP>), dim3(grid), dim3(256), 0, 0, tab, (uint32_t)(records - 1), iters / 8 + 1, out);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k_gather<HALVES, U, DEP>), dim3(grid), dim3(256), 0, 0, tab, (uint32_t)(records - 1), iters, out);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double recs = (double)grid * 256.0 * iters * U, loads = recs * HALVES;
  printf("%-34s table %8.2f MB  %d x 16 B per record, %2d in flight%s: %8.3f ms  %6.3f lane-loads / CU-cycle  %8.1f G records/s  %8.1f GB/s of records\n", what,
         records * 32.0 / 1048576.0, HALVES, U, DEP ? " (dependent chain)" : "", ms, loads / (ms * 1e-3 * 2.4e9 * numCU), recs / (ms * 1e-3) / 1e9,
         recs * 32.0 / (ms * 1e-3) / 1e9);
  hipEventDestroy(e0); hipEventDestroy(e1);
}
int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int numCU = p.multiProcessorCount;
  printf("%s, %d CUs, clock %d kHz\n", p.name, numCU, p.clockRate);
  const size_t maxRec = (size_t)1 << 26;  // 2 GiB of 32-byte records
  uint4* tab; uint32_t* out;
  hipMalloc((void**)&tab, maxRec * 32);
  hipMalloc((void**)&out, 64);
  hipMemset(tab, 1, maxRec * 32);
  struct { const char* name; size_t rec; } sizes[] = {{"L1-resident (16 KB)", 512}, {"L2-resident (2 MB)", 1 << 16}, {"8 XCD L2s (24 MB)", 3 << 18},
                                                      {"Infinity Cache (110 MB, C2's scene)", (size_t)1 << 22}, {"HBM (2 GiB)", maxRec}};
  for (auto& s : sizes) {
    size_t r = 1; while (r * 2 <= s.rec) r *= 2;  // power of two (mask)
    run<1, 4, false>(tab, r, numCU, out, s.name);
    run<2, 4, false>(tab, r, numCU, out, s.name);
    run<2, 1, false>(tab, r, numCU, out, s.name);
    run<2, 1, true>(tab, r, numCU, out, s.name);
  }
  hipFree(tab); hipFree(out);
  return 0;
}
