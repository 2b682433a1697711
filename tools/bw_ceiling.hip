// Read-only HBM ceiling for the stream kernel's access pattern on THIS box, at 1, 2 and 3 workgroups per CU: 512-thread
// workgroups, one 16 KiB tile per wave (64 lanes x 16 B x 16 iterations), three 16-byte non-temporal loads in flight per lane,
// runs of 16 tiles drawn from an atomic cursor — hg_stream_kernel's memory side with the filter, the newline count and the queue
// taken out.  The denominator for "82 % alone / 67 % beside the side passes".  hipcc --offload-arch=gfx950 -O3 tools/bw_ceiling.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <bool NT>
__global__ __launch_bounds__(512) void k_read(const u32x4 *__restrict__ t, uint32_t ntiles, uint32_t *cursor, uint32_t *out) {
  extern __shared__ uint32_t pad[];  // sized by the host so that exactly k workgroups fit on a CU
  __shared__ uint32_t s_run;
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t acc = 0;
  for (;;) {
    if (threadIdx.x == 0) s_run = atomicAdd(cursor, 16u);
    __syncthreads();
    const uint32_t r0 = s_run;
    __syncthreads();
    if (r0 >= ntiles) break;
    for (uint32_t r = r0 + wave; r < r0 + 16 && r < ntiles; r += 8) {
      const u32x4 *p = t + static_cast<uint64_t>(r) * 1024 + lane;
      u32x4 buf[3];
#pragma unroll
      for (int d = 0; d < 3; d++) buf[d] = NT ? __builtin_nontemporal_load(p + d * 64) : p[d * 64];
#pragma unroll
      for (int it = 0; it < 16; it++) {
        const u32x4 cur = buf[it % 3];
        if (it + 3 < 16) buf[it % 3] = NT ? __builtin_nontemporal_load(p + (it + 3) * 64) : p[(it + 3) * 64];
        acc ^= cur.x ^ cur.y ^ cur.z ^ cur.w;
      }
    }
  }
  if (acc == 0x12345u) out[0] = acc + pad[0];
}

int main(int argc, char **argv) {
  const uint64_t bytes = (argc > 1 ? atoll(argv[1]) : 8ull) << 30;
  u32x4 *t;
  uint32_t *out, *cursor;
  CHECK(hipMalloc(&t, bytes));
  CHECK(hipMalloc(&out, 64));
  CHECK(hipMalloc(&cursor, 64));
  CHECK(hipMemset(t, 1, bytes));
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const uint32_t ntiles = static_cast<uint32_t>(bytes >> 14);
  printf("# %s, %d CUs, %.0f GiB read per launch, best of 5; peak 8000 GB/s (MI355X_MICROARCH.md)\n", prop.name, cus, bytes / 1073741824.0);
  for (int nt = 1; nt >= 0; nt--)
    for (int k = 1; k <= 3; k++) {
      const size_t lds = (160 * 1024) / k - 2048;  // k workgroups fill a CU's LDS
      auto kern = nt ? k_read<true> : k_read<false>;
      CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
      int per_cu = 0;
      CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 512, lds));
      hipEvent_t a, b;
      CHECK(hipEventCreate(&a));
      CHECK(hipEventCreate(&b));
      float best = 1e9;
      for (int r = 0; r < 6; r++) {
        CHECK(hipMemsetAsync(cursor, 0, 4));
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL(kern, dim3(cus * k), dim3(512), lds, 0, t, ntiles, cursor, out);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms;
        CHECK(hipEventElapsedTime(&ms, a, b));
        if (r && ms < best) best = ms;
      }
      printf("%-14s %d workgroup(s) per CU (occupancy query: %d)  %8.3f ms  %7.1f GB/s  %.3f of peak\n", nt ? "non-temporal" : "default loads", k, per_cu, best, bytes / best / 1e6,
             bytes / best / 1e6 / 8000.0);
    }
  return 0;
}
